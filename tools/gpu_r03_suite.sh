#!/bin/bash
# round 3: whole GPU suite + default bench (with the PCIe-inclusive sample) + TopK at the large-v3 default dictionary
set -u
O=$PWD/gpurun_out/r03_suite
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt
timeout 600 python bench.py > $O/default.json 2> $O/default.err
python - <<PY
import json
d=json.loads(open('$O/default.json').read().strip().splitlines()[-1])
print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel_launches'], d.get('cpu_baseline',{}).get('value'), d.get('pcie_inclusive'))
PY
timeout 600 python bench.py --steps 20 --warmup 5 > $O/driver_style.json 2> $O/driver_style.err
tail -c 600 $O/driver_style.json
B="timeout 600 python bench.py --no-cpu-baseline"
$B --variant topk --d 1280 --n 40960 --k 32 --steps 20 --warmup 5 --dead-threshold 1e15 --breakdown > $O/topk_d1280_n40960.json 2> $O/topk_d1280.err
python - <<PY
import json
d=json.loads(open('$O/topk_d1280_n40960.json').read().strip().splitlines()[-1])
print('topk d1280 n40960', d['ms_per_step'], d['value'], {k:v for k,v in (d.get('kernel_ms') or {}).items() if v})
PY
$B --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e15 --breakdown > $O/c3.json 2> $O/c3.err
python - <<PY
import json
d=json.loads(open('$O/c3.json').read().strip().splitlines()[-1])
print('c3', d['ms_per_step'], d['value'])
PY

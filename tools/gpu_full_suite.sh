#!/bin/bash
set -u
O=$PWD/gpurun_out/full_suite
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
B="python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --breakdown"
$B --dead-threshold 1e5 > $O/c3_auxk.json 2> $O/c3_auxk.err
python - <<PY
import json
d=json.loads(open('$O/c3_auxk.json').read().strip().splitlines()[-1])
km=d.get('kernel_ms') or {}
print('c3_auxk', round(d['ms_per_step'],3), d.get('loss'), {k:v for k,v in km.items() if v})
PY
python bench.py > $O/default.json 2> $O/default.err; python -c "
import json
d=json.loads(open('$O/default.json').read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline'])"

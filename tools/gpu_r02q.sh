#!/bin/bash
set -u
O=$PWD/gpurun_out/r02q
mkdir -p $O
timeout 900 python -m pytest tests/test_topk_gpu.py -m gpu -q -x -p no:cacheprovider -k "auxk or dacts or real_dict" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
B="python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --breakdown"
$B --dead-threshold 1e5 > $O/c3_auxk.json 2> $O/c3_auxk.err
for f in c3_auxk; do python - <<PY
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
km=d.get('kernel_ms') or {}
print('$f', round(d['ms_per_step'],3), d.get('loss'), {k:v for k,v in km.items() if v})
PY
done

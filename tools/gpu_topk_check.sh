#!/bin/bash
set -u
O=$PWD/gpurun_out/topk_check
mkdir -p $O
timeout 1500 python -m pytest tests/test_topk_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
B="python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --breakdown"
$B --dead-threshold 1e15 > $O/c3.json 2> $O/c3.err
$B --dead-threshold 1e5 > $O/c3_auxk.json 2> $O/c3_auxk.err
$B --dead-threshold 1e5 --dbg 76 > $O/c3_auxk_old.json 2> $O/c3_auxk_old.err
for f in c3 c3_auxk c3_auxk_old; do python - <<PY
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', round(d['ms_per_step'],3), d.get('loss'), json.dumps(d.get('kernel_ms')))
PY
done

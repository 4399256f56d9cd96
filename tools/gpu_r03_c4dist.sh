#!/bin/bash
set -u
O=$PWD/gpurun_out/r03_c4dist
mkdir -p $O
timeout 1200 python -m pytest tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider -k "two_processes" > $O/pytest_dp.txt 2>&1
tail -4 $O/pytest_dp.txt
B="timeout 600 python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3 --breakdown"
for v in "c4" "c4_fd_p2p --force-dist --dp p2p" "c4_fd_rccl --force-dist --dp rccl" "c4_fd_host --force-dist --dp host" "c4_b" "c4_fd_p2p_b --force-dist --dp p2p"; do
  set -- $v; name=$1; shift
  $B "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    km=d.get('kernel_ms') or {}
    print('$name', round(d['ms_per_step'],3), {k:v for k,v in km.items() if v and v>0.2})
except Exception as e:
    print('$name FAILED', e, open('$O/$name.err').read()[-500:])
PY
done

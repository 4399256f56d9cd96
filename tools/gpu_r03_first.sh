#!/bin/bash
# round 3, first GPU pass: the new two-process exchange tests + dist / range-split bench lines
set -u
O=$PWD/gpurun_out/r03_first
mkdir -p $O
timeout 1500 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_dp.txt 2>&1
tail -15 $O/pytest_dp.txt
timeout 900 python -m pytest tests/test_topk_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest_topk.txt 2>&1
tail -8 $O/pytest_topk.txt
B="timeout 300 python bench.py --no-cpu-baseline"
run() { name=$1; shift; $B "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), d['roofline']['kernel_launches'], d['config'].get('dp'))
except Exception as e:
    print('$name FAILED', e, open('$O/$name.err').read()[-600:])
PY
}
run default
run default_b
run driver_style --steps 20 --warmup 5
run fd_p2p --force-dist --dp p2p
run fd_p2p_bf16 --force-dist --dp p2p --dp-payload bfloat16
run fd_p2p_ov2 --force-dist --dp p2p --dp-overlap 2
run fd_p2p_ov3 --force-dist --dp p2p --dp-overlap 3
run fd_rccl --force-dist --dp rccl
run fd_host --force-dist --dp host
run ov2_nodist --dp-overlap 2
run n12288 --n 12288 --steps 100 --warmup 10
run n12288_ov2 --n 12288 --steps 100 --warmup 10 --force-dist --dp p2p --dp-overlap 2
run default_c

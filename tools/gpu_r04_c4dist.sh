#!/bin/bash
# round 4: round-sized column chunks of the data-parallel weight-gradient GEMM at C4: tests, then plain / p2p / rccl / host on one rank,
# interleaved (the equal chunks of round 3: --dbg 84)
set -u
O=gpurun_out/r04_c4dist; mkdir -p $O
timeout 1200 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "rccl_single_rank or l1_generic" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3"
for i in 1 2; do
  for mode in "" "--force-dist" "--force-dist --dp rccl" "--force-dist --dbg 84" "--force-dist --dp rccl --dbg 84" "--force-dist --dp-host"; do
    echo -n "[C4 ${mode:-plain}] "
    $B $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms', d['config'].get('dp'))"
  done
done > $O/ab_c4_forcedist.txt 2>&1; cat $O/ab_c4_forcedist.txt

#!/bin/bash
# round 4: the hardened peer exchange on the GPU -- (1) the data-parallel GPU tests (2- and 4-process runs with audits and replica
# checks, fault injection caught by self-test / audit / checksums, late-peer poisoning, absent peer), (2) one-rank protocol overhead
# interleaved on one box (tools/ab_forcedist.sh), (3) the same with FREUD_P2P_FINEGRAINED=1 (fine-grained G / Gb / stats)
set -u
O=gpurun_out/r04_dp; mkdir -p $O
timeout 1500 python -m pytest tests/test_dp_gpu.py -x -q -m gpu > $O/pytest_dp.txt 2>&1; tail -15 $O/pytest_dp.txt
bash tools/ab_forcedist.sh > $O/ab_forcedist.txt 2>&1; cat $O/ab_forcedist.txt
for i in 1 2 3; do
  for fg in 0 1; do
    echo -n "[finegrained=$fg] "
    FREUD_P2P_FINEGRAINED=$fg python bench.py --no-cpu-baseline --steps 400 --warmup 20 --force-dist 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,1), 'us', d['config'].get('dp'))"
  done
done > $O/ab_finegrained.txt 2>&1; cat $O/ab_finegrained.txt

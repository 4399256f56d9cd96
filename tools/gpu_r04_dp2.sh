#!/bin/bash
# round 4, second data-parallel GPU call: dp tests after the auditor / fence trims, one-rank overhead interleaved, kernel trace of
# the forced-dist step, C4 with fine-grained G (cost of FREUD_P2P_FINEGRAINED at 210 MB of gradient), GEMM tile-walk patch shapes
set -u
O=gpurun_out/r04_dp2; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_dp_gpu.py -x -q -m gpu > $O/pytest_dp.txt 2>&1; tail -5 $O/pytest_dp.txt
bash tools/ab_forcedist.sh > $O/ab_forcedist.txt 2>&1; cat $O/ab_forcedist.txt
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/trace_fd -o fd --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-pcie-sample --force-dist --steps 400 --warmup 20 --spinup 0.2 > $R/$O/bench_fd_under_rocprof.json 2> $R/$O/trace_fd.log)
python3 tools/trace_timeline.py $(find $O/trace_fd -name "*kernel_trace.csv" | head -1) 400 > $O/timeline_forcedist_p2p.txt 2>&1
rm -f $(find $O/trace_fd -name "*kernel_trace.csv"); cat $O/timeline_forcedist_p2p.txt | head -40
for i in 1 2; do
  for fg in 0 1; do
    echo -n "[C4 finegrained=$fg] "
    FREUD_P2P_FINEGRAINED=$fg python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3 --force-dist 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms', d['config'].get('dp'))"
  done
done > $O/ab_finegrained_c4.txt 2>&1; cat $O/ab_finegrained_c4.txt
for i in 1 2; do
  for lib in "" build/ab/libfreud_sae_grp4.so build/ab/libfreud_sae_grp16.so; do
    echo -n "[C4 ${lib:-group8}] "
    FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3 --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read(); k=json.loads(re.search(r'\{.*?\}', t[t.index('per-kernel'):]).group(0)); m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1)
print('step', m, {x:k[x] for x in k if k[x]>0.3})"
  done
done > $O/ab_group_m_c4.txt 2>&1; cat $O/ab_group_m_c4.txt

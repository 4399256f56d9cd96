#!/bin/bash
# round 5: TopK -- topk_de_kernel with eight rows' loads in flight and its switches as template tags, topk_decode_kernel's epilogue loads batched
# (current) against the build before (prede): tests, then C3 without / with dead latents on one box
O=gpurun_out/r05_tksmall; mkdir -p $O
timeout 2400 python -m pytest tests/test_topk_gpu.py tests/test_engine_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
DT=1e15 bash tools/ab_topk.sh build/ab/libfreud_sae_prede.so > $O/ab_c3.txt 2>&1
bash tools/ab_topk.sh build/ab/libfreud_sae_prede.so > $O/ab_c3_auxk.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --spinup 0.3 --dead-threshold 1e15 > /dev/null 2> $GRAFT_REPO_ROOT/$O/stats.log
cd $GRAFT_REPO_ROOT
rm -f $(find $O/stats -name "*kernel_trace.csv")
tail -3 $O/tests.txt; cat $O/ab_c3.txt $O/ab_c3_auxk.txt; python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r05_tksmall/stats/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY

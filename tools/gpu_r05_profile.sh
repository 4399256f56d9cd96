#!/bin/bash
# round 5: fp8 single-store upper bound, then the profile set (tools/profile_round.sh r05) and the counter passes of C3 / C4 / C5
O=gpurun_out/r05_fp8skip; mkdir -p $O
for rep in 1 2; do
  for dbg in 0 87; do
    echo -n "[C5fp8 dbg=$dbg] "; python bench.py --d 1280 --n 81920 --steps 6 --warmup 2 --precision fp8 --no-cpu-baseline --dbg $dbg 2>&1 | grep -E "per-kernel" | sed -e 's/.*"enc_fwd_gemm": \([0-9.]*\).*"dec_fwd_gemm": \([0-9.]*\).*/enc \1 dec \2/'
  done
done > $O/fp8skip.txt 2>&1
cat $O/fp8skip.txt
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
bash tools/gpu_r04_pmc_workloads.sh > gpurun_out/pmc_workloads_r05.log 2>&1
tail -5 gpurun_out/profile_round_r05.log

#!/bin/bash
# round 4 against round 3's final build (build/ab/libfreud_sae_r3final.so, built from commit b0676df) on ONE box, interleaved:
# C2 per-kernel (level-2 profile) and driver-style lines (boxes differ by +-3 %, builds must be compared on one)
set -u
O=gpurun_out/r04_vs_r03; mkdir -p $O
bash tools/ab_bench.sh build/ab/libfreud_sae_r3final.so > $O/ab_c2.txt 2>&1; cat $O/ab_c2.txt
for i in 1 2 3; do
  for lib in "" build/ab/libfreud_sae_r3final.so; do
    echo -n "[driver-style ${lib:-r04}] "
    FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --no-pcie-sample --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,1), 'us step;', round(d['roofline']['kernel_avg_ms']*1000,1), 'us bwd_fused; frac', round(d['roofline']['frac'],4))"
  done
done > $O/ab_driver_style.txt 2>&1; cat $O/ab_driver_style.txt
for i in 1 2; do
  for lib in "" build/ab/libfreud_sae_r3final.so; do
    echo -n "[C4 ${lib:-r04}] "
    FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms')"
  done
done > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt

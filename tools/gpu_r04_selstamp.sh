#!/bin/bash
# round 4: where the compact AuxK select spends a row's 40 k cycles (diagnostic build with s_memtime stamps)
set -u
O=gpurun_out/r04_selstamp; mkdir -p $O
for i in 1 2; do
FREUD_SAE_LIB=build/ab/libfreud_sae_selstamp.so timeout 600 python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e5 --dbg 67 > $O/out_$i.json 2> $O/err_$i.txt
grep "AuxK compact select" $O/err_$i.txt | tee -a $O/stamps.txt; tail -3 $O/err_$i.txt
done

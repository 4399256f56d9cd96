#!/bin/bash
# round 5: fused forward, stamps split by launch round (workgroups 0-255 start together, 256-511 as CUs come free)
O=gpurun_out/r05_rounds; mkdir -p $O
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 65 2>&1 | grep -E "^fwd "; done > $O/stamps.txt 2>&1
bash tools/ab_fwd.sh build/ab/libfreud_sae_mtestv4.so > $O/ab_fwd.txt 2>&1
cat $O/stamps.txt $O/ab_fwd.txt

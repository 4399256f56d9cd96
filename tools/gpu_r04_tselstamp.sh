#!/bin/bash
# phase stamps of the tile-driven TopK select (C3, no dead latents)
set -u
O=gpurun_out/r04_tselstamp; mkdir -p $O
for lib in tselstamp; do
  echo "== $lib"
  FREUD_SAE_LIB=build/ab/libfreud_sae_$lib.so timeout 600 python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 10 --warmup 3 --dead-threshold 1e15 --dbg 68 2>&1 | grep -E "tile-driven|ms_per_step" | cut -c1-700
done > $O/stamps.txt 2>&1
cat $O/stamps.txt

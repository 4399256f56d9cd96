#!/bin/bash
O=gpurun_out/r05_epistamp; mkdir -p $O
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 65 2>&1 | grep -E "fwd per-workgroup|fwd in-kernel|fwd epilogue"; done > $O/stamps.txt 2>&1
cat $O/stamps.txt

#!/bin/bash
# round 6: running source pointers in the 16x16x32 loops (G2_RUNPTR; gemm_k16r / gemm_a3r) against pointers formed anew per K tile
# (gemm_k16n / gemm_a3n): hashes / bit-compare, then timing, stand-alone on random dense operands
O=gpurun_out/r06_runptr; mkdir -p $O
{
for shp in "1280 4096 8192 1 1" "1280 4096 8192 1 3" "768 2048 4160 1 5" "256 256 64 1 1" "256 512 192 1 2"; do
  for b in gemm_k16n gemm_k16r; do echo -n "$b [$shp]: "; KB_HASH=1 timeout 120 build/kbench/$b $shp | tr '\n' ' '; echo; done
done
for b in gemm_a3n gemm_a3r; do
  echo -n "$b compare K=1280: "; timeout 120 build/kbench/$b 4096 4096 1280 3
  echo -n "$b compare K=4096: "; timeout 120 build/kbench/$b 2048 1280 4096 3
  echo -n "$b compare K=128:  "; timeout 120 build/kbench/$b 1024 1024 128 3
done
for rep in 1 2 3; do
  for b in gemm_k16n gemm_k16r; do
    echo -n "$b: "; timeout 120 build/kbench/$b 1280 40960 131072 1 5
    echo -n "$b: "; timeout 120 build/kbench/$b 1280 81920 131072 1 3
    echo -n "$b: "; timeout 120 build/kbench/$b 768 24576 65536 1 3
  done
  for b in gemm_a3n gemm_a3r; do
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 1280 40960 0
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 1280 81920 0
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 768 24576 0
  done
done
} > $O/kbench.txt 2>&1
cat $O/kbench.txt

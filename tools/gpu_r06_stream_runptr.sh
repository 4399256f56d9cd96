#!/bin/bash
# round 6: running source pointers in the streaming GEMM's 16x16x32 loop (G2S_RUNPTR; gemm_s16r) against formed pointers (gemm_s16n):
# bit-compare against the tile form (mode 5), then timing (mode 4), stand-alone on random dense operands
O=gpurun_out/r06_stream_runptr; mkdir -p $O
{
for b in gemm_s16n gemm_s16r; do
  echo "== $b: bit-compare stream vs tile form"
  timeout 120 build/kbench/$b 8192 8192 1280 5
  timeout 120 build/kbench/$b 4096 12288 768 5
  timeout 120 build/kbench/$b 2048 4096 256 5
  timeout 120 build/kbench/$b 2048 4096 384 5
done
for rep in 1 2 3; do
  for b in gemm_s16n gemm_s16r; do
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 40960 1280 4
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 81920 1280 4
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 24576 768 4
  done
done
} > $O/kbench.txt 2>&1
cat $O/kbench.txt

#!/bin/bash
# round 6: the tile-form 256x256 GEMM's row x row K loop on v_mfma_f32_16x16x32_bf16 (G2_M16) against the 32x32x16 loop, stand-alone:
#   gemm_a3_m16 / gemm_a3_m32   decoder form (three-deep A ring, one-step bf16 epilogue)   [65536 x 1280], K = 40960 and [65536 x 768], K = 24576
#   gemm_row_m16 / gemm_row_m32 plain row x row (two stages, fp32 two-pass epilogue)        K = 1280 / 768 shapes
# mode 3 = bitwise comparison against the 4-wave reference kernel (tools/kbench/gemm256w4.h)
O=gpurun_out/r06_tile_m16; mkdir -p $O
{
for b in gemm_a3_m16 gemm_a3_m32 gemm_row_m16 gemm_row_m32; do
  echo -n "$b compare K=1280: "; timeout 120 build/kbench/$b 4096 4096 1280 3
  echo -n "$b compare K=4096: "; timeout 120 build/kbench/$b 2048 1280 4096 3
  echo -n "$b compare K=128:  "; timeout 120 build/kbench/$b 1024 1024 128 3      # (the 4-wave reference walks K tiles in pairs: even counts only)
done
for rep in 1 2 3; do
  for b in gemm_a3_m32 gemm_a3_m16; do
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 1280 40960 0
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 768 24576 0
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 1280 81920 0
  done
  for b in gemm_row_m32 gemm_row_m16; do
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 40960 1280 0
    echo -n "$b: "; timeout 120 build/kbench/$b 65536 24576 768 0
  done
done
} > $O/kbench.txt 2>&1
cat $O/kbench.txt

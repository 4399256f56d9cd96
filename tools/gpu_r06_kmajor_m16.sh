#!/bin/bash
# round 6: the k-major x k-major K loop (weight gradient, half ring) on v_mfma_f32_16x16x32_bf16 (-DG2_M16K=1: gemm_k16) against the
# 32x32x16 loop (gemm_k32), stand-alone: FNV hash of the fp32 slabs (KB_HASH: the two builds must print the same value), then timing
O=gpurun_out/r06_kmajor_m16; mkdir -p $O
{
for shp in "1280 4096 8192 1 1" "1280 4096 8192 1 3" "768 2048 4160 1 5" "256 256 64 1 1" "256 512 192 1 2"; do
  for b in gemm_k32 gemm_k16; do echo -n "$b [$shp]: "; KB_HASH=1 timeout 120 build/kbench/$b $shp | tr '\n' ' '; echo; done
done
for rep in 1 2 3; do
  for b in gemm_k32 gemm_k16; do
    echo -n "$b: "; timeout 120 build/kbench/$b 1280 40960 131072 1 5
    echo -n "$b: "; timeout 120 build/kbench/$b 1280 81920 131072 1 3
    echo -n "$b: "; timeout 120 build/kbench/$b 768 24576 65536 1 3
  done
done
} > $O/kbench.txt 2>&1
cat $O/kbench.txt

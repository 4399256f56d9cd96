#!/bin/bash
# round 6: the streaming GEMM stand-alone on RANDOM dense operands (tools/kbench/gemm_bench.hip): mode 5 = bit-compare against the tile form,
# mode 4 = timing.  gemm_s32 = static-stage loop on 32x32x16, gemm_s16 = the same on 16x16x32 (G2S_M16), gemm_sdyn = round 5's loop
O=gpurun_out/r06_kbench16; mkdir -p $O
{
for b in gemm_s16 gemm_s32; do
  echo "== $b: bit-compare stream vs tile form"
  build/kbench/$b 8192 8192 1280 5
  build/kbench/$b 4096 12288 768 5
  build/kbench/$b 2048 1024 128 5
done
for rep in 1 2 3; do
  for b in gemm_sdyn gemm_s32 gemm_s16; do
    echo -n "$b: "; build/kbench/$b 65536 40960 1280 4
    echo -n "$b: "; build/kbench/$b 65536 24576 768 4
  done
done
} > $O/kbench_m16.txt 2>&1
cat $O/kbench_m16.txt

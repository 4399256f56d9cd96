#!/bin/bash
# round 6: 16x16x32 timing PROXY (results wrong) on the TILE-FORM 256x256 GEMM, stand-alone on random dense operands:
#   weight-gradient shape  [1280 x 40960] = A^T B, K = 131072 rows (k-major x k-major, half ring, 5-way split-K)   gemm_t32 / gemm_t16, mode 1
#   decoder shape          [65536 x 1280], K = 40960 (row x row, three-deep A ring)                                 gemm_a3_32 / gemm_a3_16, mode 0
O=gpurun_out/r06_kbench_tile16; mkdir -p $O
{
for rep in 1 2 3; do
  for b in gemm_t32 gemm_t16; do echo -n "$b: "; build/kbench/$b 1280 40960 131072 1 5; done
  for b in gemm_a3_32 gemm_a3_16; do echo -n "$b: "; build/kbench/$b 65536 1280 40960 0; done
done
} > $O/kbench.txt 2>&1
cat $O/kbench.txt

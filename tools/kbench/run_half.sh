#!/bin/bash
# A/B of the stage hand-over of the 256x256 GEMM: whole stages (-DG2_HALF_ROW=0 / -DG2_HALF_KMAJOR=0 builds) vs k halves.
#   build/kbench/gemm_hrow0, gemm_hrow1 [, gemm_hrow1_stamp = -DG2X_STAMP -DG2X_WAITSTAMP]
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/kbench_half_row.txt
: > $out
for v in 0 1; do
  echo "== compare hrow$v vs w4" >> $out
  timeout 120 build/kbench/gemm_hrow$v 4096 4096 1280 3 >> $out 2>&1
  timeout 120 build/kbench/gemm_hrow$v 512 256 64 3 >> $out 2>&1
  timeout 120 build/kbench/gemm_hrow$v 512 512 192 3 >> $out 2>&1
done
for rep in 1 2; do
for v in 0 1; do
  echo "== hrow$v rep $rep" >> $out
  timeout 120 build/kbench/gemm_hrow$v 65536 40960 1280 0 >> $out 2>&1
  timeout 120 build/kbench/gemm_hrow$v 65536 1280 40960 0 >> $out 2>&1
  timeout 120 build/kbench/gemm_hrow$v 65536 24576 768 0 >> $out 2>&1
  timeout 120 build/kbench/gemm_hrow$v 65536 768 24576 0 >> $out 2>&1
  timeout 120 build/kbench/gemm_hrow$v 65536 81920 1280 0 >> $out 2>&1
done
done
echo "== stamps hrow1" >> $out
timeout 120 build/kbench/gemm_hrow1_stamp 65536 1280 40960 0 >> $out 2>&1
timeout 120 build/kbench/gemm_hrow1_stamp 65536 40960 1280 0 >> $out 2>&1
cat $out

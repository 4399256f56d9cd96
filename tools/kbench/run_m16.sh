#!/bin/bash
# A/B of the two MFMA shapes of the 256x256 GEMM (build/kbench/gemm_bench_m0 = 32x32x16, _m1 = 16x16x32)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/kbench_m16.txt
: > $out
for v in 1 0; do
  b=build/kbench/gemm_bench_m$v
  echo "== compare m$v vs w4" >> $out
  timeout 120 $b 4096 4096 1280 3 >> $out 2>&1
done
for rep in 1 2; do
for v in 0 1; do
  b=build/kbench/gemm_bench_m$v
  echo "== m$v rep $rep" >> $out
  timeout 120 $b 65536 40960 1280 0 >> $out 2>&1
  timeout 120 $b 65536 1280 40960 0 >> $out 2>&1
  timeout 120 $b 65536 24576 768 0 >> $out 2>&1
  timeout 120 $b 65536 768 24576 0 >> $out 2>&1
  timeout 120 $b 40960 1280 65536 1 2 >> $out 2>&1
  timeout 120 $b 24576 768 65536 1 3 >> $out 2>&1
done
done
cat $out

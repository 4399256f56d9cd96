#!/bin/bash
# A/B of the three-deep A ring of the row-major 256x256 GEMM: build/kbench/gemm_cur (two whole stages) vs gemm_a3 (-DKB_A3)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/kbench_a3.txt
: > $out
for b in gemm_cur gemm_a3; do
  echo "== compare $b vs w4" >> $out
  timeout 120 build/kbench/$b 4096 4096 1280 3 >> $out 2>&1
  timeout 120 build/kbench/$b 2048 1024 8192 3 >> $out 2>&1
done
for rep in 1 2; do
for b in gemm_cur gemm_a3; do
  echo "== $b rep $rep" >> $out
  timeout 120 build/kbench/$b 65536 1280 40960 0 >> $out 2>&1
  timeout 120 build/kbench/$b 65536 1280 81920 0 >> $out 2>&1
  timeout 120 build/kbench/$b 65536 768 24576 0 >> $out 2>&1
  timeout 120 build/kbench/$b 65536 40960 1280 0 >> $out 2>&1
  timeout 120 build/kbench/$b 65536 24576 768 0 >> $out 2>&1
done
done
echo "== stamps a3" >> $out
timeout 120 build/kbench/gemm_a3_stamp 65536 1280 40960 0 >> $out 2>&1
cat $out

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/kbench_a3_kmajor.txt
: > $out
for b in gemm_cur gemm_a3; do
  echo "== hash $b" >> $out
  KB_HASH=1 timeout 120 build/kbench/$b 1024 512 4096 1 2 >> $out 2>&1
  KB_HASH=1 timeout 120 build/kbench/$b 512 256 320 1 5 >> $out 2>&1
done
for rep in 1 2; do
for b in gemm_cur gemm_a3; do
  echo "== $b rep $rep" >> $out
  timeout 120 build/kbench/$b 40960 1280 65536 1 2 >> $out 2>&1
  timeout 120 build/kbench/$b 24576 768 65536 1 3 >> $out 2>&1
  timeout 120 build/kbench/$b 81920 1280 65536 1 1 >> $out 2>&1
done
done
echo "== stamps a3" >> $out
timeout 120 build/kbench/gemm_a3_stamp 40960 1280 65536 1 2 >> $out 2>&1
cat $out

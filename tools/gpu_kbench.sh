#!/bin/bash
build/kbench/gemm_m32 512 512 256 3
build/kbench/gemm_m32 4096 4096 1280 3
for v in m32 m32_NOGLOAD m32_NOSTAGE; do
  echo "== $v"
  build/kbench/gemm_$v 65536 40960 1280 2
  build/kbench/gemm_$v 65536 1280 40960 2
done
build/kbench/gemm_m32 65536 24576 768 2

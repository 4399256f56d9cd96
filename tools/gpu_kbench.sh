#!/bin/bash
# 8-wave 256x256 GEMM: committed K loop against the variants named in KB_VARIANTS (build/kbench/gemm_<name>)
for v in bench ${KB_VARIANTS:-stagger}; do
  echo "== $v"
  build/kbench/gemm_$v 65536 40960 1280 0
  build/kbench/gemm_$v 65536 1280 40960 0
  build/kbench/gemm_$v 65536 24576 768 0
  build/kbench/gemm_$v 1280 40960 131072 1 8
done

#!/bin/bash
for v in ${KB_VARIANTS:-v2}; do
  echo "== $v"
  build/kbench/gemm_$v 65536 40960 1280
  build/kbench/gemm_$v 65536 24576 768
  build/kbench/gemm_$v 1280 40960 131072 1 8
done

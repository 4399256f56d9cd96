#!/bin/bash
# round 5 experiment: the weight-gradient GEMM shape with its d-side operand K-contiguous (row-major x k-major) against k-major x k-major
O=gpurun_out/r05_mix; mkdir -p $O
{
for rep in 1 2; do
  echo "== rep $rep (M=1280 N=40960 K=131072 splits=5)"
  echo -n "kmajor x kmajor, half ring:        "; build/kbench/gemm_mix 1280 40960 131072 1 5
  echo -n "kmajor x kmajor, whole stages:     "; build/kbench/gemm_mix_nohalf 1280 40960 131072 1 5
  echo -n "row x kmajor, whole stages:        "; build/kbench/gemm_mix 1280 40960 131072 6 5
  echo -n "row x kmajor, three-deep A ring:   "; build/kbench/gemm_mix_a3 1280 40960 131072 6 5
done
} > $O/mix.txt 2>&1
cat $O/mix.txt

#!/bin/bash
# A/B timing of two engine builds on the same GPU box (interleaved runs): $1 = alternate .so
for i in 1 2 3; do
  for lib in "" "$1"; do
    echo -n "${lib:-current} "
    FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --steps 200 --warmup 20 --breakdown 2>&1 | grep -E "per-kernel" | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*/fwd \1 bwd \2/'
  done
done

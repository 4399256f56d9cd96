#!/bin/bash
export FREUD_SAE_ALLOW_OLD_LIB=1      # freud_amd/engine.py: an older build may lack entry points of the current header
# same-box A/B of the fused forward: in-tree library against the given alternates (level-2 per-kernel HIP events)
for i in 1 2; do
  for lib in "" "$@"; do
    echo -n "${lib:-current} "
    FREUD_SAE_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*"reduce_grads": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/fwd \1 bwd \2 reduce \3 step \4/'
    echo
  done
done

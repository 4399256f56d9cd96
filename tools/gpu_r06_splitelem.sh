#!/bin/bash
# round 6 experiment: the forward's MFMA order D E E D instead of D E D E (-DFF2_EEDD=1): does the encoder's accumulation chain stall on its own
# predecessor when an independent MFMA sits between two of its links?  Bitwise test of the launch forms, then same-box A/B
O=gpurun_out/r06_splitelem; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
FREUD_SAE_LIB=build/ab/libfreud_sae_splitelem.so timeout 600 python -m pytest tests/test_engine_gpu.py -q -x -m gpu -k "launch_forms or golden or masked" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for i in 1 2 3; do for lib in "" build/ab/libfreud_sae_splitelem.so; do for args in "" "--data normal"; do
  echo -n "${lib:-current} [$args] " >> $O/ab.txt
  FREUD_SAE_LIB=$lib python3 bench.py --no-cpu-baseline --no-pcie-sample --steps 200 --warmup 20 --breakdown $args 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*"reduce_grads": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/fwd \1 bwd \2 reduce \3 step \4/' >> $O/ab.txt
  echo >> $O/ab.txt
done; done; done; cat $O/ab.txt
FREUD_SAE_LIB=build/ab/libfreud_sae_splitelem.so python3 bench.py --no-cpu-baseline --no-pcie-sample --dbg 65 --steps 100 --warmup 20 2>&1 | grep "^fwd"

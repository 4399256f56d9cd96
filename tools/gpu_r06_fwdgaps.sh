#!/bin/bash
# round 6: the forward's gaps rebalanced -- current = FF2_SPLIT_ELEM (a pair's latent arithmetic over four gaps), dmaspread = that + the six
# LDS-DMA pieces one per gap, four gaps apart, splitoff = neither (the build of two hours ago), r5loop = round 5's loops.  Tests first.
O=gpurun_out/r06_fwdgaps; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
for lib in "" build/ab/libfreud_sae_dmaspread.so; do
  FREUD_SAE_LIB=$lib timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_trajectory_gpu.py -q -x -m gpu -k "not c4 and not 1280" > $O/tests_$(basename ${lib:-current}).txt 2>&1; tail -2 $O/tests_$(basename ${lib:-current}).txt
done
for i in 1 2 3; do for lib in "" build/ab/libfreud_sae_dmaspread.so build/ab/libfreud_sae_splitoff.so build/ab/libfreud_sae_r5loop.so; do for args in "" "--data normal"; do
  echo -n "${lib:-current} [$args] " >> $O/ab.txt
  FREUD_SAE_LIB=$lib python3 bench.py --no-cpu-baseline --no-pcie-sample --steps 200 --warmup 20 --breakdown $args 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*"reduce_grads": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/fwd \1 bwd \2 reduce \3 step \4/' >> $O/ab.txt
  echo >> $O/ab.txt
done; done; done; cat $O/ab.txt
for lib in "" build/ab/libfreud_sae_dmaspread.so; do echo "== ${lib:-current}"; FREUD_SAE_LIB=$lib python3 bench.py --no-cpu-baseline --no-pcie-sample --dbg 65 --steps 100 --warmup 20 2>&1 | grep "^fwd"; done | tee $O/stamps.txt

#!/bin/bash
# round 6: backward LDS-DMA as 14 single pieces in gaps 1, 3, ..., 27 (-DBF_DMA_SINGLE=1) against 7 pairs in gaps 1, 5, ..., 25; then the
# profile round on the final build
O=gpurun_out/r06_bwdsingle; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
FREUD_SAE_LIB=build/ab/libfreud_sae_bwdsingle.so timeout 600 python -m pytest tests/test_engine_gpu.py -q -x -m gpu -k "golden or launch_forms or masked" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for i in 1 2 3; do for lib in "" build/ab/libfreud_sae_bwdsingle.so; do for args in "" "--data normal"; do
  echo -n "${lib:-current} [$args] " >> $O/ab.txt
  FREUD_SAE_LIB=$lib python3 bench.py --no-cpu-baseline --no-pcie-sample --steps 200 --warmup 20 --breakdown $args 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*"reduce_grads": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/fwd \1 bwd \2 reduce \3 step \4/' >> $O/ab.txt
  echo >> $O/ab.txt
done; done; done; cat $O/ab.txt
FREUD_SAE_LIB=build/ab/libfreud_sae_bwdsingle.so python3 bench.py --no-cpu-baseline --no-pcie-sample --dbg 66 --steps 100 --warmup 20 2>&1 | grep "^bwd"
python3 bench.py --no-cpu-baseline --no-pcie-sample --dbg 66 --steps 100 --warmup 20 2>&1 | grep "^bwd"
unset FREUD_SAE_ALLOW_OLD_LIB
bash tools/profile_round.sh r06b > gpurun_out/prof_r06b.txt 2>&1; tail -3 gpurun_out/prof_r06b.txt

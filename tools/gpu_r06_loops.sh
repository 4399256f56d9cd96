#!/bin/bash
# round 6: the issue-port work on the two fused loops.  Correctness first (the engine / launch-form / resume tests on the new build), then
# same-box A/B, three alternations, on the headline batch, on rotating batches and on N(0,1):
#   current            = new forward loop (L1 adds kept in their gaps, pair rounding, paired fragment requests, M0 clobbered) + new backward
#                        loop (static stages, interleaved LDS layout, running row pointers)
#   fwdold / bwdold    = one of the two loops as in round 5
#   r5loop             = both as in round 5
mkdir -p gpurun_out/r06_loops2
O=gpurun_out/r06_loops2
export FREUD_SAE_ALLOW_OLD_LIB=1
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_resume_gpu.py tests/test_trajectory_gpu.py -q -x -m gpu > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt
tail -3 $O/tests.txt
B="python3 bench.py --no-cpu-baseline --no-pcie-sample"
for i in 1 2 3; do
  for lib in "" build/ab/libfreud_sae_gatesame.so build/ab/libfreud_sae_r5loop.so; do
    for args in "" "--rotate 4" "--data normal"; do
      echo -n "${lib:-current} [$args] " >> $O/ab.txt
      FREUD_SAE_LIB=$lib $B --steps 200 --warmup 20 --breakdown $args 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*"reduce_grads": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/fwd \1 bwd \2 reduce \3 step \4/' >> $O/ab.txt
      echo >> $O/ab.txt
    done
  done
done
cat $O/ab.txt
for lib in "" build/ab/libfreud_sae_r5loop.so; do
  FREUD_SAE_LIB=$lib $B --dbg 65 --steps 100 --warmup 20 2>&1 | grep -E "^fwd" >> $O/stamps_$(basename ${lib:-current}).txt
  FREUD_SAE_LIB=$lib $B --dbg 66 --steps 100 --warmup 20 2>&1 | grep -E "^bwd" >> $O/stamps_$(basename ${lib:-current}).txt
done
cat $O/stamps_*.txt
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-sample > $O/driver_style_$i.json 2>/dev/null; done
python3 -c "
import json
for i in (1,2):
    d=json.load(open('$O/driver_style_%d.json'%i)); print('driver-style', d['ms_per_step'], d['roofline']['frac'], d['step_mfma_frac'])
"

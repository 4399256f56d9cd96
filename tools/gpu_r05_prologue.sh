#!/bin/bash
# round 5: prologues -- fused forward: first biases by LDS-DMA (was a load + vmcnt(0) + LDS write in wave 0 behind its x loads; biasv1 = that);
# fused backward: table entry + count partials loaded together (epipk = the build before, three latencies in a row)
O=gpurun_out/r05_prologue; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py tests/test_resume_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
for i in 1 2 3; do
  for lib in current build/ab/libfreud_sae_biasv1.so build/ab/libfreud_sae_epipk.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo "== $lib"; python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 65 2>&1 | grep -E "^fwd (per-workgroup|prologue)"
    python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 66 2>&1 | grep -E "^bwd "
  done
done > $O/stamps.txt 2>&1
unset FREUD_SAE_LIB
bash tools/ab_fwd.sh build/ab/libfreud_sae_epipk.so > $O/ab_fwd.txt 2>&1
tail -3 $O/tests.txt; cat $O/stamps.txt $O/ab_fwd.txt

#!/bin/bash
# round 5: fused forward -- dx_hat publication with eight LDS reads in flight per batch (current) against one read-store pair at a time (prepub)
O=gpurun_out/r05_pub; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py tests/test_resume_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
for i in 1 2 3; do
  for lib in current build/ab/libfreud_sae_prepub.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo "== $lib"; python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 65 2>&1 | grep -E "^fwd (per-workgroup|epilogue)"
  done
done > $O/stamps.txt 2>&1
tail -3 $O/tests.txt; cat $O/stamps.txt

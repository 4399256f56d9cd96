#!/bin/bash
# round 5: the fused forward's epilogue fills its x staging from the fragment registers instead of reloading x -- tests, stamps, A/B
O=gpurun_out/r05_tail; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
for i in 1 2; do
  for lib in current build/ab/libfreud_sae_tailv1.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo "== $lib"; python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 65 2>&1 | grep -E "fwd per-workgroup|fwd in-kernel"
  done
done > $O/stamps.txt 2>&1
unset FREUD_SAE_LIB
bash tools/ab_fwd.sh build/ab/libfreud_sae_tailv1.so > $O/ab_fwd.txt 2>&1
tail -3 $O/tests.txt; cat $O/stamps.txt $O/ab_fwd.txt

#!/bin/bash
# round 5: the fused forward's -1.0 test moved to the staging of x (branch-free residual arithmetic) against the per-group test (mtestv4 lib):
# tests (masked entries included), stamps, same-box A/B
O=gpurun_out/r05_mtest; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py tests/test_resume_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
for i in 1 2; do
  for lib in current build/ab/libfreud_sae_mtestv4.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo "== $lib"; python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 65 2>&1 | grep -E "fwd per-workgroup|fwd in-kernel|fwd epilogue"
  done
done > $O/stamps.txt 2>&1
unset FREUD_SAE_LIB
bash tools/ab_fwd.sh build/ab/libfreud_sae_mtestv4.so > $O/ab_fwd.txt 2>&1
for i in 1 2 3; do
  for lib in current build/ab/libfreud_sae_mtestv4.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo -n "$lib: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
  done
done > $O/driver_style.txt 2>&1
tail -3 $O/tests.txt; cat $O/stamps.txt $O/ab_fwd.txt $O/driver_style.txt

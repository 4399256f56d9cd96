#!/bin/bash
# round 5: fused forward -- mask scan while x is staged + branch-free residual arithmetic (current; epipk = its packed form) against the
# per-group test of round 4 (mtestv4), on the bench's data with the loader's -1.0 guard (default) and without it (--raw-cast)
O=gpurun_out/r05_prescan; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py tests/test_resume_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
for data in "" "--raw-cast"; do
for i in 1 2; do
  for lib in current build/ab/libfreud_sae_epipk.so build/ab/libfreud_sae_mtestv4.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo "== $lib $data"; python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 65 $data 2>&1 | grep -E "fwd per-workgroup|fwd in-kernel|fwd epilogue"
  done
done
done > $O/stamps.txt 2>&1
unset FREUD_SAE_LIB
bash tools/ab_fwd.sh build/ab/libfreud_sae_epipk.so build/ab/libfreud_sae_mtestv4.so > $O/ab_fwd.txt 2>&1
for i in 1 2 3; do
  for lib in current build/ab/libfreud_sae_epipk.so build/ab/libfreud_sae_mtestv4.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo -n "$lib: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
  done
done > $O/driver_style.txt 2>&1
unset FREUD_SAE_LIB
echo -n "current raw-cast: " >> $O/driver_style.txt; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --raw-cast 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'])" >> $O/driver_style.txt
tail -3 $O/tests.txt; cat $O/stamps.txt $O/ab_fwd.txt $O/driver_style.txt

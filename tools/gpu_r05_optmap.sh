#!/bin/bash
# round 5: optimizer_l1_cols_kernel with XCD-major column blocks (current) against column block = blockIdx (preoptmap)
O=gpurun_out/r05_optmap; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py tests/test_resume_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
for i in 1 2 3; do
  for lib in current build/ab/libfreud_sae_preoptmap.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo -n "$lib: "; python bench.py --no-cpu-baseline --steps 200 --warmup 20 --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*per-kernel ms[^{]*\({[^}]*}\).*"ms_per_step": \([0-9.]*\).*/\1 step \2/'; echo
  done
done > $O/ab.txt 2>&1
unset FREUD_SAE_LIB
for i in 1 2 3; do
  for lib in current build/ab/libfreud_sae_preoptmap.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo -n "$lib driver-style: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
done >> $O/ab.txt 2>&1
tail -3 $O/tests.txt; cat $O/ab.txt

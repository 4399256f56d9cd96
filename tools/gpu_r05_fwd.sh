#!/bin/bash
# round 5: packed latent arithmetic of fwd_fused2 -- tests, same-box A/B against the element-wise build
O=gpurun_out/r05_fwd; mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
bash tools/ab_fwd.sh build/ab/libfreud_sae_ff2elem.so > $O/ab_fwd.txt 2>&1
for i in 1 2 3; do
  for lib in current build/ab/libfreud_sae_ff2elem.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo -n "$lib: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
  done
done > $O/driver_style.txt 2>&1
unset FREUD_SAE_LIB
tail -3 $O/tests.txt; cat $O/ab_fwd.txt $O/driver_style.txt

#!/bin/bash
# round 4: 16-byte epilogue stores (eight columns per thread) in the fp8 encoder's epilogue: fp8 tests, then C5-fp8 A/B
set -u
O=gpurun_out/r04_wide8; mkdir -p $O
timeout 1500 python -m pytest tests/test_fp8_gpu.py -x -q -m gpu > $O/pytest_fp8.txt 2>&1; tail -3 $O/pytest_fp8.txt
for i in 1 2; do
  for lib in "" build/ab/libfreud_sae_narrow.so; do
    echo -n "[C5 fp8 ${lib:-wide8}] "
    FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 10 --warmup 2 --precision fp8 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read(); k=json.loads(re.search(r'\{.*?\}', t[t.index('per-kernel'):]).group(0)); m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1)
print('step', m, {x:k[x] for x in k if k[x]>0.5})"
  done
done > $O/ab_wide8_c5fp8.txt 2>&1; cat $O/ab_wide8_c5fp8.txt

#!/bin/bash
# round 4: 16-byte epilogue stores (eight columns per thread) in the 256x256 GEMM's bf16 epilogue: tests, then C4 / C3-shaped A/B
set -u
O=gpurun_out/r04_wide8; mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_fp8_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for i in 1 2 3; do
  for lib in "" build/ab/libfreud_sae_narrow.so; do
    echo -n "[C4 ${lib:-wide8}] "
    FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3 --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read(); k=json.loads(re.search(r'\{.*?\}', t[t.index('per-kernel'):]).group(0)); m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1)
print('step', m, {x:k[x] for x in k if k[x]>0.3})"
  done
done > $O/ab_wide8_c4.txt 2>&1; cat $O/ab_wide8_c4.txt

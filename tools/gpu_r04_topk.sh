#!/bin/bash
# round 4: AuxK select with the histogram-based k-th largest -- TopK test suite, then C3 with 31 % dead latents: current (histogram,
# occupancy 4) against occupancy 3 (no spills) and against round 3's binary search
set -u
O=gpurun_out/r04_topk; mkdir -p $O
timeout 1800 python -m pytest tests/test_topk_gpu.py -x -q -m gpu > $O/pytest_topk.txt 2>&1; tail -4 $O/pytest_topk.txt
for i in 1 2; do
  for lib in "" build/ab/libfreud_sae_occ3.so build/ab/libfreud_sae_binsearch.so; do
    echo -n "[C3 31% dead ${lib:-current}] "
    FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e5 --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read(); k=json.loads(re.search(r'\{.*?\}', t[t.index('per-kernel'):]).group(0)); m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1)
print('step', m, {x:k[x] for x in k if k[x]>0.25})"
  done
done > $O/ab_auxk_select.txt 2>&1; cat $O/ab_auxk_select.txt

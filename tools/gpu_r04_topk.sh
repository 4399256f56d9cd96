#!/bin/bash
# round 4: AuxK compact select -- radix select + candidate path for rows with few positive dead values: TopK suite, stamps, C3 with 31 % dead
set -u
O=gpurun_out/r04_topk2; mkdir -p $O
timeout 1800 python -m pytest tests/test_topk_gpu.py -x -q -m gpu > $O/pytest_topk.txt 2>&1; tail -4 $O/pytest_topk.txt
FREUD_SAE_LIB=build/ab/libfreud_sae_selstamp.so timeout 600 python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e5 --dbg 67 > /dev/null 2> $O/err.txt; grep "AuxK compact" $O/err.txt | tee $O/stamps.txt
for i in 1 2; do
  echo -n "[C3 31% dead] "
  python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e5 --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read(); k=json.loads(re.search(r'\{.*?\}', t[t.index('per-kernel'):]).group(0)); m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1); l=re.search(r'\"loss\": (\{.*?\})', t).group(1)
print('step', m, {x:k[x] for x in k if k[x]>0.25}, l)"
done > $O/c3_auxk.txt 2>&1; cat $O/c3_auxk.txt

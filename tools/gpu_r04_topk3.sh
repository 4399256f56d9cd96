#!/bin/bash
set -u
O=gpurun_out/r04_topk3; mkdir -p $O
timeout 1800 python -m pytest tests/test_topk_gpu.py -x -q -m gpu > $O/pytest_topk.txt 2>&1; tail -3 $O/pytest_topk.txt
for i in 1 2; do
 for dt in 1e15 1e5; do
  echo -n "[C3 dead-threshold $dt] "
  python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold $dt --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read(); k=json.loads(re.search(r'\{.*?\}', t[t.index('per-kernel'):]).group(0)); m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1); l=re.search(r'\"loss\": (\{.*?\})', t).group(1)
print('step', m, {x:k[x] for x in k if k[x]>0.25}, l)"
 done
done > $O/c3.txt 2>&1; cat $O/c3.txt

#!/bin/bash
# tile-driven TopK select: round 4's register / ballot form against round 3's LDS ranking (-DTSEL_V1), same box, interleaved
set -u
O=gpurun_out/r04_tsel; mkdir -p $O
timeout 1500 python -m pytest tests/test_topk_gpu.py -x -q -m gpu > $O/pytest_topk.txt 2>&1; tail -3 $O/pytest_topk.txt
run() {
  python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --breakdown "$@" 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read(); k=json.loads(re.search(r'\{.*?\}', t[t.index('per-kernel'):]).group(0)); m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1); l=re.search(r'\"loss\": (\{.*?\})', t).group(1)
print('step', m, {x:k[x] for x in k if k[x]>0.25}, l)"
}
for i in 1 2 3; do
  echo -n "[C3 new] "; run
  echo -n "[C3 TSEL_V1] "; FREUD_SAE_LIB=build/ab/libfreud_sae_tselv1.so run
  echo -n "[C3 ballot compaction] "; FREUD_SAE_LIB=build/ab/libfreud_sae_tselballot.so run
done > $O/ab_tsel_c3.txt 2>&1
for i in 1; do
  echo -n "[n=40960 d=1280 k=64 new] "; run --d 1280 --n 40960
  echo -n "[n=40960 d=1280 k=64 TSEL_V1] "; FREUD_SAE_LIB=build/ab/libfreud_sae_tselv1.so run --d 1280 --n 40960
done >> $O/ab_tsel_c3.txt 2>&1
cat $O/ab_tsel_c3.txt

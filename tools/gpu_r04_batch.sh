#!/bin/bash
# epilogue prefetch batch of the functors that read an [M x n] operand (EPI_BATCH_HEAVY): 8 (shipped) against 16, C4 + C3/AuxK, one box
set -u
O=gpurun_out/r04_batch; mkdir -p $O
run() { timeout 300 python bench.py --no-cpu-baseline "$@" --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read(); k=json.loads(re.search(r'\{.*?\}', t[t.index('per-kernel'):]).group(0)); m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1)
print('step', m, {x:k[x] for x in ('enc_fwd_gemm','dec_fwd_gemm','dpre_gemm','dw_gemm','topk_decode','topk_auxk_backward','fwd_bwd_total') if x in k and k[x] > 0})"; }
for i in 1 2 3; do
  echo -n "[C4 batch 8] "; run --d 1280 --n 40960 --steps 20 --warmup 3
  echo -n "[C4 batch 16] "; FREUD_SAE_LIB=build/ab/libfreud_sae_batch16.so run --d 1280 --n 40960 --steps 20 --warmup 3
done > $O/ab_batch.txt 2>&1
for i in 1 2; do
  echo -n "[C3+AuxK batch 8] "; run --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e5
  echo -n "[C3+AuxK batch 16] "; FREUD_SAE_LIB=build/ab/libfreud_sae_batch16.so run --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e5
done >> $O/ab_batch.txt 2>&1
cat $O/ab_batch.txt

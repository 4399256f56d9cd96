#!/bin/bash
# decoder epilogue with one vector load of x per call (EpiDec::vec_all) against four predicated scalar loads: C4 and C5-fp8, one box
set -u
O=gpurun_out/r04_dec; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_fp8_gpu.py -x -q -m gpu 2>&1 | tail -2
run() { timeout 300 python bench.py --no-cpu-baseline "$@" --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read(); k=json.loads(re.search(r'\{.*?\}', t[t.index('per-kernel'):]).group(0)); m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1)
print('step', m, {x:k[x] for x in ('enc_fwd_gemm','dec_fwd_gemm','dpre_gemm','dw_gemm','fwd_bwd_total') if x in k})"; }
for i in 1 2 3; do
  echo -n "[C4 vec_all] "; run --d 1280 --n 40960 --steps 20 --warmup 3
  echo -n "[C4 before] "; FREUD_SAE_LIB=build/ab/libfreud_sae_predec.so run --d 1280 --n 40960 --steps 20 --warmup 3
done > $O/ab_dec_c4.txt 2>&1
for i in 1 2; do
  echo -n "[C5 fp8 vec_all] "; run --d 1280 --n 81920 --steps 10 --warmup 2 --precision fp8
  echo -n "[C5 fp8 before] "; FREUD_SAE_LIB=build/ab/libfreud_sae_predec.so run --d 1280 --n 81920 --steps 10 --warmup 2 --precision fp8
done > $O/ab_dec_c5fp8.txt 2>&1
cat $O/ab_dec_c4.txt $O/ab_dec_c5fp8.txt

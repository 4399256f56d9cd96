#!/bin/bash
# round 5: the streaming form of the K = d GEMMs (gemm256s.h) -- bit-compare against the tile form, stand-alone timing, engine tests, C4 A/B
O=gpurun_out/r05_stream; mkdir -p $O
KB=build/kbench/gemm_bench_s
{
echo "== compare tile form vs streaming form (bf16 bits)"
$KB 8192 8192 1280 5
$KB 4096 12288 768 5
$KB 2048 1024 128 5
for rep in 1 2 3; do
  echo "== rep $rep"
  $KB 65536 40960 1280 0
  $KB 65536 40960 1280 4
  $KB 65536 24576 768 0
  $KB 65536 24576 768 4
done
} > $O/kbench.txt 2>&1
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -x -q > $O/engine_tests.txt 2>&1; echo "rc=$?" >> $O/engine_tests.txt
for rep in 1 2; do
  for st in 0 1; do
    echo -n "[C4 stream=$st] "
    FREUD_GEMM_STREAM=$st python bench.py --d 1280 --n 40960 --steps 10 --warmup 3 --no-cpu-baseline --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "
import sys,re,json
t=sys.stdin.read()
k=json.loads(re.search(r'level-2 profile\): (\{.*?\})', t).group(1))
m=re.search(r'\"ms_per_step\": ([0-9.]+)', t).group(1)
print('enc %.3f dec %.3f dpre %.3f dw %.3f step %s' % (k['enc_fwd_gemm'], k['dec_fwd_gemm'], k['dpre_gemm'], k['dw_gemm'], m))"
  done
done > $O/c4_ab.txt 2>&1
cat $O/kbench.txt; tail -3 $O/engine_tests.txt; cat $O/c4_ab.txt

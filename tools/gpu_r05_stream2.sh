#!/bin/bash
# round 5: streaming GEMMs, second pass -- zero-source first K step, packed epilogue math, priority experiment, stamps; TopK / fp8 forms
O=gpurun_out/r05_stream2; mkdir -p $O
{
echo "== compare tile form vs streaming form (bf16 bits)"
build/kbench/gemm_s 8192 8192 1280 5
build/kbench/gemm_sdg2s_prio1 4096 12288 768 5
echo "== stamps"
build/kbench/gemm_sdg2x_stamp 65536 40960 1280 4
build/kbench/gemm_sdg2s_prio1dg2x_stamp 65536 40960 1280 4
build/kbench/gemm_sdg2x_stamp 65536 24576 768 4
build/kbench/gemm_sdg2s_prio1dg2x_stamp 65536 24576 768 4
for rep in 1 2 3; do
  echo "== rep $rep"
  for b in gemm_s gemm_sdg2s_prio1; do
    echo -n "$b: "; build/kbench/$b 65536 40960 1280 4
    echo -n "$b: "; build/kbench/$b 65536 24576 768 4
  done
  echo -n "tile form: "; build/kbench/gemm_s 65536 40960 1280 0
done
} > $O/kbench.txt 2>&1
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_fp8_gpu.py tests/test_topk_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
parse='
import sys,re,json
t=sys.stdin.read()
k=json.loads(re.search(r"level-2 profile\): (\{.*?\})", t).group(1))
m=re.search(r"\"ms_per_step\": ([0-9.]+)", t).group(1)
print(" ".join("%s %.3f" % (n, v) for n, v in k.items() if v > 0.3 and n != "fwd_bwd_total"), "step", m)'
for rep in 1 2; do
  for st in 0 1; do
    echo -n "[C4 stream=$st] "; FREUD_GEMM_STREAM=$st python bench.py --d 1280 --n 40960 --steps 10 --warmup 3 --no-cpu-baseline --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
    echo -n "[C3 stream=$st] "; FREUD_GEMM_STREAM=$st python bench.py --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e15 --no-cpu-baseline --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
    echo -n "[C5fp8 stream=$st] "; FREUD_GEMM_STREAM=$st python bench.py --d 1280 --n 81920 --steps 6 --warmup 2 --precision fp8 --no-cpu-baseline 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
  done
done > $O/ab.txt 2>&1
cat $O/kbench.txt; tail -3 $O/tests.txt; cat $O/ab.txt

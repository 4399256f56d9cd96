#!/bin/bash
O=gpurun_out/r05_aux2; mkdir -p $O
timeout 1800 python -m pytest tests/test_topk_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
parse='
import sys,re,json
t=sys.stdin.read()
k=json.loads(re.search(r"level-2 profile\): (\{.*?\})", t).group(1))
m=re.search(r"\"ms_per_step\": ([0-9.]+)", t).group(1)
print(" ".join("%s %.3f" % (n, v) for n, v in k.items() if v > 0.25 and n != "fwd_bwd_total"), "step", m)'
{
for dl in 300 4000 14000; do
  echo -n "[TopK d1280 n40960 k32, $dl dead] "; python bench.py --variant topk --d 1280 --n 40960 --k 32 --steps 20 --warmup 5 --dead-threshold 1e15 --dead-latents $dl --no-cpu-baseline --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
done
} > $O/bench.txt 2>&1
tail -4 $O/tests.txt; cat $O/bench.txt

#!/bin/bash
# round 5: weight-gradient GEMM with the d-side operand transposed (row-major x k-major): tests, then same-box A/B at C4 / C5-fp8
O=gpurun_out/r05_dwrow; mkdir -p $O
timeout 2400 python -m pytest tests/test_engine_gpu.py tests/test_fp8_gpu.py "tests/test_dp_gpu.py::test_in_engine_rccl_single_rank_equals_plain_step" "tests/test_dp_gpu.py::test_l1_two_halves_sum_to_whole_batch" "tests/test_dp_gpu.py::test_two_processes_one_gpu_train_like_one_process" -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
parse='
import sys,re,json
t=sys.stdin.read()
k=json.loads(re.search(r"level-2 profile\): (\{.*?\})", t).group(1))
m=re.search(r"\"ms_per_step\": ([0-9.]+)", t).group(1)
print(" ".join("%s %.3f" % (n, v) for n, v in k.items() if v > 0.05 and n != "fwd_bwd_total"), "step", m)'
for rep in 1 2; do
  for ra in 0 1; do
    echo -n "[C4 rowA=$ra] "; FREUD_DW_ROWA=$ra python bench.py --d 1280 --n 40960 --steps 10 --warmup 3 --no-cpu-baseline --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
    echo -n "[C5fp8 rowA=$ra] "; FREUD_DW_ROWA=$ra python bench.py --d 1280 --n 81920 --steps 6 --warmup 2 --precision fp8 --no-cpu-baseline 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
    echo -n "[C4 rowA=$ra one-rank p2p] "; FREUD_DW_ROWA=$ra python bench.py --d 1280 --n 40960 --steps 10 --warmup 3 --no-cpu-baseline --force-dist 2>&1 | grep -E "ms_per_step" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['dp_timing']['plain_ms_per_step'])"
  done
done > $O/ab.txt 2>&1
tail -3 $O/tests.txt; cat $O/ab.txt

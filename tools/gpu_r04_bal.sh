#!/bin/bash
# round 4: the balanced fused backward (bwd_fused.h) -- correctness (engine / train / dp suites), then same-box A/B against the
# uniform row ranges (--dbg 83) at C2 and at n = 12 288
set -u
O=gpurun_out/r04_bal; mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py -x -q -m gpu > $O/pytest_engine.txt 2>&1; tail -4 $O/pytest_engine.txt
for i in 1 2 3; do
  for dbg in 0 83; do
    echo -n "[C2 dbg=$dbg] "
    python bench.py --no-cpu-baseline --no-pcie-sample --steps 200 --warmup 20 --dbg $dbg --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*"reduce_grads": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/fwd \1 bwd \2 reduce \3 step \4/'
    echo
  done
done > $O/ab_balanced_c2.txt 2>&1; cat $O/ab_balanced_c2.txt
for i in 1 2; do
  for dbg in 0 83; do
    echo -n "[n=12288 dbg=$dbg] "
    python bench.py --no-cpu-baseline --no-pcie-sample --n 12288 --steps 100 --warmup 10 --dbg $dbg --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*"reduce_grads": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/fwd \1 bwd \2 reduce \3 step \4/'
    echo
  done
done > $O/ab_balanced_n12288.txt 2>&1; cat $O/ab_balanced_n12288.txt
python bench.py --no-cpu-baseline --no-pcie-sample --steps 20 --warmup 5 > $O/bench_driver_style.json 2>/dev/null; cat $O/bench_driver_style.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
timeout 1500 python -m pytest tests/test_dp_gpu.py -x -q -m gpu > $O/pytest_dp.txt 2>&1; tail -4 $O/pytest_dp.txt

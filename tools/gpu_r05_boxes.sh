#!/bin/bash
# round 5: where the final build stands on one more box: three driver-style runs (bench.py --steps 20 --warmup 5) and one default run of C2,
# per-kernel HIP events of a 200-step run
O=gpurun_out/r05_boxes; mkdir -p $O
tag=$(date +%s)
{
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style', round(d['ms_per_step'],4), 'bwd', round(d['roofline']['kernel_avg_ms'],4), 'frac', round(d['roofline']['frac'],3))"; done
python bench.py --no-cpu-baseline --steps 200 --warmup 20 --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*"reduce_grads": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/200 steps: fwd \1 bwd \2 reduce \3 step \4/'; echo
} > $O/box_$tag.txt 2>&1
cat $O/box_$tag.txt

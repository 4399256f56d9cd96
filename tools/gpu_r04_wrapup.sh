#!/bin/bash
# round 4 wrap-up on the final build: long-run sanity (L1 3000 steps, TopK 400 steps twice: bitwise equal), counter passes of the
# C4 / C5-fp8 workloads, three more driver-style C2 lines.  Every step under its own timeout.
set -u
O=gpurun_out/r04_wrapup; mkdir -p $O
timeout 300 python tools/longrun_sanity.py > $O/longrun_l1.txt 2>&1; tail -4 $O/longrun_l1.txt
timeout 600 python tools/longrun_topk.py > $O/longrun_topk.txt 2>&1; tail -6 $O/longrun_topk.txt
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/driver_style_$i.json; python3 -c "
import json; d=json.loads(open('$O/driver_style_$i.json').read()); print('driver-style', round(d['ms_per_step'],4), 'ms; fused backward', round(d['roofline']['kernel_avg_ms'],4), 'ms; whole step', round(d['step_mfma_frac'],3))"; done
timeout 900 bash tools/pmc_workload.sh c4 --d 1280 --n 40960 --steps 4 --warmup 2 2>&1 | tail -2
timeout 900 bash tools/pmc_workload.sh c5fp8 --d 1280 --n 81920 --steps 3 --warmup 1 --precision fp8 2>&1 | tail -2
for w in c4 c5fp8; do rm -rf gpurun_out/pmc_$w/pmc_FETCH_SIZE gpurun_out/pmc_$w/pmc_WRITE_SIZE gpurun_out/pmc_$w/pmc_SQ; done

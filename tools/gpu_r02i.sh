#!/bin/bash
set -u
O=$PWD/gpurun_out/r02i
mkdir -p $O
python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_a.json 2> $O/bench_a.err
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_b.err
python bench.py --no-cpu-baseline --steps 200 --warmup 20 --force-dist > $O/bench_fd.json 2> $O/bench_fd.err
for f in bench_a bench_driver_style bench_fd; do python -c "
import json
d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['step_mfma_frac'])"; done

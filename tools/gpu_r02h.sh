#!/bin/bash
set -u
O=$PWD/gpurun_out/r02h
mkdir -p $O
python -m pytest tests/test_topk_gpu.py tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -25
python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e15 --breakdown > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e5 --breakdown > $O/bench_c3_aux.json 2> $O/bench_c3_aux.err
grep -h "per-kernel" $O/*.err
for f in c3 c3_aux; do python -c "
import json
d=json.load(open('$O/bench_$f.json')); print('$f', d['ms_per_step'], d['value'], d['loss'])"; done

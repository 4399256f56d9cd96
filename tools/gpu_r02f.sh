#!/bin/bash
set -u
O=$PWD/gpurun_out/r02f
mkdir -p $O
python tools/fp8dbg.py 2>&1 | tail -30
python -m pytest tests/test_dp_gpu.py tests/test_train_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5
python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --steps 200 --warmup 20 --force-dist > $O/bench_forcedist.json 2> $O/bench_forcedist.err
python bench.py --no-cpu-baseline --steps 200 --warmup 20 --force-dist --dp-payload bfloat16 > $O/bench_forcedist_bf16.json 2> $O/bench_forcedist_bf16.err
for f in default forcedist forcedist_bf16; do python -c "
import json
d=json.load(open('$O/bench_$f.json')); print('$f', d['ms_per_step'], d['value'], d['loss'])"; done

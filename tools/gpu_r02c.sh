#!/bin/bash
set -u
O=gpurun_out/r02c
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
python bench.py --no-cpu-baseline --steps 100 --warmup 10 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 6 --warmup 2 --breakdown > $O/bench_c5_bf16.json 2> $O/bench_c5_bf16.err
python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 6 --warmup 2 --precision fp8 > $O/bench_c5_fp8.json 2> $O/bench_c5_fp8.err
grep -h "per-kernel" $O/*.err

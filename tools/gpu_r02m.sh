#!/bin/bash
set -u
O=$PWD/gpurun_out/r02m
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 10 --warmup 3 --breakdown > $O/c4.json 2> $O/c4.err
grep -h "per-kernel" $O/c4.err; python -c "
import json
d=json.load(open('$O/c4.json')); print('c4', d['ms_per_step'], d['value'], d['step_mfma_frac'])"

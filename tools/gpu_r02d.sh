#!/bin/bash
set -u
O=gpurun_out/r02d
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --steps 200 --warmup 20 --force-dist > $O/bench_forcedist.json 2> $O/bench_forcedist.err
python bench.py --no-cpu-baseline --steps 200 --warmup 20 --force-dist --dp-host > $O/bench_forcedist_host.json 2> $O/bench_forcedist_host.err
for f in default forcedist forcedist_host; do python -c "
import json
d=json.load(open('$O/bench_$f.json')); print('$f', d['ms_per_step'], d['value'])"; done

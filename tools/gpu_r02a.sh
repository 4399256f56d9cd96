#!/bin/bash
# round-2 GPU pass A: full -m gpu suite, default bench line, C3 / C4 shape bench lines, forced data-parallel path, profiles
set -u
O=gpurun_out/r02a
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1
tail -30 $O/pytest.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --force-dist > $O/bench_forcedist.json 2> $O/bench_forcedist.err
python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e15 --breakdown > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 10 --warmup 3 --breakdown > $O/bench_c4.json 2> $O/bench_c4.err
bash tools/profile_round.sh r02a > $O/profile.log 2>&1
cat $O/bench_default.json

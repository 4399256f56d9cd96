#!/bin/bash
set -u
O=$PWD/gpurun_out/r02j
mkdir -p $O
python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 6 --warmup 2 --breakdown > $O/c4.json 2> $O/c4.err
python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 6 --warmup 2 --breakdown --dbg 70 > $O/c4_nostore.json 2> $O/c4_nostore.err
grep -h "per-kernel" $O/*.err

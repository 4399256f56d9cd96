#!/bin/bash
set -u
O=gpurun_out/r02b
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
# bias-ring forward A/B at the headline shape (debug flag 67 forces the ring) and the default-expansion shape
python bench.py --no-cpu-baseline --steps 200 --warmup 20 --breakdown > $O/bench_noring.json 2> $O/bench_noring.err
python bench.py --no-cpu-baseline --steps 200 --warmup 20 --breakdown --dbg 67 > $O/bench_ring.json 2> $O/bench_ring.err
python bench.py --no-cpu-baseline --steps 100 --warmup 10 --n 12288 --breakdown > $O/bench_n12288.json 2> $O/bench_n12288.err
grep -h "per-kernel" $O/*.err

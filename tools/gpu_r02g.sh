#!/bin/bash
set -u
O=$PWD/gpurun_out/r02g
mkdir -p $O
python -m pytest tests/test_fp8_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
export TMPDIR=/tmp
cd /tmp
for mode in plain fd; do
  extra=""; [ $mode = fd ] && extra="--force-dist"
  rocprofv3 --kernel-trace -d $O/trace_$mode -o t --output-format csv -- python3 $OLDPWD/bench.py --no-cpu-baseline --steps 50 --warmup 10 $extra > $O/$mode.json 2> $O/$mode.err
  python3 $OLDPWD/tools/trace_timeline.py $(find $O/trace_$mode -name "*kernel_trace.csv" | head -1) 50 > $O/${mode}_timeline.txt
  head -14 $O/${mode}_timeline.txt
done

#!/bin/bash
set -u
O=$PWD/gpurun_out/r02e
mkdir -p $O
python -m pytest tests/test_fp8_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d $O/trace_fd -o t --output-format csv -- python3 $OLDPWD/bench.py --no-cpu-baseline --steps 50 --warmup 10 --force-dist > $O/fd.json 2> $O/fd.err
cd $OLDPWD
python tools/trace_timeline.py $(find $O/trace_fd -name "*kernel_trace.csv" | head -1) 50 > $O/fd_timeline.txt
cat $O/fd_timeline.txt

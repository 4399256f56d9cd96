#!/bin/bash
# vendor-BLAS measuring stick for the generic GEMM shapes + the kernel names it picked
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/blas; mkdir -p $O
timeout 600 python tools/blas_probe.py > $O/blas_c4.json 2> $O/blas_c4.err
timeout 600 python tools/blas_probe.py --d 768 --n 24576 > $O/blas_c3.json 2>> $O/blas_c4.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o blas --output-format csv -- python3 tools/blas_probe.py > $O/blas_rocprof.json 2> $O/blas_rocprof.err
rm -f $(find $O/stats -name "*kernel_trace.csv")
cat $O/blas_c4.json $O/blas_c3.json
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-260

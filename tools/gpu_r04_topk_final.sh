#!/bin/bash
# round 4, after the TopK select work: the whole GPU suite, then the C3 / C3 + AuxK / TopK d=1280 evidence refreshed on the final
# build (bench lines, rocprofv3 --stats, counter passes).  Every step under its own timeout.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out/r04_topk_final; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
B="python3 $ROOT/bench.py"
timeout 300 $B --no-cpu-baseline --variant topk --d 1280 --n 40960 --k 32 --steps 20 --warmup 5 --dead-threshold 1e15 --breakdown > "$OUT/bench_topk_d1280_n40960.json" 2> "$OUT/bench_topk_d1280.err"
timeout 300 $B --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 50 --warmup 5 --dead-threshold 1e15 --breakdown > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
timeout 300 $B --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 50 --warmup 5 --dead-threshold 1e5 --breakdown > "$OUT/bench_c3_auxk.json" 2>> "$OUT/bench_c3.err"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c3" -o stats --output-format csv -- $B --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --spinup 0.3 --dead-threshold 1e15 > "$OUT/bench_c3_under_rocprof.json" 2> "$OUT/stats_c3.log"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c3auxk" -o stats --output-format csv -- $B --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --spinup 0.3 --dead-threshold 1e5 > "$OUT/bench_c3auxk_under_rocprof.json" 2> "$OUT/stats_c3auxk.log"
for d in stats_c3 stats_c3auxk; do rm -f $(find "$OUT/$d" -name "*kernel_trace.csv"); done
cd $ROOT
timeout 900 bash tools/pmc_workload.sh c3 --variant topk --d 768 --n 24576 --k 64 --steps 6 --warmup 3 --dead-threshold 1e15 2>&1 | tail -2
timeout 900 bash tools/pmc_workload.sh c3auxk --variant topk --d 768 --n 24576 --k 64 --steps 6 --warmup 3 --dead-threshold 1e5 2>&1 | tail -2
for w in c3 c3auxk; do rm -rf gpurun_out/pmc_$w/pmc_FETCH_SIZE gpurun_out/pmc_$w/pmc_WRITE_SIZE gpurun_out/pmc_$w/pmc_SQ; done
for f in bench_c3.json bench_c3_auxk.json bench_topk_d1280_n40960.json; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), 'ms', d.get('kernel_ms'))"; done

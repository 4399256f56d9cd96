#!/bin/bash
# rocprofv3 evidence for one round, run ON the GPU box from the repo root (gpurun -- bash tools/profile_round.sh r02):
#   1. bench lines (un-profiled): default C2, forced data-parallel (peer exchange fp32 / bf16 payload, RCCL, host-driven),
#      C3 (TopK), C4, C5 bf16 and fp8 shapes                                  -> gpurun_out/prof_<tag>/bench_*.json
#   2. --kernel-trace --stats of the default bench (program directly after "--", a SHORT spin-up of 0.2 s and 3000 timed
#      steps: the --stats averages cover every launch of the run, and the ~300 launches on ramping clocks of the spin-up
#      must stay a small share of them for the CSV itself to agree with the HIP-event figure) + the timed-region
#      timeline / averages from the trace (tools/trace_timeline.py)          -> stats/, timeline.txt
#   3. short --kernel-trace --stats runs of the C3 / C4 / C5-fp8 shapes       -> stats_c3/, stats_c4/, stats_c5fp8/
#   4. PMC passes of the default bench, each in its own run (never combined with traces): FETCH_SIZE, WRITE_SIZE, SQ
# tools/parse_pmc.py turns the counter CSVs into profiles/<tag>_hbm_traffic.json afterwards.
set -u
TAG=${1:-r04}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
$B > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
$B --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_default_driver_style.json" 2>> "$OUT/bench_default.err"
$B --no-cpu-baseline --force-dist > "$OUT/bench_forcedist.json" 2> "$OUT/bench_forcedist.err"
$B --no-cpu-baseline --force-dist --dp-payload bfloat16 > "$OUT/bench_forcedist_bf16.json" 2>> "$OUT/bench_forcedist.err"
$B --no-cpu-baseline --force-dist --dp-host > "$OUT/bench_forcedist_host.json" 2>> "$OUT/bench_forcedist.err"
$B --no-cpu-baseline --force-dist --dp rccl > "$OUT/bench_forcedist_rccl.json" 2>> "$OUT/bench_forcedist.err"
$B --no-cpu-baseline --variant topk --d 1280 --n 40960 --k 32 --steps 20 --warmup 5 --dead-threshold 1e15 --breakdown > "$OUT/bench_topk_d1280_n40960.json" 2> "$OUT/bench_topk_d1280.err"
$B --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 50 --warmup 5 --dead-threshold 1e15 --breakdown > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
$B --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 50 --warmup 5 --dead-threshold 1e5 --breakdown > "$OUT/bench_c3_auxk.json" 2>> "$OUT/bench_c3.err"
$B --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3 --breakdown > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
$B --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3 --force-dist --breakdown > "$OUT/bench_c4_forcedist.json" 2>> "$OUT/bench_c4.err"
$B --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3 --force-dist --dp rccl --breakdown > "$OUT/bench_c4_forcedist_rccl.json" 2>> "$OUT/bench_c4.err"
$B --no-cpu-baseline --d 1280 --n 40960 --steps 20 --warmup 3 --force-dist --dp-host --breakdown > "$OUT/bench_c4_forcedist_host.json" 2>> "$OUT/bench_c4.err"
$B --no-cpu-baseline --d 1280 --n 81920 --steps 10 --warmup 2 --breakdown > "$OUT/bench_c5_bf16.json" 2> "$OUT/bench_c5.err"
$B --no-cpu-baseline --d 1280 --n 81920 --steps 10 --warmup 2 --precision fp8 > "$OUT/bench_c5_fp8.json" 2>> "$OUT/bench_c5.err"
$B --no-cpu-baseline --n 12288 --steps 100 --warmup 10 > "$OUT/bench_d384_n12288.json" 2> "$OUT/bench_n12288.err"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $B --no-cpu-baseline --steps 3000 --warmup 20 --spinup 0.2 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
python3 $ROOT/tools/trace_timeline.py $(find "$OUT/stats" -name "*kernel_trace.csv" | head -1) 3000 > "$OUT/timeline.txt" 2>&1
rm -f $(find "$OUT/stats" -name "*kernel_trace.csv")        # 20 MB of per-dispatch rows; the summary above is what is kept
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c3" -o stats --output-format csv -- $B --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --spinup 0.3 --dead-threshold 1e15 > "$OUT/bench_c3_under_rocprof.json" 2> "$OUT/stats_c3.log"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c3auxk" -o stats --output-format csv -- $B --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --spinup 0.3 --dead-threshold 1e5 > "$OUT/bench_c3auxk_under_rocprof.json" 2> "$OUT/stats_c3auxk.log"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c4" -o stats --output-format csv -- $B --no-cpu-baseline --d 1280 --n 40960 --steps 10 --warmup 2 --spinup 0.3 > "$OUT/bench_c4_under_rocprof.json" 2> "$OUT/stats_c4.log"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c5fp8" -o stats --output-format csv -- $B --no-cpu-baseline --d 1280 --n 81920 --steps 6 --warmup 2 --spinup 0.3 --precision fp8 > "$OUT/bench_c5fp8_under_rocprof.json" 2> "$OUT/stats_c5fp8.log"
for d in stats_c3 stats_c3auxk stats_c4 stats_c5fp8; do rm -f $(find "$OUT/$d" -name "*kernel_trace.csv"); done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d "$OUT/pmc_$c" -o pmc --output-format csv -- $B --no-cpu-baseline --steps 5 --warmup 2 --spinup 0 > /dev/null 2> "$OUT/pmc_$c.log"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d "$OUT/pmc_SQ" -o pmc --output-format csv -- $B --no-cpu-baseline --steps 5 --warmup 2 --spinup 0 > /dev/null 2> "$OUT/pmc_SQ.log"
cd $ROOT
python3 tools/parse_pmc.py "$OUT" "$OUT/$TAG" > "$OUT/parse_pmc.txt" 2>&1
ls "$OUT"

#!/bin/bash
# rocprofv3 evidence for one round, run ON the GPU box from the repo root (gpurun -- tools/profile_round.sh r01b):
#   1. --kernel-trace --stats of the default bench           -> gpurun_out/prof_<tag>/stats
#   2. PMC passes, each in its own run (never combined with traces): FETCH_SIZE, WRITE_SIZE, SQ counters
# tools/parse_pmc.py turns the counter CSVs into profiles/<tag>_hbm_traffic.json afterwards.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the program itself after "--" (no shell hop), default 1 s spin-up and 200 timed steps, so that the profiler's per-kernel
# averages are taken on sustained clocks like the HIP-event figures of the un-profiled bench line
BENCH="python3 $PWD/bench.py --no-cpu-baseline --steps 200 --warmup 20"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/pmc_$c" -o pmc --output-format csv -- python3 $OLDPWD/bench.py --no-cpu-baseline --steps 5 --warmup 2 --spinup 0 > /dev/null 2> "$OUT/pmc_$c.log"
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d "$OUT/pmc_SQ" -o pmc --output-format csv -- python3 $OLDPWD/bench.py --no-cpu-baseline --steps 5 --warmup 2 --spinup 0 > /dev/null 2> "$OUT/pmc_SQ.log"
find "$OUT" -name "*.csv" | head -20

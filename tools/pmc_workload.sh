#!/bin/bash
# rocprofv3 counter passes for ONE bench workload, each counter group in its own run (never combined with traces):
#   bash tools/pmc_workload.sh <name> <bench.py args...>        (run ON the GPU box from the repo root)
# -> gpurun_out/pmc_<name>/{pmc_FETCH_SIZE,pmc_WRITE_SIZE,pmc_SQ}/...counter_collection.csv ; tools/parse_pmc_all.py reads them
set -u
NAME=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_$NAME
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d "$OUT/pmc_$c" -o pmc --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --spinup 0 "$@" > /dev/null 2> "$OUT/pmc_$c.log"
done
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d "$OUT/pmc_SQ" -o pmc --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --spinup 0 "$@" > /dev/null 2> "$OUT/pmc_SQ.log"
cd $ROOT
python3 tools/parse_pmc_all.py "$OUT" "$OUT/hbm_traffic_$NAME.json" "$@"

#!/bin/bash
# SQ counters of the tile-driven TopK select (C3, no dead latents), current build: instruction mix and wait shares
set -u
ROOT=$PWD; O=$ROOT/gpurun_out/r04_tselpmc; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ARGS="--no-cpu-baseline --spinup 0 --variant topk --d 768 --n 24576 --k 64 --steps 6 --warmup 2 --dead-threshold 1e15"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
  -d $O/p1 -o pmc --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $O/p1.log
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES \
  -d $O/p2 -o pmc --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $O/p2.log
cd $ROOT
python3 - <<'PY' > $O/sq_tile_select.txt
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/r04_tselpmc/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(p, "no counter file"); continue
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if "select_tiles" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(agg):
        print("%-22s %14.0f per launch (%d launches)" % (k, agg[k] / n[k], n[k]))
PY
cat $O/sq_tile_select.txt; tail -3 $O/p1.log $O/p2.log | cut -c1-300

#!/bin/bash
# SQ counters of the TopK kernels at the C3 shape with AuxK active (run ON the GPU box from the repo root)
OUT=$PWD/gpurun_out/prof_topk
mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
  -d "$OUT/pmc_SQ" -o pmc --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 6 --warmup 6 --spinup 0 --dead-threshold 1e5 > /dev/null 2> "$OUT/pmc_SQ.log"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_SQ/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v[-4:]) / len(v[-4:]) for c, v in d.items()}     # the last launches (AuxK active)
    if m.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    w = m["SQ_WAVE_CYCLES"]
    print("%-70s wavecyc %.3g wait_any %.3f wait_inst %.3f active_inst %.3f valu %.3g salu %.3g lds %.3g" % (
        k, w, m["SQ_WAIT_ANY"] / w, m["SQ_WAIT_INST_ANY"] / w, m["SQ_ACTIVE_INST_ANY"] / w, m["SQ_INSTS_VALU"], m["SQ_INSTS_SALU"], m["SQ_INSTS_LDS"]))
PY

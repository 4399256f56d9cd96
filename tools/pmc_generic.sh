#!/bin/bash
# SQ counters of the generic GEMM kernels at the d=1280 n=40960 shape (run ON the GPU box from the repo root)
OUT=$PWD/gpurun_out/prof_generic
mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d "$OUT/pmc_SQ" -o pmc --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 2 --warmup 1 --spinup 0 > /dev/null 2> "$OUT/pmc_SQ.log"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_SQ/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "gemm" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, "mfma_busy %.3f wait_any %.3f wait_inst %.3f lds_conflict %.3f valu/wavecyc %.3f" % (
        m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_WAVE_CYCLES"]), m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1),
        m["SQ_INSTS_VALU"] / m["SQ_WAVE_CYCLES"]))
PY

#!/bin/bash
# round 4: why the balanced backward gains 1 % where the row count says 6 %: in-kernel cycles per workgroup and HBM traffic, balanced
# against uniform (FREUD_BWD_UNIFORM=1).  Every profiler call under its own timeout (a TCC_HIT_sum pass hung a box for 30 minutes).
set -u
O=gpurun_out/r04_balpmc; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
for mode in bal uni bal uni; do
  if [ $mode = uni ]; then export FREUD_BWD_UNIFORM=1; else unset FREUD_BWD_UNIFORM; fi
  timeout 300 python bench.py --no-cpu-baseline --no-pcie-sample --steps 100 --warmup 10 --dbg 66 2>> $O/stamps_$mode.txt > /dev/null; grep "bwd in-kernel" $O/stamps_$mode.txt | tail -1
done
for mode in bal uni; do
  if [ $mode = uni ]; then export FREUD_BWD_UNIFORM=1; else unset FREUD_BWD_UNIFORM; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $c -d $R/$O/pmc_${c}_$mode -o pmc --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-pcie-sample --steps 5 --warmup 2 --spinup 0 > /dev/null 2> $R/$O/pmc_${c}_$mode.log)
  done
done
unset FREUD_BWD_UNIFORM
python3 - <<'PY'
import csv, glob, collections
for mode in ("bal", "uni"):
    for kind in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"gpurun_out/r04_balpmc/pmc_{kind}_{mode}/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if "bwd_fused" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print(mode, kind, {k: round(sum(v) / len(v), 1) for k, v in acc.items()})
PY
rm -rf $O/pmc_*_bal $O/pmc_*_uni

#!/bin/bash
# per-kernel time of the tile-driven TopK select in three builds (C3, no dead latents), rocprofv3 --stats, one box
set -u
ROOT=$PWD; O=$ROOT/gpurun_out/r04_tselprof; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for lib in current tselfixed; do
  if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$ROOT/build/ab/libfreud_sae_$lib.so; fi
  for rep in 1 2; do
    rm -rf /tmp/tsp
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tsp -o s --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --spinup 0.3 --dead-threshold 1e15 > /dev/null 2> $O/log_$lib.txt
    f=$(find /tmp/tsp -name "*kernel_stats.csv" | head -1)
    python3 - "$lib" "$f" <<'PY'
import csv, sys
rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
pick = lambda key: next((r for n, r in rows.items() if key in n), None)
t, e = pick("topk_select_tiles"), pick("EpiTopkEnc")
print("[%s] tile-driven select %.1f us avg over %s launches | encoder GEMM %.1f us" % (sys.argv[1], float(t["AverageNs"]) / 1e3, t["Calls"], float(e["AverageNs"]) / 1e3))
PY
  done
done > $O/tiles_select_builds.txt 2>&1
cat $O/tiles_select_builds.txt

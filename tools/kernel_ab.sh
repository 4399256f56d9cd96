#!/bin/bash
export FREUD_SAE_ALLOW_OLD_LIB=1      # freud_amd/engine.py: an older build may lack entry points of the current header
# Same-box comparison of per-kernel times between engine builds (run ON the GPU box from the repo root):
#   LIBS="current name1 name2" KERNELS="substr1,substr2" bash tools/kernel_ab.sh <out dir under gpurun_out/> <bench.py args...>
# "current" = freud_amd/lib/libfreud_sae.so, other names = build/ab/libfreud_sae_<name>.so (tools/build_variant.sh).
# Prints, per build and run (two runs each), the rocprofv3 --kernel-trace --stats average of every kernel whose name contains one
# of the substrings, and the step time of that (profiled) run.
set -u
ROOT=$PWD; O=$ROOT/gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for lib in ${LIBS:-current}; do
  if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$ROOT/build/ab/libfreud_sae_$lib.so; fi
  for rep in 1 2; do
    rm -rf /tmp/kab
    timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kab -o s --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --spinup 0.3 "$@" > $O/bench_$lib.json 2> $O/log_$lib.txt
    f=$(find /tmp/kab -name "*kernel_stats.csv" | head -1)
    python3 - "$lib" "$f" $O/bench_$lib.json "${KERNELS:-}" <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[2])))
ms = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["ms_per_step"]
out = []
for key in [k for k in sys.argv[4].split(",") if k]:
    for r in rows:
        if key in r["Name"]:
            out.append("%s %.1f us x%s" % (key, float(r["AverageNs"]) / 1e3, r["Calls"]))
print("[%s] %s | step %.3f ms (profiled run)" % (sys.argv[1], " | ".join(out), ms))
PY
  done
done > $O/kernel_ab.txt 2>&1
cat $O/kernel_ab.txt

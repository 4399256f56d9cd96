#!/bin/bash
# compact AuxK select (topk_select_reg_kernel<12, true>) at C3 with 31 % dead latents: builds compared on one box, rocprofv3 --stats
set -u
ROOT=$PWD; O=$ROOT/gpurun_out/r04_auxsel; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_topk_gpu.py -x -q -m gpu 2>&1 | tail -2
cd /tmp
for lib in "$@"; do
  if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$ROOT/build/ab/libfreud_sae_$lib.so; fi
  for rep in 1 2; do
    rm -rf /tmp/tsp
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tsp -o s --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --spinup 0.3 --dead-threshold 1e5 > $O/bench_$lib.json 2> $O/log_$lib.txt
    f=$(find /tmp/tsp -name "*kernel_stats.csv" | head -1)
    python3 - "$lib" "$f" $O/bench_$lib.json <<'PY'
import csv, sys, json
rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
pick = lambda key: next((r for n, r in rows.items() if key in n), None)
a, t = pick("topk_select_reg_kernelILi12ELb1"), pick("topk_select_tiles")
ms = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["ms_per_step"]
print("[%s] compact AuxK select %.1f us avg over %s launches | tile-driven select %.1f us | step %.3f ms (under the profiler)" % (sys.argv[1], float(a["AverageNs"]) / 1e3, a["Calls"], float(t["AverageNs"]) / 1e3, ms))
PY
  done
done > $O/auxk_select_builds.txt 2>&1
cat $O/auxk_select_builds.txt

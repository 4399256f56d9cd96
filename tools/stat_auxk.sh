cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/stat_auxk; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --spinup 0.3 --dead-threshold 1e5 > $O/b.json 2> $O/b.err
rm -f $(find $O -name "*kernel_trace.csv")
python3 - <<PY
import csv,glob
f=glob.glob('$O/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MaxNs'])/1e3,1))
PY

#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/r02p
mkdir -p $O
timeout 900 python -m pytest tests/test_topk_gpu.py -m gpu -q -x -p no:cacheprovider -k "auxk or dacts" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 20 --warmup 5 --spinup 0.2 --dead-threshold 1e5 > $O/c3_auxk.json 2> $O/c3_auxk.err
rm -f $(find $O/stats -name "*kernel_trace.csv")
python - <<PY
import csv,glob
f=glob.glob('$O/stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(r['Name'][:110], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
python -c "
import json
d=json.loads(open('$O/c3_auxk.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['loss'])"

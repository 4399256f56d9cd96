#!/bin/bash
# round 3: data-parallel protocol overhead on one rank + timelines
set -u
O=$PWD/gpurun_out/r03_dist
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest_dp.txt 2>&1
tail -5 $O/pytest_dp.txt
B="timeout 300 python bench.py --no-cpu-baseline"
run() { name=$1; shift; $B "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), d['roofline']['kernel_launches'], d['config'].get('dp'))
except Exception as e:
    print('$name FAILED', e, open('$O/$name.err').read()[-600:])
PY
}
run default
run fd_p2p --force-dist --dp p2p
run fd_p2p_bf16 --force-dist --dp p2p --dp-payload bfloat16
run fd_rccl --force-dist --dp rccl
run default_b
run fd_p2p_b --force-dist --dp p2p
ROOT=$PWD
cd /tmp
for v in "default" "fd_p2p --force-dist --dp p2p" "fd_rccl --force-dist --dp rccl"; do
  set -- $v; name=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/trace_$name -o t --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --steps 300 --warmup 20 --spinup 0.2 "$@" > $O/trace_$name.json 2> $O/trace_$name.log
  python3 $ROOT/tools/trace_timeline.py $(find $O/trace_$name -name "*kernel_trace.csv" | head -1) 300 > $O/timeline_$name.txt 2>&1
  rm -f $(find $O/trace_$name -name "*kernel_trace.csv")
  head -16 $O/timeline_$name.txt
done

#!/bin/bash
# what the driver runs at the end of a round, in its order: smoke(), the GPU test suite, the default bench line
set -u
O=gpurun_out/round_end; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('metric','value','ms_per_step','dtype','scaling')}, d['roofline']['frac'], d['cpu_baseline']['value'], d['pcie_inclusive']['value'])"

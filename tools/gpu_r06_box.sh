#!/bin/bash
# one box of the survey: the driver's bench call on the round's last build (+ optionally a test subset: TESTS=1)
O=gpurun_out/r06_box_$1; mkdir -p $O
if [ "${TESTS:-0}" = 1 ]; then timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_topk_gpu.py tests/test_fp8_gpu.py -q -x -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt; fi
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
ds=d.get('data_sensitivity',{})
print('box $1: step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'bwd', round(d['roofline']['kernel_avg_ms'],4), {k:round(v['ms_per_step'],4) for k,v in ds.items() if isinstance(v,dict)})"

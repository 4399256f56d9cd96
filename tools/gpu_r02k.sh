#!/bin/bash
set -u
O=$PWD/gpurun_out/r02k
mkdir -p $O
python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
python tools/bench_loader.py --dtype float16 > $O/loader_f16_direct.json 2> $O/l1.err
python tools/bench_loader.py --dtype float32 --files 300 > $O/loader_f32_direct.json 2> $O/l2.err
python tools/bench_loader.py --dtype float32 --files 300 --direct 0 > $O/loader_f32_staged.json 2> $O/l3.err
python tools/bench_loader.py --dtype float32 --files 300 --deliver bfloat16 > $O/loader_f32_bf16.json 2> $O/l4.err
for f in loader_f16_direct loader_f32_direct loader_f32_staged loader_f32_bf16; do python -c "
import json
d=json.load(open('$O/$f.json')); print('$f', d['mode'], 'loader-only', round(d['loader_only_act_per_s']/1e6,1), 'M act/s', round(d['loader_only_GB_per_s'],1), 'GB/s; train loop', round(d['train_loop_act_per_s']/1e6,1), 'M act/s', round(d['train_loop_ms_per_step'],3), 'ms; engine only', round(d['engine_only_ms_per_step'],3))"; done

#!/bin/bash
set -u
O=$PWD/gpurun_out/r02l
mkdir -p $O
for dbg in 0 71 72 73 74; do
python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 10 --warmup 3 --dead-threshold 1e15 --breakdown --dbg $dbg > $O/c3_$dbg.json 2> $O/c3_$dbg.err
echo dbg $dbg $(grep -h "per-kernel" $O/c3_$dbg.err | python -c "
import sys,json
l=sys.stdin.read(); d=json.loads(l[l.index('{'):]); print('select', d['topk_select'], 'enc', d['topk_enc_gemm'])")
done

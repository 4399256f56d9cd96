#!/bin/bash
# round 4: where the loader-fed rate stops on THIS box: raw H2D of the link, loader alone over gather threads x ring depth, loader + step;
# then the 16x16x32 TIMING PROXY of the fused backward (build/ab/libfreud_sae_proxy16.so: wrong numbers, same FLOPs / traffic)
set -u
O=gpurun_out/r04_loader; mkdir -p $O
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt; cat $O/host.txt
timeout 900 python tools/bench_loader.py --dtype float32 --deliver bfloat16 --direct 0 --sweep > $O/loader_f32_bf16_sweep.json 2> $O/loader.err; tail -1 $O/loader_f32_bf16_sweep.json | python -c "import json,sys; print(json.dumps(json.loads(sys.stdin.read()), indent=1)[:3000])"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2>> $O/loader.err; python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['pcie_inclusive']))"
if [ -f build/ab/libfreud_sae_proxy16.so ]; then bash tools/ab_bench.sh build/ab/libfreud_sae_proxy16.so > $O/ab_proxy16.txt 2>&1; cat $O/ab_proxy16.txt; fi

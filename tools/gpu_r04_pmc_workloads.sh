#!/bin/bash
# round 4: counter passes (FETCH_SIZE / WRITE_SIZE / SQ) of the C3 (with and without dead latents), C4 and C5-fp8 workloads on the final build
set -u
bash tools/pmc_workload.sh c4 --d 1280 --n 40960 --steps 4 --warmup 2 2>&1 | tail -3
bash tools/pmc_workload.sh c3 --variant topk --d 768 --n 24576 --k 64 --steps 6 --warmup 3 --dead-threshold 1e15 2>&1 | tail -3
bash tools/pmc_workload.sh c3auxk --variant topk --d 768 --n 24576 --k 64 --steps 6 --warmup 3 --dead-threshold 1e5 2>&1 | tail -3
bash tools/pmc_workload.sh c5fp8 --d 1280 --n 81920 --steps 3 --warmup 2 --precision fp8 2>&1 | tail -3
for w in c4 c3 c3auxk c5fp8; do rm -rf gpurun_out/pmc_$w/pmc_FETCH_SIZE gpurun_out/pmc_$w/pmc_WRITE_SIZE gpurun_out/pmc_$w/pmc_SQ; done

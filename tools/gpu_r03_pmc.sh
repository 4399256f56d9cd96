#!/bin/bash
bash tools/pmc_workload.sh c4 --d 1280 --n 40960 --steps 2 --warmup 1
bash tools/pmc_workload.sh c3 --variant topk --d 768 --n 24576 --k 64 --steps 3 --warmup 2 --dead-threshold 1e15
bash tools/pmc_workload.sh c5fp8 --d 1280 --n 81920 --steps 2 --warmup 1 --precision fp8
bash tools/pmc_workload.sh c2 --steps 5 --warmup 2

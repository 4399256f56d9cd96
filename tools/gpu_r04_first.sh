#!/bin/bash
# round 4, first GPU call: (1) correctness of the new staging swizzle (engine tests incl. the bitwise v1 == v2 forward check),
# (2) same-box A/B of the old swizzle (build/ab/libfreud_sae_swzv1.so) against the new one, (3) tools/gpu_r04_clock.sh
set -u
mkdir -p gpurun_out/r04_first
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/r04_first/pytest_engine.txt 2>&1
tail -3 gpurun_out/r04_first/pytest_engine.txt
bash tools/ab_bench.sh build/ab/libfreud_sae_swzv1.so > gpurun_out/r04_first/ab_swizzle.txt 2>&1
cat gpurun_out/r04_first/ab_swizzle.txt
bash tools/gpu_r04_clock.sh

#!/bin/bash
# round 6, third GPU call: the fp32 evaluation tests, the trajectory tests with their bounds from the first run, bench smoke
mkdir -p gpurun_out/r06_eval32
timeout 1200 python -m pytest tests/test_eval_fp32_gpu.py -q -s -m gpu > gpurun_out/r06_eval32/eval32.txt 2>&1
echo "eval32 rc $?" >> gpurun_out/r06_eval32/eval32.txt
timeout 1200 python -m pytest tests/test_trajectory_gpu.py -q -s -m gpu > gpurun_out/r06_eval32/trajectory.txt 2>&1
echo "trajectory rc $?" >> gpurun_out/r06_eval32/trajectory.txt
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py tests/test_models_gpu.py -q -x -m gpu > gpurun_out/r06_eval32/engine.txt 2>&1
echo "engine rc $?" >> gpurun_out/r06_eval32/engine.txt
grep -v amdgpu.ids gpurun_out/r06_eval32/eval32.txt | tail -40; tail -5 gpurun_out/r06_eval32/trajectory.txt gpurun_out/r06_eval32/engine.txt

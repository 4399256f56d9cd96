#!/bin/bash
# round 6: the whole -m gpu suite + smoke + the driver's bench call on the final build
mkdir -p gpurun_out/r06_final
timeout 3300 python -m pytest tests -q -x -m gpu > gpurun_out/r06_final/gpu_suite.txt 2>&1
echo "suite rc $?" >> gpurun_out/r06_final/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final/smoke.txt 2>&1
echo "smoke rc $?" >> gpurun_out/r06_final/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final/bench_driver_call.json 2> gpurun_out/r06_final/bench_driver_call.err
grep -v amdgpu.ids gpurun_out/r06_final/gpu_suite.txt | tail -4; tail -2 gpurun_out/r06_final/smoke.txt; head -c 400 gpurun_out/r06_final/bench_driver_call.json

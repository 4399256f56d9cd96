#!/bin/bash
# round 6, first GPU call: the new trajectory tests, the self-launching bench, the default line with its data_sensitivity block,
# and the full-length N(0,1) / rotating-batch / raw-cast lines SURVEY 8(d) / VERDICT r5 item 1a ask for
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_trajectory_gpu.py -x -q -s -m gpu > gpurun_out/r06/trajectory.txt 2>&1
echo "trajectory rc $?" >> gpurun_out/r06/trajectory.txt
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "without_a_launcher or several_ranks" > gpurun_out/r06/selflaunch.txt 2>&1
echo "selflaunch rc $?" >> gpurun_out/r06/selflaunch.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_default_driver_style.json 2> gpurun_out/r06/bench_default_driver_style.err
python bench.py --no-cpu-baseline > gpurun_out/r06/bench_lowrank.json 2>/dev/null
python bench.py --no-cpu-baseline --data normal > gpurun_out/r06/bench_normal.json 2>/dev/null
python bench.py --no-cpu-baseline --rotate 4 > gpurun_out/r06/bench_rotate4.json 2>/dev/null
python bench.py --no-cpu-baseline --raw-cast > gpurun_out/r06/bench_raw_cast.json 2>/dev/null
python bench.py --no-cpu-baseline --data normal --rotate 4 > gpurun_out/r06/bench_normal_rotate4.json 2>/dev/null
tail -3 gpurun_out/r06/trajectory.txt gpurun_out/r06/selflaunch.txt

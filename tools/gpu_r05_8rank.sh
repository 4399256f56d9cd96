#!/bin/bash
O=gpurun_out/r05_8rank; mkdir -p $O
for i in 1 2 3; do
  timeout 900 python -m pytest "tests/test_dp_gpu.py::test_eight_processes_one_gpu_train_like_one_process" -m gpu -q -x > $O/try$i.txt 2>&1
  echo "try $i rc=$?" >> $O/try$i.txt; tail -2 $O/try$i.txt
done

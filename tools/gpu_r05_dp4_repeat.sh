#!/bin/bash
# round 5: the four-process one-GPU training test hung once (900 s, no output) in a full-suite run: repeat it to catch where the ranks stand
O=gpurun_out/r05_dp4_repeat; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -k "four_processes" > $O/run_$i.txt 2>&1; echo "run $i rc=$?" | tee -a $O/summary.txt
  tail -1 $O/run_$i.txt >> $O/summary.txt
done
cat $O/summary.txt

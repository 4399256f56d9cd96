#!/bin/bash
# round 5, first GPU call: the new resume / 8-rank / dp_timing tests, a sanity bench line, the 4- and 8-rank share-GPU bench lines
O=gpurun_out/r05_first; mkdir -p $O
timeout 1500 python -m pytest tests/test_resume_gpu.py -m gpu -x -q > $O/resume.txt 2>&1; echo "resume rc=$?" >> $O/resume.txt
timeout 2400 python -m pytest tests/test_dp_gpu.py -m gpu -q > $O/dp.txt 2>&1; echo "dp rc=$?" >> $O/dp.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
for W in 4 8; do
  P=$((29600 + W))
  for r in $(seq 0 $((W-1))); do
    FREUD_BENCH_SHARE_GPU=1 RANK=$r LOCAL_RANK=$r WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=$P \
      python bench.py --gpus $W --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_ranks${W}_r$r.json 2> $O/bench_ranks${W}_r$r.err &
  done
  wait
done
tail -3 $O/resume.txt $O/dp.txt; cat $O/bench_c2.json | cut -c1-400

#!/bin/bash
# NOTE: the engine side of this experiment (EpiEncG / EpiDpreG, the FREUD_GATE_MASK switch) was measured and REMOVED -- DESIGN.md section 7 item 5,
# profiles/r06_gate_mask_experiment.txt; the script is kept as the record of what was run.
# round 6: the L1 backward's ReLU gate as a bit mask from the streaming encoder (EpiEncG -> EpiDpreG) against the latent read (FREUD_GATE_MASK=0),
# same library, same box, interleaved: tests through both, then C4 / C5 bf16 step + kernel times
O=gpurun_out/r06_gatemask; mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_trajectory_gpu.py tests/test_resume_gpu.py -q -x -m gpu > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt; tail -3 $O/tests.txt
FREUD_GATE_MASK=0 timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -m gpu > $O/tests_off.txt 2>&1; echo "tests (mask off) rc $?" >> $O/tests_off.txt; tail -2 $O/tests_off.txt
for shape in "1280 40960 20 3" "1280 81920 6 2" "768 24576 20 3"; do set -- $shape
for i in 1 2 3; do for gm in 1 0; do
  echo -n "d=$1 n=$2 FREUD_GATE_MASK=$gm " >> $O/ab.txt
  FREUD_GATE_MASK=$gm python bench.py --no-cpu-baseline --d $1 --n $2 --steps $3 --warmup $4 --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v and k in ('enc_fwd_gemm','dec_fwd_gemm','dpre_gemm','dw_gemm')})" >> $O/ab.txt
done; done; done; cat $O/ab.txt

#!/bin/bash
# round 6: tile-form row x row K loop on 16x16x32 -- stand-alone compare + timing, the engine tests through it, C4 / C3 / C5 same-box A/B
bash tools/gpu_r06_tile_m16.sh
O=gpurun_out/r06_tile_m16
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_topk_gpu.py tests/test_fp8_gpu.py tests/test_models_gpu.py -q -x -m gpu > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt; tail -3 $O/tests.txt
bash tools/ab_c4.sh build/ab/libfreud_sae_tile32.so > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt
DT=1e15 bash tools/ab_topk.sh build/ab/libfreud_sae_tile32.so > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt
export FREUD_SAE_ALLOW_OLD_LIB=1
for i in 1 2; do for lib in "" build/ab/libfreud_sae_tile32.so; do
  echo -n "${lib:-current} " >> $O/ab_c5.txt
  FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 6 --warmup 2 --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v})" >> $O/ab_c5.txt
done; done; cat $O/ab_c5.txt

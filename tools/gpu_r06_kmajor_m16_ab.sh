#!/bin/bash
# round 6: k-major 16x16x32 loop (build/ab/libfreud_sae_m16k.so = -DG2_M16K=1) against the shipped build, same box: C4, C5 bf16, d = 768 L1
O=gpurun_out/r06_kmajor_m16; mkdir -p $O
bash tools/ab_c4.sh build/ab/libfreud_sae_m16k.so > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt
export FREUD_SAE_ALLOW_OLD_LIB=1
for i in 1 2; do for lib in "" build/ab/libfreud_sae_m16k.so; do
  echo -n "${lib:-current} " >> $O/ab_c5.txt
  FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 6 --warmup 2 --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v})" >> $O/ab_c5.txt
done; done; cat $O/ab_c5.txt
for i in 1 2; do for lib in "" build/ab/libfreud_sae_m16k.so; do
  echo -n "${lib:-current} " >> $O/ab_d768.txt
  FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 768 --n 24576 --steps 20 --warmup 3 --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v})" >> $O/ab_d768.txt
done; done; cat $O/ab_d768.txt
FREUD_SAE_LIB=build/ab/libfreud_sae_m16k.so timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -m gpu > $O/tests_m16k.txt 2>&1; tail -2 $O/tests_m16k.txt

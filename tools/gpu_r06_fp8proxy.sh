#!/bin/bash
# round 6: would the 16x16x128 shape lift the fp8 encoder like 16x16x32 lifted the bf16 GEMMs?  Timing proxy (-DG8S_PROXY16, results wrong):
# only enc_fwd_gemm of the proxy build counts, first steps only (the garbage it produces changes the later operands)
O=gpurun_out/r06_fp8proxy; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
for i in 1 2 3; do for lib in "" build/ab/libfreud_sae_g8sproxy16.so; do
  echo -n "${lib:-current} " >> $O/ab_c5fp8.txt
  FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 4 --warmup 1 --spinup 0.3 --precision fp8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v})" >> $O/ab_c5fp8.txt
done; done; cat $O/ab_c5fp8.txt

#!/bin/bash
# round 6, last build (k-major 16x16x32 loop + running source pointers in): the whole -m gpu suite, smoke, the driver's bench call, then the
# same-box A/B against the build before both (build/ab/libfreud_sae_pre16k.so = -DG2_M16K=0 -DG2_RUNPTR=0): C4, C5 bf16, C5 fp8
bash tools/gpu_r06_final.sh
O=gpurun_out/r06_final2; mkdir -p $O
bash tools/ab_c4.sh build/ab/libfreud_sae_pre16k.so > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt
export FREUD_SAE_ALLOW_OLD_LIB=1
for prec in bf16 fp8; do
for i in 1 2; do for lib in "" build/ab/libfreud_sae_pre16k.so; do
  echo -n "${lib:-current} " >> $O/ab_c5_$prec.txt
  FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 6 --warmup 2 --breakdown $( [ $prec = fp8 ] && echo "--precision fp8" ) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v})" >> $O/ab_c5_$prec.txt
done; done; cat $O/ab_c5_$prec.txt
done

#!/bin/bash
# round 6: fp8 K loops with their side work one piece per MFMA gap + carried source origins (G8_SPREAD), the bf16 streaming loop with carried
# origins (G2S_RUNPTR) -- tests through them, then same-box A/B against build/ab/libfreud_sae_g8old.so (-DG8_SPREAD=0 -DG2S_RUNPTR=0)
O=gpurun_out/r06_fp8spread; mkdir -p $O
timeout 1500 python -m pytest tests/test_fp8_gpu.py tests/test_engine_gpu.py tests/test_topk_gpu.py -q -x -m gpu > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt; tail -3 $O/tests.txt
export FREUD_SAE_ALLOW_OLD_LIB=1
for prec in fp8 bf16; do
for i in 1 2 3; do for lib in "" build/ab/libfreud_sae_g8old.so; do
  echo -n "${lib:-current} " >> $O/ab_c5_$prec.txt
  FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 6 --warmup 2 --breakdown $( [ $prec = fp8 ] && echo "--precision fp8" ) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v})" >> $O/ab_c5_$prec.txt
done; done; cat $O/ab_c5_$prec.txt
done
bash tools/ab_c4.sh build/ab/libfreud_sae_g8old.so > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt
DT=1e15 bash tools/ab_topk.sh build/ab/libfreud_sae_g8old.so > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt

#!/bin/bash
# round 6: the streaming GEMM's K loop -- compile-time stages (G2S_STATIC) and the 16x16x32 MFMA shape (G2S_M16) -- against the static
# 32x32x16 loop (libfreud_sae_g2s32.so) and round 5's loop (libfreud_sae_g2sdyn.so): stand-alone bit-compare + timing on random dense
# operands, the engine's parity tests that run through it, then C4 / C3 / C5-bf16 same-box A/B
O=gpurun_out/r06_g2s16; mkdir -p $O
{
echo "== gemm_s16: bit-compare stream (16x16x32) vs tile form (32x32x16)"
build/kbench/gemm_s16 8192 8192 1280 5
build/kbench/gemm_s16 4096 12288 768 5
build/kbench/gemm_s16 2048 1024 128 5
for rep in 1 2 3; do
  for b in gemm_sdyn gemm_s32 gemm_s16; do
    echo -n "$b: "; build/kbench/$b 65536 40960 1280 4
    echo -n "$b: "; build/kbench/$b 65536 24576 768 4
  done
done
} > $O/kbench.txt 2>&1
cat $O/kbench.txt
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_topk_gpu.py tests/test_fp8_gpu.py tests/test_trajectory_gpu.py -q -x -m gpu > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt; tail -3 $O/tests.txt
bash tools/ab_c4.sh build/ab/libfreud_sae_g2s32.so build/ab/libfreud_sae_g2sdyn.so > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt
DT=1e15 bash tools/ab_topk.sh build/ab/libfreud_sae_g2s32.so build/ab/libfreud_sae_g2sdyn.so > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt
export FREUD_SAE_ALLOW_OLD_LIB=1
for i in 1 2; do for lib in "" build/ab/libfreud_sae_g2s32.so build/ab/libfreud_sae_g2sdyn.so; do
  echo -n "${lib:-current} " >> $O/ab_c5.txt
  FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 6 --warmup 2 --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v})" >> $O/ab_c5.txt
done; done; cat $O/ab_c5.txt

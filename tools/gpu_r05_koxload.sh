#!/bin/bash
# round 5: fused forward prologue -- knock-outs of the x loads (results wrong): koxload = fully coalesced, koxsame = every workgroup reads block 0 (x from the L2)
O=gpurun_out/r05_koxload2; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
for i in 1 2; do
  for lib in current build/ab/libfreud_sae_koxload.so build/ab/libfreud_sae_koxsame.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo "== $lib"; python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 65 2>&1 | grep -E "^fwd "
  done
done > $O/stamps.txt 2>&1
cat $O/stamps.txt

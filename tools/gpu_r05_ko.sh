#!/bin/bash
# round 5: knock-out series on the fused forward's residual-arithmetic phase (-DFF2_KO=mask, results wrong by construction; stamps only)
O=gpurun_out/r05_ko; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
for i in 1 2; do
  for lib in current ko1 ko2 ko4 ko8 ko15; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=build/ab/libfreud_sae_$lib.so; fi
    echo "== $lib"; python bench.py --no-cpu-baseline --steps 50 --warmup 10 --dbg 65 2>&1 | grep -E "fwd per-workgroup|fwd epilogue"
  done
done > $O/stamps.txt 2>&1
cat $O/stamps.txt

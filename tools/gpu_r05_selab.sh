#!/bin/bash
# round 5: the select kernels before / after the template-constant loops and occupancy 3 (same box, interleaved)
O=gpurun_out/r05_selab; mkdir -p $O
export FREUD_SAE_ALLOW_OLD_LIB=1
parse='
import sys,re,json
t=sys.stdin.read()
k=json.loads(re.search(r"level-2 profile\): (\{.*?\})", t).group(1))
m=re.search(r"\"ms_per_step\": ([0-9.]+)", t).group(1)
print(" ".join("%s %.3f" % (n, v) for n, v in k.items() if v > 0.25 and n != "fwd_bwd_total"), "step", m)'
{
for rep in 1 2; do
  for lib in current build/ab/libfreud_sae_presel.so; do
    if [ $lib = current ]; then unset FREUD_SAE_LIB; else export FREUD_SAE_LIB=$lib; fi
    echo -n "[$lib C3 31% dead] "; python bench.py --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e5 --no-cpu-baseline --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
    echo -n "[$lib TopK d1280 n40960 k32, 4000 dead] "; python bench.py --variant topk --d 1280 --n 40960 --k 32 --steps 20 --warmup 5 --dead-threshold 1e15 --dead-latents 4000 --no-cpu-baseline --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
    echo -n "[$lib TopK d1280 n40960 k32, 300 dead] "; python bench.py --variant topk --d 1280 --n 40960 --k 32 --steps 20 --warmup 5 --dead-threshold 1e15 --dead-latents 300 --no-cpu-baseline --breakdown 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
  done
done
} > $O/ab.txt 2>&1
cat $O/ab.txt

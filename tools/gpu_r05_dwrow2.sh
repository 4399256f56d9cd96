#!/bin/bash
O=gpurun_out/r05_dwrow2; mkdir -p $O
build/kbench/lds_tr_probe > $O/lds_probe.txt 2>&1
parse='
import sys,re,json
t=sys.stdin.read()
k=json.loads(re.search(r"level-2 profile\): (\{.*?\})", t).group(1))
m=re.search(r"\"ms_per_step\": ([0-9.]+)", t).group(1)
print(" ".join("%s %.3f" % (n, v) for n, v in k.items() if v > 0.2 and n != "fwd_bwd_total"), "step", m)'
for rep in 1 2; do
  for dbg in 0 81; do
    for ra in 0 1; do
      echo -n "[C4 rowA=$ra dbg=$dbg] "; FREUD_DW_ROWA=$ra python bench.py --d 1280 --n 40960 --steps 10 --warmup 3 --no-cpu-baseline --breakdown --dbg $dbg 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | python -c "$parse"
    done
  done
done > $O/ab.txt 2>&1
cat $O/lds_probe.txt $O/ab.txt

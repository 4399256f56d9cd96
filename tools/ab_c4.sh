#!/bin/bash
export FREUD_SAE_ALLOW_OLD_LIB=1      # freud_amd/engine.py: an older build may lack entry points of the current header
# same-box A/B of the C4 shape (L1, d=1280, n=40960): in-tree library against the given alternates
for i in 1 2; do
  for lib in "" "$@"; do
    echo -n "${lib:-current} "
    FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 40960 --steps 10 --warmup 3 --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v})"
  done
done

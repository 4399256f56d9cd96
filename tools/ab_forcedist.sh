#!/bin/bash
export FREUD_SAE_ALLOW_OLD_LIB=1      # freud_amd/engine.py: an older build may lack entry points of the current header
# same-box, interleaved: the C2 step alone against the same step with the whole data-parallel protocol on one rank
# (peer exchange fp32 / bf16 payload, in-engine RCCL, host-driven) -- separate bench runs differ by +-1.5 %, pairs do not
for i in 1 2 3; do
  for mode in "" "--force-dist" "--force-dist --dp-payload bfloat16" "--force-dist --dp rccl" "--force-dist --dp-host"; do
    echo -n "[${mode:-plain}] "
    python bench.py --no-cpu-baseline --steps 400 --warmup 20 $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,1), 'us', d['config'].get('dp'))"
  done
done

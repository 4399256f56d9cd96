#!/bin/bash
# C3 TopK step with a chosen number of latents marked dead at the start (AuxK active from step 1): how the AuxK branch's
# cost scales with the dead set.  Run ON the GPU box from the repo root.
for nd in ${DEAD_LIST:-0 300 1000 3000}; do
  python bench.py --no-cpu-baseline --variant topk --d 768 --n 24576 --k 64 --steps 30 --warmup 5 --dead-threshold 1e15 \
    --dead-latents $nd --breakdown 2>/dev/null | python tools/print_topk_line.py "dead $nd"
done

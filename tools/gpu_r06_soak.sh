#!/bin/bash
# round 6, last build: soak runs -- C2 (3000 steps), TopK with a changing dead set (two runs, bitwise), the generic L1 path at the C4 shape in bf16 and fp8
O=gpurun_out/r06_soak; mkdir -p $O
{ python tools/longrun_sanity.py; python tools/longrun_topk.py; python tools/longrun_generic.py; } > $O/soak.txt 2>&1
grep -v amdgpu.ids $O/soak.txt | tail -32

#!/bin/bash
# round 5: streaming GEMM epilogues without a branch per s_apply call (EpiEnc: skip_store a build switch; EpiTopkEnc: no null test, no lane
# predicate on the tile-maximum store) against the build before (prebranch): tests, then C4 / C3 / C3-no-dead A/B on one box
O=gpurun_out/r05_epibranch; mkdir -p $O
timeout 2400 python -m pytest tests/test_engine_gpu.py tests/test_topk_gpu.py tests/test_fp8_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
bash tools/ab_c4.sh build/ab/libfreud_sae_prebranch.so > $O/ab_c4.txt 2>&1
DT=1e15 bash tools/ab_topk.sh build/ab/libfreud_sae_prebranch.so > $O/ab_c3.txt 2>&1
bash tools/ab_topk.sh build/ab/libfreud_sae_prebranch.so > $O/ab_c3_auxk.txt 2>&1
tail -3 $O/tests.txt; cat $O/ab_c4.txt $O/ab_c3.txt $O/ab_c3_auxk.txt

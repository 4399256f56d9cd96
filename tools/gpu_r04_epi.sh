#!/bin/bash
# round 4: lean forward epilogue (fwd_fused2.h) -- engine / train suites (incl. the new full-size C4 test), A/B against round 3's
# epilogue (build/ab/libfreud_sae_epiv1.so)
set -u
O=gpurun_out/r04_epi; mkdir -p $O
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py tests/test_models_gpu.py -x -q -m gpu > $O/pytest_engine.txt 2>&1; tail -4 $O/pytest_engine.txt
bash tools/ab_bench.sh build/ab/libfreud_sae_epiv1.so > $O/ab_epilogue.txt 2>&1; cat $O/ab_epilogue.txt

#!/bin/bash
# round 6: EpiEnc8's streaming epilogue in packed arithmetic (28 instead of ~70 vector instructions per 8 latents): fp8 tests, then C5 fp8 same-box
# A/B against build/ab/libfreud_sae_g8old.so (the build before the fp8 loop work of this evening)
O=gpurun_out/r06_enc8; mkdir -p $O
timeout 1500 python -m pytest tests/test_fp8_gpu.py tests/test_trajectory_gpu.py -q -x -m gpu > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt; tail -3 $O/tests.txt
timeout 600 python -m pytest "tests/test_engine_gpu.py::test_streaming_gemms_equal_the_tile_form" -q -x -m gpu >> $O/tests.txt 2>&1; tail -2 $O/tests.txt
export FREUD_SAE_ALLOW_OLD_LIB=1
for i in 1 2 3; do for lib in "" build/ab/libfreud_sae_g8old.so; do
  echo -n "${lib:-current} " >> $O/ab_c5_fp8.txt
  FREUD_SAE_LIB=$lib python bench.py --no-cpu-baseline --d 1280 --n 81920 --steps 6 --warmup 2 --breakdown --precision fp8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); km=d['kernel_ms']
print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in km.items() if v})" >> $O/ab_c5_fp8.txt
done; done; cat $O/ab_c5_fp8.txt

#!/bin/bash
# round 6, the last build: whole -m gpu suite + smoke + the driver's bench call, then the profile round (kernel stats, PMC, timelines, C2-C5 lines)
bash tools/gpu_r06_final.sh
bash tools/profile_round.sh r06f > gpurun_out/r06f_profile_round.txt 2>&1
tail -3 gpurun_out/r06f_profile_round.txt

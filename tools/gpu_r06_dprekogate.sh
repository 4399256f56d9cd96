#!/bin/bash
# round 6, costing experiment: EpiDpre without its read of the latent (-DDPRE_KO_GATE: results wrong) -- the upper bound of a 1-bit gate mask
O=gpurun_out/r06_dprekogate; mkdir -p $O
bash tools/ab_c4.sh build/ab/libfreud_sae_dprekogate.so > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt

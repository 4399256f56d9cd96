#!/bin/bash
# round 6: what would the 16x16x32 MFMA shape give the streaming K = d GEMMs (two waves per SIMD) under the power limit?  A timing
# PROXY (results wrong): every 32x32x16 MFMA as two 16x16x32 on the same operands (-DG2S_PROXY16).  Only enc_fwd_gemm / dpre_gemm count.
O=gpurun_out/r06_proxy16; mkdir -p $O
bash tools/ab_c4.sh build/ab/libfreud_sae_g2sproxy16.so > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt

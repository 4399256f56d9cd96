#!/bin/bash
# device assembly of the shipped fused forward (bf16 activations, no padding, no stamps) -> /tmp/ff2b.s, with the numbers that matter:
# scratch use, MFMA count, the instruction histogram of the tile loop.  tools/ff2_asm.sh [extra hipcc flags]
cd /root/repo/build/asm || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function --cuda-device-only "$@" -S -o engine_dev.s /root/repo/freud_amd/csrc/engine.hip 2>/dev/null
L=$(grep -n "^_Z22fwd_fused2_d384_kernelIDF16bLb0ELb0E" engine_dev.s | cut -d: -f1)
awk -v l=$L 'NR>=l' engine_dev.s | awk '/^\.Lfunc_end/{exit} {print}' > /tmp/ff2b.s
echo "scratch instructions: $(grep -c scratch_ /tmp/ff2b.s); MFMAs: $(grep -c v_mfma /tmp/ff2b.s); lines: $(wc -l < /tmp/ff2b.s)"
grep -n "private_segment_fixed_size\|amdhsa_next_free_vgpr\|amdhsa_accum_offset" /tmp/ff2b.s
grep -n "Loop Header" /tmp/ff2b.s | head

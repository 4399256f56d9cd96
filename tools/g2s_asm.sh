#!/bin/bash
# device assembly of the streaming GEMM (encoder functor) -> /tmp/g2s.s.  tools/g2s_asm.sh [extra hipcc flags]
cd /root/repo/build/asm || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function --cuda-device-only "$@" -S -o engine_dev.s /root/repo/freud_amd/csrc/engine.hip 2>/tmp/g2s_asm.err || { grep -E "error" /tmp/g2s_asm.err | head; exit 1; }
L=$(grep -n "^_Z20gemm256s_bf16_kernelI6EpiEncE" engine_dev.s | head -1 | cut -d: -f1)
awk -v l=$L 'NR>=l' engine_dev.s | awk '/^\.Lfunc_end/{exit} {print}' > /tmp/g2s.s
echo "scratch instructions: $(grep -c scratch_ /tmp/g2s.s); MFMAs: $(grep -c v_mfma /tmp/g2s.s); lines: $(wc -l < /tmp/g2s.s)"
grep -n "private_segment_fixed_size\|amdhsa_next_free_vgpr\|amdhsa_next_free_sgpr" /tmp/g2s.s
grep -n "Loop Header" /tmp/g2s.s | head

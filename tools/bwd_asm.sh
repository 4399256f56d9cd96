#!/bin/bash
# device assembly of the fused backward -> /tmp/bwd.s (scratch use, registers, loop headers).  tools/bwd_asm.sh [extra hipcc flags]
cd /root/repo/build/asm || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function --cuda-device-only "$@" -S -o engine_dev.s /root/repo/freud_amd/csrc/engine.hip 2>/tmp/bwd_asm.err || { grep -E "error" /tmp/bwd_asm.err | head; exit 1; }
L=$(grep -n "^_Z21bwd_fused_d384_kernel" engine_dev.s | head -1 | cut -d: -f1)
awk -v l=$L 'NR>=l' engine_dev.s | awk '/^\.Lfunc_end/{exit} {print}' > /tmp/bwd.s
echo "scratch instructions: $(grep -c scratch_ /tmp/bwd.s); MFMAs: $(grep -c v_mfma /tmp/bwd.s); lines: $(wc -l < /tmp/bwd.s)"
grep -n "private_segment_fixed_size\|amdhsa_next_free_vgpr\|amdhsa_next_free_sgpr\|amdhsa_accum_offset" /tmp/bwd.s
grep -n "Loop Header" /tmp/bwd.s | head

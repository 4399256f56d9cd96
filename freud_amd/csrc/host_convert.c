// libfreud_host.so -- host-side helper of the activation loader (freud_amd/loader.py): fp32 shard rows -> bf16 while
// they are gathered into the pinned staging ring, so that the host -> HBM link carries 2 instead of 4 bytes per value.
// (The reference's collector writes fp32 shards; the engine's GEMMs read bf16(x) either way.)
//
// Rounding: round to nearest even, like the device's v_cvt_pk_bf16_f32.  One guard keeps the reference's semantics:
// mse_loss masks entries that are EXACTLY -1.0 (src/models/l1autoencoder.py:31), so a value that is not -1.0 but would
// round to it is moved to the neighbouring bf16 instead (it stays unmasked, off by one more ulp); NaNs stay NaNs.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t f32_to_bf16(uint32_t u) {        // branch-free (selects), so that the loop vectorises
  uint32_t b = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
  const uint32_t nan = (u & 0x7FFFFFFFu) > 0x7F800000u;
  b = nan ? ((u >> 16) | 0x0040u) : b;                  // NaN: keep it a (quiet) NaN
  const uint32_t guard = (b == 0xBF80u) & (u != 0xBF800000u);
  b = guard ? ((u > 0xBF800000u) ? 0xBF81u : 0xBF7Fu) : b;   // -1.0 is reserved for the mask
  return b;
}

// n values src (fp32) -> dst (bf16 bit patterns)
void freud_f32_to_bf16(const float* src, uint16_t* dst, size_t n) {
  const uint32_t* s = (const uint32_t*)src;
  for (size_t i = 0; i < n; ++i) dst[i] = (uint16_t)f32_to_bf16(s[i]);
}

// rows idx[0..count) of a [*, row_elems] fp32 matrix -> consecutive bf16 rows of dst
void freud_gather_f32_to_bf16(const float* base, const int64_t* idx, size_t count, size_t row_elems, uint16_t* dst) {
  for (size_t j = 0; j < count; ++j) freud_f32_to_bf16(base + (size_t)idx[j] * row_elems, dst + j * row_elems, row_elems);
}

int freud_host_version(void) { return 1; }

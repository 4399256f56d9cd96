// Index arithmetic of the peer exchange (p2p_exchange.h), free of any HIP type so that the SAME functions compile for the host:
// tests/test_p2p_index.py builds them with g++ and checks, for worlds 1-8, every grid size and ragged totals (total % world != 0,
// total < world, total < world * grid), that the slices (shard q, workgroup b) cover every vector of a segment exactly once and
// that a vector's element address stays inside its 2-D block.  The 8-rank index paths have no other CPU-side witness.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define P2P_HD __host__ __device__ __forceinline__
#else
#define P2P_HD inline
#endif

// Vectors [v0, v1) of slice (shard q, workgroup b) of a segment of `total` 16-byte vectors cut into `world` shards of `nblocks`
// slices.  Shards are ceil(total / world) vectors (the last ones may be short or empty), slices ceil(shard / nblocks).
P2P_HD void p2p_slice_of(unsigned total, unsigned world, unsigned nblocks, unsigned q, unsigned b, unsigned& v0, unsigned& v1) {
  const unsigned shard = (total + world - 1) / world, slice = (shard + nblocks - 1) / nblocks;
  v0 = q * shard + b * slice;
  v1 = v0 + slice;
  if (v1 > (q + 1) * shard) v1 = (q + 1) * shard;
  if (v1 > total) v1 = total;
  if (v0 > v1) v0 = v1;          // (an empty slice: the loops over [v0, v1) must not wrap)
}

// Element offset (in elements of the buffer) of vector v of a 2-D block: `rows` rows of `cols` elements, `pitch` elements apart,
// first element `off`; epv = elements per 16-byte vector (cols % epv == 0, so a vector never straddles a row).
P2P_HD int64_t p2p_elem_of(int64_t off, int64_t pitch, int rows, int cols, int epv, unsigned v) {
  if (rows == 1) return off + (int64_t)v * epv;
  const unsigned vpr = (unsigned)(cols / epv), row = v / vpr;
  return off + (int64_t)row * pitch + (int64_t)(v - row * vpr) * epv;
}

// libfreud_host.so -- host-side helper of the activation loader (freud_amd/loader.py): fp32 shard rows -> bf16 while
// they are gathered into the pinned staging ring, so that the host -> HBM link carries 2 instead of 4 bytes per value.
// (The reference's collector writes fp32 shards; the engine's GEMMs read bf16(x) either way.)
//
// Rounding: round to nearest even, like the device's v_cvt_pk_bf16_f32.  One guard keeps the reference's semantics:
// mse_loss masks entries that are EXACTLY -1.0 (src/models/l1autoencoder.py:31), so a value that is not -1.0 but would
// round to it is moved to the neighbouring bf16 instead (it stays unmasked, off by one more ulp); NaNs stay NaNs.
//
// Round 4 (the loader delivered 30.5 GB/s on the driver's box, the gather threads were the limit): an
// AVX-512 form of the same integer arithmetic (bit-identical to the portable loop, checked by tests/test_loader.py) that
// leaves through NON-TEMPORAL 64-byte stores -- the pinned ring is read next by the DMA engine, not by this core, and a
// streaming store saves the read-for-ownership of every destination line (6 instead of 8 bytes of memory traffic per value).
// Chosen at run time (__builtin_cpu_supports); the portable loop (auto-vectorised for AVX2) stays as the fallback and as the
// reference of the test.  (vcvtne2ps2bf16 is deliberately NOT used: it flushes fp32 denormals to zero, the device does not.)
#include <immintrin.h>
#include <pthread.h>
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

static void convert_portable(const float* src, uint16_t* dst, size_t n) {
  const uint32_t* s = (const uint32_t*)src;
  for (size_t i = 0; i < n; ++i) dst[i] = (uint16_t)f32_to_bf16(s[i]);
}

__attribute__((target("avx512f,avx512bw,avx512vl"))) static inline __m256i cvt16_avx512(__m512i u) {
  const __m512i lsb = _mm512_and_si512(_mm512_srli_epi32(u, 16), _mm512_set1_epi32(1));
  __m512i b = _mm512_srli_epi32(_mm512_add_epi32(_mm512_add_epi32(u, _mm512_set1_epi32(0x7FFF)), lsb), 16);
  const __mmask16 nan = _mm512_cmpgt_epu32_mask(_mm512_and_si512(u, _mm512_set1_epi32(0x7FFFFFFF)), _mm512_set1_epi32(0x7F800000));
  b = _mm512_mask_mov_epi32(b, nan, _mm512_or_si512(_mm512_srli_epi32(u, 16), _mm512_set1_epi32(0x40)));
  const __mmask16 guard = _mm512_cmpeq_epu32_mask(b, _mm512_set1_epi32(0xBF80)) & _mm512_cmpneq_epu32_mask(u, _mm512_set1_epi32((int)0xBF800000u));
  const __mmask16 above = _mm512_cmpgt_epu32_mask(u, _mm512_set1_epi32((int)0xBF800000u));
  b = _mm512_mask_mov_epi32(b, guard & above, _mm512_set1_epi32(0xBF81));
  b = _mm512_mask_mov_epi32(b, guard & ~above, _mm512_set1_epi32(0xBF7F));
  return _mm512_cvtepi32_epi16(b);
}

__attribute__((target("avx512f,avx512bw,avx512vl"))) static void convert_avx512(const float* src, uint16_t* dst, size_t n) {
  size_t i = 0;
  // head: up to the first 64-byte boundary of dst (streaming stores want whole aligned lines)
  while (i < n && ((uintptr_t)(dst + i) & 63)) {
    uint32_t u;
    memcpy(&u, src + i, 4);
    dst[i] = (uint16_t)f32_to_bf16(u);
    ++i;
  }
  for (; i + 32 <= n; i += 32) {
    const __m256i lo = cvt16_avx512(_mm512_loadu_si512((const void*)(src + i)));
    const __m256i hi = cvt16_avx512(_mm512_loadu_si512((const void*)(src + i + 16)));
    _mm512_stream_si512((__m512i*)(dst + i), _mm512_inserti64x4(_mm512_castsi256_si512(lo), hi, 1));
  }
  for (; i < n; ++i) {
    uint32_t u;
    memcpy(&u, src + i, 4);
    dst[i] = (uint16_t)f32_to_bf16(u);
  }
  _mm_sfence();            // the streaming stores are globally visible before the caller hands the buffer to the DMA
}

static int g_impl = -1;    // 0 portable, 1 avx512
static int pick_impl(void) {
  if (g_impl < 0) {
    __builtin_cpu_init();
    g_impl = (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl")) ? 1 : 0;
  }
  return g_impl;
}

// n values src (fp32) -> dst (bf16 bit patterns)
void freud_f32_to_bf16(const float* src, uint16_t* dst, size_t n) {
  if (pick_impl() == 1) convert_avx512(src, dst, n);
  else convert_portable(src, dst, n);
}

// the portable loop by name (tests compare the two forms; also what a host without AVX-512 runs)
void freud_f32_to_bf16_portable(const float* src, uint16_t* dst, size_t n) { convert_portable(src, dst, n); }

// rows idx[0..count) of a [*, row_elems] fp32 matrix -> consecutive bf16 rows of dst
void freud_gather_f32_to_bf16(const float* base, const int64_t* idx, size_t count, size_t row_elems, uint16_t* dst) {
  for (size_t j = 0; j < count; ++j) freud_f32_to_bf16(base + (size_t)idx[j] * row_elems, dst + j * row_elems, row_elems);
}

// elements [e0, e1) of row idx of the shard -> the same elements of dst_row: the unit of work of the loader's gather pool
// (rows are 1-8 MB: pieces of rows balance 40 rows over 16-24 threads where whole rows cannot)
void freud_convert_piece(const float* base, int64_t idx, size_t row_elems, size_t e0, size_t e1, uint16_t* dst_row) {
  freud_f32_to_bf16(base + (size_t)idx * row_elems + e0, dst_row + e0, e1 - e0);
}

// ---- a small persistent thread pool (pthreads; no OpenMP: the process already hosts torch's OpenMP runtime, and a second one
// in a helper library is a known source of oversubscription and fork trouble).  One job at a time (callers queue on a mutex: a process may run
// a training and a validation loader); workers sleep on a condition variable between jobs; work items are handed out by an atomic counter; the
// calling thread works too.
#define POOL_MAX 64
typedef void (*item_fn)(long item, void* arg);
static struct {
  pthread_t th[POOL_MAX];
  int nth;                         // workers started
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_done;
  unsigned long gen;               // job generation (under mu)
  item_fn fn;
  void* arg;
  long items;
  int use;                         // workers that may join this job
  volatile long next;              // next item (atomic)
  volatile int active;             // workers still inside the job (atomic)
  int shutdown;
} g_pool = {.mu = PTHREAD_MUTEX_INITIALIZER, .cv_work = PTHREAD_COND_INITIALIZER, .cv_done = PTHREAD_COND_INITIALIZER};

static void pool_run_items(void) {
  for (;;) {
    const long it = __atomic_fetch_add(&g_pool.next, 1, __ATOMIC_RELAXED);
    if (it >= g_pool.items) break;
    g_pool.fn(it, g_pool.arg);
  }
}

static void* pool_worker(void* idp) {
  const int id = (int)(intptr_t)idp;
  unsigned long seen = 0;
  pthread_mutex_lock(&g_pool.mu);
  for (;;) {
    while (!g_pool.shutdown && (g_pool.gen == seen || id >= g_pool.use)) {
      if (g_pool.gen != seen && id >= g_pool.use) seen = g_pool.gen;      // a job this worker is not part of
      pthread_cond_wait(&g_pool.cv_work, &g_pool.mu);
    }
    if (g_pool.shutdown) break;
    seen = g_pool.gen;
    pthread_mutex_unlock(&g_pool.mu);
    pool_run_items();
    pthread_mutex_lock(&g_pool.mu);
    if (__atomic_sub_fetch(&g_pool.active, 1, __ATOMIC_ACQ_REL) == 0) pthread_cond_signal(&g_pool.cv_done);
  }
  pthread_mutex_unlock(&g_pool.mu);
  return NULL;
}

static pthread_mutex_t g_job_mu = PTHREAD_MUTEX_INITIALIZER;      // one job at a time: a process may run two loaders (training + validation)

// fork(): the child inherits g_pool.nth > 0 and possibly locked mutexes, but none of the worker threads -- its first job would wait
// for `active` to reach 0 for ever (multiprocessing's 'fork' start method, torch DataLoader workers).  The child starts
// with an empty pool and fresh locks; its workers are created by its own first job.
static void pool_atfork_child(void) {
  pthread_mutex_init(&g_pool.mu, NULL);
  pthread_cond_init(&g_pool.cv_work, NULL);
  pthread_cond_init(&g_pool.cv_done, NULL);
  pthread_mutex_init(&g_job_mu, NULL);
  g_pool.nth = 0;
  g_pool.gen = 0;
  g_pool.use = 0;
  g_pool.active = 0;
  g_pool.shutdown = 0;
}
static pthread_once_t g_atfork_once = PTHREAD_ONCE_INIT;
static void pool_register_atfork(void) { pthread_atfork(NULL, NULL, pool_atfork_child); }

static void pool_parallel_for(long items, int nthreads, item_fn fn, void* arg) {
  if (nthreads > POOL_MAX + 1) nthreads = POOL_MAX + 1;
  if (nthreads <= 1 || items <= 1) {
    for (long it = 0; it < items; ++it) fn(it, arg);
    return;
  }
  pthread_once(&g_atfork_once, pool_register_atfork);
  pthread_mutex_lock(&g_job_mu);
  pthread_mutex_lock(&g_pool.mu);
  while (g_pool.nth < nthreads - 1) {              // the caller is the last "thread" of the job
    if (pthread_create(&g_pool.th[g_pool.nth], NULL, pool_worker, (void*)(intptr_t)g_pool.nth) != 0) break;
    g_pool.nth++;
  }
  const int use = g_pool.nth < nthreads - 1 ? g_pool.nth : nthreads - 1;
  g_pool.fn = fn; g_pool.arg = arg; g_pool.items = items; g_pool.use = use;
  __atomic_store_n(&g_pool.next, 0, __ATOMIC_RELAXED);
  __atomic_store_n(&g_pool.active, use, __ATOMIC_RELEASE);
  g_pool.gen++;
  pthread_cond_broadcast(&g_pool.cv_work);
  pthread_mutex_unlock(&g_pool.mu);
  pool_run_items();
  pthread_mutex_lock(&g_pool.mu);
  while (__atomic_load_n(&g_pool.active, __ATOMIC_ACQUIRE) != 0) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
  pthread_mutex_unlock(&g_pool.mu);
  pthread_mutex_unlock(&g_job_mu);
}

// A whole batch in ONE call (round 4): rows idx[0..count) -> consecutive rows of dst, cut into pieces and spread over `nthreads`
// threads of the pool above.  The loader's Python thread pool spent more time handing pieces to threads (GIL hand-offs: 24 Python
// threads delivered HALF of what 8 did on a 256-thread host, profiles/r04_loader_*.json) than converting them; here the
// interpreter is out of the loop and the call releases the GIL for its whole duration.
typedef struct {
  const char* base; const int64_t* idx; size_t row_elems, per, piece, dst_pitch; char* dst; int convert;
} gather_job;

static void gather_item(long it, void* argp) {
  const gather_job* g = (const gather_job*)argp;
  const size_t j = (size_t)it / g->per, e0 = ((size_t)it % g->per) * g->piece;
  const size_t e1 = e0 + g->piece < g->row_elems ? e0 + g->piece : g->row_elems;
  if (g->convert)      // fp32 -> bf16: row_elems / piece / dst_pitch in ELEMENTS
    freud_f32_to_bf16((const float*)g->base + (size_t)g->idx[j] * g->row_elems + e0, (uint16_t*)g->dst + j * g->dst_pitch + e0, e1 - e0);
  else                 // plain copy: the same three in BYTES
    memcpy(g->dst + j * g->dst_pitch + e0, g->base + (size_t)g->idx[j] * g->row_elems + e0, e1 - e0);
}

void freud_gather_batch_f32_to_bf16(const float* base, const int64_t* idx, size_t count, size_t row_elems, uint16_t* dst,
                                    size_t dst_pitch, size_t piece, int nthreads) {
  if (piece == 0 || piece > row_elems) piece = row_elems;
  gather_job g = {(const char*)base, idx, row_elems, (row_elems + piece - 1) / piece, piece, dst_pitch, (char*)dst, 1};
  (void)pick_impl();
  pool_parallel_for((long)(count * g.per), nthreads, gather_item, &g);
}

// the same for shards that travel as they are (fp16 / bf16 shards, fp32 delivered natively): a parallel row gather
void freud_gather_batch_copy(const char* base, const int64_t* idx, size_t count, size_t row_bytes, char* dst, size_t dst_pitch_bytes,
                             size_t piece_bytes, int nthreads) {
  if (piece_bytes == 0 || piece_bytes > row_bytes) piece_bytes = row_bytes;
  gather_job g = {base, idx, row_bytes, (row_bytes + piece_bytes - 1) / piece_bytes, piece_bytes, dst_pitch_bytes, dst, 0};
  pool_parallel_for((long)(count * g.per), nthreads, gather_item, &g);
}

int freud_host_impl(void) { return pick_impl(); }
int freud_host_version(void) { return 4; }

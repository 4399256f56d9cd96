// libfreud_sae.so -- context management and the C ABI declared in include/freud_sae.h.
// Host orchestration of the gfx950 kernels for one SAE optimizer step
// (reference: src/scripts/train_sae.py:429-451).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "../../include/freud_sae.h"
#include "l1_kernels.h"
#include "gemm256.h"
#include "gemm256s.h"
#include "gemm256_fp8.h"
#include "l1_fp8.h"
#include "dp_kernels.h"
#include "p2p_exchange.h"
#include <rccl/rccl.h>
#include <mutex>
#include <unistd.h>

static thread_local bool g_force_gemm128 = false;   // (the CURRENT call's context: set by use_device, like g_device -- ADVICE r5)
                                                    // sae_config.force_gemm128: keep every GEMM on the 128x128 kernel (A/B timing, tests)
static thread_local bool g_no_stream = false;       // FREUD_GEMM_STREAM=0 / debug_flags 86: the K = d GEMMs in gemm256.h's tile form (A/B timing, tests)
static thread_local int g_device = 0;  // device of the context the current call works on (set by use_device)
#include "bwd_fused.h"
#include "fwd_fused.h"
#include "fwd_fused2.h"
#include "topk_kernels.h"
#include "topk_sparse.h"
#include "topk_aux.h"
#include "eval_fp32.h"

#ifndef G2_PERSIST_STATIC
#define G2_PERSIST_STATIC 512      // resident workgroups of the big static 256x256 GEMM launches (0: one workgroup per tile)
#endif

// ------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define NCCL_TRY(expr)                                                                                      \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess) return fail(SAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), \
                                       __FILE__, __LINE__);                                                 \
  } while (0)
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(SAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                              \
  } while (0)

// dynamic-LDS opt-in (> 64 KiB) is a per-device function attribute: remember (function, device) pairs
static int ensure_lds_attr(const void* fn, int bytes, int device) {
  static std::vector<std::pair<const void*, int>> done;
  static std::mutex mu;                     // contexts of several host threads share the cache
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& e : done)
    if (e.first == fn && e.second == device) return SAE_OK;
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.emplace_back(fn, device);
  return SAE_OK;
}
#define LDS_ATTR(fn, bytes, device)                                           \
  do {                                                                        \
    int rc_ = ensure_lds_attr(reinterpret_cast<const void*>(fn), (bytes), (device)); \
    if (rc_) return rc_;                                                      \
  } while (0)

// ------------------------------------------------------------------------------------------
// kernel ids for profiling
// ------------------------------------------------------------------------------------------
enum KernelId {
  KID_PREP_W = 0,
  KID_PREP_X,
  KID_ENC_FWD,
  KID_DEC_FWD,
  KID_FWD_FUSED,
  KID_DPRE,
  KID_DW,
  KID_BWD_FUSED,
  KID_REDUCE,
  KID_OPT,
  KID_TK_ENC,
  KID_TK_SELECT,
  KID_TK_DECODE,
  KID_TK_DDENSE,
  KID_TK_DWD,
  KID_TK_DWE,
  KID_TK_DSAE,
  KID_TK_AUX,
  KID_EXCHANGE,        // data parallel: the gradient exchange (peer-exchange kernel or ncclAllReduce), on the stream it runs on
  KID_STATS_XCHG,      // data parallel: the batch statistics (push inside finalize_losses, or the statistics kernel + its exchange)
  KID_STEP_TOTAL,
  KID_COUNT
};
static const char* kKernelNames[KID_COUNT] = {"prep_w", "prep_x", "enc_fwd_gemm", "dec_fwd_gemm", "fwd_fused_gemm", "dpre_gemm",
                                              "dw_gemm", "bwd_fused_gemm", "reduce_grads", "clip_adam", "topk_enc_gemm", "topk_select",
                                              "topk_decode", "topk_ddense_gemm", "topk_dwdec_gemm", "topk_dwenc_gemm",
                                              "topk_dsaein_colsum", "topk_auxk_backward", "dp_exchange", "dp_stats_exchange",
                                              "fwd_bwd_total"};
constexpr int EV_RING = 64;

struct EvRing {
  hipEvent_t beg[EV_RING], end[EV_RING];
  int n = 0;  // recorded since last read
};

struct sae_ctx {
  sae_config cfg;
  int d, n, d_p, n_p;
  int64_t max_rows_p;
  int64_t nW;       // d_p * n_p
  int64_t nparams;  // nW + n_p
  // flat parameter / state buffers: [W (d_p*n_p) | b (n_p)]
  float *P = nullptr, *Mom = nullptr, *Var = nullptr;
  float* G = nullptr;  // [grads (nparams) | metrics (8)]
  bf16_t *Wb = nullptr, *Wt = nullptr;
  // SAE_PREC_FP8: encoder / decoder GEMMs on e4m3 operands (l1_fp8.h); every padded dimension is a multiple of 256 then
  bool fp8 = false;
  bool fp8_bwd = false;         // SAE_PREC_FP8_BWD: the dpre GEMM of the backward on e4m3 operands too (dxh8 = e4m3(dx_hat s_g))
  unsigned char* dxh8 = nullptr;
  int row_pad = 128;            // M_p = round_up(M, row_pad)
  unsigned char *x8 = nullptr, *c8 = nullptr, *W8 = nullptr, *W8t = nullptr;
  float *scal8 = nullptr, *x8_part = nullptr;
  bool use_fused_fwd = false;
  int fwd_variant = 2;          // fused forward: 2 = fwd_fused2.h (decoder split along d), 1 = fwd_fused.h (FREUD_FWD=1: A/B, stamps)
  bf16_t *xb = nullptr, *c = nullptr, *dxh = nullptr, *dpre = nullptr;
  // generic L1 backward (round 5): dx_hat^T and x^T, [d_p][M_p] -- the weight-gradient GEMM's d-side operand K-contiguous, so that its
  // fragments come by ds_read_b128 instead of twice as many transposing reads (null: the k-major x k-major form)
  bf16_t *dxhT = nullptr, *xT = nullptr;
  float *slab = nullptr, *db_part = nullptr, *l1_part = nullptr, *sq_part = nullptr, *scal = nullptr;
  double* gn_part = nullptr;
  float* cn_part = nullptr;
  bool cn_valid = false;       // cn_part holds the column-norm partials of the CURRENT weights (left by optimizer_l1_kernel)
  bool wb_valid = false;       // TopK: We_b / Wd_b are the bf16 copies of the CURRENT weights (left by optimizer_kernel)
  // L1, d_p <= 384 (optimizer_l1_cols_kernel): the update left the fp32 master UN-normalised (the reference's state after
  // optimizer.step()), the column denominators in cnorm and the bf16 copies Wb / Wt of the NORMALISED weights.
  //   wn_pending              that state holds;
  //   wn_fwd_seen             ... and one forward has used the copies since: the reference's W is normalised by now
  //                           (l1autoencoder.py:71-73), ours is so in effect -- the next update divides while loading, any other
  //                           reader of the master (settle_weights) has the division carried out first.
  float* cnorm = nullptr;
  bool wn_pending = false, wn_fwd_seen = false;
  sae_grad_ready_fn grad_ready = nullptr;   // data-parallel overlap hook (sae_set_grad_ready_callback)
  void* grad_ready_user = nullptr;
  int dw_chunk_rows = 0;       // generic L1 path: rows (of d_p) per weight-gradient GEMM launch when a hook is set
  int dw_chunk_splits = 1;     // its split-K factor
  int dw_col_chunks = 1;       // ... with the peer exchange: COLUMN chunks of dW instead (each reads only its own columns of the
  int dw_col_splits = 1;       //     latent / dpre streams, where a row chunk re-reads all of them), and their split-K factor
  float* cnt_part = nullptr;   // fused forward: per-workgroup masked-entry counts
  int gn_blocks = 0;
  bool step_fused_call = false; // set by sae_step: forward_backward and optimizer_step back to back
  bool gn_valid = false;       // gn_part holds the sum of squares of the UNSCALED local gradient
  bool metrics_fresh = false;  // the loss scalars were written by a forward and not yet averaged by an optimizer step
  const bf16_t* xb_cur = nullptr;   // bf16 GEMM copy of the current batch (== the caller's x when no copy is needed)
  unsigned int* masked = nullptr;
  int dw_splits = 1;
  // plain (single-launch) weight-gradient GEMM of the generic L1 path: whole tiles written straight into the gradient, the tiles
  // left over after whole rounds in K pieces through dw_tail (gemm_tail_plan); 0 = uniform split-K through the slabs
  int dw_tail_tiles = 0, dw_tail_pieces = 0;
  float* dw_tail = nullptr;
  int bwd_splits = 1;       // row ranges of the fused backward
  int bwd_range_splits = 1; // ... when it is launched in column-tile ranges (bwd_ranges > 1)
  bool use_fused_bwd = false;
  // ---- TopK variant (topkautoencoder.py): flat params [We (n_p*d_p) | be (n_p) | Wd (n_p*d_p) | bd (d_p)]
  bool topk = false;
  int k = 0, k_aux_cap = 0;
  int64_t rows_per_file = 0;          // T: x is [B][T][d] for x.mean(0); 0 = the whole batch is one file
  double dead_threshold = 1e300;
  bf16_t *We_b = nullptr, *Wd_b = nullptr, *xs = nullptr, *pre = nullptr, *dense = nullptr, *aux_dense = nullptr;
  bf16_t *de_b = nullptr, *dh_b = nullptr;
  float *e = nullptr, *dh = nullptr, *e2_part = nullptr, *a2_part = nullptr, *dbd_part = nullptr, *ds_part = nullptr, *tkf = nullptr;
  int *top_idx = nullptr, *aux_idx = nullptr, *tk = nullptr;
  bf16_t *top_vals = nullptr, *aux_vals = nullptr, *multi_vals = nullptr;   // selected activations, compact [M_p][kcap]
  unsigned short* tile_max = nullptr;   // [M_p][n_p / 64] maxima of the pre-activation tiles (tile-driven select)
  unsigned char* sel_flag = nullptr;    // [M_p] rows the tile-driven select left to the general kernel
  bool dense_valid = false;     // the masked dense rows of the last forward were written (else: topk_densify on demand)
  bool multi_dense_valid = false;   // the same for the multi-TopK (4k) rows
  // AuxK on the compacted dead set (topk_aux.h)
  float* be_r = nullptr;            // encoder bias rounded to bf16 (as float), refreshed every step
  int slab_splits = 1;              // slabs allocated in c->slab (TopK)
  bool aux_compact = false;
  int *tkd = nullptr, *dead_cols = nullptr, *vec_rank = nullptr;
  unsigned char* vec_bits = nullptr;
  bf16_t* Wdd_b = nullptr;          // [n_p][d_p] rows of W_dec of the dead latents, compact
  float* aux_dbe_part = nullptr;    // [M_p / 128][n_p] column sums of the AuxK d pre-activations, compact columns
  int* dead_hint = nullptr;     // pinned host copy of tk[0] (number of dead latents), refreshed asynchronously every step
  double* tv_part = nullptr;
  long long* nfsf = nullptr;
  long long* dbe_fx = nullptr;  // [n_p] fixed-point d b_enc accumulator of the sparse d-activation kernel
  bool topk_sparse_da = false;  // sparse d pre-activations (topk_dacts_kernel) instead of the dense ddense GEMM + mask
  // CSC backward (topk_sparse.h): the selection sorted by latent, all three gradients as gathered weighted row sums
  bool topk_csc = false;
  unsigned short* csc_counts = nullptr;
  unsigned int *csc_block_off = nullptr, *csc_total = nullptr, *csc_start = nullptr, *csc_item_start = nullptr, *csc_item_latent = nullptr;
  unsigned int* csc_multi = nullptr;      // [0] = number of latents with no or several work items, [4 ...] = those latents (csc_items_kernel)
  CscEntry* csc_entries = nullptr;
  float *csc_part = nullptr, *csc_pbe = nullptr;
  int64_t csc_max_items = 0;
  // cfg.multi_topk (topkautoencoder.py:134-140): a second selection of 4k latents, its decode and its FVU / 8 in the loss
  bool multi = false;
  int k4 = 0;
  bf16_t *multi_dense = nullptr, *dm_b = nullptr;
  int* multi_idx = nullptr;
  float *em = nullptr, *m2_part = nullptr;
  unsigned char* dead = nullptr;
  // ---- data parallel (dp_kernels.h): batch statistics the losses normalise by, summed over the ranks
  double* stats = nullptr;      // [stats_cap] doubles
  int64_t stats_cap = 0, stats_n = 0;   // capacity / doubles the last sae_batch_stats wrote
  unsigned int* stats_part = nullptr;
  int dp_world = 0;             // > 0: forward_backward normalises by `stats` (which the host, or the engine's own RCCL
                                // communicator, has summed over dp_world ranks) instead of this rank's own batch
  // in-engine RCCL (sae_dist_init): statistics and gradient ranges are all-reduced on a communication stream
  bool dist = false;
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_x = nullptr, ev_stats = nullptr, ev_done = nullptr;
  hipEvent_t ev_range[16] = {};
  int ev_range_i = 0;
  int payload = SAE_DTYPE_F32;  // SAE_DTYPE_BF16: the fused d = 384 path all-reduces a bf16 copy of the gradient (sae_dist_set_payload)
  bf16_t* Gb = nullptr;         // that copy
  bool grads_in_bf16 = false;   // the current step's summed gradient lives in Gb (the optimizer step converts it back)
  int dist_error = 0;           // a collective of the current call failed to enqueue (message in dist_errmsg)
  char dist_errmsg[256] = "";
  // peer exchange over hipIpc mappings (p2p_exchange.h): the exchange of the in-engine protocol when sae_p2p_init was called
  bool p2p = false;
  int p2p_rank = 0;
  float* p2p_G[P2P_MAX_WORLD] = {};
  bf16_t* p2p_Gb[P2P_MAX_WORLD] = {};
  double* p2p_stats[P2P_MAX_WORLD] = {};
  unsigned long long* p2p_sig[P2P_MAX_WORLD] = {};
  void* p2p_opened[4 * P2P_MAX_WORLD] = {};      // peer mappings to close
  int p2p_nopened = 0;
  unsigned long long* sig = nullptr;             // this rank's flag block (uncached device memory)
  unsigned int* p2p_status = nullptr;            // device words: [0] sticky failure bits the exchange kernels set (P2P_ST_*), [1] self-test mismatches
  unsigned int* p2p_status_host = nullptr;       // host-mapped mirror of word 0 (hipHostMalloc): sae_dist_poll reads it without a sync
  unsigned int* p2p_status_hostdev = nullptr;    // its device address
  int p2p_fault_kind = 0, p2p_fault_rank = -1;   // FREUD_P2P_FAULT=skip_phase2:<rank>[:<after exchanges>] (tests: the guards must catch it)
  unsigned long long p2p_fault_after = 0;
  float* col_stage = nullptr;                    // in-engine RCCL, d >= 1024: contiguous staging of the dW column chunks ([d_p x n_p] floats)
  float* audit_dst = nullptr;                    // sae_dist_audit: every gradient exchange first copies its segments here
  bool finegrained = false;                      // FREUD_P2P_FINEGRAINED=1: G / Gb / stats in fine-grained device memory
  unsigned long long p2p_epoch[P2P_CHANNELS] = {};
  unsigned long long p2p_epoch_push = 0;         // epoch of the statistics push inside finalize_losses_kernel
  // 120 s of the 100 MHz clock (FREUD_P2P_TIMEOUT_MS): a LIVENESS bound, not a skew budget -- a rank may be late by a loader
  // stall, a validation pass or a checkpoint write; a failed context fails fast afterwards (sticky status, poisoned flags)
  unsigned long long p2p_timeout_ticks = 12000000000ull;
  bool gn_from_exchange = false;                 // gn_part holds the sum of squares of the EXCHANGED gradient
  int bal_m = 0;                // balanced fused backward (bwd_fused.h): quanta per workgroup (0 = uniform row ranges), workgroups,
  int bal_grid = 0;             //   and the blockIdx -> workgroup table (XCD-aware)
  short* bal_map = nullptr;
  int bwd_ranges = 1;           // fused d = 384 backward: column-tile ranges launched one after the other, each range reduced and
                                // exchanged on the communication stream under the next range's backward (sae_dist_set_overlap)
  // fp32 evaluation forward (eval_fp32.h; sae_set_eval_precision): buffers sized lazily for the largest evaluation batch seen
  bool stats_inline = true;     // fused d = 384 L1 path with the peer exchange: statistics pushed inside finalize_losses_kernel (see inline_stats)
  bool no_stream = false;       // this context's K = d GEMMs stay on the tile form (FREUD_GEMM_STREAM=0 / debug_flags 86 at ITS creation)
  int eval_prec = 0;            // 0 = the training kernels' arithmetic (bf16 operands, fp32 accumulate), 1 = fp32 end to end
  bool last_fwd_e32 = false;    // the last forward was an fp32 evaluation: its per-feature maxima live in e32_colmax
  int64_t e32_rows = 0;
  float *e32_x = nullptr, *e32_pre = nullptr, *e32_sel = nullptr, *e32_xhat = nullptr;
  double* e32_part = nullptr;
  int* e32_colmax = nullptr;
  int64_t step = 0;
  int64_t last_M = 0, last_M_p = 0;
  int last_dtype = 0;
  int profile = 0;
  int64_t prof_tick = 0;        // forward/backward calls since sae_profile (level 1 samples every prof_period-th)
  int prof_period = 8;
  EvRing ev[KID_COUNT];
  bool ev_init = false;
};

// fused d = 384 L1 path with the peer exchange: the batch statistics are exchanged inside finalize_losses_kernel (StatsPush)
// (FREUD_DP_STATS=stream: the statistics leave the critical path instead -- a pass over x and their exchange on the communication stream
// UNDER the weight preparation and the forward, which depend on x only: the step then has ONE cross-GPU round trip in line, the gradient
// exchange)
static bool inline_stats(const sae_ctx* c) { return c->p2p && c->use_fused_fwd && !c->topk && c->stats_inline; }

static int use_device(const sae_ctx* c) {
  HIP_TRY(hipSetDevice(c->cfg.device_id));
  g_device = c->cfg.device_id;
  g_force_gemm128 = c->cfg.force_gemm128 == 1;      // per-context launch choices travel with the call, not with the last sae_create
  g_no_stream = c->no_stream;
  return SAE_OK;
}
#define USE_DEVICE(c)          \
  do {                         \
    int rc_ = use_device(c);   \
    if (rc_) return rc_;       \
  } while (0)

static int dominant_kid(const sae_ctx* c) { return c->topk ? KID_TK_ENC : (c->use_fused_bwd ? KID_BWD_FUSED : KID_DW); }
// Level 1 brackets the dominant kernel and the whole step on every PROF_PERIOD-th step only: an event record is a packet of
// its own in the queue and costs ~6 us of idle GPU between two dependent kernels (kernel trace of the C2 step: 3 records
// per step were 3 % of it), so the timed region of bench.py samples instead of instrumenting every step (sae_profile_period).
static bool ev_on(const sae_ctx* c, int kid) {
  if (c->profile >= 2) return true;
  return c->profile == 1 && (kid == dominant_kid(c) || kid == KID_STEP_TOTAL) && (c->prof_tick % c->prof_period) == 0;
}
static void ev_begin(sae_ctx* c, int kid, hipStream_t s) {
  if (ev_on(c, kid)) {
    EvRing& r = c->ev[kid];
    (void)hipEventRecord(r.beg[r.n % EV_RING], s);   // a failure stays sticky and is reported by the hipGetLastError that ends the call
  }
}
static void ev_end(sae_ctx* c, int kid, hipStream_t s) {
  if (ev_on(c, kid)) {
    EvRing& r = c->ev[kid];
    (void)hipEventRecord(r.end[r.n % EV_RING], s);
    r.n++;
  }
}

extern "C" const char* sae_last_error(void) { return g_err; }
extern "C" int sae_version(void) { return 1; }
extern "C" const char* sae_kernel_name(int id) { return (id >= 0 && id < KID_COUNT) ? kKernelNames[id] : nullptr; }
extern "C" int sae_dominant_kernel(sae_ctx* c) { return c ? dominant_kid(c) : KID_DW; }

// ------------------------------------------------------------------------------------------
// TopK variant: buffers
// ------------------------------------------------------------------------------------------
// Split-K factor of a weight-gradient GEMM with an [r128 x c128] grid of 128x128 output tiles and `ktiles` K tiles: the
// launch runs in rounds of one workgroup per CU (256x256 kernel, both dimensions even) or two (128x128 kernel); pick
// the factor that minimises rounds / factor (plus a small price per slab for the reduction pass).
static int choose_splits(int r128, int c128, int64_t ktiles) {
  const bool big = !g_force_gemm128 && r128 % 2 == 0 && c128 % 2 == 0;
  const int tiles = big ? (r128 / 2) * (c128 / 2) : r128 * c128, slots = big ? 256 : 512;
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 16; ++sp) {
    if (sp > 1 && ktiles / sp < 8) break;
    const int rounds = (tiles * sp + slots - 1) / slots;
    const double cost = (double)rounds / sp + 0.04 * sp;
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = sp;
    }
  }
  return best;
}

// Row ranges ("splits") of the fused d = 384 backward for `ntiles` column tiles of 128 and `steps` 32-row steps: the launch has
// ntiles x splits workgroups, one per CU at a time (each wave owns its SIMD's register file), so it runs in
// ceil(ntiles x splits / 256) rounds of steps / splits steps each, plus a prologue + epilogue worth ~5 steps per workgroup,
// plus one more [384 x cols] fp32 slab written and read back per split.  256 / ntiles, the old rule, left CUs idle whenever
// ntiles does not divide 256 (n = 12 288: 96 x 2 = 192 workgroups on 256 CUs, a quarter of the backward).
static int fused_bwd_splits(int ntiles, int steps, int cols) {
  const double slab_steps = (double)cols * BF_D * 8.0 / 4.0e12 / 1.43e-6;      // slab write + read at ~4 TB/s, in 1.43 us steps
  int best = 1;
  double best_cost = 1e300;
  for (int sp = 1; sp <= 64 && sp <= steps; ++sp) {
    const int rounds = (ntiles * sp + 255) / 256;
    const double cost = rounds * ((double)((steps + sp - 1) / sp) + 5.0) + sp * slab_steps;
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = sp;
    }
  }
  return best;
}

// allocation of a buffer the peers of a data-parallel run map (G, Gb, stats)
static hipError_t peer_visible_malloc(const sae_ctx* c, void** p, size_t bytes) {
  if (c->finegrained) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
  return hipMalloc(p, bytes);
}

static int create_events(sae_ctx* c) {
  for (auto& r : c->ev)
    for (int i = 0; i < EV_RING; ++i) r.beg[i] = r.end[i] = nullptr;
  c->ev_init = true;               // from here on sae_destroy releases whatever was created
  for (auto& r : c->ev)
    for (int i = 0; i < EV_RING; ++i) {
      HIP_TRY(hipEventCreate(&r.beg[i]));
      HIP_TRY(hipEventCreate(&r.end[i]));
    }
  return SAE_OK;
}

static int topk_create(sae_ctx* c, int64_t Mp) {
  c->k = c->cfg.k;
  c->k_aux_cap = c->d / 2 > 0 ? c->d / 2 : 1;           // k_aux = x.shape[-1] // 2 (topkautoencoder.py:110)
  if (c->k_aux_cap > 1024) return fail(SAE_ERR_INVALID, "topk: d_model/2 = %d aux latents exceed the 1024 supported", c->k_aux_cap);
  c->nparams = 2 * c->nW + c->n_p + c->d_p;
  const int64_t ntail = SAE_NUM_METRICS + c->n_p;       // metrics + did_fire flags ride in the all-reduced buffer
  g_force_gemm128 = c->cfg.force_gemm128 == 1;
  c->no_stream = g_no_stream = c->cfg.debug_flags == 86 || (getenv("FREUD_GEMM_STREAM") && atoi(getenv("FREUD_GEMM_STREAM")) == 0);
  const int splits = choose_splits(c->n_p / 128, c->d_p / 128, Mp / 64);
  c->dw_splits = splits;
#define TALLOC(ptr, bytes)                                                                                   \
  do {                                                                                                       \
    hipError_t e_ = hipMalloc((void**)&(ptr), (size_t)(bytes));                                              \
    if (e_ != hipSuccess)                                                                                    \
      return fail(SAE_ERR_HIP, "hipMalloc(%lld bytes) for %s failed: %s", (long long)(bytes), #ptr, hipGetErrorString(e_)); \
  } while (0)
  TALLOC(c->P, c->nparams * 4);
  TALLOC(c->Mom, c->nparams * 4);
  TALLOC(c->Var, c->nparams * 4);
  {
    hipError_t e_ = peer_visible_malloc(c, (void**)&c->G, (size_t)(c->nparams + ntail) * 4);
    if (e_ != hipSuccess) return fail(SAE_ERR_HIP, "allocation of the gradient buffer failed: %s", hipGetErrorString(e_));
  }
  TALLOC(c->We_b, c->nW * 2);
  TALLOC(c->Wd_b, c->nW * 2);
  TALLOC(c->xs, Mp * c->d_p * 2);
  TALLOC(c->pre, Mp * c->n_p * 2);
  TALLOC(c->dense, Mp * c->n_p * 2);
  TALLOC(c->aux_dense, Mp * c->n_p * 2);
  TALLOC(c->dpre, Mp * c->n_p * 2);
  TALLOC(c->de_b, Mp * c->d_p * 2);
  TALLOC(c->dh_b, Mp * c->d_p * 2);
  TALLOC(c->e, Mp * c->d_p * 4);
  TALLOC(c->dh, Mp * c->d_p * 4);
  TALLOC(c->e2_part, Mp * 4);
  TALLOC(c->a2_part, Mp * 4);
  TALLOC(c->dbd_part, (Mp / 128 + 1) * c->d_p * 4);
  TALLOC(c->ds_part, (int64_t)((c->n_p + 63) / 64) * c->d_p * 4);
  TALLOC(c->db_part, (Mp / 128) * c->n_p * 4);
  TALLOC(c->tkf, 64);
  TALLOC(c->tk, 64);
  TALLOC(c->top_idx, Mp * c->k * 4);
  TALLOC(c->aux_idx, Mp * c->k_aux_cap * 4);
  TALLOC(c->top_vals, Mp * c->k * 2);
  TALLOC(c->tile_max, Mp * (c->n_p / TSEL_TILE) * 2);
  TALLOC(c->sel_flag, Mp);
  TALLOC(c->aux_vals, Mp * c->k_aux_cap * 2);
  HIP_TRY(hipHostMalloc((void**)&c->dead_hint, 64, hipHostMallocDefault));
  c->dead_hint[0] = 0;
  TALLOC(c->tv_part, ((Mp * c->d + 255) / 256 + 1) * 8);
  TALLOC(c->nfsf, (size_t)c->n_p * 8);
  TALLOC(c->dbe_fx, (size_t)c->n_p * 8);
  c->stats_cap = DP_STATS_HEAD + 2 * (int64_t)c->cfg.max_rows * c->d;     // column sums / sums of squares: at most one file
  {
    hipError_t e_ = peer_visible_malloc(c, (void**)&c->stats, (size_t)c->stats_cap * 8);
    if (e_ != hipSuccess) return fail(SAE_ERR_HIP, "allocation of the statistics buffer failed: %s", hipGetErrorString(e_));
  }
  // topk_dense_backward keeps the dense ddense GEMM (tests cover both)
  c->topk_sparse_da = (c->d_p == 384 || c->d_p == 768 || c->d_p == 1280) && c->cfg.topk_dense_backward != 1;
  TALLOC(c->dead, c->n_p);
  TALLOC(c->be_r, (size_t)c->n_p * 4);
  c->multi = c->cfg.multi_topk != 0;
  c->k4 = 4 * c->k;
  if (c->multi) {
    TALLOC(c->multi_dense, Mp * c->n_p * 2);
    TALLOC(c->multi_idx, Mp * c->k4 * 4);
    TALLOC(c->multi_vals, Mp * c->k4 * 2);
    TALLOC(c->em, Mp * c->d_p * 4);
    TALLOC(c->dm_b, Mp * c->d_p * 2);
    TALLOC(c->m2_part, Mp * 4);
  }
  // reserved switches: topk_dense_backward = 1 dense ddense GEMM, 2 = sparse d pre-activations + dense weight-gradient GEMMs
  c->topk_csc = c->topk_sparse_da && c->cfg.topk_dense_backward == 0;
  if (c->topk_csc) {
    const int64_t nb = (Mp + CSC_ROWS - 1) / CSC_ROWS;
    // (the AuxK entries enter the CSC lists only where the compact dead-set path does not apply)
    const bool auxc_will = c->cfg.auxk_alpha != 0.0 && c->n_p <= 2048 * 44 && c->cfg.debug_flags != 76;
    const int64_t emax = Mp * (int64_t)(c->k + (auxc_will ? 0 : c->k_aux_cap) + (c->multi ? c->k4 : 0));
    c->csc_max_items = c->n_p + emax / CSC_CHUNK + 1;
    TALLOC(c->csc_counts, nb * c->n_p * 2);
    TALLOC(c->csc_block_off, nb * c->n_p * 4);
    TALLOC(c->csc_total, (size_t)c->n_p * 4);
    TALLOC(c->csc_start, (size_t)(c->n_p + 1) * 4);
    TALLOC(c->csc_item_start, (size_t)(c->n_p + 1) * 4);
    TALLOC(c->csc_item_latent, (size_t)c->csc_max_items * 4);
    TALLOC(c->csc_multi, (size_t)(c->n_p + 4) * 4);
    TALLOC(c->csc_entries, (size_t)emax * sizeof(CscEntry));
    TALLOC(c->csc_part, (size_t)c->csc_max_items * 2 * c->d_p * 4);
    TALLOC(c->csc_pbe, (size_t)c->csc_max_items * 4);
  }
  // AuxK as dense GEMMs over the compacted dead latents: with the CSC main path and the register select kernel
  c->aux_compact = c->topk_csc && c->cfg.auxk_alpha != 0.0 && c->n_p <= 2048 * 44 && c->cfg.debug_flags != 76;
  if (c->aux_compact) {
    TALLOC(c->tkd, 64);
    TALLOC(c->dead_cols, (size_t)c->n_p * 4);
    TALLOC(c->vec_rank, (size_t)(c->n_p / 8) * 4);
    TALLOC(c->vec_bits, (size_t)vec_bits_t_offset(c->n_p) + VEC_BITS_T_BYTES);     // the table + its transposed copy (topk_aux.h)
    TALLOC(c->Wdd_b, c->nW * 2);
    TALLOC(c->aux_dbe_part, (Mp / 128) * c->n_p * 4);
    HIP_TRY(hipMemset(c->tkd, 0, 64));
  }
  // the multi-TopK weight gradient is a second GEMM launch into its own split-K slabs
  // (the AuxK weight-gradient GEMMs over a SMALL dead set split K up to 32 times to fill the chip: slabs [split][n_p][d_p])
  c->slab_splits = (splits > 1 ? splits : 1) * (c->multi ? 2 : 1);
  if (c->aux_compact && c->slab_splits < 32) c->slab_splits = 32;
  TALLOC(c->slab, (int64_t)c->slab_splits * c->nW * 4);
  TALLOC(c->gn_part, 1024 * 8);
#undef TALLOC
  HIP_TRY(hipMemset(c->P, 0, c->nparams * 4));
  HIP_TRY(hipMemset(c->Mom, 0, c->nparams * 4));
  HIP_TRY(hipMemset(c->Var, 0, c->nparams * 4));
  HIP_TRY(hipMemset(c->G, 0, (c->nparams + ntail) * 4));
  HIP_TRY(hipMemset(c->nfsf, 0, (size_t)c->n_p * 8));
  int rc_ev = create_events(c);
  if (rc_ev) return rc_ev;
  HIP_TRY(hipDeviceSynchronize());
  return SAE_OK;
}

extern "C" void sae_destroy(sae_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device_id);   // teardown: nothing useful can be done about a failure here
  void* ptrs[] = {c->P,    c->Mom,     c->Var,     c->G,       c->Wb,   c->Wt,      c->xb,    c->c, c->dxh, c->dxhT, c->xT,
                  c->dpre, c->slab,    c->db_part, c->l1_part, c->sq_part, c->scal, c->gn_part, c->masked, c->cn_part, c->cnt_part,
                  c->We_b, c->Wd_b, c->xs, c->pre, c->dense, c->aux_dense, c->de_b, c->dh_b, c->e, c->dh, c->e2_part,
                  c->a2_part, c->dbd_part, c->ds_part, c->tkf, c->top_idx, c->aux_idx, c->tk, c->tv_part, c->nfsf, c->dead, c->dbe_fx,
                  c->multi_dense, c->multi_idx, c->em, c->dm_b, c->m2_part, c->x8, c->c8, c->W8, c->W8t, c->scal8, c->x8_part, c->dxh8,
                  c->stats, c->stats_part, c->Gb, c->top_vals, c->aux_vals, c->multi_vals, c->tile_max, c->sel_flag, c->csc_counts, c->csc_block_off, c->csc_total, c->csc_start, c->csc_item_start,
                  c->csc_item_latent, c->csc_entries, c->csc_part, c->csc_pbe, c->tkd, c->dead_cols, c->vec_rank, c->vec_bits, c->Wdd_b,
                  c->aux_dbe_part, c->be_r, c->cnorm, c->dw_tail, c->csc_multi};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (c->dead_hint) (void)hipHostFree(c->dead_hint);
  for (int i = 0; i < c->p2p_nopened; ++i) (void)hipIpcCloseMemHandle(c->p2p_opened[i]);
  if (c->sig) (void)hipFree(c->sig);
  if (c->p2p_status) (void)hipFree(c->p2p_status);
  if (c->p2p_status_host) (void)hipHostFree(c->p2p_status_host);
  if (c->col_stage) (void)hipFree(c->col_stage);
  if (c->bal_map) (void)hipFree(c->bal_map);
  for (void* p : {(void*)c->e32_x, (void*)c->e32_pre, (void*)c->e32_sel, (void*)c->e32_xhat, (void*)c->e32_part, (void*)c->e32_colmax})
    if (p) (void)hipFree(p);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  for (hipEvent_t e : {c->ev_x, c->ev_stats, c->ev_done})
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_range)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_init)
    for (auto& r : c->ev)
      for (int i = 0; i < EV_RING; ++i) {
        if (r.beg[i]) (void)hipEventDestroy(r.beg[i]);
        if (r.end[i]) (void)hipEventDestroy(r.end[i]);
      }
  delete c;
}

extern "C" int sae_create(const sae_config* cfg, sae_ctx** out) {
  if (!cfg || !out) return fail(SAE_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->variant != SAE_VARIANT_L1 && cfg->variant != SAE_VARIANT_TOPK)
    return fail(SAE_ERR_INVALID, "Invalid autoencoder variant: %d, must be 'l1' or 'topk'", cfg->variant);
  if (cfg->variant == SAE_VARIANT_TOPK && (cfg->k <= 0 || cfg->k > cfg->n_dict || cfg->k > 1024))
    return fail(SAE_ERR_INVALID, "topk: k=%d must be in [1, min(n_dict, 1024)]", cfg->k);
  if (cfg->variant == SAE_VARIANT_TOPK && cfg->multi_topk && (4 * cfg->k > cfg->n_dict || 4 * cfg->k > 1024))
    return fail(SAE_ERR_INVALID, "topk: multi_topk selects 4k = %d latents, must be <= min(n_dict, 1024)", 4 * cfg->k);
  if (cfg->variant != SAE_VARIANT_TOPK && cfg->multi_topk)
    return fail(SAE_ERR_INVALID, "multi_topk is a TopK option");
  if (cfg->variant == SAE_VARIANT_TOPK && cfg->d_model > 1536)
    return fail(SAE_ERR_INVALID, "topk: d_model=%d above the 1536 the sparse decoder is built for", cfg->d_model);
  if (cfg->d_model <= 0 || cfg->n_dict <= 0 || cfg->max_rows <= 0)
    return fail(SAE_ERR_INVALID, "d_model, n_dict and max_rows must be positive");
  if (cfg->optimizer != SAE_OPT_RADAM && cfg->optimizer != SAE_OPT_ADAM)
    return fail(SAE_ERR_INVALID, "Invalid optimizer: %d, must be 'radam' or 'adam'", cfg->optimizer);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(SAE_ERR_INVALID, "device_id %d out of range", cfg->device_id);
  HIP_TRY(hipSetDevice(cfg->device_id));
  g_device = cfg->device_id;

  if (cfg->precision != SAE_PREC_BF16 && cfg->precision != SAE_PREC_FP8 && cfg->precision != SAE_PREC_FP8_BWD)
    return fail(SAE_ERR_INVALID, "Invalid precision: %d, must be SAE_PREC_BF16, SAE_PREC_FP8 or SAE_PREC_FP8_BWD", cfg->precision);
  if (cfg->precision != SAE_PREC_BF16 && cfg->variant != SAE_VARIANT_L1)
    return fail(SAE_ERR_INVALID, "fp8 encoder / decoder GEMMs exist for the L1 variant only");

  sae_ctx* c = new sae_ctx();
  c->cfg = *cfg;
  c->d = cfg->d_model;
  c->n = cfg->n_dict;
  c->fp8 = cfg->precision != SAE_PREC_BF16;
  c->fp8_bwd = cfg->precision == SAE_PREC_FP8_BWD;
  const int pad = c->fp8 ? 256 : 128;      // the fp8 GEMM has 256x256 tiles only
  c->row_pad = pad;
  c->d_p = (int)round_up(c->d, pad);
  c->n_p = (int)round_up(c->n, pad);
  c->max_rows_p = round_up(cfg->max_rows, pad);
  c->nW = (int64_t)c->d_p * c->n_p;
  c->nparams = c->nW + c->n_p;
  c->topk = cfg->variant == SAE_VARIANT_TOPK;
  // FREUD_P2P_FINEGRAINED=1: the three buffers the peers read (G, Gb, stats) in fine-grained device memory -- never held
  // non-coherently in a cache, so the peer exchange needs no cache maintenance to be correct.  The second in-engine option for a
  // node where the start-up self-test of the default (coarse-grained + fences) fails; costs local bandwidth.
  if (const char* fg = getenv("FREUD_P2P_FINEGRAINED")) c->finegrained = atoi(fg) == 1;
  const int64_t Mp = c->max_rows_p;
  if (c->topk) {
    int rc_tk = topk_create(c, Mp);
    if (rc_tk) {
      sae_destroy(c);
      return rc_tk;
    }
    *out = c;
    return SAE_OK;
  }
  g_force_gemm128 = cfg->force_gemm128 == 1;
  c->no_stream = g_no_stream = cfg->debug_flags == 86 || (getenv("FREUD_GEMM_STREAM") && atoi(getenv("FREUD_GEMM_STREAM")) == 0);
  c->dw_splits = choose_splits(c->d_p / 128, c->n_p / 128, 2 * Mp / 64);
  if (const char* ov = getenv("FREUD_DW_SPLITS")) {       // timing sweeps of the split-K factor of the weight-gradient GEMM
    const int v = atoi(ov);
    if (v >= 1 && v <= 64) c->dw_splits = v;
  }
  // with a gradient-ready hook the weight-gradient GEMM is issued in 512-row chunks (d_p >= 1024 only: smaller
  // models finish their gradient in one piece); the chunk's split-K factor keeps its launch rounds full
  c->dw_chunk_rows = c->d_p >= 1024 ? 512 : c->d_p;
  if (!g_force_gemm128 && c->d_p % 256 == 0 && c->n_p % 256 == 0 && (c->d_p / 256) * (c->n_p / 256) >= 256 && !getenv("FREUD_DW_SPLITS") &&
      c->cfg.debug_flags != 81) {
    gemm_tail_plan((c->d_p / 256) * (c->n_p / 256), (int)(2 * Mp / 64), c->dw_tail_tiles, c->dw_tail_pieces);
    if ((c->d_p / 256) * (c->n_p / 256) % 256 == 0) c->dw_tail_tiles = c->dw_tail_pieces = -1;     // whole rounds: one split, no tail
  }
  c->dw_chunk_splits = choose_splits(c->dw_chunk_rows / 128, c->n_p / 128, 2 * Mp / 64);
  if (c->d_p >= 1024 && c->n_p % 256 == 0)
    for (int q = 4; q >= 2; --q)
      if ((c->n_p / 256) % q == 0) {
        c->dw_col_chunks = q;
        break;
      }
  c->dw_col_splits = choose_splits(c->d_p / 128, c->n_p / 128 / c->dw_col_chunks, 2 * Mp / 64);
  // fused backward (bwd_fused.h) is specialised for a padded d_model of 384; force_generic forces the
  // generic three-GEMM path (used by the tests to cover both)
  c->use_fused_bwd = (c->d_p == BF_D) && cfg->force_generic != 1 && !c->fp8;
  c->use_fused_fwd = (c->d_p == FF_D) && cfg->force_generic != 1 && !c->fp8;
  if (const char* fv = getenv("FREUD_FWD")) c->fwd_variant = atoi(fv) == 1 ? 1 : 2;
  c->bwd_splits = fused_bwd_splits(c->n_p / BF_BN, (int)(Mp / BF_BM), c->n_p);
  // column-tile ranges (sae_dist_set_overlap, up to 4): a range of ntiles / r tiles is split into more row ranges
  c->bwd_range_splits = c->bwd_splits;
  for (int r = 2; r <= 4; ++r)
    if ((c->n_p / BF_BN) % r == 0) {
      const int sp = fused_bwd_splits(c->n_p / BF_BN / r, (int)(Mp / BF_BM), c->n_p / r);
      if (sp > c->bwd_range_splits) c->bwd_range_splits = sp;
    }
  int slab_splits = c->use_fused_bwd ? (c->bwd_range_splits > c->dw_splits ? c->bwd_range_splits : c->dw_splits) : c->dw_splits;
  if (c->use_fused_bwd && c->cfg.debug_flags != 83 && !getenv("FREUD_BWD_UNIFORM")) {     // (both: A/B switches for the uniform row ranges)
    // balanced form: 32 quanta per column tile, m per workgroup so that at most one workgroup per CU is needed
    const int ntiles = c->n_p / BF_BN, cus = 256;
    const int m = (32 * ntiles + cus - 1) / cus, grid = (32 * ntiles + m - 1) / m;
    int pmax = 1;
    for (int j = 0; j < ntiles; ++j) pmax = bal_pieces(j, m) > pmax ? bal_pieces(j, m) : pmax;
    // cost in 32-row steps at the largest batch, as in fused_bwd_splits: rows per workgroup + prologue / epilogue (+ one more of
    // each for a workgroup that changes tiles) + the slab pieces written and read back
    const int steps = (int)(Mp / BF_BM), q = (steps + 31) / 32;
    const double slab_steps = (double)c->n_p * BF_D * 8.0 / 4.0e12 / 1.43e-6;
    const double cost_bal = (double)m * q + 5.0 + (32 % m ? 3.5 : 0.0) + (32.0 / m + (32 % m ? 1.0 : 0.0)) * slab_steps;
    const int sp = c->bwd_splits, rounds = (ntiles * sp + 255) / 256;
    const double cost_uni = rounds * ((double)((steps + sp - 1) / sp) + 5.0) + sp * slab_steps;
    if (cost_bal < cost_uni * 0.995 && pmax <= 64 && grid <= 4096) {
      c->bal_m = m;
      c->bal_grid = grid;
      if (pmax > slab_splits) slab_splits = pmax;
      // blockIdx -> workgroup: workgroups of equal start phase (k m mod 32) walk the same rows at the same time; sorted by phase
      // and dealt to the XCDs in contiguous runs (blocks b and b + 8 share an XCD), an XCD streams as few row ranges as possible
      std::vector<int> order(grid);
      for (int k = 0; k < grid; ++k) order[k] = k;
      std::stable_sort(order.begin(), order.end(), [m](int a_, int b_) { return (a_ * m) % 32 < (b_ * m) % 32; });
      std::vector<short> map(grid);
      int pos = 0;
      for (int x = 0; x < 8; ++x) {
        const int len = grid / 8 + (x < grid % 8 ? 1 : 0);
        for (int i = 0; i < len; ++i) map[8 * i + x] = (short)order[pos++];
      }
      hipError_t e_ = hipMalloc((void**)&c->bal_map, (size_t)grid * 2);
      if (e_ == hipSuccess) e_ = hipMemcpy(c->bal_map, map.data(), (size_t)grid * 2, hipMemcpyHostToDevice);
      if (e_ != hipSuccess) {
        int rc_ = fail(SAE_ERR_HIP, "balanced backward: workgroup table: %s", hipGetErrorString(e_));
        sae_destroy(c);
        return rc_;
      }
    }
  }
  {   // a chunk launch writes splits x (chunk rows x n_p) floats at the chunk's row offset of each slab
    const int64_t need = (int64_t)(c->dw_chunk_splits > c->dw_col_splits ? c->dw_chunk_splits : c->dw_col_splits);
    if (need > slab_splits) slab_splits = (int)need;
  }
  const int64_t db_rows = (Mp / 128) > 64 ? (Mp / 128) : 64;

#define ALLOC(ptr, bytes)                                   \
  do {                                                      \
    hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));     \
    if (e_ != hipSuccess) {                                 \
      int rc_ = fail(SAE_ERR_HIP, "hipMalloc(%lld bytes) for %s failed: %s", (long long)(bytes), #ptr, \
                     hipGetErrorString(e_));                \
      sae_destroy(c);                                       \
      return rc_;                                           \
    }                                                       \
  } while (0)
  ALLOC(c->P, c->nparams * 4);
  ALLOC(c->Mom, c->nparams * 4);
  ALLOC(c->Var, c->nparams * 4);
  if (c->finegrained) {
    hipError_t e_ = peer_visible_malloc(c, (void**)&c->G, (size_t)(c->nparams + SAE_NUM_METRICS) * 4);
    if (e_ == hipSuccess) e_ = peer_visible_malloc(c, (void**)&c->stats, (size_t)DP_STATS_HEAD * 8);
    if (e_ != hipSuccess) {
      int rc_ = fail(SAE_ERR_HIP, "fine-grained allocation of the gradient / statistics buffers failed: %s", hipGetErrorString(e_));
      sae_destroy(c);
      return rc_;
    }
  } else {
    ALLOC(c->G, (c->nparams + SAE_NUM_METRICS) * 4);
  }
  ALLOC(c->Wb, c->nW * 2);
  ALLOC(c->Wt, c->nW * 2);
  ALLOC(c->xb, Mp * c->d_p * 2);
  ALLOC(c->c, Mp * c->n_p * 2 + 4096);   // + a dummy line the fused forward parks its first two stores on
  ALLOC(c->dxh, Mp * c->d_p * 2);
  ALLOC(c->dpre, Mp * c->n_p * 2);
  // (opt-in, FREUD_DW_ROWA=1: measured SLOWER in the engine -- C4 weight gradient 9.70 -> 10.75 ms, profiles/r05_ab_dw_row_a.txt --
  // although the stand-alone shape on random dense operands ran 10 % faster, profiles/r05_kbench_dw_row_a.txt)
  if (!c->use_fused_bwd && !g_force_gemm128 && c->d_p % 256 == 0 && c->n_p % 256 == 0 && Mp % 64 == 0 &&
      getenv("FREUD_DW_ROWA") && atoi(getenv("FREUD_DW_ROWA")) == 1) {
    ALLOC(c->dxhT, Mp * c->d_p * 2);
    ALLOC(c->xT, Mp * c->d_p * 2);
  }
  ALLOC(c->slab, (int64_t)slab_splits * c->nW * 4);
  ALLOC(c->db_part, db_rows * c->n_p * 4);
  ALLOC(c->l1_part, (Mp / 128) * (c->n_p / 128) * 4);
  ALLOC(c->sq_part, (Mp / 128) * (c->d_p / 128) * 2 * 4);
  ALLOC(c->scal, 16 * 4);
  ALLOC(c->gn_part, 1024 * 8);
  ALLOC(c->cn_part, (int64_t)(c->d_p / 32) * c->n_p * 4);
  ALLOC(c->cnorm, (int64_t)c->n_p * 4);
  // overflow buffer of the tail-split weight-gradient launches: at most one round of pieces (256 tiles of 256 x 256 fp32); the
  // round-sized column chunks of the data-parallel form (below) use it for their last, partial chunk as well
  if (c->dw_tail_tiles > 0 || (c->d_p >= 1024 && c->d_p % 256 == 0 && c->n_p % 256 == 0)) ALLOC(c->dw_tail, (int64_t)256 * 65536 * 4);
  ALLOC(c->masked, 2048 * 4);
  ALLOC(c->cnt_part, (Mp / 128 + 1) * 4);
  c->stats_cap = DP_STATS_HEAD;
  if (!c->stats) ALLOC(c->stats, c->stats_cap * 8);
  ALLOC(c->stats_part, 1024 * 4);
  if (c->fp8) {
    ALLOC(c->x8, Mp * c->d_p);
    ALLOC(c->c8, Mp * c->n_p);
    ALLOC(c->W8, c->nW);
    ALLOC(c->W8t, c->nW);
    ALLOC(c->scal8, S8_COUNT * 4);
    ALLOC(c->x8_part, 2 * 1024 * 4);
    if (c->fp8_bwd) ALLOC(c->dxh8, Mp * c->d_p);
  }
#undef ALLOC
  {
    int rc_init = [&]() -> int {
      HIP_TRY(hipMemset(c->P, 0, c->nparams * 4));
      HIP_TRY(hipMemset(c->Mom, 0, c->nparams * 4));
      HIP_TRY(hipMemset(c->Var, 0, c->nparams * 4));
      HIP_TRY(hipMemset(c->G, 0, (c->nparams + SAE_NUM_METRICS) * 4));
      int rc_ev = create_events(c);
      if (rc_ev) return rc_ev;
      // > 64 KiB dynamic LDS is opted in to lazily at the first launch of every kernel instantiation (LDS_ATTR)
      HIP_TRY(hipDeviceSynchronize());
      return SAE_OK;
    }();
    if (rc_init) {
      sae_destroy(c);
      return rc_init;
    }
  }
  *out = c;
  return SAE_OK;
}

// ------------------------------------------------------------------------------------------
// parameter / optimizer-state transfer (reference layouts <-> padded internal layout)
// ------------------------------------------------------------------------------------------
static int copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, int to_internal,
                  int is_device) {
  hipMemcpyKind kind = is_device ? hipMemcpyDeviceToDevice : (to_internal ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
  HIP_TRY(hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind));
  return SAE_OK;
}

static int xfer_flat(sae_ctx* c, float* internal, float* w_ext, float* b_ext, int to_internal, int is_device) {
  int rc;
  if (w_ext) {
    if (to_internal)
      rc = copy2d(internal, (size_t)c->n_p * 4, w_ext, (size_t)c->n * 4, (size_t)c->n * 4, c->d, 1, is_device);
    else
      rc = copy2d(w_ext, (size_t)c->n * 4, internal, (size_t)c->n_p * 4, (size_t)c->n * 4, c->d, 0, is_device);
    if (rc) return rc;
  }
  if (b_ext) {
    hipMemcpyKind kind = is_device ? hipMemcpyDeviceToDevice : (to_internal ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
    if (to_internal)
      HIP_TRY(hipMemcpy(internal + c->nW, b_ext, (size_t)c->n * 4, kind));
    else
      HIP_TRY(hipMemcpy(b_ext, internal + c->nW, (size_t)c->n * 4, kind));
  }
  return SAE_OK;
}

// TopK flat layout: [We [n_p][d_p] | be [n_p] | Wd [n_p][d_p] | bd [d_p]]; reference tensors We[n][d], be[n], Wd[n][d], bd[d]
static int xfer_flat_topk(sae_ctx* c, float* internal, float* const ext[4], int to_internal, int is_device) {
  const int64_t off[4] = {0, c->nW, c->nW + c->n_p, 2 * c->nW + c->n_p};
  hipMemcpyKind kind = is_device ? hipMemcpyDeviceToDevice : (to_internal ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) {
    if (!ext[i]) continue;
    float* in = internal + off[i];
    if (i == 0 || i == 2) {
      int rc = to_internal ? copy2d(in, (size_t)c->d_p * 4, ext[i], (size_t)c->d * 4, (size_t)c->d * 4, c->n, 1, is_device)
                           : copy2d(ext[i], (size_t)c->d * 4, in, (size_t)c->d_p * 4, (size_t)c->d * 4, c->n, 0, is_device);
      if (rc) return rc;
    } else {
      const size_t bytes = (size_t)(i == 1 ? c->n : c->d) * 4;
      if (to_internal) HIP_TRY(hipMemcpy(in, ext[i], bytes, kind));
      else HIP_TRY(hipMemcpy(ext[i], in, bytes, kind));
    }
  }
  return SAE_OK;
}

// Brings the fp32 master of an L1 context to what the reference holds at this point (see sae_ctx::wn_pending): if a forward
// has run since the last update, the in-place normalisation it stands for is carried out.  Afterwards the next forward takes
// the plain path (column norms + normalize_cast), exactly as if the folded update had never been used.
static void settle_weights(sae_ctx* c, hipStream_t s) {
  if (!c->wn_pending) return;
  if (c->wn_fwd_seen) {
    const int64_t n4 = c->nW / 4;
    int grid = (int)((n4 + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(normalize_inplace_kernel, dim3(grid), dim3(256), 0, s, c->P, c->cnorm, n4, c->n_p);
  }
  c->wn_pending = c->wn_fwd_seen = false;
  c->cn_valid = false;
}

extern "C" int sae_set_params(sae_ctx* c, const float* p0, const float* p1, const float* p2, const float* p3, int is_device) {
  if (!c || !p0 || !p1) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(c->P, 0, c->nparams * 4));
  c->cn_valid = false;
  c->wb_valid = false;
  c->wn_pending = c->wn_fwd_seen = false;
  if (c->topk) {
    if (!p2 || !p3) return fail(SAE_ERR_INVALID, "topk needs 4 parameter tensors");
    float* const ext[4] = {const_cast<float*>(p0), const_cast<float*>(p1), const_cast<float*>(p2), const_cast<float*>(p3)};
    return xfer_flat_topk(c, c->P, ext, 1, is_device);
  }
  return xfer_flat(c, c->P, const_cast<float*>(p0), const_cast<float*>(p1), 1, is_device);
}

extern "C" int sae_get_params(sae_ctx* c, float* p0, float* p1, float* p2, float* p3, int is_device) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  if (c->wn_pending) {
    settle_weights(c, nullptr);
    HIP_TRY(hipDeviceSynchronize());
  }
  if (c->topk) {
    float* const ext[4] = {p0, p1, p2, p3};
    return xfer_flat_topk(c, c->P, ext, 0, is_device);
  }
  return xfer_flat(c, c->P, p0, p1, 0, is_device);
}

extern "C" int sae_set_opt_state(sae_ctx* c, int64_t step, const float* const exp_avg[4], const float* const exp_avg_sq[4],
                                 int is_device) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  c->step = step;
  int rc = SAE_OK;
  if (c->topk) {
    if (exp_avg) {
      float* const ext[4] = {const_cast<float*>(exp_avg[0]), const_cast<float*>(exp_avg[1]), const_cast<float*>(exp_avg[2]), const_cast<float*>(exp_avg[3])};
      rc = xfer_flat_topk(c, c->Mom, ext, 1, is_device);
      if (rc) return rc;
    }
    if (exp_avg_sq) {
      float* const ext[4] = {const_cast<float*>(exp_avg_sq[0]), const_cast<float*>(exp_avg_sq[1]), const_cast<float*>(exp_avg_sq[2]), const_cast<float*>(exp_avg_sq[3])};
      rc = xfer_flat_topk(c, c->Var, ext, 1, is_device);
    }
    return rc;
  }
  if (exp_avg) rc = xfer_flat(c, c->Mom, const_cast<float*>(exp_avg[0]), const_cast<float*>(exp_avg[1]), 1, is_device);
  if (rc) return rc;
  if (exp_avg_sq) rc = xfer_flat(c, c->Var, const_cast<float*>(exp_avg_sq[0]), const_cast<float*>(exp_avg_sq[1]), 1, is_device);
  return rc;
}

extern "C" int sae_get_opt_state(sae_ctx* c, int64_t* step, float* const exp_avg[4], float* const exp_avg_sq[4],
                                 int is_device) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  if (step) *step = c->step;
  int rc = SAE_OK;
  if (c->topk) {
    if (exp_avg) rc = xfer_flat_topk(c, c->Mom, exp_avg, 0, is_device);
    if (rc) return rc;
    if (exp_avg_sq) rc = xfer_flat_topk(c, c->Var, exp_avg_sq, 0, is_device);
    return rc;
  }
  if (exp_avg) rc = xfer_flat(c, c->Mom, exp_avg[0], exp_avg[1], 0, is_device);
  if (rc) return rc;
  if (exp_avg_sq) rc = xfer_flat(c, c->Var, exp_avg_sq[0], exp_avg_sq[1], 0, is_device);
  return rc;
}

extern "C" int sae_set_grad_ready_callback(sae_ctx* c, sae_grad_ready_fn fn, void* user) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  c->grad_ready = fn;
  c->grad_ready_user = user;
  return SAE_OK;
}

// ------------------------------------------------------------------------------------------
// the exchange of the in-engine data-parallel protocol: peer exchange (p2p_exchange.h) or RCCL
// ------------------------------------------------------------------------------------------
static void dist_fail(sae_ctx* c, const char* what, const char* detail) {
  if (!c->dist_error) snprintf(c->dist_errmsg, sizeof(c->dist_errmsg), "%s: %s", what, detail ? detail : "");
  c->dist_error = 1;
}

static int p2p_grid(int64_t vectors) {
  int64_t g = vectors / 1024;
  if (g < 1) g = 1;
  if (g > 48) g = 48;
  return (int)g;
}

// enqueue one peer exchange of `nseg` segments on stream s (channel 0: statistics, doubles; channel 1: gradient buffer)
static void p2p_launch(sae_ctx* c, int channel, const P2PSeg* segs, int nseg, int grid, double* gn_part, hipStream_t s) {
  P2PArgs a{};
  for (int r = 0; r < c->dp_world; ++r) {
    a.buf[r] = channel == 0 ? (void*)c->p2p_stats[r] : (void*)c->p2p_G[r];
    a.bbuf[r] = c->p2p_Gb[r];
    a.sig[r] = c->p2p_sig[r];
  }
  for (int i = 0; i < nseg; ++i) a.seg[i] = segs[i];
  a.nseg = nseg; a.rank = c->p2p_rank; a.world = c->dp_world; a.channel = channel;
  a.epoch = ++c->p2p_epoch[channel];
  a.timeout_ticks = c->p2p_timeout_ticks;
  a.status = c->p2p_status;
  a.status_host = c->p2p_status_hostdev;
  a.gn_part = gn_part;
  if (channel == 1 && c->p2p_fault_kind && c->p2p_fault_rank == c->p2p_rank && a.epoch > c->p2p_fault_after) a.fault = c->p2p_fault_kind;
  if (channel == 1 && c->audit_dst) {        // sae_dist_audit: this rank's contribution as it stands before the exchange
    for (int i = 0; i < nseg; ++i) {
      const P2PSeg& g = segs[i];
      if (hipMemcpy2DAsync(c->audit_dst + g.off, (size_t)g.pitch * 4, c->G + g.off, (size_t)g.pitch * 4, (size_t)g.cols * 4, (size_t)g.rows,
                           hipMemcpyDeviceToDevice, s) != hipSuccess)
        dist_fail(c, "exchange audit", "snapshot copy failed");
    }
  }
  // (level-2 profile only: the exchange's own duration -- launch to last barrier, i.e. including the wait for the slowest peer --
  // on the stream it runs on; bench.py reports it as dp_timing.exchange_ms)
  const int kid = channel == 0 ? KID_STATS_XCHG : KID_EXCHANGE;
  ev_begin(c, kid, s);
  if (channel == 0) hipLaunchKernelGGL(p2p_allreduce_kernel<2>, dim3(grid), dim3(P2P_THREADS), 0, s, a);
  else hipLaunchKernelGGL(p2p_allreduce_kernel<4>, dim3(grid), dim3(P2P_THREADS), 0, s, a);
  ev_end(c, kid, s);
  if (hipGetLastError() != hipSuccess) dist_fail(c, "peer exchange launch", "hipLaunchKernel failed");
}

// all-reduce (sum) of the contiguous range [offset, offset + count) of the gradient buffer on stream s
static void exchange_range(sae_ctx* c, int64_t offset, int64_t count, hipStream_t s) {
  if (c->p2p) {
    P2PSeg g{};
    g.off = offset; g.pitch = count; g.rows = 1; g.cols = (int)count; g.kind = P2P_F32; g.in_norm = 0;
    if (count > 0x7ffffff0ll) {      // (cols is an int: cut very long ranges into rows of 2^20 elements + a tail segment)
      const int64_t rows = count >> 20;
      P2PSeg t = g;
      g.rows = (int)rows; g.cols = 1 << 20; g.pitch = 1 << 20;
      t.off = offset + (rows << 20); t.cols = (int)(count - (rows << 20)); t.pitch = t.cols;
      P2PSeg two[2] = {g, t};
      p2p_launch(c, 1, two, t.cols > 0 ? 2 : 1, p2p_grid(count / 4), nullptr, s);
      return;
    }
    p2p_launch(c, 1, &g, 1, p2p_grid(count / 4), nullptr, s);
    return;
  }
  ev_begin(c, KID_EXCHANGE, s);
  const ncclResult_t r = ncclAllReduce(c->G + offset, c->G + offset, (size_t)count, ncclFloat, ncclSum, c->comm, s);
  ev_end(c, KID_EXCHANGE, s);
  if (r != ncclSuccess) dist_fail(c, "ncclAllReduce of a gradient range", ncclGetErrorString(r));
}

// A contiguous range of the gradient buffer is final in stream order: tell the host (its own all-reduce), or -- with the
// engine's own communicator / peer mappings -- all-reduce it now on the communication stream, under the backward kernels still
// to come.  (The fused d = 384 backward announces its column ranges itself: fused_exchange.)
static inline void notify_grads(sae_ctx* c, int64_t offset, int64_t count, hipStream_t s) {
  if (count <= 0) return;
  if (c->dist && c->dp_world > 0) {
    hipEvent_t ev = c->ev_range[c->ev_range_i++ & 15];
    if (hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(c->comm_stream, ev, 0) != hipSuccess) {
      dist_fail(c, "gradient range hand-over", "event record / wait failed");
      return;
    }
    exchange_range(c, offset, count, c->comm_stream);
    return;
  }
  if (c->grad_ready) c->grad_ready(c->grad_ready_user, offset, count, (void*)s);
}

// the same for a 2-D block of the gradient buffer (rows x cols at element offset `off`, `pitch` elements between rows):
// peer exchange only (callers check c->p2p)
static inline void exchange_block(sae_ctx* c, int64_t off, int rows, int cols, int64_t pitch, hipStream_t s) {
  hipEvent_t ev = c->ev_range[c->ev_range_i++ & 15];
  if (hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(c->comm_stream, ev, 0) != hipSuccess) {
    dist_fail(c, "gradient block hand-over", "event record / wait failed");
    return;
  }
  P2PSeg g{};
  g.off = off; g.pitch = pitch; g.rows = rows; g.cols = cols; g.kind = P2P_F32; g.in_norm = 0;
  p2p_launch(c, 1, &g, 1, p2p_grid((int64_t)rows * (cols / 4)), nullptr, c->comm_stream);
}

// RCCL form of exchange_block: the block was summed into the contiguous `stage` ([rows x cols]); all-reduce it there and copy
// the sums into their strided place of the gradient buffer, both on the communication stream
static inline void exchange_staged(sae_ctx* c, float* stage, int64_t off, int rows, int cols, int64_t pitch, hipStream_t s) {
  hipEvent_t ev = c->ev_range[c->ev_range_i++ & 15];
  if (hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(c->comm_stream, ev, 0) != hipSuccess) {
    dist_fail(c, "gradient block hand-over", "event record / wait failed");
    return;
  }
  ev_begin(c, KID_EXCHANGE, c->comm_stream);
  const ncclResult_t r = ncclAllReduce(stage, stage, (size_t)rows * cols, ncclFloat, ncclSum, c->comm, c->comm_stream);
  if (r != ncclSuccess) {
    dist_fail(c, "ncclAllReduce of a staged gradient block", ncclGetErrorString(r));
    return;
  }
  if (hipMemcpy2DAsync(c->G + off, (size_t)pitch * 4, stage, (size_t)cols * 4, (size_t)cols * 4, (size_t)rows, hipMemcpyDeviceToDevice,
                       c->comm_stream) != hipSuccess)
    dist_fail(c, "staged gradient block", "copy back failed");
  ev_end(c, KID_EXCHANGE, c->comm_stream);
}

// Fused d = 384 backward, data parallel: column range [c0, c0 + cols) of dW and of db is final on stream s (`last`: with it
// the loss scalars).  One range = the whole gradient: exchanged in line on the COMPUTE stream (nothing is left to overlap
// with, and two event hops cost ~20 us of a 600 us step).  Several ranges: each goes to the communication stream and runs
// under the backward of the next range; the caller joins the streams after the last one.
static void fused_exchange(sae_ctx* c, int range, int nranges, int c0, int cols, bool last, hipStream_t s) {
  hipStream_t xs = s;
  if (nranges > 1) {
    hipEvent_t ev = c->ev_range[c->ev_range_i++ & 15];
    if (hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(c->comm_stream, ev, 0) != hipSuccess) {
      dist_fail(c, "gradient range hand-over", "event record / wait failed");
      return;
    }
    xs = c->comm_stream;
  }
  const int64_t tail = SAE_NUM_METRICS;
  if (c->p2p) {
    const int kind = c->payload == SAE_DTYPE_BF16 ? P2P_BF16 : P2P_F32;
    P2PSeg seg[3]{};
    int ns = 0;
    if (nranges == 1) { seg[ns].off = 0; seg[ns].pitch = c->nW; seg[ns].rows = 1; seg[ns].cols = (int)c->nW; }
    else { seg[ns].off = c0; seg[ns].pitch = c->n_p; seg[ns].rows = c->d_p; seg[ns].cols = cols; }
    seg[ns].kind = kind; seg[ns].in_norm = 1; ++ns;
    seg[ns].off = c->nW + c0; seg[ns].pitch = cols; seg[ns].rows = 1; seg[ns].cols = cols; seg[ns].kind = kind; seg[ns].in_norm = 1; ++ns;
    if (last) { seg[ns].off = c->nparams; seg[ns].pitch = tail; seg[ns].rows = 1; seg[ns].cols = (int)tail; seg[ns].kind = P2P_F32; seg[ns].in_norm = 0; ++ns; }
    const int grid = p2p_grid((int64_t)(c->d_p + 1) * (cols / 4));
    p2p_launch(c, 1, seg, ns, grid, c->gn_part + (int64_t)range * grid, xs);
    c->gn_blocks = nranges * grid;
    c->gn_from_exchange = true;
    return;
  }
  // RCCL: contiguous buffers only -- the whole gradient in one piece (sae_dist_set_overlap refuses ranges without peer mappings)
  if (c->payload == SAE_DTYPE_BF16) {     // bf16 copy of the parameters' gradient + the fp32 scalars, one RCCL group
    ev_begin(c, KID_EXCHANGE, xs);
    ncclResult_t r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclAllReduce(c->Gb, c->Gb, (size_t)c->nparams, ncclBfloat16, ncclSum, c->comm, xs);
    if (r == ncclSuccess) r = ncclAllReduce(c->G + c->nparams, c->G + c->nparams, (size_t)tail, ncclFloat, ncclSum, c->comm, xs);
    const ncclResult_t e = ncclGroupEnd();
    ev_end(c, KID_EXCHANGE, xs);
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) dist_fail(c, "ncclAllReduce of the bf16 gradient", ncclGetErrorString(r));
    c->grads_in_bf16 = true;
  } else {
    exchange_range(c, 0, c->nparams + tail, xs);
  }
}

extern "C" int sae_get_topk_state(sae_ctx* c, int64_t* out, int64_t n) {
  if (!c || !out) return fail(SAE_ERR_INVALID, "null argument");
  if (!c->topk) return fail(SAE_ERR_INVALID, "sae_get_topk_state: not a TopK context");
  if (n != c->n) return fail(SAE_ERR_INVALID, "sae_get_topk_state: n = %lld, context has %d latents", (long long)n, c->n);
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, c->nfsf, (size_t)n * 8, hipMemcpyDeviceToHost));
  return SAE_OK;
}

extern "C" int sae_set_topk_state(sae_ctx* c, const int64_t* in, int64_t n) {
  if (!c || !in) return fail(SAE_ERR_INVALID, "null argument");
  if (!c->topk) return fail(SAE_ERR_INVALID, "sae_set_topk_state: not a TopK context");
  if (n != c->n) return fail(SAE_ERR_INVALID, "sae_set_topk_state: n = %lld, context has %d latents", (long long)n, c->n);
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(c->nfsf, in, (size_t)n * 8, hipMemcpyHostToDevice));
  return SAE_OK;
}

extern "C" int sae_grad_buffer(sae_ctx* c, void** dev_ptr, int64_t* n_floats) {
  if (!c || !dev_ptr || !n_floats) return fail(SAE_ERR_INVALID, "null argument");
  *dev_ptr = c->G;
  *n_floats = c->nparams + SAE_NUM_METRICS + (c->topk ? c->n_p : 0);   // TopK: + did_fire flags (OR == sum > 0)
  return SAE_OK;
}

// ------------------------------------------------------------------------------------------
// data parallel: batch statistics (dp_kernels.h) and the engine's own RCCL communicator
// ------------------------------------------------------------------------------------------
template <typename T>
static int batch_stats_impl(sae_ctx* c, const T* x, int64_t M, hipStream_t s) {
  if (c->topk) {
    const int64_t T_rows = (c->rows_per_file > 0 && M % c->rows_per_file == 0) ? c->rows_per_file : M;
    const int B = (int)(M / T_rows);
    const int64_t TD = T_rows * c->d, need = DP_STATS_HEAD + 2 * TD;
    if (need > c->stats_cap) return fail(SAE_ERR_INVALID, "batch statistics need %lld doubles, capacity %lld", (long long)need, (long long)c->stats_cap);
    hipLaunchKernelGGL(dp_topk_stats_kernel<T>, dim3((unsigned)((TD + 255) / 256)), dim3(256), 0, s, x, B, TD, M, c->stats);
    c->stats_n = need;
  } else {
    const int64_t total = M * c->d;
    int grid = (int)((total + 256 * 16 - 1) / (256 * 16));
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(dp_count_masked_kernel<T>, dim3(grid), dim3(256), 0, s, x, total, c->stats_part);
    hipLaunchKernelGGL(dp_l1_stats_kernel, dim3(1), dim3(256), 0, s, c->stats_part, grid, M, c->d, c->stats);
    c->stats_n = DP_STATS_HEAD;
  }
  HIP_TRY(hipGetLastError());
  return SAE_OK;
}

static int batch_stats_dispatch(sae_ctx* c, const void* x, int64_t M, int x_dtype, hipStream_t s) {
  switch (x_dtype) {
    case SAE_DTYPE_F32: return batch_stats_impl<float>(c, (const float*)x, M, s);
    case SAE_DTYPE_F16: return batch_stats_impl<_Float16>(c, (const _Float16*)x, M, s);
    case SAE_DTYPE_BF16: return batch_stats_impl<bf16_t>(c, (const bf16_t*)x, M, s);
    default: return fail(SAE_ERR_INVALID, "unknown x_dtype %d", x_dtype);
  }
}

extern "C" int sae_batch_stats(sae_ctx* c, const void* x, int64_t M, int x_dtype, void* stream) {
  if (!c || !x) return fail(SAE_ERR_INVALID, "null argument");
  if (M <= 0 || M > c->cfg.max_rows) return fail(SAE_ERR_INVALID, "M=%lld outside (0, max_rows=%lld]", (long long)M, (long long)c->cfg.max_rows);
  USE_DEVICE(c);
  return batch_stats_dispatch(c, x, M, x_dtype, (hipStream_t)stream);
}

extern "C" int sae_stats_buffer(sae_ctx* c, void** dev_ptr, int64_t* n_doubles) {
  if (!c || !dev_ptr || !n_doubles) return fail(SAE_ERR_INVALID, "null argument");
  *dev_ptr = c->stats;
  *n_doubles = c->stats_n;
  return SAE_OK;
}

extern "C" int sae_set_dp_world(sae_ctx* c, int world) {
  if (!c || world < 0) return fail(SAE_ERR_INVALID, "bad argument");
  if (c->dist && world != c->dp_world) return fail(SAE_ERR_STATE, "the context owns a communicator of %d ranks", c->dp_world);
  c->dp_world = world;
  return SAE_OK;
}

static int dist_streams_create(sae_ctx* c) {
  if (c->comm_stream) return SAE_OK;
  HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_x, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_stats, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  for (auto& e : c->ev_range) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return SAE_OK;
}

extern "C" int sae_dist_unique_id(void* out, int64_t capacity) {
  if (!out || capacity < (int64_t)sizeof(ncclUniqueId)) return fail(SAE_ERR_INVALID, "need %d bytes", (int)sizeof(ncclUniqueId));
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memcpy(out, &id, sizeof(id));
  return SAE_OK;
}

extern "C" int sae_dist_init(sae_ctx* c, const void* unique_id, int64_t id_bytes, int rank, int world) {
  if (!c || !unique_id) return fail(SAE_ERR_INVALID, "null argument");
  if (id_bytes != (int64_t)sizeof(ncclUniqueId)) return fail(SAE_ERR_INVALID, "unique id must be %d bytes", (int)sizeof(ncclUniqueId));
  if (world < 1 || rank < 0 || rank >= world) return fail(SAE_ERR_INVALID, "rank %d / world %d", rank, world);
  if (c->dist) return fail(SAE_ERR_STATE, "communicator already initialised");
  USE_DEVICE(c);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  if (c->bwd_ranges > 1) return fail(SAE_ERR_STATE, "column ranges (sae_dist_set_overlap) need the peer exchange: RCCL sums contiguous buffers");
  NCCL_TRY(ncclCommInitRank(&c->comm, world, id, rank));
  {
    int rc_s = dist_streams_create(c);
    if (rc_s) return rc_s;
  }
  c->dist = true;
  c->dp_world = world;
  return SAE_OK;
}

extern "C" int sae_dist_world(sae_ctx* c) { return (c && c->dist) ? c->dp_world : 0; }

extern "C" int sae_dist_set_payload(sae_ctx* c, int dtype) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  if (dtype != SAE_DTYPE_F32 && dtype != SAE_DTYPE_BF16) return fail(SAE_ERR_INVALID, "payload must be SAE_DTYPE_F32 or SAE_DTYPE_BF16");
  USE_DEVICE(c);
  if (dtype == SAE_DTYPE_BF16 && !c->Gb) HIP_TRY(peer_visible_malloc(c, (void**)&c->Gb, (size_t)c->nparams * 2));
  c->payload = dtype;
  return SAE_OK;
}

// ---- peer exchange over hipIpc mappings (p2p_exchange.h) ----
struct P2PBlob {                      // what one rank publishes (sae_p2p_export) -- plain bytes, travels by any host channel
  hipIpcMemHandle_t G, Gb, stats, sig;
  int64_t n_grad, n_stats;            // floats in G, doubles in stats: must agree between the ranks
  int32_t pid, device;
  int32_t magic, pad_;
};
constexpr int32_t P2P_MAGIC = 0x50325031;

extern "C" int sae_p2p_blob_bytes(void) { return (int)sizeof(P2PBlob); }

extern "C" int sae_p2p_export(sae_ctx* c, void* out, int64_t capacity) {
  if (!c || !out) return fail(SAE_ERR_INVALID, "null argument");
  if (capacity < (int64_t)sizeof(P2PBlob)) return fail(SAE_ERR_INVALID, "need %d bytes", (int)sizeof(P2PBlob));
  if (c->dist) return fail(SAE_ERR_STATE, "the context already runs a data-parallel protocol");
  USE_DEVICE(c);
  if (!c->Gb) HIP_TRY(peer_visible_malloc(c, (void**)&c->Gb, (size_t)c->nparams * 2));
  if (!c->sig) {
    // flags the PEERS write and this rank polls: uncached device memory, so that a poll always reaches memory
    // [barrier flags of the exchange kernels | inbox of the statistics push: 2 parities x 8 sources x 4 words]
    HIP_TRY(hipExtMallocWithFlags((void**)&c->sig, (size_t)(P2P_SIG_WORDS + 64) * 8, hipDeviceMallocUncached));
    HIP_TRY(hipMemset(c->sig, 0, (size_t)(P2P_SIG_WORDS + 64) * 8));
    HIP_TRY(hipMalloc((void**)&c->p2p_status, 64));
    HIP_TRY(hipMemset(c->p2p_status, 0, 64));
    // a mirror of the failure word the host can read without touching the stream (sae_dist_poll); best effort: without it
    // the failure still shows at the next sae_dist_check
    if (hipHostMalloc((void**)&c->p2p_status_host, 64, hipHostMallocMapped) == hipSuccess) {
      memset(c->p2p_status_host, 0, 64);
      if (hipHostGetDevicePointer((void**)&c->p2p_status_hostdev, c->p2p_status_host, 0) != hipSuccess) c->p2p_status_hostdev = nullptr;
    } else {
      c->p2p_status_host = nullptr;
      (void)hipGetLastError();
    }
    HIP_TRY(hipDeviceSynchronize());
  }
  P2PBlob b{};
  HIP_TRY(hipIpcGetMemHandle(&b.G, c->G));
  HIP_TRY(hipIpcGetMemHandle(&b.Gb, c->Gb));
  HIP_TRY(hipIpcGetMemHandle(&b.stats, c->stats));
  HIP_TRY(hipIpcGetMemHandle(&b.sig, c->sig));
  b.n_grad = c->nparams + SAE_NUM_METRICS + (c->topk ? c->n_p : 0);
  b.n_stats = c->stats_cap;
  b.pid = (int32_t)getpid();
  b.device = c->cfg.device_id;
  b.magic = P2P_MAGIC;
  memcpy(out, &b, sizeof(b));
  return SAE_OK;
}

static int p2p_selftest(sae_ctx* c);
static void p2p_leave(sae_ctx* c) {
  c->p2p = false;
  c->dist = false;
  c->dp_world = 0;
  c->audit_dst = nullptr;
  (void)hipDeviceSynchronize();
  for (int i = 0; i < c->p2p_nopened; ++i) (void)hipIpcCloseMemHandle(c->p2p_opened[i]);
  c->p2p_nopened = 0;
  if (c->p2p_status) (void)hipMemset(c->p2p_status, 0, 64);
  if (c->p2p_status_host) c->p2p_status_host[0] = 0;
  // (the flag block keeps its epochs: a later sae_p2p_init continues the counters, never reuses a value)
}

extern "C" int sae_p2p_init(sae_ctx* c, const void* blobs, int64_t bytes_per_rank, int rank, int world) {
  if (!c || !blobs) return fail(SAE_ERR_INVALID, "null argument");
  if (bytes_per_rank != (int64_t)sizeof(P2PBlob)) return fail(SAE_ERR_INVALID, "blob must be %d bytes per rank", (int)sizeof(P2PBlob));
  if (world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world)
    return fail(SAE_ERR_INVALID, "rank %d / world %d (the peer exchange serves up to %d ranks of one node)", rank, world, P2P_MAX_WORLD);
  if (c->dist) return fail(SAE_ERR_STATE, "the context already runs a data-parallel protocol");
  if (!c->sig) return fail(SAE_ERR_STATE, "sae_p2p_export must be called first");
  USE_DEVICE(c);
  const P2PBlob* all = reinterpret_cast<const P2PBlob*>(blobs);
  const int64_t n_grad = c->nparams + SAE_NUM_METRICS + (c->topk ? c->n_p : 0);
  for (int r = 0; r < world; ++r) {
    if (all[r].magic != P2P_MAGIC) return fail(SAE_ERR_INVALID, "rank %d: not a sae_p2p_export blob", r);
    if (all[r].n_grad != n_grad || all[r].n_stats != c->stats_cap)
      return fail(SAE_ERR_INVALID, "rank %d runs a different model (gradient buffer %lld floats, here %lld)", r,
                  (long long)all[r].n_grad, (long long)n_grad);
  }
  int rc = dist_streams_create(c);
  if (rc) return rc;
  for (int r = 0; r < world; ++r) {
    if (r == rank) {
      c->p2p_G[r] = c->G; c->p2p_Gb[r] = c->Gb; c->p2p_stats[r] = c->stats; c->p2p_sig[r] = c->sig;
      continue;
    }
    if (all[r].pid == (int32_t)getpid()) return fail(SAE_ERR_INVALID, "rank %d lives in this process: one process per rank", r);
    void* m[4] = {};
    const hipIpcMemHandle_t* h[4] = {&all[r].G, &all[r].Gb, &all[r].stats, &all[r].sig};
    for (int i = 0; i < 4; ++i) {
      hipError_t e = hipIpcOpenMemHandle(&m[i], *h[i], hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) return fail(SAE_ERR_HIP, "hipIpcOpenMemHandle (rank %d, buffer %d) failed: %s", r, i, hipGetErrorString(e));
      c->p2p_opened[c->p2p_nopened++] = m[i];
    }
    c->p2p_G[r] = (float*)m[0]; c->p2p_Gb[r] = (bf16_t*)m[1]; c->p2p_stats[r] = (double*)m[2]; c->p2p_sig[r] = (unsigned long long*)m[3];
  }
  if (const char* t = getenv("FREUD_P2P_TIMEOUT_MS")) {
    const long ms = atol(t);
    if (ms > 0) c->p2p_timeout_ticks = (unsigned long long)ms * 100000ull;
  }
  if (const char* f = getenv("FREUD_P2P_FAULT")) {     // test hook: "skip_phase2:<rank>[:<after this many gradient exchanges>]"
    int fr = -1;
    long long after = 0;
    if (sscanf(f, "skip_phase2:%d:%lld", &fr, &after) >= 1) {
      c->p2p_fault_kind = 1;
      c->p2p_fault_rank = fr;
      c->p2p_fault_after = (unsigned long long)after;
    }
  }
  if (const char* m = getenv("FREUD_DP_STATS")) c->stats_inline = strcmp(m, "stream") != 0;      // "inline" (default) | "stream"
  c->p2p = true;
  c->p2p_rank = rank;
  c->dist = true;
  c->dp_world = world;
  rc = p2p_selftest(c);
  if (rc) p2p_leave(c);     // the caller may fall back to another exchange with this context
  return rc;
}

// Leave the peer exchange again (a peer's self-test failed although this rank's passed: every rank must fall back together).
extern "C" int sae_p2p_leave(sae_ctx* c) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  if (!c->p2p) return SAE_OK;
  USE_DEVICE(c);
  p2p_leave(c);
  return SAE_OK;
}

// Start-up self-test of the peer exchange (collective: every rank runs it inside sae_p2p_init).  P2P_SELFTEST_EPOCHS exchanges
// of EACH payload form over the SAME addresses, every one with a different rank-dependent pattern (p2p_exchange.h), checked on
// the device after every exchange:
//   kind 0  fp32, the whole parameter gradient as one contiguous segment (the fused d = 384 path)
//   kind 1  bf16 payload of the same segment (bf16 copies read from the peers, fp32 results)
//   kind 2  fp32, a 2-D strided block (rows x cols at an offset, pitch n_p: the column chunks of the d >= 1024 path); the
//           elements OUTSIDE the block must keep this rank's own values
//   kind 3  fp64 on the statistics channel (its own flags and epochs)
//   kind 4  the statistics push through the inbox (finalize_losses_kernel's StatsPush)
// A peer line that survives in a cache from one exchange to the next, a flag that overtakes its data, a mapping to the wrong
// buffer: each gives wrong sums here, before the first training step.  A rank that cannot reach its peers times out (status
// word) instead of corrupting a run.  FREUD_P2P_FAULT=skip_phase2:<rank> is caught here (tests/test_dp_gpu.py).
constexpr int P2P_SELFTEST_EPOCHS = 4;
static int p2p_selftest_body(sae_ctx* c) {
  hipStream_t s = c->comm_stream;
  const int64_t n = c->nparams;
  const int64_t ns = c->stats_cap < 4096 ? c->stats_cap : 4096;        // doubles of the statistics buffer exercised (>= DP_STATS_HEAD)
  std::vector<float> keep((size_t)n);
  std::vector<double> keep_stats((size_t)ns);
  HIP_TRY(hipMemcpy(keep.data(), c->G, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(keep_stats.data(), c->stats, (size_t)ns * 8, hipMemcpyDeviceToHost));
  unsigned int bad_total = 0, bad_kind[5] = {};
  // whatever way this function is left, the gradient tail and the statistics head get back what they held: a context that falls
  // back to RCCL or the host carrier must not inherit self-test patterns
  struct Restore {
    sae_ctx* c; std::vector<float>& g; std::vector<double>& st;
    ~Restore() {
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(c->G, g.data(), g.size() * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(c->stats, st.data(), st.size() * 8, hipMemcpyHostToDevice);
    }
  } restore{c, keep, keep_stats};
  auto collect = [&](int kind) -> int {
    unsigned int st[2] = {};
    HIP_TRY(hipMemcpyAsync(st, c->p2p_status, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (st[0] & P2P_ST_TIMEOUT) return fail(SAE_ERR_HIP, "peer exchange self-test: a peer did not arrive within %.1f s", c->p2p_timeout_ticks / 1e8);
    if (st[0] & P2P_ST_POISONED) return fail(SAE_ERR_HIP, "peer exchange self-test: a peer left the protocol (its self-test failed or timed out)");
    bad_total += st[1];
    bad_kind[kind] += st[1];
    return SAE_OK;
  };
  // the 2-D block: the second quarter of the columns of every W row (L1: [d_p x n_p]; TopK: the first [d_p rows] of the flat buffer
  // read with pitch n_p -- the exchange kernel does not care what the floats mean)
  P2PSeg blk{};
  blk.pitch = c->n_p; blk.rows = (int)((n / c->n_p) < c->d_p ? (n / c->n_p) : c->d_p); blk.cols = (c->n_p / 4) & ~3; blk.off = blk.cols;
  blk.kind = P2P_F32; blk.in_norm = 0;
  for (int e = 0; e < P2P_SELFTEST_EPOCHS; ++e) {
    for (int kind = 0; kind < 5; ++kind) {
      HIP_TRY(hipMemsetAsync(c->p2p_status + 1, 0, 4, s));
      if (kind <= 2) {
        hipLaunchKernelGGL(p2p_selftest_fill_kernel, dim3(256), dim3(256), 0, s, c->G, c->Gb, n, c->p2p_rank, e, kind);
        P2PSeg g{};
        g.off = 0; g.pitch = n; g.rows = 1; g.cols = (int)n; g.kind = kind == 1 ? P2P_BF16 : P2P_F32; g.in_norm = 1;
        if (kind == 2) g = blk;
        if (kind == 2 && (blk.cols < 4 || blk.rows < 1)) continue;
        p2p_launch(c, 1, &g, 1, p2p_grid((int64_t)g.rows * (g.cols / 4)), kind == 2 ? nullptr : c->gn_part, s);
        hipLaunchKernelGGL(p2p_selftest_check_kernel, dim3(256), dim3(256), 0, s, c->G, n, g, c->p2p_rank, c->dp_world, e, kind, c->p2p_status + 1);
      } else if (kind == 3) {
        hipLaunchKernelGGL(p2p_selftest_fill64_kernel, dim3(16), dim3(256), 0, s, c->stats, ns, c->p2p_rank, e, kind);
        P2PSeg g{};
        g.off = 0; g.pitch = ns; g.rows = 1; g.cols = (int)(ns & ~1ll); g.kind = P2P_F64;
        p2p_launch(c, 0, &g, 1, p2p_grid(g.cols / 2), nullptr, s);
        hipLaunchKernelGGL(p2p_selftest_check64_kernel, dim3(16), dim3(256), 0, s, c->stats, (int64_t)g.cols, c->dp_world, e, kind, c->p2p_status + 1);
      } else {
        StatsPush push{};
        for (int r = 0; r < c->dp_world; ++r) push.inbox[r] = c->p2p_sig[r] + P2P_SIG_WORDS;
        push.rank = c->p2p_rank; push.world = c->dp_world; push.epoch = ++c->p2p_epoch_push;
        push.timeout_ticks = c->p2p_timeout_ticks; push.status = c->p2p_status; push.status_host = c->p2p_status_hostdev; push.gstats_out = nullptr;
        hipLaunchKernelGGL(p2p_selftest_push_kernel, dim3(1), dim3(64), 0, s, push, e, c->p2p_status + 1);
      }
      HIP_TRY(hipGetLastError());
      const int rc = collect(kind);
      if (rc) return rc;
    }
  }
  if (c->dist_error) return fail(SAE_ERR_HIP, "peer exchange self-test: %s", c->dist_errmsg);
  if (bad_total) {
    // (a rank whose own sums were right must not go on with one whose sums were wrong: the ranks agree on the outcome over the
    // host channel -- freud_amd/dp.py: _all_agree -- and the ones that passed leave again with sae_p2p_leave)
    return fail(SAE_ERR_HIP, "peer exchange self-test: %u wrong values (fp32 %u, bf16 payload %u, strided block %u, fp64 statistics %u, statistics push %u) "
                             "in %d exchanges per payload form", bad_total, bad_kind[0], bad_kind[1], bad_kind[2], bad_kind[3], bad_kind[4], P2P_SELFTEST_EPOCHS);
  }
  return SAE_OK;
}
// The ranks enter the self-test right after the hand-shake that exchanged their handles, i.e. within milliseconds of each other:
// a peer whose flags do not arrive within 40 s never will (mappings that do not reach the other device), and the caller falls back
// to another exchange that much sooner than after the run-time limit (FREUD_P2P_TIMEOUT_MS, 120 s: a step may wait for a peer
// that writes a checkpoint).
static int p2p_selftest(sae_ctx* c) {
  const unsigned long long keep = c->p2p_timeout_ticks;
  const unsigned long long limit = 4000000000ull;            // 40 s of the 100 MHz counter
  if (c->p2p_timeout_ticks > limit) c->p2p_timeout_ticks = limit;
  const int rc = p2p_selftest_body(c);
  c->p2p_timeout_ticks = keep;
  return rc;
}

extern "C" int sae_dist_set_overlap(sae_ctx* c, int nranges) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  if (nranges < 1 || nranges > 4) return fail(SAE_ERR_INVALID, "1 to 4 column ranges");
  if (nranges > 1 && c->dist && !c->p2p) return fail(SAE_ERR_STATE, "column ranges need the peer exchange (RCCL sums contiguous buffers)");
  if (nranges > 1 && !(c->use_fused_bwd && !c->topk)) return fail(SAE_ERR_INVALID, "column ranges belong to the fused d = 384 backward");
  if ((c->n_p / BF_BN) % nranges != 0) return fail(SAE_ERR_INVALID, "%d column tiles do not divide into %d ranges", c->n_p / BF_BN, nranges);
  c->bwd_ranges = nranges;
  if (nranges > 1)     // (the slab buffer was sized for the largest of these at creation)
    c->bwd_range_splits = fused_bwd_splits(c->n_p / BF_BN / nranges, (int)(c->max_rows_p / BF_BM), c->n_p / nranges);
  return SAE_OK;
}

static int p2p_status_error(sae_ctx* c, unsigned int st) {
  if (st & P2P_ST_TIMEOUT)
    return fail(SAE_ERR_HIP, "peer exchange: a peer did not arrive within %.1f s (the replicas are out of step: results after the last "
                             "successful check are invalid)", c->p2p_timeout_ticks / 1e8);
  if (st & P2P_ST_POISONED) return fail(SAE_ERR_HIP, "peer exchange: a peer left the protocol (timed out or failed); results after the last successful check are invalid");
  return fail(SAE_ERR_HIP, "peer exchange: failure word %u", st);
}

// Synchronises the context's streams and reports a failure of the in-engine exchange (a peer that never arrived: the
// exchange kernels give up after the timeout instead of hanging the GPU).
extern "C" int sae_dist_check(sae_ctx* c) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  if (!c->dist) return SAE_OK;
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  if (c->p2p) {
    unsigned int st = 0;
    HIP_TRY(hipMemcpy(&st, c->p2p_status, 4, hipMemcpyDeviceToHost));
    if (st) return p2p_status_error(c, st);
  }
  if (c->dist_error) return fail(SAE_ERR_HIP, "%s", c->dist_errmsg);
  return SAE_OK;
}

// layout of the gradient buffer: out[0] floats of parameter gradients, out[1] loss scalars behind them (SAE_NUM_METRICS; some
// are written AFTER the exchange, e.g. the clipped gradient norm), out[2] did_fire flags behind those (TopK)
extern "C" int sae_grad_layout(sae_ctx* c, int64_t out[3]) {
  if (!c || !out) return fail(SAE_ERR_INVALID, "null argument");
  out[0] = c->nparams;
  out[1] = SAE_NUM_METRICS;
  out[2] = c->topk ? c->n_p : 0;
  return SAE_OK;
}

extern "C" int sae_dist_audit(sae_ctx* c, float* snapshot_dev) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  if (snapshot_dev && !c->p2p) return fail(SAE_ERR_STATE, "the audit snapshots the peer exchange's inputs: call sae_p2p_init first");
  c->audit_dst = snapshot_dev;
  return SAE_OK;
}

extern "C" int sae_param_checksum(sae_ctx* c, uint64_t out[4]) {
  if (!c || !out) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  // (the raw buffers as they stand: with the folded weight preparation of the L1 path the master holds the un-normalised update
  // on EVERY replica alike -- nothing is settled here, the call does not change what the next step launches)
  unsigned long long* dev = nullptr;
  HIP_TRY(hipMalloc((void**)&dev, 32));
  hipError_t e = hipMemset(dev, 0, 32);
  const float* bufs[3] = {c->P, c->Mom, c->Var};
  for (int i = 0; i < 3 && e == hipSuccess; ++i) {
    hipLaunchKernelGGL(checksum_kernel, dim3(512), dim3(256), 0, nullptr, bufs[i], c->nparams, dev + i);
    e = hipGetLastError();
  }
  unsigned long long h[4] = {};
  if (e == hipSuccess) e = hipMemcpy(h, dev, 24, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(SAE_ERR_HIP, "parameter checksum failed: %s", hipGetErrorString(e));
  h[3] = (unsigned long long)c->step;
  for (int i = 0; i < 4; ++i) out[i] = h[i];
  return SAE_OK;
}

// The same check WITHOUT synchronising: reads the host-mapped mirror of the failure word (the exchange kernels store it there
// when they give up).  Free to call after every step; a failure it does not see yet shows at the next sae_dist_check.
extern "C" int sae_dist_poll(sae_ctx* c) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  if (c->dist_error) return fail(SAE_ERR_HIP, "%s", c->dist_errmsg);
  if (c->p2p && c->p2p_status_host) {
    const unsigned int st = *(volatile unsigned int*)c->p2p_status_host;
    if (st) return p2p_status_error(c, st);
  }
  return SAE_OK;
}

// ------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------
// true: launch_gemm runs this GEMM in the streaming form of gemm256s.h (the K = d GEMMs: row-major x row-major, one K segment,
// thousands of static output tiles) -- callers that size a per-workgroup output of the functor ask first
template <int AM, int BM_, class Epi>
static bool gemm_streams(const GemmArgs& g) {
  if constexpr (!(G2_STREAM && epi_stream<Epi>::value && AM == OP_ROW && BM_ == OP_ROW)) return false;
  if (g_no_stream || g_force_gemm128 || g.nbm % 2 != 0 || g.nbn % 2 != 0) return false;
  if (G2S_STATIC && g.ktiles % 2 != 0) return false;       // (the static-stage K loop walks the K tiles in pairs: gemm256s.h)
  return !g.dyn && g.splits == 1 && g.tail_tiles == 0 && g.ktiles == g.ktiles0 && g.ktiles >= 2 && g.seg1_gate == nullptr && g.lda == g.ldb &&
         (g.nbm / 2) * (g.nbn / 2) >= 4 * G2_PERSIST_STATIC;
}

template <int AM, int BM_, class Epi>
static int launch_gemm(const GemmArgs& g, const Epi& epi, hipStream_t s) {
  if constexpr (G2_STREAM && epi_stream<Epi>::value && AM == OP_ROW && BM_ == OP_ROW) {
    if (gemm_streams<AM, BM_, Epi>(g)) {
      auto kerns = gemm256s_bf16_kernel<Epi>;
      LDS_ATTR(kerns, G2S_LDS_BYTES, g_device);
      GemmArgs g2 = g;
      g2.nbm = g.nbm / 2;
      g2.nbn = g.nbn / 2;
      hipLaunchKernelGGL(kerns, dim3(G2_PERSIST_STATIC), dim3(512), G2S_LDS_BYTES, s, g2, epi);
      HIP_TRY(hipGetLastError());
      return SAE_OK;
    }
  }
  if (!g_force_gemm128 && g.nbm % 2 == 0 && g.nbn % 2 == 0) {   // both output dimensions are multiples of 256
    auto kern256 = gemm256_bf16_kernel<AM, BM_, Epi>;
    constexpr bool a3 = G2_A3 && epi_deep_a_ring<Epi>::value;     // (gemm256.h: three A slots + two B slots)
    constexpr int lds256 = a3 ? G2_A3_LDS_BYTES : (epi_rounds_first<Epi>::value && G2_BF16_LDS_BYTES > G2_LDS_BYTES) ? G2_BF16_LDS_BYTES : G2_LDS_BYTES;
    static_assert(G2_A3_LDS_BYTES >= G2_BF16_LDS_BYTES && G2_A3_LDS_BYTES >= G2_LDS_BYTES, "the epilogues reuse the ring's LDS");
    LDS_ATTR(kern256, lds256, g_device);
    GemmArgs g2 = g;
    g2.nbm = g.nbm / 2;
    g2.nbn = g.nbn / 2;
    const int grid_max = (g.dyn && g.grid_cover > 0) ? g.grid_cover
                         : g2.tail_tiles > 0    ? g2.nbm * g2.nbn + g2.tail_tiles * (g2.tail_pieces - 1)
                                                : g2.nbm * g2.nbn * g2.splits;
    const int grid_est = (((g.grid_hint + 3) / 4 + 7) / 8) * 8;          // 256x256 tiles, a multiple of 8
    // (the caller sets grid_hint only for a SMALL estimated extent: the persistent instantiation's tile loop costs the K loop
    // ~10 %, while the workgroups that start only to exit are cheap until they are the great majority)
    if (G2_PERSIST_STATIC > 0 && !g.dyn && g2.splits == 1 && g2.tail_tiles == 0 && grid_max >= 4 * G2_PERSIST_STATIC) {
      // big static launches (the K = d GEMMs: tens of thousands of tiles) as G2_PERSIST_STATIC resident workgroups that walk
      // the tiles: encoder / dpre -2 %, TopK encoder -2.5 % against one workgroup per tile (same box; 256 / 512 / 1024 measure alike)
      auto kernp = gemm256_bf16_kernel<AM, BM_, Epi, true>;
      LDS_ATTR(kernp, lds256, g_device);
      hipLaunchKernelGGL(kernp, dim3(G2_PERSIST_STATIC), dim3(512), lds256, s, g2, epi);
    } else if (g.dyn && g.grid_hint > 0 && grid_est < grid_max) {
      auto kernp = gemm256_bf16_kernel<AM, BM_, Epi, true>;
      LDS_ATTR(kernp, lds256, g_device);
      hipLaunchKernelGGL(kernp, dim3(grid_est < 256 ? 256 : grid_est), dim3(512), lds256, s, g2, epi);
    } else {
      hipLaunchKernelGGL(kern256, dim3(grid_max), dim3(512), lds256, s, g2, epi);
    }
    HIP_TRY(hipGetLastError());
    return SAE_OK;
  }
  auto kern = gemm_bf16_kernel<AM, BM_, Epi>;
  LDS_ATTR(kern, GEMM_LDS_BYTES, g_device);
  const int grid = (g.dyn && g.grid_cover > 0) ? g.grid_cover : g.nbm * g.nbn * g.splits;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), GEMM_LDS_BYTES, s, g, epi);
  HIP_TRY(hipGetLastError());
  return SAE_OK;
}

template <class Epi>
static bool gemm8_streams(const Gemm8Args& g) {
  if constexpr (!(G2_STREAM && epi_stream<Epi>::value)) return false;
  return !g_no_stream && g.ktiles >= 2 && g.lda == g.ldb && g.nbm * g.nbn >= 4 * G2_PERSIST_STATIC;
}

template <class Epi>
static int launch_gemm8(const Gemm8Args& g, const Epi& epi, hipStream_t s) {
  if constexpr (G2_STREAM && epi_stream<Epi>::value) {
    if (gemm8_streams<Epi>(g)) {
      auto kerns = gemm256s_fp8_kernel<Epi>;
      LDS_ATTR(kerns, G2S_LDS_BYTES, g_device);
      hipLaunchKernelGGL(kerns, dim3(G2_PERSIST_STATIC), dim3(512), G2S_LDS_BYTES, s, g, epi);
      HIP_TRY(hipGetLastError());
      return SAE_OK;
    }
  }
  auto kern = gemm256_fp8_kernel<Epi>;
  constexpr int lds = g8_lds_bytes<Epi>();
  LDS_ATTR(kern, lds, g_device);
  hipLaunchKernelGGL(kern, dim3(g.nbm * g.nbn), dim3(512), lds, s, g, epi);
  HIP_TRY(hipGetLastError());
  return SAE_OK;
}

// The in-place column normalisation every L1 forward starts with (l1autoencoder.py:71-73), and the bf16 copies of the result.
static void prep_weights_l1(sae_ctx* c, hipStream_t s) {
  const int d_p = c->d_p, n_p = c->n_p;
  float* W = c->P;
  if (c->wn_pending && !c->wn_fwd_seen) {
    // the update of the last training step already wrote Wb / Wt for the normalised weights (optimizer_l1_cols_kernel): this
    // forward IS the reference's in-place normalisation -- no kernel
    c->wn_fwd_seen = true;
  } else {
    settle_weights(c, s);       // (a second forward since the update: the first one's normalisation becomes real, then the usual one)
    // (a training step's optimizer left the column-norm partials of the weights it wrote: optimizer_l1_kernel)
    if (!c->cn_valid) hipLaunchKernelGGL(colnorm_partial_kernel, dim3(n_p / 128, d_p / 32), dim3(256), 0, s, W, c->cn_part, n_p);
    c->cn_valid = false;        // normalize_cast rewrites W in place: the partials describe the weights before it
    hipLaunchKernelGGL(normalize_cast_kernel, dim3(n_p / 64, d_p / 64), dim3(256), 0, s, W, c->cn_part, d_p / 32, c->Wb,
                       c->Wt, d_p, n_p);
  }
}

template <typename T> static constexpr bool x_dtype_is_bf16() { return false; }
template <> constexpr bool x_dtype_is_bf16<bf16_t>() { return true; }

template <typename T>
static int forward_impl(sae_ctx* c, const T* x, int64_t M, int64_t Mp, hipStream_t s, bool need_backward) {
  const int d = c->d, d_p = c->d_p, n_p = c->n_p;
  const float alpha = (float)c->cfg.recon_alpha;
  float* W = c->P;
  float* b = c->P + c->nW;

  ev_begin(c, KID_PREP_W, s);
  prep_weights_l1(c, s);
  if (c->fp8) hipLaunchKernelGGL(fp8_cast_w_kernel, dim3(n_p / 64, d_p / 64), dim3(256), 0, s, W, c->W8, c->W8t, d_p, n_p);
  ev_end(c, KID_PREP_W, s);

  ev_begin(c, KID_PREP_X, s);
  // bf16 activations that already have the padded GEMM shape need no copy; with the fused forward the masked-entry
  // count comes from the forward's own epilogue, so the whole pass over x disappears.
  const bool alias_x = c->use_fused_fwd && sizeof(T) == 2 && x_dtype_is_bf16<T>() && d == d_p && M == Mp &&
                       (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  c->xb_cur = alias_x ? reinterpret_cast<const bf16_t*>(x) : c->xb;
  if (!alias_x) {
    const int64_t chunks = Mp * (d_p / 8);
    int grid = (int)((chunks + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (d % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
      hipLaunchKernelGGL((prep_x_kernel<T, true>), dim3(grid), dim3(256), 0, s, x, c->xb, c->masked, M, d, Mp, d_p);
    else
      hipLaunchKernelGGL((prep_x_kernel<T, false>), dim3(grid), dim3(256), 0, s, x, c->xb, c->masked, M, d, Mp, d_p);
    if (!c->use_fused_fwd) {
      const double* gs = (need_backward && c->dp_world > 0) ? c->stats : nullptr;
      if (gs && c->dist) HIP_TRY(hipStreamWaitEvent(s, c->ev_stats, 0));     // the summed statistics have arrived
      hipLaunchKernelGGL(finalize_count_kernel, dim3(1), dim3(256), 0, s, c->masked, grid, c->scal, M, d, alpha, gs);
    }
  }
  if (c->fp8) {   // per-tensor power-of-two scales from the batch (max |x|, max row norm) and the bias, then x8 = e4m3(xb s_x)
    const int sgrid = (int)((Mp / 4) < 1024 ? (Mp / 4) : 1024);
    hipLaunchKernelGGL(fp8_x_stats_kernel, dim3(sgrid), dim3(256), 0, s, c->xb, Mp, d_p, c->x8_part);
    hipLaunchKernelGGL(fp8_scales_kernel, dim3(1), dim3(256), 0, s, c->x8_part, sgrid, b, c->n, c->scal8);
    const int64_t n8 = Mp * (d_p / 8);
    int qgrid = (int)((n8 + 255) / 256);
    if (qgrid > 2048) qgrid = 2048;
    hipLaunchKernelGGL(fp8_quant_x_kernel, dim3(qgrid), dim3(256), 0, s, c->xb, c->x8, n8, c->scal8);
  }
  ev_end(c, KID_PREP_X, s);

  int rc;
  if (c->use_fused_fwd) {
    const int lds = FF_LDS_BYTES;
    FwdFusedArgs a{};
    a.xb = c->xb_cur; a.x = x; a.Wt = c->Wt; a.bias = b; a.cnt_part = c->cnt_part; a.c = c->c; a.dxh = c->dxh;
    a.l1_part = c->l1_part; a.sq_part = c->sq_part; a.M = M; a.d = d; a.n_p = n_p; a.ntiles = n_p / FF_BN;
    a.c_rows = c->max_rows_p;
    // full workgroups (all 128 rows < M) run the mask-free instantiation; a ragged last workgroup the padded one
    const int full_wgs = (int)(M / FF_BM), all_wgs = (int)(Mp / FF_BM);
    auto launch = [&](auto kern, int blocks, int block_offset) -> int {
      LDS_ATTR(kern, 160 * 1024, g_device);     // opt in to > 64 KiB dynamic LDS once per instantiation and device
      a.block_offset = block_offset;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a);
      return SAE_OK;
    };
    ev_begin(c, KID_FWD_FUSED, s);
    int rcl = SAE_OK;
    if (full_wgs > 0 && c->cfg.debug_flags == 65 && c->fwd_variant == 1) {   // diagnostic: stamps into the (unused here) dpre buffer
      a.stamps = reinterpret_cast<unsigned long long*>(c->dpre);
      rcl = launch(fwd_fused_d384_kernel<T, false, true>, full_wgs, 0);
    } else if (c->fwd_variant == 2) {               // the second decomposition (fwd_fused2.h)
      if (full_wgs > 0 && c->cfg.debug_flags == 65) {
        a.stamps = reinterpret_cast<unsigned long long*>(c->dpre);
        rcl = launch(fwd_fused2_d384_kernel<T, false, true>, full_wgs, 0);
      } else if (full_wgs > 0) rcl = launch(fwd_fused2_d384_kernel<T, false>, full_wgs, 0);
      if (!rcl && all_wgs > full_wgs) rcl = launch(fwd_fused2_d384_kernel<T, true>, all_wgs - full_wgs, full_wgs);
      if (rcl) return rcl;
      ev_end(c, KID_FWD_FUSED, s);
      HIP_TRY(hipGetLastError());
      return SAE_OK;
    } else if (full_wgs > 0) {
      rcl = launch(fwd_fused_d384_kernel<T, false, false>, full_wgs, 0);
    }
    if (!rcl && all_wgs > full_wgs) rcl = launch(fwd_fused_d384_kernel<T, true, false>, all_wgs - full_wgs, full_wgs);
    if (rcl) return rcl;
    ev_end(c, KID_FWD_FUSED, s);
    HIP_TRY(hipGetLastError());
    return SAE_OK;
  }
  if (c->fp8) {
    {  // c = relu(bf16((x8 W8t) / (s_x s_w)) + b); also stored as c8 = e4m3(c s_c)
      Gemm8Args g{};
      g.A = c->x8; g.B = c->W8t; g.lda = d_p; g.ldb = d_p;
      g.nbm = (int)(Mp / 256); g.nbn = n_p / 256; g.ktiles = d_p / 128;
      EpiEnc8 e{};
      e.c = c->c; e.c8 = c->c8; e.bias = b; e.scal8 = c->scal8; e.l1_part = c->l1_part; e.M = M; e.n_p = n_p; e.nbn = n_p / 128;
      // (streaming form: one L1 partial per workgroup, the rest of the per-tile range zeroed -- as for EpiEnc below)
      if (gemm8_streams<EpiEnc8>(g)) HIP_TRY(hipMemsetAsync(c->l1_part, 0, (size_t)(Mp / 128) * (n_p / 128) * 4, s));
      ev_begin(c, KID_ENC_FWD, s);
      rc = launch_gemm8(g, e, s);
      ev_end(c, KID_ENC_FWD, s);
      if (rc) return rc;
    }
    {  // x_hat = bf16((c8 W8^T) / (s_c s_w)), residual, dx_hat
      Gemm8Args g{};
      g.A = c->c8; g.B = c->W8; g.lda = n_p; g.ldb = n_p;
      g.nbm = (int)(Mp / 256); g.nbn = d_p / 256; g.ktiles = n_p / 128;
      EpiDec<T> e{};
      e.x = x; e.dxh = c->dxh; e.scal = c->scal; e.sq_part = c->sq_part; e.M = M; e.d = d; e.d_p = d_p; e.nbn = d_p / 128;
      e.vec_all = (M == Mp && d == d_p && d % 4 == 0 && (reinterpret_cast<uintptr_t>(x) % (4 * sizeof(T))) == 0) ? 1 : 0;
      e.vscale = c->scal8 + S8_INV_DEC;
      ev_begin(c, KID_DEC_FWD, s);
      rc = launch_gemm8(g, e, s);
      ev_end(c, KID_DEC_FWD, s);
      if (rc) return rc;
    }
    if (need_backward && c->dxhT) {
      hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)(Mp / 64), d_p / 64), dim3(256), 0, s, c->dxh, c->dxhT, Mp, d_p);
      hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)(Mp / 64), d_p / 64), dim3(256), 0, s, c->xb_cur, c->xT, Mp, d_p);
      HIP_TRY(hipGetLastError());
    }
    return SAE_OK;
  }
  {  // c = relu(x W + b)
    GemmArgs g{};
    g.A0 = c->xb_cur; g.B0 = c->Wt; g.lda = d_p; g.ldb = d_p;
    g.nbm = (int)(Mp / 128); g.nbn = n_p / 128; g.ktiles0 = g.ktiles = d_p / 64; g.splits = 1;
    EpiEnc e{};
    e.c = c->c; e.bias = b; e.l1_part = c->l1_part; e.M = M; e.n_p = n_p; e.nbn = g.nbn;
    // (streaming form: one L1 partial per WORKGROUP in l1_part[0 .. grid); finalize_losses sums the whole per-tile range, so the
    // rest of it is zeroed -- 4 bytes per 128x128 tile)
    if (gemm_streams<OP_ROW, OP_ROW, EpiEnc>(g)) HIP_TRY(hipMemsetAsync(c->l1_part, 0, (size_t)g.nbm * g.nbn * 4, s));
    ev_begin(c, KID_ENC_FWD, s);
    rc = launch_gemm<OP_ROW, OP_ROW>(g, e, s);
    ev_end(c, KID_ENC_FWD, s);
    if (rc) return rc;
  }
  {  // x_hat = c W^T, residual, dx_hat
    GemmArgs g{};
    g.A0 = c->c; g.B0 = c->Wb; g.lda = n_p; g.ldb = n_p;
    g.nbm = (int)(Mp / 128); g.nbn = d_p / 128; g.ktiles0 = g.ktiles = n_p / 64; g.splits = 1;
    g.group_m = 4;       // (gemm.h: GemmArgs::group_m)
    EpiDec<T> e{};
    e.x = x; e.dxh = c->dxh; e.scal = c->scal; e.sq_part = c->sq_part; e.M = M; e.d = d; e.d_p = d_p; e.nbn = g.nbn;
    e.vec_all = (M == Mp && d == d_p && d % 4 == 0 && (reinterpret_cast<uintptr_t>(x) % (4 * sizeof(T))) == 0) ? 1 : 0;
    ev_begin(c, KID_DEC_FWD, s);
    rc = launch_gemm<OP_ROW, OP_ROW>(g, e, s);
    ev_end(c, KID_DEC_FWD, s);
    if (rc) return rc;
  }
  if (need_backward && c->dxhT) {
    // the weight-gradient GEMM's d-side operands K-contiguous: dx_hat^T and x^T, 2 x (M_p x d_p) bf16 read + written (section 4)
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)(Mp / 64), d_p / 64), dim3(256), 0, s, c->dxh, c->dxhT, Mp, d_p);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)(Mp / 64), d_p / 64), dim3(256), 0, s, c->xb_cur, c->xT, Mp, d_p);
    HIP_TRY(hipGetLastError());
  }
  return SAE_OK;
}

template <typename T>
static int fwd_bwd_impl(sae_ctx* c, const T* x, int64_t M, hipStream_t s, bool backward) {
  const int d = c->d, d_p = c->d_p, n_p = c->n_p;
  const int64_t Mp = round_up(M, c->row_pad);
  const float alpha = (float)c->cfg.recon_alpha;
  const double* gs = (backward && c->dp_world > 0) ? c->stats : nullptr;    // data parallel: global normalisers
  ev_begin(c, KID_STEP_TOTAL, s);
  int rc = forward_impl<T>(c, x, M, Mp, s, backward);
  if (rc) return rc;
  // single GPU, fused forward + fused backward in ONE launch: the backward takes alpha/count from the forward's counts itself and
  // the loss scalars are finalised by reduce_grads_kernel's last block (5 us less between the forward and the backward)
  const bool fold_finalize = c->use_fused_fwd && c->use_fused_bwd && backward && gs == nullptr && c->bwd_ranges == 1 &&
                             c->cfg.debug_flags != 78;
  if (c->use_fused_fwd && !fold_finalize) {  // the fused forward counted the masked entries itself: scal[] and the losses are due now
    StatsPush push{};
    if (gs && inline_stats(c)) {       // peer exchange: the statistics travel inside this kernel (no pass over x, no second stream)
      for (int r = 0; r < c->dp_world; ++r) push.inbox[r] = c->p2p_sig[r] + P2P_SIG_WORDS;
      push.rank = c->p2p_rank; push.world = c->dp_world; push.epoch = ++c->p2p_epoch_push;
      push.timeout_ticks = c->p2p_timeout_ticks; push.status = c->p2p_status; push.status_host = c->p2p_status_hostdev; push.gstats_out = c->stats;
    } else if (gs && c->dist) {
      HIP_TRY(hipStreamWaitEvent(s, c->ev_stats, 0));      // the summed statistics have arrived
    }
    if (push.world > 0) ev_begin(c, KID_STATS_XCHG, s);
    hipLaunchKernelGGL(finalize_losses_kernel, dim3(1), dim3(1024), 0, s, c->l1_part, (int)(Mp / 128), c->sq_part,
                       (int)(Mp / 128), c->scal, c->G + c->nparams, M, d, alpha, c->cnt_part, (int)(Mp / 128),
                       push.world > 0 ? (const double*)nullptr : gs, push);
    if (push.world > 0) ev_end(c, KID_STATS_XCHG, s);
  }
  bool dw_chunked_any = false;
  if (backward) {
    int splits = c->dw_splits;
    int db_rows = (int)(Mp / 128);
    bool dw_chunked = false;
    if (c->use_fused_bwd) {
      BwdFusedArgs a{};
      a.dxh = c->dxh; a.xb = c->xb_cur; a.c = c->c; a.Wt = c->Wt; a.scal = c->scal; a.slab = c->slab; a.db_part = c->db_part;
      a.unscaled = c->use_fused_fwd ? 1 : 0;
      if (fold_finalize) { a.cnt_part = c->cnt_part; a.n_cnt = (int)(Mp / 128); a.alpha = alpha; a.M = M; a.d = d; }
      a.n_p = n_p; a.steps_total = (int)(Mp / BF_BM);
      // diagnostic clock stamps go to the second half of the (unused on this path) dpre buffer
      a.clk = c->cfg.debug_flags == 66 ? reinterpret_cast<unsigned long long*>(c->dpre) + (1 << 16) : nullptr;
      // Column-tile ranges (sae_dist_set_overlap; one range otherwise): each range is a launch of its own over ALL rows, split
      // into enough row ranges to fill the chip; its slabs are reduced and the range's gradient exchanged on the communication
      // stream while the next range's backward runs.
      const int nranges = c->bwd_ranges, rtiles = (n_p / BF_BN) / nranges;
      splits = nranges == 1 ? c->bwd_splits : c->bwd_range_splits;
      if (splits > a.steps_total) splits = a.steps_total;
      a.ntiles = rtiles; a.splits = splits;
      db_rows = splits;
      const bool balanced = nranges == 1 && c->bal_m > 0 && a.steps_total >= 64;
      if (balanced) {
        a.bal_m = c->bal_m;
        a.bal_q = (a.steps_total + 31) / 32;
        a.wg_map = c->bal_map;
      }
      const bool dp_now = c->dist && c->dp_world > 0;
      bf16_t* gb = (dp_now && c->payload == SAE_DTYPE_BF16) ? c->Gb : (bf16_t*)nullptr;
      ev_begin(c, KID_BWD_FUSED, s);
      LDS_ATTR(bwd_fused_d384_kernel, BF_LDS_BYTES, g_device);
      for (int r = 0; r < nranges; ++r) {
        a.tile0 = r * rtiles;
        hipLaunchKernelGGL(bwd_fused_d384_kernel, dim3(balanced ? c->bal_grid : a.ntiles * a.splits), dim3(256), BF_LDS_BYTES, s, a);
        if (nranges > 1) {
          const int c0 = a.tile0 * BF_BN, cols = rtiles * BF_BN;
          hipLaunchKernelGGL(reduce_grads_range_kernel, dim3(256), dim3(256), 0, s, c->slab, c->nW, splits, c->db_part, db_rows, n_p,
                             d_p, c0, cols, c->G, c->gn_part + 256 * r, gb);
          if (dp_now && r + 1 < nranges) fused_exchange(c, r, nranges, c0, cols, false, s);
        }
      }
      ev_end(c, KID_BWD_FUSED, s);
      HIP_TRY(hipGetLastError());
    } else {
      if (c->fp8_bwd) {  // dpre = (bf16((dxh8 W8t^T) / (s_g s_w)) + 1/M) [c > 0]: dx_hat quantised with a scale from its own maximum
        const int sgrid = (int)((Mp / 4) < 1024 ? (Mp / 4) : 1024);
        hipLaunchKernelGGL(fp8_x_stats_kernel, dim3(sgrid), dim3(256), 0, s, c->dxh, Mp, d_p, c->x8_part);
        hipLaunchKernelGGL(fp8_g_scale_kernel, dim3(1), dim3(256), 0, s, c->x8_part, sgrid, c->scal8);
        const int64_t n8 = Mp * (d_p / 8);
        int qgrid = (int)((n8 + 255) / 256);
        if (qgrid > 2048) qgrid = 2048;
        hipLaunchKernelGGL(fp8_quant_x_kernel, dim3(qgrid), dim3(256), 0, s, c->dxh, c->dxh8, n8, c->scal8, (int)S8_SG);
        Gemm8Args g{};
        g.A = c->dxh8; g.B = c->W8t; g.lda = d_p; g.ldb = d_p;
        g.nbm = (int)(Mp / 256); g.nbn = n_p / 256; g.ktiles = d_p / 128;
        EpiDpre8 e{};
        e.c = c->c; e.dpre = c->dpre; e.db_part = c->db_part; e.scal = c->scal; e.scal8 = c->scal8; e.n_p = n_p;
        ev_begin(c, KID_DPRE, s);
        rc = launch_gemm8(g, e, s);
        ev_end(c, KID_DPRE, s);
        if (rc) return rc;
      } else {  // dpre = (dx_hat W + 1/M) [c > 0]
        GemmArgs g{};
        g.A0 = c->dxh; g.B0 = c->Wt; g.lda = d_p; g.ldb = d_p;
        g.nbm = (int)(Mp / 128); g.nbn = n_p / 128; g.ktiles0 = g.ktiles = d_p / 64; g.splits = 1;
        EpiDpre e{};
        e.c = c->c; e.dpre = c->dpre; e.db_part = c->db_part; e.scal = c->scal; e.n_p = n_p;
        ev_begin(c, KID_DPRE, s);
        rc = launch_gemm<OP_ROW, OP_ROW>(g, e, s);
        ev_end(c, KID_DPRE, s);
        if (rc) return rc;
      }
      // dW = dx_hat^T c + x^T dpre   (K = 2 M, split-K partial slabs).  With a gradient-ready hook the GEMM is issued in
      // row chunks of dW (contiguous ranges of the gradient buffer): each chunk is reduced and announced as soon as it
      // is enqueued, so its all-reduce runs under the GEMM of the next chunk.
      // Peer exchange: COLUMN chunks of dW (a [d_p x cols] block of the gradient is a strided 2-D segment, which the exchange
      // kernel takes as it is): every chunk reads only its own columns of the latent / dpre streams, so chunking costs no
      // extra HBM traffic -- row chunks (the contiguous ranges RCCL and the host callback need) re-read both 5.4 GB streams
      // once per chunk: +1.5 ms of the 30 ms C4 step on one rank.
      // In-engine RCCL (round 4): the same column chunks, each summed over its split-K slabs into a CONTIGUOUS staging block,
      // all-reduced there and copied back into its strided place on the communication stream (two passes over 52 MB per chunk
      // at C4 against re-reading two 5.4 GB streams per row chunk).  Host-driven exchange: row chunks (the callback's contract
      // is a contiguous range of the gradient buffer).
      // dW = dx_hat^T c + x^T dpre (rows r0.. of d).  Round 5 experiment (FREUD_DW_ROWA=1, off by default): with transposed copies of
      // the d-side operands (forward_impl) the A operand is row-major -- [d_p][M_p], K = the batch rows contiguous -- and its fragments
      // come by ds_read_b128 instead of twice as many ds_read_b64_tr_b16.  The stand-alone shape on random dense operands ran 10.6
      // against 11.7 ms (tools/kbench mode 6); in the engine, on the real latent (half zeros) the k-major x k-major form with its
      // k-half ring is FASTER: 9.70 against 10.75 ms at C4 (profiles/r05_ab_dw_row_a.txt).  Kept as a switch for that record.
      auto launch_dw = [&](GemmArgs g, const EpiSlab& e, int r0) -> int {
        if (c->dxhT) {
          g.A0 = c->dxhT + (int64_t)r0 * Mp; g.A1 = c->xT + (int64_t)r0 * Mp; g.lda = Mp;
          return launch_gemm<OP_ROW, OP_KMAJOR>(g, e, s);
        }
        return launch_gemm<OP_KMAJOR, OP_KMAJOR>(g, e, s);
      };
      const bool col_staged = c->dist && c->dp_world > 0 && !c->p2p && c->dw_col_chunks > 1 && c->cfg.debug_flags != 82;
      if (col_staged && !c->col_stage) {
        hipError_t e_ = hipMalloc((void**)&c->col_stage, (size_t)c->nW * 4);
        if (e_ != hipSuccess) return fail(SAE_ERR_HIP, "staging buffer of the column chunks: %s", hipGetErrorString(e_));
      }
      const bool col_chunked = c->dist && c->dp_world > 0 && (c->p2p || col_staged) && c->dw_col_chunks > 1;
      // Round 4: chunks sized by ROUNDS.  A chunk of floor(256 / (d_p / 256)) tile columns is one round of whole 256x256 tiles --
      // it needs no split-K, writes straight into the gradient (or the staging block) and costs what the same tiles cost in the
      // plain launch; the columns left over run like the plain launch's tail (K pieces through the overflow buffer).  Round 3's
      // four equal chunks were 200 tiles each: 0.78 of a round, bought full with split-K slabs and a reduction pass (+2.8 % on
      // one rank at C4; now +1.5 % with the peer exchange, +1.1 % with RCCL, section 5).  debug_flags 84 = the equal chunks.
      const int nbm256 = d_p / 256, per_round = nbm256 > 0 ? 256 / nbm256 : 0;
      const bool round_chunks = col_chunked && d_p % 256 == 0 && n_p % 256 == 0 && per_round >= 1 && n_p / 256 > per_round &&
                                c->dw_tail != nullptr && c->cfg.debug_flags != 84 && !g_force_gemm128;
      if (round_chunks) {
        ev_begin(c, KID_DW, s);
        const int nbn256 = n_p / 256;
        for (int ct0 = 0; ct0 < nbn256;) {
          const int nct = nbn256 - ct0 >= per_round ? per_round : nbn256 - ct0;
          const int col0 = ct0 * 256, cols = nct * 256;
          float* stage = c->p2p ? nullptr : c->col_stage + (int64_t)col0 * d_p;        // contiguous [d_p x cols] block (RCCL)
          GemmArgs g{};
          g.A0 = c->dxh; g.B0 = c->c + col0; g.A1 = c->xb_cur; g.B1 = c->dpre + col0; g.lda = d_p; g.ldb = n_p;
          g.nbm = d_p / 128; g.nbn = cols / 128; g.ktiles0 = (int)(Mp / 64); g.ktiles = 2 * g.ktiles0;
          g.splits = 1;
          gemm_tail_plan(nbm256 * nct, g.ktiles, g.tail_tiles, g.tail_pieces);
          if (nbm256 * nct >= 255) g.tail_tiles = g.tail_pieces = 0;      // (a full chunk: one round, nothing to split)
          EpiSlab e{};
          e.slab = c->p2p ? c->G + col0 : stage; e.slab_stride = 0; e.ld = c->p2p ? n_p : cols; e.tail = c->dw_tail;
          rc = launch_dw(g, e, 0);
          if (rc) return rc;
          if (g.tail_tiles > 0)
            hipLaunchKernelGGL(reduce_tail_kernel, dim3(g.tail_tiles, 16), dim3(256), 0, s, c->dw_tail, e.slab, e.ld, nbm256, nct,
                               g.tail_tiles, g.tail_pieces);
          if (c->p2p) exchange_block(c, col0, d_p, cols, n_p, s);
          else exchange_staged(c, stage, col0, d_p, cols, n_p, s);
          ct0 += nct;
        }
        ev_end(c, KID_DW, s);
        dw_chunked = dw_chunked_any = true;
      } else if (col_chunked) {
        const int cols = n_p / c->dw_col_chunks;
        ev_begin(c, KID_DW, s);
        for (int q = 0; q < c->dw_col_chunks; ++q) {
          const int col0 = q * cols;
          GemmArgs g{};
          g.A0 = c->dxh; g.B0 = c->c + col0; g.A1 = c->xb_cur; g.B1 = c->dpre + col0; g.lda = d_p; g.ldb = n_p;
          g.nbm = d_p / 128; g.nbn = cols / 128; g.ktiles0 = (int)(Mp / 64); g.ktiles = 2 * g.ktiles0;
          g.splits = c->dw_col_splits > g.ktiles ? g.ktiles : c->dw_col_splits;
          EpiSlab e{};
          e.slab = c->slab + col0; e.slab_stride = c->nW; e.ld = n_p;
          rc = launch_dw(g, e, 0);
          if (rc) return rc;
          if (c->p2p) {
            hipLaunchKernelGGL(reduce_slabs_range_kernel, dim3(512), dim3(256), 0, s, c->slab, c->nW, g.splits, n_p, d_p, col0, cols, c->G,
                               (int64_t)n_p, col0);
            exchange_block(c, col0, d_p, cols, n_p, s);
          } else {
            float* stage = c->col_stage + (int64_t)q * d_p * cols;
            hipLaunchKernelGGL(reduce_slabs_range_kernel, dim3(512), dim3(256), 0, s, c->slab, c->nW, g.splits, n_p, d_p, col0, cols, stage,
                               (int64_t)cols, 0);
            exchange_staged(c, stage, col0, d_p, cols, n_p, s);
          }
        }
        ev_end(c, KID_DW, s);
        dw_chunked = dw_chunked_any = true;
      }
      const bool chunked = !col_chunked && (c->grad_ready != nullptr || (c->dist && c->dp_world > 0)) && c->dw_chunk_rows < d_p;
      const int chunk_rows = chunked ? c->dw_chunk_rows : d_p;
      if (chunked) splits = c->dw_chunk_splits;
      // one launch, no exchange to feed: whole tiles go straight into the gradient buffer (no slabs, no reduction pass), the
      // tiles left over after whole rounds of 256 workgroups in K pieces through the small overflow buffer
      const bool direct = !col_chunked && !chunked && c->dw_tail_tiles != 0 && (int64_t)c->dw_tail_tiles * c->dw_tail_pieces <= 256 &&
                          2 * Mp / 64 >= 16 * (c->dw_tail_pieces > 0 ? c->dw_tail_pieces : 1);
      if (!col_chunked) ev_begin(c, KID_DW, s);
      for (int r0 = 0; r0 < d_p && !col_chunked; r0 += chunk_rows) {
        const int rows = d_p - r0 < chunk_rows ? d_p - r0 : chunk_rows;
        GemmArgs g{};
        g.A0 = c->dxh + r0; g.B0 = c->c; g.A1 = c->xb_cur + r0; g.B1 = c->dpre; g.lda = d_p; g.ldb = n_p;
        g.nbm = rows / 128; g.nbn = n_p / 128; g.ktiles0 = (int)(Mp / 64); g.ktiles = 2 * g.ktiles0;
        if (splits > g.ktiles) splits = g.ktiles;
        g.splits = splits;
        EpiSlab e{};
        e.slab = c->slab + (int64_t)r0 * n_p; e.slab_stride = c->nW; e.ld = n_p;
        if (direct) {
          g.splits = 1;
          g.tail_tiles = c->dw_tail_tiles > 0 ? c->dw_tail_tiles : 0;
          g.tail_pieces = c->dw_tail_tiles > 0 ? c->dw_tail_pieces : 0;
          e.slab = c->G; e.slab_stride = 0; e.tail = c->dw_tail;
        }
        rc = launch_dw(g, e, r0);
        if (rc) return rc;
        if (direct && g.tail_tiles > 0)
          hipLaunchKernelGGL(reduce_tail_kernel, dim3(g.tail_tiles, 16), dim3(256), 0, s, c->dw_tail, c->G, n_p, d_p / 256, n_p / 256,
                             g.tail_tiles, g.tail_pieces);
        if (chunked) {
          const int64_t off4 = (int64_t)r0 * n_p / 4, n4 = (int64_t)rows * n_p / 4;
          hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, c->slab + 4 * off4,
                             c->G + 4 * off4, n4, c->nW / 4, splits);
          notify_grads(c, 4 * off4, 4 * n4, s);
        }
      }
      if (!col_chunked) {
        ev_end(c, KID_DW, s);
        dw_chunked_any = chunked;
        dw_chunked = chunked || direct;     // (direct: the gradient is already in place -- no slab reduction; nothing was announced)
      }
    }
    ev_begin(c, KID_REDUCE, s);
    if (c->use_fused_bwd && c->bwd_ranges > 1) {   // (every range was reduced right behind its launch)
      c->gn_blocks = 256 * c->bwd_ranges;
      c->gn_valid = true;
    } else if (c->use_fused_bwd) {   // one pass: slabs + db partials -> grads, plus the local gradient sum of squares
      const int64_t nW4 = c->nW / 4, n4 = c->nparams / 4;
      int blocks = (int)((n4 + 255) / 256);
      if (blocks > 1024) blocks = 1024;
      LossFinalize fin{};
      if (fold_finalize) {
        fin.l1_part = c->l1_part; fin.sq_part = c->sq_part; fin.cnt_part = c->cnt_part; fin.n_parts = (int)(Mp / 128);
        fin.scal = c->scal; fin.metrics = c->G + c->nparams; fin.M = M; fin.d = d; fin.alpha = alpha;
      }
      hipLaunchKernelGGL(reduce_grads_kernel, dim3(blocks), dim3(256), 0, s, c->slab, nW4, splits, c->db_part, db_rows, n_p,
                         c->G, nW4, n4, c->gn_part,
                         (c->dist && c->dp_world > 0 && c->payload == SAE_DTYPE_BF16) ? c->Gb : (bf16_t*)nullptr, fin,
                         (c->bwd_ranges == 1 && c->bal_m > 0 && (int)(Mp / BF_BM) >= 64) ? c->bal_m : 0);
      c->gn_blocks = blocks;
      c->gn_valid = true;
    } else {
      const int64_t n4 = c->nW / 4;
      if (!dw_chunked)
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, c->slab, c->G, n4, n4,
                           splits);
      hipLaunchKernelGGL(reduce_db_kernel, dim3(n_p / 32), dim3(256), 0, s, c->db_part, c->G + c->nW, db_rows, n_p);
      c->gn_valid = false;
    }
    ev_end(c, KID_REDUCE, s);
  }
  if (!c->use_fused_fwd)
    hipLaunchKernelGGL(finalize_losses_kernel, dim3(1), dim3(1024), 0, s, c->l1_part, (int)((Mp / 128) * (n_p / 128)),
                       c->sq_part, (int)((Mp / 128) * (d_p / 128)), c->scal, c->G + c->nparams, M, d, alpha,
                       (const float*)nullptr, 0, gs, StatsPush{});
  if (backward) {   // everything that was not announced chunk by chunk: [dW] | db | loss scalars
    const int64_t total = c->nparams + SAE_NUM_METRICS;
    if (c->use_fused_bwd && c->dist && c->dp_world > 0) {      // the last (or only) column range, with the loss scalars
      const int nr = c->bwd_ranges, cols = n_p / nr;
      fused_exchange(c, nr - 1, nr, (nr - 1) * cols, cols, true, s);
    } else if (dw_chunked_any) notify_grads(c, c->nW, total - c->nW, s);
    else notify_grads(c, 0, total, s);
  }
  ev_end(c, KID_STEP_TOTAL, s);
  HIP_TRY(hipGetLastError());
  c->last_M = M;
  c->last_M_p = Mp;
  c->metrics_fresh = true;
  c->prof_tick++;
  return SAE_OK;
}

// ------------------------------------------------------------------------------------------
// TopK variant: one train step's forward + backward (topkautoencoder.py:93-151, train_sae.py:436-448)
// ------------------------------------------------------------------------------------------
template <typename T>
static int topk_fwd_bwd(sae_ctx* c, const T* x, int64_t M, hipStream_t s, bool backward) {
  const int d = c->d, n = c->n, d_p = c->d_p, n_p = c->n_p, k = c->k;
  const int64_t Mp = round_up(M, 128);
  const int64_t T_rows = (c->rows_per_file > 0 && M % c->rows_per_file == 0) ? c->rows_per_file : M;
  const int B = (int)(M / T_rows);
  const float alpha = (float)c->cfg.auxk_alpha;
  float* We = c->P;
  float* be = c->P + c->nW;
  float* Wd = c->P + c->nW + n_p;
  float* bd = c->P + 2 * c->nW + n_p;
  float* gWe = c->G;
  float* gbe = c->G + c->nW;
  float* gWd = c->G + c->nW + n_p;
  float* gbd = c->G + 2 * c->nW + n_p;
  float* metrics = c->G + c->nparams;
  float* did_fire = metrics + SAE_NUM_METRICS;
  const double* gs = (backward && c->dp_world > 0) ? c->stats : nullptr;    // data parallel: global normalisers
  int rc;
  ev_begin(c, KID_STEP_TOTAL, s);

  // dead mask from num_frames_since_fired.  Whether the AuxK branch runs (num_dead > 0, topkautoencoder.py:109) is decided
  // ON THE DEVICE: the aux kernels are always enqueued and exit at once when tk[0] == 0, the weight-gradient GEMM drops its
  // second K segment through GemmArgs::seg1_gate -- no device->host copy, no stream synchronisation in the step.
  // validate() passes no dead_mask (train_sae.py:168-171): without a backward the AuxK branch is off.
  if (c->aux_compact)
    hipLaunchKernelGGL(dead_compact_kernel, dim3(1), dim3(1024), 0, s, c->nfsf, c->dead, did_fire, n, n_p, c->dead_threshold, d, c->tk,
                       c->tkf, c->tkd, c->dead_cols, c->vec_rank, c->vec_bits);
  else
    hipLaunchKernelGGL(dead_mask_kernel, dim3(1), dim3(1024), 0, s, c->nfsf, c->dead, did_fire, n, n_p, c->dead_threshold, d, c->tk,
                       c->tkf);
  const bool aux = alpha != 0.f && backward;      // "possible": the device gates it on tk[0] > 0
  // AuxK as dense GEMMs over the compacted dead latents (topk_aux.h) next to the CSC backward of the main selection
  const bool auxc = aux && c->aux_compact;
  // Which backward: the CSC one (topk_sparse.h) wins by a wide margin while no latent is dead, the dense-GEMM one once the
  // AuxK pass brings d/2 more entries per row.  Both are correct for any number of dead latents (each gates its AuxK part
  // on the device), so the choice may rest on a STALE count: the last value of tk[0] that an asynchronous copy happened to
  // land in pinned host memory (this step's copy is enqueued now and read by some later step; never waited for).
  const bool use_csc = c->topk_csc && backward && (auxc || !(aux && c->dead_hint[0] > 0));
  HIP_TRY(hipMemcpyAsync(c->dead_hint, c->tk, 4, hipMemcpyDeviceToHost, s));
  // the masked DENSE rows [M x n] are only written for those who read them: the dense fallbacks and validation / inference
  const bool write_dense = !use_csc;
  c->dense_valid = write_dense;
  c->multi_dense_valid = write_dense;
  // tile-driven main select (topk_select_tiles_kernel): when no dense row is wanted, every column is a candidate and k fits
  // (rows it cannot take -- too many qualifying tiles or candidates -- fall to the register kernel: n_p <= 2048 * 44)
  const bool tile_select = !write_dense && n == n_p && k <= n_p / TSEL_TILE && n_p / TSEL_TILE <= TSEL_MAX_TILES && n_p <= 2048 * 44 &&
                           c->cfg.debug_flags != 75;

  {
    const int64_t n8 = c->nW / 8;
    int grid = (int)((n8 + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (!c->wb_valid) {        // (a training step's optimizer already wrote the bf16 copies of the weights it updated)
      hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, s, We, c->We_b, n8);
      hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, s, Wd, c->Wd_b, n8);
    }
    if (auxc) hipLaunchKernelGGL(aux_gather_rows_kernel, dim3(n_p / 4), dim3(256), 0, s, c->Wd_b, c->dead_cols, c->tkd, c->Wdd_b, d_p);
    const int64_t chunks = Mp * (d_p / 8);
    int g2 = (int)((chunks + 255) / 256);
    if (g2 > 4096) g2 = 4096;
    if (d % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
      hipLaunchKernelGGL((topk_prep_x_kernel<T, true>), dim3(g2), dim3(256), 0, s, x, bd, c->xs, M, d, Mp, d_p);
    else
      hipLaunchKernelGGL((topk_prep_x_kernel<T, false>), dim3(g2), dim3(256), 0, s, x, bd, c->xs, M, d, Mp, d_p);
    const int64_t TD = T_rows * d;
    if (!gs)   // (data parallel: the variance comes from the column statistics summed over the ranks, below)
    {
      bool vec = false;
      if constexpr (std::is_same<T, bf16_t>::value) {
        if (B <= TV_MAXB && TD % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
          vec = true;
          if (B <= 32)
            hipLaunchKernelGGL((total_variance_vec_kernel<32, 8>), dim3((unsigned)((TD + 2047) / 2048)), dim3(256), 0, s, x, B, TD, c->tv_part);
          else
            hipLaunchKernelGGL((total_variance_vec_kernel<64, 4>), dim3((unsigned)((TD + 1023) / 1024)), dim3(256), 0, s, x, B, TD, c->tv_part);
        }
      }
      if (!vec) hipLaunchKernelGGL(total_variance_kernel<T>, dim3((unsigned)((TD + 255) / 256)), dim3(256), 0, s, x, B, TD, c->tv_part);
    }
  }
  {  // pre = relu(sae_in We^T + be)
    GemmArgs g{};
    g.A0 = c->xs; g.B0 = c->We_b; g.lda = d_p; g.ldb = d_p;
    g.nbm = (int)(Mp / 128); g.nbn = n_p / 128; g.ktiles0 = g.ktiles = d_p / 64; g.splits = 1;
    hipLaunchKernelGGL(round_bias_kernel, dim3((n_p + 255) / 256), dim3(256), 0, s, be, c->be_r, n_p);
    EpiTopkEnc e{};
    e.pre = c->pre; e.bias = c->be_r; e.M = M; e.n_p = n_p;
    e.tmax = c->tile_max;      // (always written: a null test per s_apply call would split the streaming epilogue into basic blocks)
    ev_begin(c, KID_TK_ENC, s);
    rc = launch_gemm<OP_ROW, OP_ROW>(g, e, s);
    ev_end(c, KID_TK_ENC, s);
    if (rc) return rc;
  }
  ev_begin(c, KID_TK_SELECT, s);
  {
    const unsigned char* only_flagged = nullptr;
    auto launch_select = [&](bf16_t* dense_out, int* idx_out, bf16_t* vals_out, float* fire, const unsigned char* dead_mask,
                             const int* k_ptr, int k_fixed, int kcap) {
      unsigned short* vo = reinterpret_cast<unsigned short*>(vals_out);
      const int wd = write_dense ? 1 : 0;
      if (n_p <= 2048 * 12)
        hipLaunchKernelGGL(topk_select_reg_kernel<12>, dim3((unsigned)Mp), dim3(256), 0, s, c->pre, dense_out, idx_out, fire,
                           dead_mask, k_ptr, k_fixed, kcap, n, n_p, M, vo, wd, only_flagged);
      else if (n_p <= 2048 * 44)
        hipLaunchKernelGGL(topk_select_reg_kernel<44>, dim3((unsigned)Mp), dim3(256), 0, s, c->pre, dense_out, idx_out, fire,
                           dead_mask, k_ptr, k_fixed, kcap, n, n_p, M, vo, wd, only_flagged);
      else
        hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)Mp), dim3(256), 0, s, c->pre, dense_out, idx_out, fire, dead_mask,
                           k_ptr, k_fixed, kcap, n, n_p, M, vo, wd);
    };
    // did_fire follows out.encoded.top_indices (train_sae.py:442), which forward() re-binds to the 4k selection when
    // cfg.multi_topk is set (topkautoencoder.py:135)
    if (tile_select) {
      if (n_p <= 4096 * 8)
        hipLaunchKernelGGL(topk_select_tiles_kernel<8>, dim3((unsigned)Mp), dim3(256), 0, s, c->pre, c->tile_max, c->top_idx,
                           reinterpret_cast<unsigned short*>(c->top_vals), c->multi ? (float*)nullptr : did_fire, k, k, n_p, M, c->sel_flag);
      else
        hipLaunchKernelGGL(topk_select_tiles_kernel<32>, dim3((unsigned)Mp), dim3(256), 0, s, c->pre, c->tile_max, c->top_idx,
                           reinterpret_cast<unsigned short*>(c->top_vals), c->multi ? (float*)nullptr : did_fire, k, k, n_p, M, c->sel_flag);
      only_flagged = c->sel_flag;      // the general kernel below then only takes the rows flagged as left over
    }
    launch_select(c->dense, c->top_idx, c->top_vals, c->multi ? (float*)nullptr : did_fire, nullptr, nullptr, k, k);
    only_flagged = nullptr;
    if (c->multi) launch_select(c->multi_dense, c->multi_idx, c->multi_vals, did_fire, nullptr, nullptr, c->k4, c->k4);
    if (auxc) {         // compact masked rows over the dead columns only
      unsigned short* vo = reinterpret_cast<unsigned short*>(c->aux_vals);
      if (n_p <= 2048 * 12)
        hipLaunchKernelGGL((topk_select_reg_kernel<12, true>), dim3((unsigned)Mp), dim3(256), 0, s, c->pre, c->aux_dense,
                           c->aux_idx, (float*)nullptr, c->dead, c->tk + 1, 0, c->k_aux_cap, n, n_p, M, vo, 0,
                           (const unsigned char*)nullptr, c->vec_rank, c->vec_bits, c->tkd, c->dead_cols);
      else if (n_p <= 2048 * 44) {
        // dictionaries above 24 576 latents (round 5): copy the dead columns into the compact rows, then the GENERAL select in
        // place on those rows -- <12> for up to 24 576 dead latents, <44> beyond; the device-side count picks (topk_kernels.h:
        // compact_mode).  The one-kernel form needed 44 key vectors per thread: 7.8 ms per step at n = 40 960 with 4000 dead.
        hipLaunchKernelGGL((topk_select_reg_kernel<12, true>), dim3((unsigned)Mp), dim3(256), 0, s, c->pre, c->aux_dense,
                           c->aux_idx, (float*)nullptr, c->dead, c->tk + 1, 0, c->k_aux_cap, n, n_p, M, vo, 0,
                           (const unsigned char*)nullptr, c->vec_rank, c->vec_bits, c->tkd, c->dead_cols, 1);
        hipLaunchKernelGGL((topk_select_reg_kernel<12>), dim3((unsigned)Mp), dim3(256), 0, s, c->aux_dense, c->aux_dense,
                           c->aux_idx, (float*)nullptr, (const unsigned char*)nullptr, c->tk + 1, 0, c->k_aux_cap, n, n_p, M, vo, 1,
                           (const unsigned char*)nullptr, (const int*)nullptr, (const unsigned char*)nullptr, c->tkd, (const int*)nullptr, 2);
        hipLaunchKernelGGL((topk_select_reg_kernel<44>), dim3((unsigned)Mp), dim3(256), 0, s, c->aux_dense, c->aux_dense,
                           c->aux_idx, (float*)nullptr, (const unsigned char*)nullptr, c->tk + 1, 0, c->k_aux_cap, n, n_p, M, vo, 1,
                           (const unsigned char*)nullptr, (const int*)nullptr, (const unsigned char*)nullptr, c->tkd, (const int*)nullptr, 2);
      } else {
        return fail(SAE_ERR_INVALID, "the compact AuxK path serves dictionaries up to %d latents", 2048 * 44);
      }
    } else if (aux) {
      launch_select(c->aux_dense, c->aux_idx, c->aux_vals, nullptr, c->dead, c->tk + 1, 0, c->k_aux_cap);
    }
  }
  ev_end(c, KID_TK_SELECT, s);
  ev_begin(c, KID_TK_DECODE, s);
  {
    auto launch_decode = [&](auto np_tag) {
      constexpr int NP = decltype(np_tag)::value;
      hipLaunchKernelGGL((topk_decode_kernel<T, NP>), dim3((unsigned)(Mp / 4)), dim3(256), 0, s, x, c->top_vals, c->top_idx, k, c->Wd_b,
                         bd, c->e, c->dh, c->e2_part, M, d, d_p, n_p, 0, (const int*)nullptr);
      if (aux && !auxc)
        hipLaunchKernelGGL((topk_decode_kernel<T, NP>), dim3((unsigned)(Mp / 4)), dim3(256), 0, s, x, c->aux_vals, c->aux_idx,
                           c->k_aux_cap, c->Wd_b, bd, c->e, c->dh, c->a2_part, M, d, d_p, n_p, 1, (const int*)c->tk);
      if (c->multi)     // e_multi = decode(top-4k) - x into its own residual buffer
        hipLaunchKernelGGL((topk_decode_kernel<T, NP>), dim3((unsigned)(Mp / 4)), dim3(256), 0, s, x, c->multi_vals, c->multi_idx,
                           c->k4, c->Wd_b, bd, c->em, c->dh, c->m2_part, M, d, d_p, n_p, 0, (const int*)nullptr);
    };
    if (d_p == 384) launch_decode(std::integral_constant<int, 3>{});
    else if (d_p == 768) launch_decode(std::integral_constant<int, 6>{});
    else if (d_p == 1280) launch_decode(std::integral_constant<int, 10>{});
    else launch_decode(std::integral_constant<int, 0>{});
    if (auxc) {   // e_hat = A_aux W_dec[dead] over the compact dead set: K = ND_p (device side), e_hat - e and its squares in the epilogue
      HIP_TRY(hipMemsetAsync(c->a2_part, 0, (size_t)Mp * 4, s));
      GemmArgs g{};
      g.A0 = c->aux_dense; g.B0 = c->Wdd_b; g.lda = n_p; g.ldb = d_p;
      g.nbm = (int)(Mp / 128); g.nbn = d_p / 128; g.ktiles0 = g.ktiles = n_p / 64; g.splits = 1;
      g.dyn = c->tkd; g.dyn_dim = GEMM_DYN_K;
      EpiAuxDecode e{};
      e.e = c->e; e.b_dec = bd; e.dh = c->dh; e.part = c->a2_part; e.M = M; e.d = d; e.d_p = d_p; e.nbn = d_p / 128;
      rc = launch_gemm<OP_ROW, OP_KMAJOR>(g, e, s);
      if (rc) return rc;
    }
  }
  const int64_t TD = T_rows * d;
  if (gs) {
    if (c->dist) HIP_TRY(hipStreamWaitEvent(s, c->ev_stats, 0));     // the summed statistics have arrived
    hipLaunchKernelGGL(dp_topk_tv_kernel, dim3((unsigned)((TD + 255) / 256)), dim3(256), 0, s, gs, TD, c->tv_part);
  }
  hipLaunchKernelGGL(topk_finalize_kernel, dim3(1), dim3(1024), 0, s, c->tv_part, (int)((TD + 255) / 256), c->e2_part,
                     aux ? c->a2_part : (const float*)nullptr, c->multi ? c->m2_part : (const float*)nullptr, Mp, M, d, alpha,
                     c->tk, c->tkf, metrics, (float)n, gs, c->dp_world);
  ev_end(c, KID_TK_DECODE, s);
  if (backward) {
    const int rpb = 128;
    const int nrb = (int)((Mp + rpb - 1) / rpb);
    hipLaunchKernelGGL(topk_de_kernel, dim3((d_p + 255) / 256, nrb), dim3(256), 0, s, c->e, c->dh, c->tkf, c->de_b, c->dh_b,
                       c->dbd_part, Mp, d_p, rpb, aux ? 1 : 0, c->tk, c->multi ? c->em : (const float*)nullptr, c->dm_b);
    if (use_csc) {
      // ---- CSC backward: selection sorted by latent, then d W_dec, d W_enc and d b_enc as gathered row sums (topk_sparse.h)
      SparsePasses ps{};
      if (c->multi) { ps.idx[0] = c->multi_idx; ps.vals[0] = c->multi_vals; ps.g[0] = c->dm_b; ps.kcap[0] = c->k4; }
      if (aux && !auxc) { ps.idx[1] = c->aux_idx; ps.vals[1] = c->aux_vals; ps.g[1] = c->dh_b; ps.kcap[1] = c->k_aux_cap; ps.gated[1] = 1; }
      ps.idx[2] = c->top_idx; ps.vals[2] = c->top_vals; ps.g[2] = c->de_b; ps.kcap[2] = k;
      const int nb = (int)((M + CSC_ROWS - 1) / CSC_ROWS);
      const int nseg = (n_p + CSC_MAX_NP - 1) / CSC_MAX_NP;            // dictionary segments of the counting kernels
      const int lds = (n_p < CSC_MAX_NP ? n_p : CSC_MAX_NP) * 2;
      ev_begin(c, KID_TK_DDENSE, s);
      hipLaunchKernelGGL(csc_count_kernel, dim3(nb, nseg), dim3(64), lds, s, ps, c->tk, M, n_p, c->csc_counts);
      hipLaunchKernelGGL(csc_scan_blocks_kernel, dim3((n_p + 63) / 64), dim3(1024), 0, s, c->csc_counts, nb, n_p, c->csc_block_off,
                         c->csc_total);
      hipLaunchKernelGGL(csc_scan_latents_kernel, dim3(1), dim3(1024), 0, s, c->csc_total, n_p, c->csc_start, c->csc_item_start);
      HIP_TRY(hipMemsetAsync(c->csc_multi, 0, 16, s));
      hipLaunchKernelGGL(csc_items_kernel, dim3((n_p + 255) / 256), dim3(256), 0, s, c->csc_item_start, n_p, c->csc_item_latent, c->csc_multi);
      const int nseg_fill = (n_p + CSC_FILL_NP - 1) / CSC_FILL_NP;     // (32-bit position counters: 16 384 latents per 64 KiB)
      hipLaunchKernelGGL(csc_fill_kernel, dim3(nb, nseg_fill), dim3(64), (n_p < CSC_FILL_NP ? n_p : CSC_FILL_NP) * 4, s, ps, c->tk, M, n_p,
                         c->csc_block_off, c->csc_start, c->csc_entries);
      ev_end(c, KID_TK_DDENSE, s);
      ev_begin(c, KID_TK_DWD, s);
      {
        const unsigned blocks = (unsigned)((c->csc_max_items + SB_WAVES - 1) / SB_WAVES);
        auto launch_sb = [&](auto np_tag) {
          constexpr int NP = decltype(np_tag)::value;
          hipLaunchKernelGGL(sparse_bwd_kernel<NP>, dim3(blocks), dim3(64 * SB_WAVES), 0, s, ps, c->xs, c->Wd_b, c->csc_entries, c->csc_start,
                             c->csc_item_start, c->csc_item_latent, n_p, c->csc_part, c->csc_pbe, gWd, gWe, gbe, c->db_part);
        };
        if (d_p == 384) launch_sb(std::integral_constant<int, 3>{});
        else if (d_p == 768) launch_sb(std::integral_constant<int, 6>{});
        else launch_sb(std::integral_constant<int, 10>{});
      }
      ev_end(c, KID_TK_DWD, s);
      ev_begin(c, KID_TK_DWE, s);
      hipLaunchKernelGGL(sparse_combine_kernel, dim3(4096), dim3(256), 0, s, c->csc_part, c->csc_pbe, c->csc_item_start, n_p,
                         d_p, gWd, gWe, gbe, c->db_part, c->csc_multi);
      ev_end(c, KID_TK_DWE, s);
      HIP_TRY(hipGetLastError());
      if (auxc) {
        // ---- AuxK backward over the compact dead set (every launch covers the static maximum and gates itself on the device)
        ev_begin(c, KID_TK_AUX, s);
        // launch sizes from the STALE dead count (pinned copy of an earlier step, never waited for) plus a margin: only an
        // estimate is needed (GemmArgs::grid_hint), the kernels cover whatever the real extent is
        const int nd_hint = c->dead_hint[0];
        int ndp_hint = (nd_hint + nd_hint / 4 + 511) & ~255;
        if (ndp_hint > n_p) ndp_hint = n_p;
        const bool small_dead = (int64_t)ndp_hint * 8 <= n_p;      // estimated grids (persistent instantiation) only then
        {   // d A = [selected, > 0] bf16(d e_hat W_dec[dead]^T), column sums
          GemmArgs g{};
          g.A0 = c->dh_b; g.B0 = c->Wdd_b; g.lda = d_p; g.ldb = d_p;
          g.nbm = (int)(Mp / 128); g.nbn = n_p / 128; g.ktiles0 = g.ktiles = d_p / 64; g.splits = 1;
          g.dyn = c->tkd; g.dyn_dim = GEMM_DYN_N;
          g.grid_hint = (int64_t)ndp_hint * 2 <= n_p ? (int)((Mp / 128) * (ndp_hint / 128)) : 0;   // (one exiting workgroup per
                                                            // missing tile costs more than the persistent loop up to about there)
          EpiTopkDpre e{};
          e.sel = c->aux_dense; e.dpre = c->dpre; e.dbe_part = c->aux_dbe_part; e.n_p = n_p; e.accumulate = 0; e.last = 1;
          rc = launch_gemm<OP_ROW, OP_ROW>(g, e, s);
          if (rc) return rc;
        }
        for (int which = 0; which < 2; ++which) {   // d W_dec[dead] += A_aux^T d e_hat ; d W_enc[dead] += d A^T sae_in
          GemmArgs g{};
          g.A0 = which == 0 ? c->aux_dense : c->dpre; g.B0 = which == 0 ? c->dh_b : c->xs; g.lda = n_p; g.ldb = d_p;
          g.nbm = n_p / 128; g.nbn = d_p / 128; g.ktiles0 = g.ktiles = (int)(Mp / 64);
          // split-K on the DEVICE from the real dead count (gemm_dyn_splits): a few hundred dead latents are a handful of
          // output tiles and K = M is long; between dw_splits and what the slab buffer holds
          g.splits = c->slab_splits > g.ktiles / 8 ? (g.ktiles / 8 > 0 ? g.ktiles / 8 : 1) : c->slab_splits;
          g.dyn_splits_min = 1;
          // the launch only has to cover tiles x factor for the factor the device will choose: its maximum over every possible
          // extent, not the static tile count times the largest factor (a workgroup that starts only to exit costs ~3 us of a
          // CU: 8 700 of them were a quarter of these GEMMs' time)
          {
            const bool big_ = !g_force_gemm128 && g.nbm % 2 == 0 && g.nbn % 2 == 0;
            const int tl = big_ ? 256 : 128, tn = d_p / tl, tm_max = n_p / tl;
            int cover = 1;
            for (int tmm = 1; tmm <= tm_max; ++tmm) {
              const int need = tmm * tn * gemm_dyn_splits(tmm * tn, 1, g.splits, g.ktiles);
              if (need > cover) cover = need;
            }
            g.grid_cover = cover;
          }
          g.dyn = c->tkd; g.dyn_dim = GEMM_DYN_M;
          const bool big = !g_force_gemm128 && g.nbm % 2 == 0 && g.nbn % 2 == 0;       // launch_gemm's kernel choice
          const int tile = big ? 256 : 128;
          g.grid_hint = small_dead ? (ndp_hint / 128) * (d_p / 128) *
                                         gemm_dyn_splits((ndp_hint / tile) * (d_p / tile), g.dyn_splits_min, g.splits, g.ktiles)
                                   : 0;                                                                              // (sizing only)
          EpiSlab e{};
          e.slab = c->slab; e.slab_stride = c->nW; e.ld = d_p;
          rc = launch_gemm<OP_KMAJOR, OP_KMAJOR>(g, e, s);
          if (rc) return rc;
          hipLaunchKernelGGL(aux_scatter_rows_kernel, dim3(n_p / 4), dim3(256), 0, s, c->slab, c->nW, g.dyn_splits_min, g.splits,
                             g.ktiles, tile, c->dead_cols, c->tkd, which == 0 ? gWd : gWe, d_p);
        }
        hipLaunchKernelGGL(aux_scatter_dbe_kernel, dim3((n_p + 63) / 64), dim3(1024), 0, s, c->aux_dbe_part, (int)(Mp / 128), n_p,
                           c->dead_cols, c->tkd, c->db_part, gbe);
        ev_end(c, KID_TK_AUX, s);
        HIP_TRY(hipGetLastError());
      }
      notify_grads(c, c->nW + c->n_p, c->nW, s);                 // d W_dec
    } else {
    if (c->topk_sparse_da) {   // dpre only where a latent was selected: k (+ k_aux) gathered dot products per row
        ev_begin(c, KID_TK_DDENSE, s);
        HIP_TRY(hipMemsetAsync(c->dpre, 0, (size_t)Mp * n_p * 2, s));
        HIP_TRY(hipMemsetAsync(c->dbe_fx, 0, (size_t)n_p * 8, s));
        DactsPasses ps{};
        if (c->multi) { ps.g[0] = c->dm_b; ps.vals[0] = c->multi_vals; ps.idx[0] = c->multi_idx; ps.kcap[0] = c->k4; }
        if (aux) { ps.g[1] = c->dh_b; ps.vals[1] = c->aux_vals; ps.idx[1] = c->aux_idx; ps.kcap[1] = c->k_aux_cap; ps.gated[1] = 1; }
        ps.g[2] = c->de_b; ps.vals[2] = c->top_vals; ps.idx[2] = c->top_idx; ps.kcap[2] = k;
        auto launch_dacts = [&](auto np_tag) {
          constexpr int NP = decltype(np_tag)::value;
          hipLaunchKernelGGL(topk_dacts_kernel<NP>, dim3((unsigned)(Mp / 4)), dim3(256), 0, s, ps, c->Wd_b, c->dpre, c->dbe_fx, M,
                             n_p, c->tk);
        };
        if (d_p == 384) launch_dacts(std::integral_constant<int, 3>{});
        else if (d_p == 768) launch_dacts(std::integral_constant<int, 6>{});
        else launch_dacts(std::integral_constant<int, 10>{});
        ev_end(c, KID_TK_DDENSE, s);
      } else {  // dpre = [selected] (de W_dec^T)  (+ aux part) as a dense GEMM with a masking epilogue
        // (A/B and test path only, topk_dense_backward: its second launch is a host decision, so this path reads num_dead back)
        int num_dead = 0;
        if (aux) {
          HIP_TRY(hipMemcpyAsync(&num_dead, c->tk, 4, hipMemcpyDeviceToHost, s));
          HIP_TRY(hipStreamSynchronize(s));
        }
        const bool aux_now = aux && num_dead > 0;
        GemmArgs g{};
        g.B0 = c->Wd_b; g.lda = d_p; g.ldb = d_p;
        g.nbm = (int)(Mp / 128); g.nbn = n_p / 128; g.ktiles0 = g.ktiles = d_p / 64; g.splits = 1;
        // passes in autograd's execution order (multi-TopK, AuxK, main); each adds into dpre with one bf16 rounding
        const bf16_t* gsrc[3] = {c->multi ? c->dm_b : nullptr, aux_now ? c->dh_b : nullptr, c->de_b};
        const bf16_t* sel[3] = {c->multi_dense, c->aux_dense, c->dense};
        ev_begin(c, KID_TK_DDENSE, s);
        bool first = true;
        rc = SAE_OK;
        for (int pass = 0; pass < 3 && !rc; ++pass) {
          if (!gsrc[pass]) continue;
          g.A0 = gsrc[pass];
          EpiTopkDpre e{};
          e.sel = sel[pass]; e.dpre = c->dpre; e.dbe_part = c->db_part; e.n_p = n_p; e.accumulate = first ? 0 : 1; e.last = pass == 2;
          rc = launch_gemm<OP_ROW, OP_ROW>(g, e, s);
          first = false;
        }
        ev_end(c, KID_TK_DDENSE, s);
        if (rc) return rc;
      }
      const int splits = c->dw_splits;
      {  // dW_dec[n][d] = dense^T de (+ aux_dense^T de_hat) (+ multi_dense^T dm as a second launch into its own slabs)
        GemmArgs g{};
        g.A0 = c->dense; g.B0 = c->de_b; g.A1 = c->aux_dense; g.B1 = c->dh_b; g.lda = n_p; g.ldb = d_p;
        g.nbm = n_p / 128; g.nbn = d_p / 128; g.ktiles0 = (int)(Mp / 64); g.ktiles = aux ? 2 * g.ktiles0 : g.ktiles0;
        g.seg1_gate = aux ? c->tk : nullptr;          // the AuxK pair joins only while latents are dead
        g.splits = splits > g.ktiles0 ? g.ktiles0 : splits;
        const bool slabs = g.splits > 1 || c->multi;
        EpiSlab e{};
        e.slab = slabs ? c->slab : gWd; e.slab_stride = c->nW; e.ld = d_p;
        ev_begin(c, KID_TK_DWD, s);
        rc = launch_gemm<OP_KMAJOR, OP_KMAJOR>(g, e, s);
        int nslabs = g.splits;
        if (!rc && c->multi) {
          GemmArgs g2 = g;
          g2.A0 = c->multi_dense; g2.B0 = c->dm_b; g2.A1 = nullptr; g2.B1 = nullptr; g2.ktiles = g2.ktiles0; g2.seg1_gate = nullptr;
          EpiSlab e2 = e;
          e2.slab = c->slab + (int64_t)g.splits * c->nW;
          rc = launch_gemm<OP_KMAJOR, OP_KMAJOR>(g2, e2, s);
          nslabs = 2 * g.splits;
        }
        if (!rc && slabs) {
          const int64_t n4 = c->nW / 4;
          hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, c->slab, gWd, n4, n4, nslabs);
        }
        ev_end(c, KID_TK_DWD, s);
        if (rc) return rc;
        notify_grads(c, c->nW + c->n_p, c->nW, s);     // d W_dec is final: its all-reduce runs under the d W_enc GEMM
      }
      {  // dW_enc[n][d] = dpre^T sae_in
        GemmArgs g{};
        g.A0 = c->dpre; g.B0 = c->xs; g.lda = n_p; g.ldb = d_p;
        g.nbm = n_p / 128; g.nbn = d_p / 128; g.ktiles0 = g.ktiles = (int)(Mp / 64);
        g.splits = splits > g.ktiles ? g.ktiles : splits;
        EpiSlab e{};
        e.slab = g.splits > 1 ? c->slab : gWe; e.slab_stride = c->nW; e.ld = d_p;
        ev_begin(c, KID_TK_DWE, s);
        rc = launch_gemm<OP_KMAJOR, OP_KMAJOR>(g, e, s);
        if (!rc && g.splits > 1) {
          const int64_t n4 = c->nW / 4;
          hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, c->slab, gWe, n4, n4, g.splits);
        }
        ev_end(c, KID_TK_DWE, s);
        if (rc) return rc;
      }
    }
    ev_begin(c, KID_REDUCE, s);
    if (use_csc) {
      // (d b_enc and its exact copy came out of sparse_combine_kernel)
    } else if (c->topk_sparse_da)
      hipLaunchKernelGGL(topk_dbe_from_fx_kernel, dim3((n_p + 255) / 256), dim3(256), 0, s, c->dbe_fx, gbe, c->db_part, n_p);
    else
      hipLaunchKernelGGL(reduce_db_kernel, dim3(n_p / 32), dim3(256), 0, s, c->db_part, gbe, (int)(Mp / 128), n_p);
    // d b_dec also receives -sum_rows(dpre W_enc) through sae_in = x - b_dec; that row sum is a GEMV on d b_enc
    ev_begin(c, KID_TK_DSAE, s);
    const int nchunks = (n_p + 63) / 64;
    hipLaunchKernelGGL(topk_dsae_colsum_kernel, dim3((d_p + 255) / 256, nchunks), dim3(256), 0, s,
                       (use_csc || c->topk_sparse_da) ? c->db_part : gbe, c->We_b, c->ds_part, n_p, d_p);
    ev_end(c, KID_TK_DSAE, s);
    hipLaunchKernelGGL(topk_dbd_kernel, dim3((d_p + 63) / 64), dim3(1024), 0, s, c->dbd_part, nrb, c->ds_part, nchunks, gbd, d_p);
    ev_end(c, KID_REDUCE, s);
    notify_grads(c, 0, c->nW + c->n_p, s);                                                     // d W_enc | d b_enc
    notify_grads(c, 2 * c->nW + c->n_p, c->d_p + SAE_NUM_METRICS + c->n_p, s);                 // d b_dec | scalars | did_fire
  }
  ev_end(c, KID_STEP_TOTAL, s);
  HIP_TRY(hipGetLastError());
  c->last_M = M;
  c->last_M_p = Mp;
  c->metrics_fresh = true;
  c->prof_tick++;
  return SAE_OK;
}

static int dispatch_fwd_bwd_inner(sae_ctx* c, const void* x, int64_t M, int x_dtype, hipStream_t s, bool backward);

// With the engine's own communicator (sae_dist_init) a training step is bracketed by: (1) the batch statistics and their
// all-reduce on the communication stream, concurrent with the weight preparation and the forward kernels (they depend on
// x only); (2) [the step's kernels; every gradient range is all-reduced on the communication stream as soon as it is
// final, notify_grads]; (3) the compute stream waits for the last of those all-reduces.
static int dispatch_fwd_bwd(sae_ctx* c, const void* x, int64_t M, int x_dtype, void* stream, bool backward) {
  if (!c || !x) return fail(SAE_ERR_INVALID, "null argument");
  if (M <= 0 || M > c->cfg.max_rows) return fail(SAE_ERR_INVALID, "M=%lld outside (0, max_rows=%lld]", (long long)M, (long long)c->cfg.max_rows);
  USE_DEVICE(c);
  hipStream_t s = (hipStream_t)stream;
  const bool dp = c->dist && backward;
  c->gn_from_exchange = false;
  if (dp) c->dist_error = 0;
  if (dp && !inline_stats(c)) {
    HIP_TRY(hipEventRecord(c->ev_x, s));
    HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_x, 0));
    int rc = batch_stats_dispatch(c, x, M, x_dtype, c->comm_stream);
    if (rc) return rc;
    if (c->p2p) {
      P2PSeg g{};
      g.off = 0; g.pitch = c->stats_n; g.rows = 1; g.cols = (int)c->stats_n; g.kind = P2P_F64;
      p2p_launch(c, 0, &g, 1, p2p_grid(c->stats_n / 2), nullptr, c->comm_stream);
    } else {
      ev_begin(c, KID_STATS_XCHG, c->comm_stream);
      NCCL_TRY(ncclAllReduce(c->stats, c->stats, (size_t)c->stats_n, ncclDouble, ncclSum, c->comm, c->comm_stream));
      ev_end(c, KID_STATS_XCHG, c->comm_stream);
    }
    HIP_TRY(hipEventRecord(c->ev_stats, c->comm_stream));
  }
  int rc = dispatch_fwd_bwd_inner(c, x, M, x_dtype, s, backward);
  if (rc) return rc;
  if (dp) {
    if (c->dist_error) return fail(SAE_ERR_HIP, "%s", c->dist_errmsg);
    if (!(c->use_fused_bwd && !c->topk && c->bwd_ranges == 1)) {      // (a single-range fused backward exchanged on the compute stream itself)
      HIP_TRY(hipEventRecord(c->ev_done, c->comm_stream));
      HIP_TRY(hipStreamWaitEvent(s, c->ev_done, 0));
    }
  }
  return SAE_OK;
}

static int dispatch_fwd_bwd_inner(sae_ctx* c, const void* x, int64_t M, int x_dtype, hipStream_t s, bool backward) {
  c->last_dtype = x_dtype;
  c->last_fwd_e32 = false;
  if (c->topk) {
    switch (x_dtype) {
      case SAE_DTYPE_F32: return topk_fwd_bwd<float>(c, (const float*)x, M, s, backward);
      case SAE_DTYPE_F16: return topk_fwd_bwd<_Float16>(c, (const _Float16*)x, M, s, backward);
      case SAE_DTYPE_BF16: return topk_fwd_bwd<bf16_t>(c, (const bf16_t*)x, M, s, backward);
      default: return fail(SAE_ERR_INVALID, "unknown x_dtype %d", x_dtype);
    }
  }
  switch (x_dtype) {
    case SAE_DTYPE_F32: return fwd_bwd_impl<float>(c, (const float*)x, M, s, backward);
    case SAE_DTYPE_F16: return fwd_bwd_impl<_Float16>(c, (const _Float16*)x, M, s, backward);
    case SAE_DTYPE_BF16: return fwd_bwd_impl<bf16_t>(c, (const bf16_t*)x, M, s, backward);
    default: return fail(SAE_ERR_INVALID, "unknown x_dtype %d", x_dtype);
  }
}

// ---- fp32 evaluation forward (eval_fp32.h): validate() of the reference on device='cpu' runs without autocast
static int e32_ensure(sae_ctx* c, int64_t M) {
  if (c->e32_rows >= M && c->e32_x) return SAE_OK;
  for (void** p : {(void**)&c->e32_x, (void**)&c->e32_pre, (void**)&c->e32_sel, (void**)&c->e32_xhat}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  c->e32_rows = 0;
  const int64_t rows = M;
  HIP_TRY(hipMalloc((void**)&c->e32_x, (size_t)rows * c->d_p * 4));
  HIP_TRY(hipMalloc((void**)&c->e32_pre, (size_t)rows * c->n_p * 4));
  if (c->topk) HIP_TRY(hipMalloc((void**)&c->e32_sel, (size_t)rows * c->n_p * 4));
  HIP_TRY(hipMalloc((void**)&c->e32_xhat, (size_t)rows * c->d_p * 4));
  if (!c->e32_part) HIP_TRY(hipMalloc((void**)&c->e32_part, (size_t)E32_PART_DOUBLES * 8));
  if (!c->e32_colmax) HIP_TRY(hipMalloc((void**)&c->e32_colmax, (size_t)c->n_p * 4));
  c->e32_rows = rows;
  return SAE_OK;
}

template <bool BT>
static void e32_gemm(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(e32_gemm_kernel<BT>, dim3((N + E32_BN - 1) / E32_BN, (unsigned)((M + E32_BM - 1) / E32_BM)), dim3(256), 0, s, A, lda, B, ldb,
                     C, ldc, (int)M, N, K);
}

template <typename T>
static int eval_fp32_impl(sae_ctx* c, const T* x, int64_t M, hipStream_t s) {
  const int d = c->d, n = c->n, d_p = c->d_p, n_p = c->n_p;
  int rc = e32_ensure(c, M);
  if (rc) return rc;
  // partial-sum layout of e32_part (doubles): [0, E32_L1_PARTS) latent sums | 4 x E32_RES_BLOCKS residual sums | ... multi-TopK residual | total variance
  double *l1_part = c->e32_part, *res_part = c->e32_part + E32_L1_PARTS, *res2_part = res_part + 4 * E32_RES_BLOCKS, *tv_part = res2_part + 4 * E32_RES_BLOCKS;
  int grid_x = (int)((M * d_p + 255) / 256);
  if (grid_x > 2048) grid_x = 2048;
  HIP_TRY(hipMemsetAsync(c->e32_colmax, 0, (size_t)n_p * 4, s));
  E32Final f{};
  f.M = M; f.d = d; f.alpha = (float)c->cfg.recon_alpha; f.topk = c->topk ? 1 : 0;
  if (!c->topk) {
    // W <- W / max(||W[:, j]||, 1e-12) in place first, as every forward of the reference does (l1autoencoder.py:71-73); the fp32
    // master then holds exactly the weights this forward multiplies by
    prep_weights_l1(c, s);
    settle_weights(c, s);
    const float *W = c->P, *b = c->P + c->nW;
    hipLaunchKernelGGL(e32_load_x_kernel<T>, dim3(grid_x), dim3(256), 0, s, x, c->e32_x, (const float*)nullptr, M, d, d_p);
    e32_gemm<false>(c->e32_x, d_p, W, n_p, c->e32_pre, n_p, M, n_p, d_p, s);                       // x W: W is [d_p][n_p]
    const dim3 gb((n_p + 255) / 256, (unsigned)((M + 63) / 64));
    if ((int64_t)gb.x * gb.y > E32_L1_PARTS) return fail(SAE_ERR_INVALID, "fp32 evaluation: %lld rows x %d latents exceed its partial-sum buffer", (long long)M, n_p);
    hipLaunchKernelGGL(e32_bias_relu_kernel, gb, dim3(256), 0, s, c->e32_pre, b, M, n, n_p, l1_part, c->e32_colmax);
    e32_gemm<true>(c->e32_pre, n_p, W, n_p, c->e32_xhat, d_p, M, d_p, n_p, s);                     // c W^T: W as [N = d_p][K = n_p]
    hipLaunchKernelGGL(e32_residual_kernel<T>, dim3(E32_RES_BLOCKS), dim3(256), 0, s, c->e32_xhat, d_p, (const float*)nullptr, x, M, d, res_part);
    f.l1_part = l1_part; f.n_l1 = (int)(gb.x * gb.y);
    f.res_part = res_part; f.n_res = E32_RES_BLOCKS;
  } else {
    const float *We = c->P, *be = c->P + c->nW, *Wd = c->P + c->nW + c->n_p, *bd = c->P + 2 * c->nW + c->n_p;
    hipLaunchKernelGGL(e32_load_x_kernel<T>, dim3(grid_x), dim3(256), 0, s, x, c->e32_x, bd, M, d, d_p);          // sae_in = x - b_dec
    e32_gemm<true>(c->e32_x, d_p, We, d_p, c->e32_pre, n_p, M, n_p, d_p, s);                       // sae_in W_enc^T: W_enc is [n_p][d_p]
    const dim3 gb((n_p + 255) / 256, (unsigned)((M + 63) / 64));
    if ((int64_t)gb.x * gb.y > E32_L1_PARTS) return fail(SAE_ERR_INVALID, "fp32 evaluation: %lld rows x %d latents exceed its partial-sum buffer", (long long)M, n_p);
    hipLaunchKernelGGL(e32_bias_relu_kernel, gb, dim3(256), 0, s, c->e32_pre, be, M, n, n_p, l1_part, (int*)nullptr);
    const int64_t T_rows = c->rows_per_file > 0 && M % c->rows_per_file == 0 ? c->rows_per_file : M;
    hipLaunchKernelGGL(e32_total_variance_kernel<T>, dim3(E32_RES_BLOCKS), dim3(256), 0, s, x, M / T_rows, T_rows * d, tv_part);
    f.tv_part = tv_part; f.n_tv = E32_RES_BLOCKS;
    hipLaunchKernelGGL(e32_topk_select_kernel, dim3((unsigned)M), dim3(256), 0, s, c->e32_pre, c->e32_sel, n, n_p, c->k, (int*)nullptr);
    e32_gemm<false>(c->e32_sel, n_p, Wd, d_p, c->e32_xhat, d_p, M, d_p, n_p, s);                   // dense W_dec: W_dec is [n_p][d_p]
    hipLaunchKernelGGL(e32_residual_kernel<T>, dim3(E32_RES_BLOCKS), dim3(256), 0, s, c->e32_xhat, d_p, bd, x, M, d, res_part);
    f.res_part = res_part; f.n_res = E32_RES_BLOCKS;
    if (c->multi) {              // cfg.multi_topk: the returned encoding is the 4k one (topkautoencoder.py:134-136)
      hipLaunchKernelGGL(e32_topk_select_kernel, dim3((unsigned)M), dim3(256), 0, s, c->e32_pre, c->e32_sel, n, n_p, c->k4, (int*)nullptr);
      e32_gemm<false>(c->e32_sel, n_p, Wd, d_p, c->e32_xhat, d_p, M, d_p, n_p, s);
      hipLaunchKernelGGL(e32_residual_kernel<T>, dim3(E32_RES_BLOCKS), dim3(256), 0, s, c->e32_xhat, d_p, bd, x, M, d, res2_part);
      f.res2_part = res2_part; f.n_res2 = E32_RES_BLOCKS;
    }
    hipLaunchKernelGGL(e32_colmax_kernel, dim3((n + 255) / 256, (unsigned)((M + 63) / 64)), dim3(256), 0, s, c->e32_sel, M, n, n_p, c->e32_colmax);
  }
  hipLaunchKernelGGL(e32_finalize_kernel, dim3(1), dim3(256), 0, s, f, c->G + c->nparams);
  HIP_TRY(hipGetLastError());
  c->last_M = M;
  c->last_M_p = round_up(M, c->row_pad);
  c->last_fwd_e32 = true;
  c->metrics_fresh = false;
  return SAE_OK;
}

static int eval_fp32_dispatch(sae_ctx* c, const void* x, int64_t M, int x_dtype, void* stream) {
  if (!c || !x) return fail(SAE_ERR_INVALID, "null argument");
  if (M <= 0) return fail(SAE_ERR_INVALID, "M=%lld must be positive", (long long)M);
  USE_DEVICE(c);
  hipStream_t s = (hipStream_t)stream;
  c->last_dtype = x_dtype;
  switch (x_dtype) {
    case SAE_DTYPE_F32: return eval_fp32_impl<float>(c, (const float*)x, M, s);
    case SAE_DTYPE_F16: return eval_fp32_impl<_Float16>(c, (const _Float16*)x, M, s);
    case SAE_DTYPE_BF16: return eval_fp32_impl<bf16_t>(c, (const bf16_t*)x, M, s);
    default: return fail(SAE_ERR_INVALID, "unknown x_dtype %d", x_dtype);
  }
}

extern "C" int sae_set_eval_precision(sae_ctx* c, int precision) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  if (precision != SAE_PREC_BF16 && precision != SAE_PREC_FP32)
    return fail(SAE_ERR_INVALID, "Invalid evaluation precision: %d, must be SAE_PREC_BF16 (the training kernels) or SAE_PREC_FP32", precision);
  c->eval_prec = precision == SAE_PREC_FP32 ? 1 : 0;
  return SAE_OK;
}

extern "C" int sae_forward_backward(sae_ctx* c, const void* x, int64_t M, int x_dtype, void* stream) {
  return dispatch_fwd_bwd(c, x, M, x_dtype, stream, true);
}
extern "C" int sae_eval(sae_ctx* c, const void* x, int64_t M, int x_dtype, void* stream) {
  if (c && c->eval_prec == 1) return eval_fp32_dispatch(c, x, M, x_dtype, stream);
  return dispatch_fwd_bwd(c, x, M, x_dtype, stream, false);
}

extern "C" int sae_optimizer_step(sae_ctx* c, double lr, double grad_scale, void* stream) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  hipStream_t s = (hipStream_t)stream;
  c->step += 1;
  const double t = (double)c->step, b1 = c->cfg.beta1, b2 = c->cfg.beta2;
  const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
  OptArgs a{};
  a.lr = (float)lr;
  a.grad_scale = (float)grad_scale;
  a.clip_thresh = (float)c->cfg.clip_thresh;
  a.weight_decay = (float)c->cfg.weight_decay;
  a.beta1 = (float)b1;
  a.beta2 = (float)b2;
  a.eps = (float)c->cfg.eps;
  a.one_minus_beta1 = (float)(1.0 - b1);
  a.one_minus_beta2 = (float)(1.0 - b2);
  a.bc1 = (float)bc1;
  a.bc2_sqrt = (float)sqrt(bc2);
  a.step_size = (float)(lr / bc1);
  a.is_radam = c->cfg.optimizer == SAE_OPT_RADAM;
  a.scale_metrics = c->metrics_fresh ? 1 : 0;
  c->metrics_fresh = false;
  if (a.is_radam) {
    const double rho_inf = 2.0 / (1.0 - b2) - 1.0;
    const double rho_t = rho_inf - 2.0 * t * pow(b2, t) / bc2;
    a.rectify = rho_t > 5.0;
    a.rect = a.rectify ? (float)sqrt((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t)) : 0.f;
  }
  ev_begin(c, KID_OPT, s);
  const int64_t n4 = c->nparams / 4;  // nW and n_p are multiples of 128
  int gblocks = (int)((n4 + 255) / 256);
  if (gblocks > 1024) gblocks = 1024;
  // the reduction kernel already left the local sum of squares; it is only valid when nothing (no all-reduce, no
  // rescaling) touched the gradient buffer in between, i.e. for the single-GPU sae_step path
  if (c->grads_in_bf16) {     // data parallel, bf16 payload: the summed gradient comes back to fp32 in the same pass
    hipLaunchKernelGGL(gnorm_from_bf16_kernel, dim3(gblocks), dim3(256), 0, s, c->Gb, c->G, n4, a.grad_scale, c->gn_part);
    c->grads_in_bf16 = false;
  } else if (c->gn_valid && grad_scale == 1.0 && c->step_fused_call) {
    gblocks = c->gn_blocks;
  } else {
    hipLaunchKernelGGL(gnorm_partial_kernel, dim3(gblocks), dim3(256), 0, s, c->G, n4, a.grad_scale, c->gn_part);
  }
  c->gn_valid = false;
  int oblocks = (int)((n4 + 255) / 256);
  if (oblocks > 2048) oblocks = 2048;
  if (!c->topk && !c->fp8 && c->d_p <= 384 && c->cfg.debug_flags != 79 && c->cfg.debug_flags != 80) {
    // L1, small d: the update that also prepares the next forward's weights (column norms, bf16 copies)
    const int norm_on_load = (c->wn_pending && c->wn_fwd_seen) ? 1 : 0;
    hipLaunchKernelGGL(optimizer_l1_cols_kernel, dim3(c->n_p / OPTC_COLS + c->n_p / 128), dim3(OPTC_THREADS), optc_lds_bytes(c->d_p), s, c->P,
                       c->Mom, c->Var, c->G, c->d_p, c->n_p, c->gn_part, gblocks, a, c->G + c->nparams, c->cnorm, norm_on_load, c->Wb,
                       c->Wt);
    c->wn_pending = true;
    c->wn_fwd_seen = false;
    c->cn_valid = false;
  } else if (!c->topk && c->cfg.debug_flags != 79) {      // L1: tiled update that also leaves the column-norm partials of the new weights
    settle_weights(c, s);
    hipLaunchKernelGGL(optimizer_l1_kernel, dim3(c->n_p / 128, c->d_p / 32 + 1), dim3(256), 0, s, c->P, c->Mom, c->Var, c->G, c->d_p,
                       c->n_p, c->gn_part, gblocks, a, c->G + c->nparams, c->cn_part);
    c->cn_valid = true;
  } else {
    if (!c->topk) settle_weights(c, s);
    OptCast cast{};
    if (c->topk) {       // the bf16 copies of W_enc and W_dec leave with the update: no cast pass in the next step
      cast.off4[0] = 0; cast.len4[0] = c->nW / 4; cast.dst[0] = c->We_b;
      cast.off4[1] = (c->nW + c->n_p) / 4; cast.len4[1] = c->nW / 4; cast.dst[1] = c->Wd_b;
    }
    hipLaunchKernelGGL(optimizer_kernel, dim3(oblocks), dim3(256), 0, s, c->P, c->Mom, c->Var, c->G, n4, c->gn_part, gblocks,
                       a, c->G + c->nparams, cast);
    c->cn_valid = false;
    c->wb_valid = c->topk;
  }
  if (c->topk)   // train_sae.py:443-446 with the (possibly data-parallel summed) did_fire flags
    hipLaunchKernelGGL(nfsf_update_kernel, dim3((c->n + 255) / 256), dim3(256), 0, s, c->nfsf, c->G + c->nparams + SAE_NUM_METRICS,
                       c->n, (long long)(c->last_M * (grad_scale > 0 ? (long long)llround(1.0 / grad_scale) : 1)),
                       c->dp_world > 0 ? c->stats : (const double*)nullptr);
  ev_end(c, KID_OPT, s);
  HIP_TRY(hipGetLastError());
  return SAE_OK;
}

extern "C" int sae_set_topk_options(sae_ctx* c, double dead_feature_threshold, int64_t rows_per_file) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  if (!c->topk) return fail(SAE_ERR_STATE, "not a topk context");
  c->dead_threshold = dead_feature_threshold;
  c->rows_per_file = rows_per_file;
  return SAE_OK;
}

extern "C" int sae_step(sae_ctx* c, const void* x, int64_t M, int x_dtype, double lr, void* stream) {
  int rc = sae_forward_backward(c, x, M, x_dtype, stream);
  if (rc) return rc;
  // (data parallel with global statistics: the all-reduced gradient already is the whole batch's -- grad_scale stays 1)
  // the sum of squares the backward left is the clip norm when nothing summed the gradient over ranks afterwards -- or when
  // the peer exchange itself took it from the summed values
  c->step_fused_call = c->dp_world == 0 || c->gn_from_exchange;
  rc = sae_optimizer_step(c, lr, 1.0, stream);
  c->step_fused_call = false;
  return rc;
}

extern "C" int sae_read_metrics(sae_ctx* c, float out[SAE_NUM_METRICS], void* stream) {
  if (!c || !out) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  HIP_TRY(hipMemcpyAsync(out, c->G + c->nparams, SAE_NUM_METRICS * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return SAE_OK;
}

// TopK: the masked dense rows of the last forward, written on demand when the step itself did not need them
static int ensure_dense(sae_ctx* c) {
  if (!c->topk || c->dense_valid || c->last_M <= 0) return SAE_OK;
  HIP_TRY(hipDeviceSynchronize());
  hipLaunchKernelGGL(topk_densify_kernel, dim3((unsigned)c->last_M_p), dim3(256), 0, (hipStream_t)0, c->top_vals, c->top_idx, c->k,
                     c->dense, c->n_p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  c->dense_valid = true;
  return SAE_OK;
}

extern "C" int sae_debug_read(sae_ctx* c, int which, float* out, int64_t cap) {
  if (!c || !out) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  if (which == 0) {
    int rc_d = ensure_dense(c);
    if (rc_d) return rc_d;
  }
  const int64_t M = c->last_M;
  if (which == 0 || which == 1) {
    const int cols = which == 0 ? c->n : c->d, ld = which == 0 ? c->n_p : c->d_p;
    if (cap < M * cols) return fail(SAE_ERR_INVALID, "capacity too small");
    std::vector<uint16_t> tmp((size_t)M * ld);
    HIP_TRY(hipMemcpy(tmp.data(), which == 0 ? (c->topk ? (void*)c->dense : (void*)c->c) : (c->topk ? (void*)c->de_b : (void*)c->dxh),
                      tmp.size() * 2, hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < M; ++r)
      for (int j = 0; j < cols; ++j) {
        uint32_t u = (uint32_t)tmp[(size_t)r * ld + j] << 16;
        float f;
        memcpy(&f, &u, 4);
        out[r * cols + j] = f;
      }
    return SAE_OK;
  }
  if (which >= 7 && which <= 9) {   // fp8 path: 7 = the scales scal8[8]; 8 = x8 [M][d], 9 = c8 [M][n] decoded (still scaled)
    if (!c->fp8) return fail(SAE_ERR_INVALID, "not an fp8 context");
    if (which == 7) {
      if (cap < S8_COUNT) return fail(SAE_ERR_INVALID, "capacity too small");
      HIP_TRY(hipMemcpy(out, c->scal8, S8_COUNT * 4, hipMemcpyDeviceToHost));
      return SAE_OK;
    }
    const int cols = which == 8 ? c->d : c->n, ld = which == 8 ? c->d_p : c->n_p;
    if (cap < M * cols) return fail(SAE_ERR_INVALID, "capacity too small");
    std::vector<unsigned char> tmp((size_t)M * ld);
    HIP_TRY(hipMemcpy(tmp.data(), which == 8 ? c->x8 : c->c8, tmp.size(), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < M; ++r)
      for (int j = 0; j < cols; ++j) {      // OCP e4m3fn: 1 sign, 4 exponent (bias 7), 3 mantissa bits; no infinities
        const unsigned b = tmp[(size_t)r * ld + j], e = (b >> 3) & 15, m = b & 7;
        float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, (int)e - 7);
        if (e == 15 && m == 7) v = NAN;
        out[r * cols + j] = (b & 0x80) ? -v : v;
      }
    return SAE_OK;
  }
  if (which == 6) {   // clock stamps of the fused backward (debug_flags == 66): [wg][4] as floats
    // [0 .. grid): loop stamps per workgroup {loop cycles, loop time in 10 ns ticks, steps, 0}; [grid .. 2 grid): whole-kernel cycles
    const int64_t grid = (c->bal_m > 0 && c->bwd_ranges == 1) ? c->bal_grid : (int64_t)(c->n_p / 128) * c->bwd_splits;
    const int64_t nq = grid * 4 * 2;
    if (cap < nq) return fail(SAE_ERR_INVALID, "capacity too small");
    std::vector<unsigned long long> tmp((size_t)nq);
    HIP_TRY(hipMemcpy(tmp.data(), reinterpret_cast<unsigned long long*>(c->dpre) + (1 << 16), tmp.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < nq; ++i) out[i] = (float)tmp[(size_t)i];
    return SAE_OK;
  }
#ifdef CSCF_STAMP
  if (which == 13) {   // phase stamps of csc_fill (-DCSCF_STAMP builds only): [workgroup][8] cycles since its start
    const int64_t rows = cap / 8 < 4096 ? cap / 8 : 4096;
    std::vector<unsigned long long> tmp((size_t)rows * 8);
    HIP_TRY(hipMemcpyFromSymbol(tmp.data(), HIP_SYMBOL(cscf_stamp_buf), tmp.size() * 8, 0, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < rows * 8; ++i) out[i] = (float)tmp[(size_t)i];
    return SAE_OK;
  }
#endif
#ifdef TSEL_STAMP
  if (which == 12) {   // phase stamps of the tile-driven main select (-DTSEL_STAMP builds only): [row][8] cycles since the row's start
    const int64_t rows = cap / 8 < 4096 ? cap / 8 : 4096;
    std::vector<unsigned long long> tmp((size_t)rows * 8);
    HIP_TRY(hipMemcpyFromSymbol(tmp.data(), HIP_SYMBOL(tsel_stamp_buf), tmp.size() * 8, 0, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < rows * 8; ++i) out[i] = (float)tmp[(size_t)i];
    return SAE_OK;
  }
#endif
  if (which == 11) {   // phase stamps of the compact AuxK select (-DSEL_STAMP builds only): [row][8] cycles since the row's start
    if (!c->topk || !c->aux_dense) return fail(SAE_ERR_INVALID, "no AuxK buffers");
    const int64_t rows = cap / 8 < c->last_M ? cap / 8 : c->last_M;
    std::vector<unsigned long long> tmp((size_t)rows * 8);
    HIP_TRY(hipMemcpy2D(tmp.data(), 64, reinterpret_cast<const char*>(c->aux_dense) + (size_t)c->n_p * 2 - 64, (size_t)c->n_p * 2, 64, (size_t)rows,
                        hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < rows * 8; ++i) out[i] = (float)tmp[(size_t)i];
    return SAE_OK;
  }
  if (which == 5) {   // stamp sums of the diagnostic fused forward: [wg][wave][8] as floats
    const int64_t nq = (c->last_M / FF_BM) * 4 * 8;
    if (cap < nq) return fail(SAE_ERR_INVALID, "capacity too small");
    std::vector<unsigned long long> tmp((size_t)nq);
    HIP_TRY(hipMemcpy(tmp.data(), c->dpre, tmp.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < nq; ++i) out[i] = (float)tmp[(size_t)i];
    return SAE_OK;
  }
  if (which == 2 && c->topk) {
    const int64_t nd = (int64_t)c->n * c->d;
    if (cap < 2 * nd + c->n + c->d) return fail(SAE_ERR_INVALID, "capacity too small");
    float* const ext[4] = {out, out + nd, out + nd + c->n, out + 2 * nd + c->n};
    return xfer_flat_topk(c, c->G, ext, 0, 0);
  }
  if (which == 3 && c->topk) {   // top-k indices of the last step as floats [M][k]
    if (cap < M * c->k) return fail(SAE_ERR_INVALID, "capacity too small");
    std::vector<int> tmp((size_t)M * c->k);
    HIP_TRY(hipMemcpy(tmp.data(), c->top_idx, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i) out[i] = (float)tmp[i];
    return SAE_OK;
  }
  if (which == 2) {
    if (cap < (int64_t)c->d * c->n + c->n) return fail(SAE_ERR_INVALID, "capacity too small");
    int rc = xfer_flat(c, c->G, out, out + (int64_t)c->d * c->n, 0, 0);
    return rc;
  }
  return fail(SAE_ERR_INVALID, "unknown debug tensor %d", which);
}

extern "C" int sae_latent_buffer(sae_ctx* c, void** dev_ptr, int64_t* row_stride) {
  if (!c || !dev_ptr || !row_stride) return fail(SAE_ERR_INVALID, "null argument");
  if (c->last_M <= 0) return fail(SAE_ERR_STATE, "no forward has run yet");
  USE_DEVICE(c);
  if (c->last_fwd_e32) return fail(SAE_ERR_STATE, "the last forward was an fp32 evaluation: it leaves no bf16 latent rows");
  {
    int rc_d = ensure_dense(c);
    if (rc_d) return rc_d;
  }
  *dev_ptr = c->topk ? (void*)c->dense : (void*)c->c;
  *row_stride = c->n_p;
  return SAE_OK;
}

extern "C" int sae_topk_indices(sae_ctx* c, void** dev_ptr, int* k) {
  if (!c || !dev_ptr || !k) return fail(SAE_ERR_INVALID, "null argument");
  if (!c->topk) return fail(SAE_ERR_INVALID, "sae_topk_indices: not a TopK context");
  if (c->last_M <= 0) return fail(SAE_ERR_STATE, "no forward has run yet");
  *dev_ptr = c->top_idx;
  *k = c->k;
  return SAE_OK;
}

extern "C" int sae_multi_topk_buffers(sae_ctx* c, void** dense_dev, int64_t* row_stride, void** idx_dev, int* k4) {
  if (!c || !dense_dev || !row_stride || !idx_dev || !k4) return fail(SAE_ERR_INVALID, "null argument");
  if (!c->topk || !c->multi) return fail(SAE_ERR_INVALID, "sae_multi_topk_buffers: not a TopK context with multi_topk");
  if (c->last_M <= 0) return fail(SAE_ERR_STATE, "no forward has run yet");
  if (!c->multi_dense_valid) {      // a training step on the sparse backward keeps the 4k selection compact: densify on demand
    USE_DEVICE(c);
    HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(topk_densify_kernel, dim3((unsigned)c->last_M_p), dim3(256), 0, (hipStream_t)0, c->multi_vals, c->multi_idx, c->k4,
                       c->multi_dense, c->n_p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    c->multi_dense_valid = true;
  }
  *dense_dev = c->multi_dense;
  *row_stride = c->n_p;
  *idx_dev = c->multi_idx;
  *k4 = c->k4;
  return SAE_OK;
}

extern "C" int sae_decode(sae_ctx* c, const void* latent, int latent_dtype, int64_t ld, int64_t M, float* x_hat, void* stream) {
  if (!c || !latent || !x_hat) return fail(SAE_ERR_INVALID, "null argument");
  if (M <= 0 || M > c->cfg.max_rows) return fail(SAE_ERR_INVALID, "M=%lld outside (0, max_rows=%lld]", (long long)M, (long long)c->cfg.max_rows);
  if (ld < c->n) return fail(SAE_ERR_INVALID, "row stride %lld < n_dict %d", (long long)ld, c->n);
  if (latent_dtype != SAE_DTYPE_F32 && latent_dtype != SAE_DTYPE_BF16) return fail(SAE_ERR_INVALID, "latent dtype must be f32 or bf16");
  USE_DEVICE(c);
  hipStream_t s = (hipStream_t)stream;
  const int d_p = c->d_p, n_p = c->n_p;
  const int64_t Mp = round_up(M, c->row_pad);
  bf16_t* lat = c->dpre;                       // [M_p][n_p] scratch that no forward output lives in
  const int64_t total = Mp * (int64_t)n_p;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (latent_dtype == SAE_DTYPE_F32)
    hipLaunchKernelGGL(pad_latent_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)latent, ld, M, c->n, lat, Mp, n_p);
  else
    hipLaunchKernelGGL(pad_latent_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)latent, ld, M, c->n, lat, Mp, n_p);
  GemmArgs g{};
  g.A0 = lat; g.lda = n_p;
  g.nbm = (int)(Mp / 128); g.nbn = d_p / 128; g.ktiles0 = g.ktiles = n_p / 64; g.splits = 1;
  EpiStoreF32 e{};
  e.out = x_hat; e.M = M; e.d = c->d;
  int rc;
  if (c->topk) {
    // W_dec [n_p][d_p] fp32 -> bf16 (K = n is the slow dimension of this operand: transposed-read mode)
    float* Wd = c->P + c->nW + c->n_p;
    const int64_t n8 = c->nW / 8;
    int cg = (int)((n8 + 255) / 256);
    if (cg > 2048) cg = 2048;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(cg), dim3(256), 0, s, Wd, c->Wd_b, n8);
    g.B0 = c->Wd_b; g.ldb = d_p;
    e.bias = c->P + 2 * c->nW + c->n_p;
    rc = launch_gemm<OP_ROW, OP_KMAJOR>(g, e, s);
  } else {
    // current W [d_p][n_p] as is (decode() does not renormalise); the copy is refreshed by the next forward anyway
    settle_weights(c, s);
    const int64_t n8 = c->nW / 8;
    int cg = (int)((n8 + 255) / 256);
    if (cg > 2048) cg = 2048;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(cg), dim3(256), 0, s, c->P, c->Wb, n8);
    g.B0 = c->Wb; g.ldb = n_p;
    e.bias = nullptr;
    rc = launch_gemm<OP_ROW, OP_ROW>(g, e, s);
  }
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return SAE_OK;
}

extern "C" int sae_latent_colmax(sae_ctx* c, float* out_host, int64_t capacity, void* stream) {
  if (!c || !out_host) return fail(SAE_ERR_INVALID, "null argument");
  if (capacity < c->n) return fail(SAE_ERR_INVALID, "capacity too small");
  if (c->last_M <= 0) return fail(SAE_ERR_STATE, "no forward has run yet");
  USE_DEVICE(c);
  if (c->last_fwd_e32) {
    HIP_TRY(hipMemcpyAsync(out_host, c->e32_colmax, (size_t)c->n * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return SAE_OK;
  }
  {
    int rc_d = ensure_dense(c);
    if (rc_d) return rc_d;
  }
  hipStream_t s = (hipStream_t)stream;
  int* bits = reinterpret_cast<int*>(c->db_part);   // scratch, free outside of a backward pass
  HIP_TRY(hipMemsetAsync(bits, 0, (size_t)c->n_p * 4, s));
  const int rows_per_block = 256;
  dim3 grid(c->n_p / 128 / 2 > 0 ? c->n_p / 256 : 1, (unsigned)((c->last_M + rows_per_block - 1) / rows_per_block));
  if (c->n_p % 256 != 0) grid.x = (c->n_p + 255) / 256;
  hipLaunchKernelGGL(latent_colmax_kernel, grid, dim3(256), 0, s, c->topk ? c->dense : c->c, bits, c->last_M, c->n_p,
                     rows_per_block);
  HIP_TRY(hipGetLastError());
  std::vector<float> tmp(c->n_p);
  HIP_TRY(hipMemcpyAsync(tmp.data(), bits, (size_t)c->n_p * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(out_host, tmp.data(), (size_t)c->n * 4);
  return SAE_OK;
}

// Validation without host round trips (train_sae.py:168-190 runs one file per forward and reads .item() four times per
// file): forward of one file, then its loss scalars and its per-feature maxima are left in CALLER-OWNED device rows; the
// host reads all rows once at the end.  Everything is asynchronous on `stream`.
extern "C" int sae_eval_into(sae_ctx* c, const void* x, int64_t M, int x_dtype, float* metrics_out, float* colmax_out, void* stream) {
  if (!c || !x || !metrics_out) return fail(SAE_ERR_INVALID, "null argument");
  int rc = c->eval_prec == 1 ? eval_fp32_dispatch(c, x, M, x_dtype, stream) : dispatch_fwd_bwd(c, x, M, x_dtype, stream, false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(metrics_out, c->G + c->nparams, SAE_NUM_METRICS * 4, hipMemcpyDeviceToDevice, s));
  if (colmax_out && c->last_fwd_e32) {      // the fp32 forward left the per-feature maxima (bit patterns of non-negative floats)
    HIP_TRY(hipMemcpyAsync(colmax_out, c->e32_colmax, (size_t)c->n * 4, hipMemcpyDeviceToDevice, s));
    return SAE_OK;
  }
  if (colmax_out) {
    if (c->topk && !c->dense_valid) return fail(SAE_ERR_STATE, "the evaluation forward did not leave the dense latent rows");
    HIP_TRY(hipMemsetAsync(colmax_out, 0, (size_t)c->n * 4, s));
    const int rows_per_block = 256;
    const int64_t Mr = c->last_M;
    // (columns [n_dict, n_p) are padding and are not written: the kernel covers whole 256-column blocks below n_dict only
    // when n_dict is a multiple of 256, so the tail block is guarded by the column bound)
    dim3 grid((c->n + 255) / 256, (unsigned)((Mr + rows_per_block - 1) / rows_per_block));
    hipLaunchKernelGGL(latent_colmax_bounded_kernel, grid, dim3(256), 0, s, c->topk ? c->dense : c->c, reinterpret_cast<int*>(colmax_out),
                       Mr, c->n_p, c->n, rows_per_block);
    HIP_TRY(hipGetLastError());
  }
  return SAE_OK;
}

extern "C" int sae_profile(sae_ctx* c, int level) {
  if (!c) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  c->profile = level;
  c->prof_tick = 0;
  for (auto& r : c->ev) r.n = 0;
  return SAE_OK;
}

extern "C" int sae_profile_period(sae_ctx* c, int period) {
  if (!c || period < 1) return fail(SAE_ERR_INVALID, "bad argument");
  c->prof_period = period;
  return SAE_OK;
}

extern "C" int sae_kernel_times(sae_ctx* c, float* ms_sum, int32_t* launches, int n) {
  if (!c || !ms_sum || !launches) return fail(SAE_ERR_INVALID, "null argument");
  USE_DEVICE(c);
  HIP_TRY(hipDeviceSynchronize());
  for (int k = 0; k < n; ++k) {
    ms_sum[k] = 0.f;
    launches[k] = 0;
    if (k >= KID_COUNT) continue;
    EvRing& r = c->ev[k];
    const int cnt = r.n < EV_RING ? r.n : EV_RING;
    for (int i = 0; i < cnt; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r.beg[i], r.end[i]) == hipSuccess) {
        ms_sum[k] += ms;
        launches[k]++;
      }
    }
    r.n = 0;
  }
  return SAE_OK;
}

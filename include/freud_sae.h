/*
 * freud_sae.h -- C ABI of libfreud_sae.so, the MI355X (gfx950) SAE training engine.
 *
 * This is the drop-in boundary for the one hot path of ksadov/FREUD that this repository
 * replaces: the body of the training loop in src/scripts/train_sae.py:421-453
 * (forward under autocast -> backward -> clip_grad_norm_ -> optimizer.step) together with
 * the model arithmetic it calls (src/models/l1autoencoder.py:69-95, mse_loss :29-36;
 * src/models/topkautoencoder.py:72-151).  The reference has no FFI of its own (it is pure
 * PyTorch); the entry points below are what a ctypes/cffi binding added to the reference's
 * train() would call instead of `dist_model(activations)`, `loss.backward()`,
 * `clip_grad_norm_` and `optimizer.step()` -- see INTEGRATION.md for that stub.
 *
 * Conventions
 *   - plain C types only; every pointer marked "dev" is a device (HBM) pointer owned by the
 *     caller, every pointer marked "host" is ordinary host memory;
 *   - all calls return 0 on success, a negative code on failure; sae_last_error() then
 *     returns a description (thread-local).  No exceptions cross the boundary;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued on it; only the calls documented as synchronising wait for the device;
 *   - one host thread drives one context; a context is bound to one GPU;
 *   - parameter layouts are the reference's own (row-major fp32), so checkpoints written from
 *     sae_get_params are bit-layout compatible with the reference's state_dict:
 *       L1   : decoder.weight W[d][n], encoder_bias b[n]            (l1autoencoder.py:56-60)
 *       TopK : encoder.weight We[n][d], encoder.bias be[n], W_dec Wd[n][d], b_dec bd[d]
 *                                                                  (topkautoencoder.py:62-70)
 */
#ifndef FREUD_SAE_H
#define FREUD_SAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sae_ctx sae_ctx;

enum { SAE_VARIANT_L1 = 0, SAE_VARIANT_TOPK = 1 };          /* train_sae.py:352-361 autoencoder_variant */
enum { SAE_OPT_RADAM = 0, SAE_OPT_ADAM = 1 };               /* train_sae.py:374-381 optimizer           */
enum { SAE_DTYPE_F32 = 0, SAE_DTYPE_F16 = 1, SAE_DTYPE_BF16 = 2 }; /* dtype of the activation rows x     */

enum {
  SAE_OK = 0,
  SAE_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
  SAE_ERR_HIP = -2,       /* a HIP runtime call failed                */
  SAE_ERR_STATE = -3      /* call sequence error                      */
};

/* Mirrors the knobs of train() that reach the step (train_sae.py:297-320, 356-381) and the
 * autoencoder_config dataclasses (src/models/config.py:5-28). */
typedef struct sae_config {
  int32_t variant;        /* SAE_VARIANT_*                                               */
  int32_t d_model;        /* activation_size (feat_dim, train_sae.py:65)                 */
  int32_t n_dict;         /* get_n_dict_components(...) (src/utils/models.py:1-6)         */
  int32_t k;              /* TopK only: cfg.k                                            */
  int32_t optimizer;      /* SAE_OPT_*                                                   */
  int32_t device_id;      /* HIP device ordinal                                          */
  int64_t max_rows;       /* largest M = batch_size*T this context will be asked to step */
  /* hyper-parameters are doubles because the reference holds them as Python floats */
  double recon_alpha;     /* L1 only: cfg.recon_alpha                                    */
  double auxk_alpha;      /* TopK only: cfg.auxk_alpha                                   */
  double clip_thresh;     /* clip_grad_norm_ max_norm (train_sae.py:449)                 */
  double weight_decay;    /* RAdam only (train_sae.py:375-377)                           */
  double beta1, beta2;    /* 0.9 / 0.999 (torch defaults the reference relies on)        */
  double eps;             /* 1e-5 RAdam (train_sae.py:376), 1e-8 Adam (torch default)    */
  /* development / test switches (zero in production): */
  int32_t force_generic;  /* 1: generic GEMM path even where a fused d=384 kernel exists (tests cover both)          */
  int32_t debug_flags;    /* kernel timing experiments of bench.py --dbg (65, 66, 70: results become wrong) and A/B    */
                          /* paths that stay correct: 75 = no tile-driven TopK select, 76 = AuxK through the gather   */
                          /* kernels instead of the compacted dead-set GEMMs (tests compare both), 78 = the loss         */
                          /* finalisation as its own kernel between the fused forward and backward (round 2's order),   */
                          /* 79 = the flat optimizer kernel + a separate column-norm pass for L1 (round 2's order),      */
                          /* 80 = L1 with d <= 384: the tiled update + normalize_cast in the next forward instead of the */
                          /* update that also writes the next forward's weight copies (tests compare both to the bit),  */
                          /* 81 = uniform split-K through slabs for the plain weight-gradient GEMM of the generic L1     */
                          /* path (instead of whole tiles straight into the gradient + the tail tiles in K pieces)       */
  int32_t force_gemm128;  /* 1: 128x128 GEMM tiles even where the 256x256 kernel applies (A/B timing, tests)        */
  int32_t topk_dense_backward; /* TopK backward A/B (tests): 0 CSC sparse backward, 1 dense GEMMs + mask, 2 sparse d      */
                          /* pre-activations + dense weight-gradient GEMMs                                            */
  /* model / numerics options: */
  int32_t multi_topk;     /* TopK only: cfg.multi_topk (topkautoencoder.py:134-140, loss term train_sae.py:442)     */
  int32_t precision;      /* SAE_PREC_*: operand type of the L1 encoder / decoder GEMMs                             */
  int32_t reserved[2];    /* must be zero                                                                           */
} sae_config;

/* SAE_PREC_BF16: every GEMM on bf16 operands (the reference's CPU autocast precision; the parity configuration).
 * SAE_PREC_FP8 : BASELINE configs[4] -- encoder and decoder GEMMs on OCP e4m3 operands (x, W and the latent quantised
 *                with per-tensor power-of-two scales), fp32 MFMA accumulation, outputs rounded to bf16 like the bf16 path;
 *                the backward GEMMs and the fp32 master weights / optimizer are unchanged. */
/* SAE_PREC_FP8_BWD: SAE_PREC_FP8 plus the dpre GEMM of the backward (dc = dx_hat W: a third of the backward's FLOPs) on e4m3
 *                operands -- dx_hat quantised per tensor with a power-of-two scale from its own maximum, W as above; the
 *                weight-gradient GEMMs stay bf16.  Beyond BASELINE configs[4] ("fp8 enc/dec"); an option with a stated cost:
 *                raw gradients within 5e-2 (rel-Frobenius) of the bf16 arithmetic (tests/test_fp8_gpu.py). */
enum { SAE_PREC_BF16 = 0, SAE_PREC_FP8 = 1, SAE_PREC_FP8_BWD = 2,
       SAE_PREC_FP32 = 3 /* evaluation only: sae_set_eval_precision */ };

/* Metrics of the most recent sae_forward_backward / sae_eval (all fp32). */
enum {
  SAE_M_LOSS_RECON = 0,   /* L1: out.reconstruction_loss  (l1autoencoder.py:86)  | TopK: fvu            */
  SAE_M_LOSS_L1 = 1,      /* L1: out.l1_loss              (l1autoencoder.py:85)  | TopK: auxk_loss      */
  SAE_M_MSE = 2,          /* the return_mse value         (l1autoencoder.py:93-94, topk :149-150)       */
  SAE_M_GRAD_NORM = 3,    /* total_norm returned by clip_grad_norm_ (valid after sae_optimizer_step)    */
  SAE_M_COUNT = 4,        /* number of unmasked elements (x != -1) entering the masked MSE             */
  SAE_M_DEAD_PCT = 5,     /* TopK: fraction of dead latents (train/dead_pct, train_sae.py:481-485)                      */
  SAE_M_MULTI_TOPK_FVU = 6, /* TopK with cfg.multi_topk: out.multi_topk_fvu (topkautoencoder.py:134-140), else 0      */
  SAE_M_RESERVED7 = 7,
  SAE_NUM_METRICS = 8
};

const char* sae_last_error(void);
int sae_version(void);

/* Create / destroy.  Allocates every HBM buffer the context will ever need (no allocation
 * happens inside the step calls, so they can be captured into a hipGraph). */
int sae_create(const sae_config* cfg, sae_ctx** out);
void sae_destroy(sae_ctx* ctx);

/* Parameters, reference layouts (see top).  `is_device` selects host or device pointers.
 * For L1 pass (W, b, NULL, NULL); for TopK pass (We, be, Wd, bd).  Synchronising.
 * Replaces model.load_state_dict / model.state_dict (train_sae.py:232-251, 265-294).
 * sae_get_params returns what the reference's state_dict holds at the same point of the same call sequence: for L1 the
 * decoder weight is UN-normalised right after an update (train_sae.py:450) and normalised once a forward has run
 * (l1autoencoder.py:71-73 normalises in place on every forward) -- the engine may postpone that in-place division
 * internally, never its visible effect. */
int sae_set_params(sae_ctx* ctx, const float* p0, const float* p1, const float* p2, const float* p3, int is_device);
int sae_get_params(sae_ctx* ctx, float* p0, float* p1, float* p2, float* p3, int is_device);

/* Optimizer moments (exp_avg / exp_avg_sq per parameter, same order and layouts as the
 * parameters) and the step counter.  Replaces optimizer.state_dict()/load_state_dict().
 * Pointers may be NULL to skip a tensor.  Synchronising. */
int sae_set_opt_state(sae_ctx* ctx, int64_t step, const float* const exp_avg[4], const float* const exp_avg_sq[4], int is_device);
int sae_get_opt_state(sae_ctx* ctx, int64_t* step, float* const exp_avg[4], float* const exp_avg_sq[4], int is_device);

/* Forward + backward of one batch of M activation rows x[M][d_model] (dev pointer, row-major,
 * dtype SAE_DTYPE_*).  For L1 this is: renormalise decoder columns in place, encoder GEMM +
 * bias + ReLU, decoder GEMM, masked MSE + L1, and the full backward into the gradient buffer
 * (train_sae.py:429-448).  Asynchronous on `stream`. */
int sae_forward_backward(sae_ctx* ctx, const void* x_dev, int64_t M, int x_dtype, void* stream);

/* The flat fp32 gradient buffer the backward fills and the optimizer consumes:
 * [ grads of every parameter, padded | SAE_NUM_METRICS loss scalars ].  A data-parallel host
 * all-reduces (sum) exactly this buffer between sae_forward_backward and sae_optimizer_step and
 * passes grad_scale = 1/world_size.  Pointer is stable for the life of the context. */
int sae_grad_buffer(sae_ctx* ctx, void** dev_ptr, int64_t* n_floats);

/* Data-parallel overlap (the reference has no data parallelism; this is what a DDP-style wrapper around
 * "loss.backward()", train_sae.py:448, needs).  When a callback is set, sae_forward_backward calls it on the calling
 * thread each time a contiguous range [offset, offset + count) of the gradient buffer has become final in `stream`
 * order (the kernels that produce it have been enqueued), so that the host can start that range's all-reduce on a
 * communication stream while the remaining backward kernels run.  The ranges of one call are disjoint and cover the
 * whole buffer.  The weight-gradient GEMM of the generic L1 path is issued in row chunks for this (one chunk, i.e. no
 * change, without a callback).  fn == NULL removes the callback. */
typedef void (*sae_grad_ready_fn)(void* user, int64_t offset, int64_t count, void* stream);
int sae_set_grad_ready_callback(sae_ctx* ctx, sae_grad_ready_fn fn, void* user);

/* ---- data-parallel exactness and the engine's own communicator (the reference is single-process: train_sae.py never
 * leaves one device; this is the partitioning BASELINE.json's north_star asks for).
 *
 * R ranks x batch B must train like one rank x batch R B.  The gradients are linear in the per-row contributions but the
 * losses NORMALISE by whole-batch quantities -- the number of unmasked entries of the masked MSE and the row count of the
 * L1 mean (l1autoencoder.py:29-36,85); for TopK the total variance around x.mean(0) over ALL files (topkautoencoder.py:
 * 104-106) and the rows of mse -- so those statistics are summed over the ranks FIRST (they depend on the batch only, not
 * on the model) and every rank's backward normalises by the global values; the summed gradients then ARE the whole
 * batch's and no 1/R rescaling follows (grad_scale = 1).
 *
 *   sae_batch_stats   writes this rank's statistics of batch x into the context's statistics buffer (float64, device;
 *                     L1: [unmasked entries, rows]; TopK: [rows, files, sum_b x, sum_b x^2 per (t, feature)]); asynchronous.
 *   sae_stats_buffer  the buffer and the number of doubles the last sae_batch_stats wrote: a host-driven data-parallel
 *                     loop all-reduces (sum) exactly that range before sae_forward_backward.
 *   sae_set_dp_world  world > 0: sae_forward_backward normalises by the statistics buffer (and loss scalars become this
 *                     rank's SHARE of the global losses: they sum over the ranks; dead_pct is pre-divided by world);
 *                     0 (default): by the rank's own batch.
 *
 * sae_dist_unique_id / sae_dist_init give the context its own RCCL communicator (one process per GPU; the unique id made on
 * rank 0 travels to the others by any host channel).  Afterwards sae_forward_backward / sae_step do the whole protocol
 * inside the engine, with no host code in the step: statistics + their all-reduce on a communication stream while the
 * forward runs, every gradient range all-reduced on that stream as soon as its backward kernels are enqueued (i.e. under
 * the remaining backward GEMMs), the compute stream joining before the optimizer.  sae_optimizer_step takes grad_scale 1. */
int sae_batch_stats(sae_ctx* ctx, const void* x_dev, int64_t M, int x_dtype, void* stream);
int sae_stats_buffer(sae_ctx* ctx, void** dev_ptr, int64_t* n_doubles);
int sae_set_dp_world(sae_ctx* ctx, int world);
int sae_dist_unique_id(void* out_host, int64_t capacity_bytes);       /* 128 bytes (ncclUniqueId) */
int sae_dist_init(sae_ctx* ctx, const void* unique_id_host, int64_t id_bytes, int rank, int world);
int sae_dist_world(sae_ctx* ctx);                                      /* 0 without a communicator */
/* Gradient payload of the in-engine all-reduce: SAE_DTYPE_F32 (default: R ranks == one rank exactly, up to summation
 * order) or SAE_DTYPE_BF16 -- the fused d = 384 path then sums a bf16 copy of the parameter gradients (half the bytes of
 * a latency- and link-bound 4.7 MB exchange; the loss scalars stay fp32).  The reference's own CPU autocast rounds these
 * gradients to bf16 as well (the weight-gradient GEMMs' outputs), so this stays inside the parity tolerance; paths
 * other than the fused one keep fp32. */
int sae_dist_set_payload(sae_ctx* ctx, int dtype);

/* ---- peer exchange: the same in-engine protocol with the all-reduces done by the engine's own kernels over hipIpc peer
 * mappings instead of RCCL (on the full xGMI mesh a direct reduce-scatter + all-gather pulls 1/R of
 * the payload over EVERY link at once and needs three flag round trips, where a ring moves the payload over one link per hop
 * -- the 4.7 MB gradient of the d = 384 model is latency-bound).  Up to 8 ranks of one node, one process per GPU.  Like the
 * RCCL form it stands where a DDP wrapper would stand around loss.backward() (train_sae.py:448); results are bit-identical on
 * every rank (each range is summed once, in rank order, by its owner).
 *   sae_p2p_blob_bytes  size of the blob one rank publishes;
 *   sae_p2p_export      allocates the exchange state and writes this rank's blob (hipIpc handles of the gradient buffer, its
 *                       bf16 copy, the statistics buffer and the flag block) to host memory;
 *   sae_p2p_init        takes the blobs of ALL ranks (rank order, gathered over any host channel), maps the peers and runs a
 *                       self-test (collective: every rank must call it): four exchanges of every payload form -- fp32, bf16
 *                       payload, a strided 2-D block, fp64 statistics, the statistics push -- over the SAME addresses with a
 *                       different rank-dependent pattern each time, checked on the device, so that a stale cached peer line, a
 *                       flag overtaking its data or a wrong mapping shows as wrong sums BEFORE the first step; a peer that
 *                       cannot be reached makes it FAIL after min(FREUD_P2P_TIMEOUT_MS, 40 s) (default 120000: a liveness bound of the run's steps)
 *                       instead of hanging.  Afterwards sae_forward_backward / sae_step run the data-parallel protocol
 *                       through the peer exchange; sae_dist_world() == world.  FREUD_P2P_FINEGRAINED=1 (read by sae_create)
 *                       puts the three peer-read buffers in fine-grained memory (a peer never caches their lines non-coherently, so
 *                       correctness does not rest on cache maintenance; measured free for the LOCAL kernels on one GPU at C2 and C4,
 *                       profiles/r04_ab_finegrained_c{2,4}.txt; its cross-device cost is unmeasured).  Default of train() / bench.py
 *                       for WORLD_SIZE > 1.
 *   sae_dist_set_overlap  fused d = 384 path: launch the backward in `nranges` column-tile ranges; each range's gradient is
 *                       exchanged on the communication stream under the next range's backward (needs the peer exchange:
 *                       RCCL sums contiguous buffers only).  1 (default) = one launch, exchanged in line.
 *   sae_dist_check      synchronises and reports a failed exchange (a peer that never arrived: the exchange kernels give up
 *                       after the timeout, POISON the flags they owe their peers so that no rank sails on, and the failure is
 *                       sticky -- the replicas are out of step and the run must stop);
 *   sae_dist_poll       the same report without synchronising (host-mapped mirror of the failure word): free after every step;
 *   sae_dist_audit      snapshot_dev != NULL: from now on every gradient exchange first copies the segments it is about to sum
 *                       (this rank's own contribution) into snapshot_dev (device, same layout and size as sae_grad_buffer).
 *                       The host sums the snapshots over the ranks with an INDEPENDENT carrier (torch.distributed) and
 *                       compares with the exchanged gradient: a wrong-but-identical sum, which no replica comparison can
 *                       see, shows here (freud_amd/dp.py: audit_step).  NULL switches it off.
 *   sae_param_checksum  out_host[4] = order-independent 64-bit checksums of {parameters, first moments, second moments} and the
 *                       step count.  Replicas are bit-identical by construction, so ANY difference between ranks is an
 *                       exchange bug; train() compares them over the host channel at every logging step and before every
 *                       checkpoint (no reference counterpart: train_sae.py:448-450 is single-device).  Synchronises. */
int sae_p2p_blob_bytes(void);
int sae_p2p_export(sae_ctx* ctx, void* blob_out_host, int64_t capacity_bytes);
int sae_p2p_init(sae_ctx* ctx, const void* all_blobs_host, int64_t bytes_per_rank, int rank, int world);
int sae_p2p_leave(sae_ctx* ctx);     /* undo sae_p2p_init (a PEER's self-test failed: every rank falls back together); no-op otherwise */
int sae_dist_set_overlap(sae_ctx* ctx, int nranges);
int sae_dist_check(sae_ctx* ctx);
int sae_dist_poll(sae_ctx* ctx);
int sae_dist_audit(sae_ctx* ctx, float* snapshot_dev);
int sae_grad_layout(sae_ctx* ctx, int64_t out_host[3]);   /* floats of sae_grad_buffer: {parameter gradients, loss scalars, did_fire flags} */
int sae_param_checksum(sae_ctx* ctx, uint64_t out_host[4]);

/* clip_grad_norm_ + Adam/RAdam update with learning rate `lr` (train_sae.py:449-450).
 * grad_scale multiplies every gradient (and the loss scalars) first.  Asynchronous. */
int sae_optimizer_step(sae_ctx* ctx, double lr, double grad_scale, void* stream);

/* TopK only.  dead_feature_threshold: a latent is dead when num_frames_since_fired > threshold
 * (autoencoder_config["dead_feature_threshold"], train_sae.py:436-439).  rows_per_file: T of the
 * [B][T][d] batch -- the FVU denominator is sum (x - x.mean(0))^2 with the mean over the B files
 * (topkautoencoder.py:104); 0 = treat the batch as one file. */
int sae_set_topk_options(sae_ctx* ctx, double dead_feature_threshold, int64_t rows_per_file);

/* TopK bookkeeping state num_frames_since_fired[n_dict] (int64, train_sae.py:412-415,443-446).  The reference keeps
 * it only in the live process and restarts it from zero on resume (train_sae.py:396-415 never saves it); these two
 * calls let the host persist it next to the checkpoint.  Host buffers of n_dict int64.
 * Synchronous.  SAE_ERR_INVALID on an L1 context. */
int sae_get_topk_state(sae_ctx* ctx, int64_t* num_frames_since_fired_host, int64_t n);
int sae_set_topk_state(sae_ctx* ctx, const int64_t* num_frames_since_fired_host, int64_t n);

/* Convenience: sae_forward_backward + sae_optimizer_step(lr, 1). */
int sae_step(sae_ctx* ctx, const void* x_dev, int64_t M, int x_dtype, double lr, void* stream);

/* Forward only (no parameter update except the in-place column renormalisation the reference's
 * encode() also performs in eval).  Fills the metrics.  Asynchronous. */
int sae_eval(sae_ctx* ctx, const void* x_dev, int64_t M, int x_dtype, void* stream);

/* ---- inference (the SAE inside FlyActivationDataLoader.__iter__, dataset/activations.py:
 * 96-108, and manipulate_latent, utils/activations.py:243-268).  sae_eval is encode(): afterwards
 *   sae_latent_buffer  -> the latent of the last forward, bf16 [M][*row_stride] on the device (columns [0, n_dict)):
 *                         L1: c = relu(x W + b) (l1autoencoder.py:69-75); TopK: the top-k activations scattered into a
 *                         dense row, zeros elsewhere (topkautoencoder.py:79-85 + eager_decode's buffer, :15-18);
 *   sae_topk_indices   -> TopK only: int32 [M][k] top_indices of the last forward (tie order: lowest column first).
 * Pointers are owned by the context and overwritten by the next forward.
 * sae_decode: x_hat[M][d_model] (fp32, device, dense) = latent . W^T (L1 decode(), l1autoencoder.py:77-78, with the
 * CURRENT weights, no renormalisation) or latent_dense . W_dec + b_dec (TopK decode(), topkautoencoder.py:87-91).
 * latent: device, row-major, `ld` elements per row, SAE_DTYPE_F32 or SAE_DTYPE_BF16; bf16 MFMA arithmetic like the
 * train step.  M <= max_rows.  Asynchronous on `stream`. */
int sae_latent_buffer(sae_ctx* ctx, void** dev_ptr, int64_t* row_stride);
int sae_topk_indices(sae_ctx* ctx, void** dev_ptr, int* k);
int sae_decode(sae_ctx* ctx, const void* latent_dev, int latent_dtype, int64_t ld, int64_t M, float* x_hat_dev, void* stream);
/* TopK with cfg.multi_topk only: the second selection of the last forward -- the top 4k activations scattered into a
 * dense bf16 row [M][*row_stride] and their int32 indices [M][*k4].  TopKAutoEncoder.forward() returns THESE as
 * out.encoded / out.sae_out when multi_topk is set (topkautoencoder.py:134-147 re-binds the names); encode() keeps k. */
int sae_multi_topk_buffers(sae_ctx* ctx, void** dense_dev, int64_t* row_stride, void** idx_dev, int* k4);

/* validate() without per-file host round trips (train_sae.py:168-190 reads four .item()s per file): sae_eval of one file,
 * then its SAE_NUM_METRICS loss scalars go to metrics_out_dev[8] and -- unless colmax_out_dev is NULL -- its per-feature
 * maxima of |latent| (train_sae.py:176-178) to colmax_out_dev[n_dict]; both are CALLER-OWNED device rows (one pair per
 * file), so a whole validation folder is enqueued without a single synchronisation and read back once.  Asynchronous. */
int sae_eval_into(sae_ctx* ctx, const void* x_dev, int64_t M, int x_dtype, float* metrics_out_dev, float* colmax_out_dev, void* stream);

/* Arithmetic of sae_eval / sae_eval_into.  SAE_PREC_BF16 (default): the training kernels' -- bf16 operands, fp32 accumulation, i.e.
 * CPU autocast's.  SAE_PREC_FP32: fp32 end to end, which is what the reference's validate() computes on device='cpu'
 * (train_sae.py:162-166: nullcontext() instead of autocast; L1AutoEncoder.forward l1autoencoder.py:69-95, TopKAutoEncoder.forward
 * topkautoencoder.py:93-151 without a dead mask) and what its bestval.pth selection (train_sae.py:585-595) rests on: fp32 matrix
 * instructions, fp32 bias / ReLU / top-k (ties: lowest column first), loss sums in double.  The in-place column normalisation of
 * the L1 weights happens first, exactly as in every other forward.  Training steps are not affected.  Any SAE_PREC_* other than
 * these two: SAE_ERR_INVALID. */
int sae_set_eval_precision(sae_ctx* ctx, int precision);

/* Copy the SAE_NUM_METRICS scalars to host.  Synchronises `stream`. */
int sae_read_metrics(sae_ctx* ctx, float out_host[SAE_NUM_METRICS], void* stream);

/* Per-dictionary-feature maximum of |latent| over the M rows of the last forward
 * (torch.max(torch.abs(latent), dim=0) in validate(), train_sae.py:176-178) -> out_host[n_dict].
 * Synchronises `stream`. */
int sae_latent_colmax(sae_ctx* ctx, float* out_host, int64_t capacity_floats, void* stream);

/* Test / inspection hook: copy an internal tensor of the last step to host as fp32, un-padded.
 * which: 0 = latent c [M][n]; 1 = x_hat-derived dx_hat [M][d]; 2 = raw gradients in reference
 * layouts, concatenated in parameter order.  Synchronising.  Not part of the hot path. */
int sae_debug_read(sae_ctx* ctx, int which, float* out_host, int64_t capacity_floats);

/* Timing hooks for bench.py / profiling.  level 0 = off, 1 = bracket only the dominant kernel of
 * every step with HIP events on the launch stream, 2 = bracket every kernel (diagnostic).
 * sae_kernel_times synchronises, adds up the events recorded since the last call (at most the last
 * 64 launches per kernel are kept) and returns per-kernel total milliseconds and launch counts for
 * kernel ids 0..n-1 (ids: sae_kernel_name).  Level 1 samples every 8th step: an event record costs ~6 us of idle GPU
 * between two dependent kernels, which at three records per 0.6 ms step was 3 % of the thing being measured. */
int sae_profile(sae_ctx* ctx, int level);
int sae_profile_period(sae_ctx* ctx, int period);   /* level 1 samples every `period`-th step (default 8; short runs use less) */
int sae_kernel_times(sae_ctx* ctx, float* ms_sum, int32_t* launches, int n);
const char* sae_kernel_name(int id);        /* NULL past the last id */
int sae_dominant_kernel(sae_ctx* ctx);      /* id bracketed at level 1 */

#ifdef __cplusplus
}
#endif
#endif /* FREUD_SAE_H */

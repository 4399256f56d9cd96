"""CPU oracle for the SAE train step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a CPU restatement (torch-CPU tensors, explicit hand-written backward, no
autograd) of the arithmetic that ksadov/FREUD's ``src.scripts.train_sae`` hot loop runs
for one optimizer step.  It exists only so that the HIP engine can be checked against
something that follows the reference line by line.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it; the
product package ``freud_amd`` never does (tests/test_boundary.py greps for that).

Pinning: ``tests/golden/make_golden.py`` imports the real reference from /root/reference
(in the build container only), runs it on seeded inputs and stores inputs + outputs as
``tests/golden/*.npz``; ``tests/test_oracle.py`` checks this file against those vectors on
CPU.  With ``autocast=True`` the oracle reproduces the reference's ``autocast('cpu')``
(= bf16) rounding points and agrees with the stored reference outputs to fp32 round-off
(the reference publishes no golden vectors of its own: SURVEY.md section 8c).

Reference citations are ``file:line`` relative to the reference repository root.

Shapes: x is flat [M, d] (the reference's [B, T, d] with M = B*T; the only places the
3-D shape matters are ``norm(c, 1, dim=2).mean()`` = mean over the M rows, and the TopK
``x.mean(0)`` which is over B -- handled explicitly below).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch

BF16 = torch.bfloat16


def _r(t: torch.Tensor) -> torch.Tensor:
    """Round an fp32 tensor to bf16 and back (one autocast rounding point)."""
    return t.to(BF16).to(torch.float32)


# How a bf16 x bf16 matmul is evaluated.  "native": torch's own CPU bf16 kernel (oneDNN) -- what the reference runs, and
# what the golden fixtures pin bit for bit IN THE BUILD CONTAINER.  Its result is host dependent: on the GPU box's host
# (other ISA extensions, other oneDNN kernels) the same call on a [512 x 40960] operand moved a gradient norm by 0.6 %.
# "fp32": the definition both follow -- products of the bf16 operands accumulated in fp32, ONE rounding of the result to
# bf16 -- evaluated with an fp32 matmul: host independent up to fp32 summation order.  Parity tests at sizes beyond the
# fixtures use "fp32" (tests/conftest.py sets it for the GPU suite); tests/test_oracle.py checks the two agree to bf16
# round-off on the fixtures.
MATMUL_MODE = "native"
# The same for the gradient norm of clip_grad_norm_: "native" = torch's fp32 vector_norm (bit-exact against the fixtures
# here; on the GPU box's host the fp32 reduction over a 52 M element gradient read 0.6 % low), "float64" = the norm taken
# in double and rounded to fp32 once.
NORM_MODE = "native"


def _mm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """bf16 [.., k] @ bf16 [k, n] -> bf16."""
    if MATMUL_MODE == "native":
        return a @ b
    return (a.to(torch.float32) @ b.to(torch.float32)).to(BF16)


def _linear(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """bf16 addmm of nn.Linear under autocast: bf16(a @ w^T + bias), one rounding."""
    if MATMUL_MODE == "native":
        return torch.nn.functional.linear(a, w, bias)
    return (a.to(torch.float32) @ w.to(torch.float32).t() + bias.to(torch.float32)).to(BF16)


def get_n_dict_components(activation_size: int, expansion_factor: int, n_dict_components: int) -> int:
    """src/utils/models.py:1-6."""
    if n_dict_components == 0:
        return activation_size * expansion_factor
    return n_dict_components


# --------------------------------------------------------------------------------------
# L1 (tied-weight) SAE
# --------------------------------------------------------------------------------------
def normalize_columns(W: torch.Tensor) -> torch.Tensor:
    """src/models/l1autoencoder.py:71-73: W <- F.normalize(W, dim=0): every dictionary
    column divided by max(||col||_2, 1e-12).  Runs before *every* forward (train and eval)."""
    return torch.nn.functional.normalize(W, dim=0)


E4M3_MAX = 448.0
FP8_W_SCALE = 256.0


def _q8(t: torch.Tensor) -> torch.Tensor:
    """Round to OCP e4m3fn (round to nearest even, subnormals kept) and back to fp32."""
    return t.to(torch.float8_e4m3fn).to(torch.float32)


def _pow2_scale(v: float) -> float:
    """Largest power of two p with p * v <= 448, i.e. 2^floor(log2(448 / v)) (1 for v <= 0)."""
    if not v > 0.0:
        return 1.0
    q = torch.tensor(E4M3_MAX, dtype=torch.float32) / torch.tensor(v, dtype=torch.float32)
    _, e = math.frexp(float(q))
    return math.ldexp(1.0, max(-100, min(100, e - 1)))


def fp8_scales(x: torch.Tensor, b: torch.Tensor) -> Tuple[float, float]:
    """Per-tensor power-of-two scales of the fp8 forward (BASELINE configs[4]; freud_amd/csrc/l1_fp8.h): s_x from max |x|
    of the bf16 activations; s_c from the bound (max_row ||x_row||_2 + max(0, max b)) * 1.15 >= every latent."""
    xb = x.to(BF16).to(torch.float32)
    amax = float(xb.abs().max())
    rmax = float((xb * xb).sum(dim=1).max())
    bound = (torch.sqrt(torch.tensor(rmax, dtype=torch.float32)) + max(float(b.max()), 0.0)) * torch.tensor(1.15, dtype=torch.float32)
    return _pow2_scale(amax), _pow2_scale(float(bound))


def l1_forward(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor, recon_alpha: float,
               autocast: bool = True, precision: str = "bf16", dp_count: Optional[float] = None,
               dp_rows: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """Forward of L1AutoEncoder on flat rows.  ``W`` must already be column-normalised.

    src/models/l1autoencoder.py:69-95 (+ mse_loss :29-36).  With ``autocast`` the dtype flow
    of ``torch.autocast('cpu')`` (src/scripts/train_sae.py:431) is reproduced:
      pre   = bf16( bf16(x) @ bf16(W) )            GEMM 1, fp32 accumulate, one rounding
      c     = relu( fp32(pre) + b )                fp32          (:74)
      x_hat = bf16( bf16(c) @ bf16(W)^T )          GEMM 2        (:84)
      l1    = mean_rows( sum_j |c| )               fp32          (:85)
      recon = alpha * mean_{x != -1}( (fp32(x_hat) - x)^2 )      (:86, :29-36)

    precision="fp8" (BASELINE configs[4], not a reference mode): the two GEMMs take OCP e4m3 operands
      x8 = e4m3(bf16(x) s_x), W8 = e4m3(W 2^8), c8 = e4m3(c s_c)   (power-of-two per-tensor scales, fp8_scales()),
      pre = bf16((x8 @ W8) / (s_x 2^8)),  x_hat = bf16((c8 @ W8^T) / (s_c 2^8)),  fp32 accumulation,
    everything else (bias add, ReLU, losses, and the backward on the bf16 latent) as in the bf16 path.

    dp_count / dp_rows (data parallel, not in the reference): this call sees one rank's rows of a larger batch whose
    unmasked-entry count and row number are given; losses are then this rank's SHARE of the whole batch's losses
    (they sum over the ranks) and the backward normalises by the global numbers, so that the ranks' gradients sum to
    the gradient of the whole batch (freud_amd/csrc/dp_kernels.h).
    """
    x = x.to(torch.float32)
    M = x.shape[0]
    if precision == "fp8":
        sx, sc = fp8_scales(x, b)
        x8 = _q8(x.to(BF16).to(torch.float32) * sx)
        W8 = _q8(W * FP8_W_SCALE)
        pre = _r((x8 @ W8) / (sx * FP8_W_SCALE))
        c = torch.relu(pre + b)
        c8 = _q8(c * sc)
        x_hat = _r((c8 @ W8.t()) / (sc * FP8_W_SCALE))
    elif autocast:
        xb, Wb = x.to(BF16), W.to(BF16)
        pre = _mm(xb, Wb).to(torch.float32)
        c = torch.relu(pre + b)
        cb = c.to(BF16)
        x_hat = _mm(cb, Wb.t()).to(torch.float32)
    else:
        c = torch.relu(x @ W + b)
        x_hat = c @ W.t()
    keep = x != -1.0                                     # mse_loss: mask = target == ignored_index
    count = keep.sum()
    diff = torch.where(keep, x_hat - x, torch.zeros_like(x))
    sq_sum = (diff.double() ** 2).sum()
    rows = M
    if dp_count is None:
        mse_masked = (diff[keep] ** 2).mean() if int(count) > 0 else torch.tensor(float("nan"))
        l1 = c.abs().sum(dim=1).mean()
        mse_plain = ((x_hat - x) ** 2).mean()            # return_mse path (:93-94), unmasked
    else:
        count, rows = torch.tensor(float(dp_count), dtype=torch.float64), float(dp_rows)
        mse_masked = (sq_sum / count).to(torch.float32)
        l1 = (c.abs().sum().double() / rows).to(torch.float32)
        mse_plain = (((x_hat - x).double() ** 2).sum() / (rows * x.shape[1])).to(torch.float32)
    recon = recon_alpha * mse_masked
    out = {"c": c, "x_hat": x_hat, "l1_loss": l1, "reconstruction_loss": recon, "rows": rows,
           "mse": mse_plain, "keep": keep, "count": count, "local_count": keep.sum(), "diff": diff, "sq_sum": sq_sum}
    if precision == "fp8":
        out.update({"x8": x8, "c8": c8, "W8": W8, "s_x": sx, "s_c": sc})
    return out


def l1_backward(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor, fwd: Dict[str, torch.Tensor],
                recon_alpha: float, autocast: bool = True, precision: str = "bf16") -> Tuple[torch.Tensor, torch.Tensor]:
    """Hand-written backward of loss = reconstruction_loss + l1_loss (train_sae.py:434,448).

      dx_hat = alpha * 2 * (x_hat - x) * [x != -1] / count
      dc     = dx_hat @ W + sign(c)/M ;  dpre = dc * [c > 0]
      dW     = dx_hat^T @ c  +  x^T @ dpre          (tied weights: one gradient)
      db     = sum_rows dpre
    Under autocast every GEMM takes bf16 operands and rounds its output to bf16 once; the two
    weight-gradient GEMMs meet at the single autocast-cached bf16 copy of W, so autograd sums
    them *in bf16* before the cast back to fp32 (observed bit-exact against the reference).
    """
    x = x.to(torch.float32)
    M = fwd.get("rows", x.shape[0])                      # (data parallel: the rows of the whole batch)
    c, diff, count = fwd["c"], fwd["diff"], fwd["count"]
    g = (recon_alpha / count.to(torch.float32))          # d loss / d each squared term
    dx_hat = (diff * 2.0) * g                            # zero where masked
    gate = (c > 0).to(torch.float32)
    if autocast:
        xb, Wb, cb = x.to(BF16), W.to(BF16), c.to(BF16)
        dxb = dx_hat.to(BF16)
        if precision == "fp8bwd":
            # not a reference mode (include/freud_sae.h, SAE_PREC_FP8_BWD): the dx_hat W product on e4m3 operands --
            # g8 = e4m3(bf16(dx_hat) s_g) with the power-of-two scale of its own maximum, W8 = e4m3(W 2^8), fp32 accumulation,
            # one rounding to bf16 where the bf16 GEMM rounds; the weight-gradient GEMMs below stay bf16
            sg = _pow2_scale(float(dxb.to(torch.float32).abs().max()))
            g8 = _q8(dxb.to(torch.float32) * sg)
            W8 = _q8(W * FP8_W_SCALE)
            dc = _r((g8 @ W8) / (sg * FP8_W_SCALE)) + torch.sign(c) / M
        else:
            dc = _mm(dxb, Wb).to(torch.float32) + torch.sign(c) / M
        dpre = dc * gate
        dW_dec = _mm(dxb.t(), cb)                        # bf16 [d, n]
        dW_enc = _mm(xb.t(), dpre.to(BF16))              # bf16 [d, n]
        dW = (dW_dec + dW_enc).to(torch.float32)         # summed in bf16 at the shared cast
    else:
        dc = dx_hat @ W + torch.sign(c) / M
        dpre = dc * gate
        dW = dx_hat.t() @ c + x.t() @ dpre
    db = dpre.sum(dim=0)
    return dW, db


# --------------------------------------------------------------------------------------
# TopK SAE (untied)
# --------------------------------------------------------------------------------------
def topk_forward(x3: torch.Tensor, W_enc: torch.Tensor, b_enc: torch.Tensor, W_dec: torch.Tensor,
                 b_dec: torch.Tensor, k: int, dead_mask: Optional[torch.Tensor] = None,
                 auxk_alpha: float = 0.0, autocast: bool = True, multi_topk: bool = False,
                 stable_ties: bool = False, dp_stats: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """src/models/topkautoencoder.py:72-151 on x3 = [B, T, d] (B matters for x.mean(0), :104).

      pre   = relu( (x - b_dec) @ W_enc^T + b_enc )                   (:72-77)
      top   = pre.topk(k)                                             (:79-81)
      x_hat = scatter(top) @ W_dec + b_dec                            (:15-18, :87-91)
      fvu   = sum (x_hat - x)^2 / sum (x - mean_B x)^2                (:104-106, :131-132)
      auxk  = scale * sum (decode(top-k_aux dead) - e)^2 / total_var  (:109-129) times alpha (:146)
      multi = sum (decode(top-4k) - x)^2 / total_var  if cfg.multi_topk   (:134-140); the returned sae_out / encoded
              are then the 4k ones (the names are re-bound at :135-136), which is what did_fire sees (train_sae.py:442)
    autocast: ``encoder`` is a Linear -> bf16 addmm (bias included, output bf16); relu keeps bf16;
    top_acts bf16; scatter buffer bf16; decode matmul bf16 -> + b_dec (fp32) -> fp32.

    stable_ties: ``torch.topk`` has no rule for equal values at the k-th place (bf16 pre-activations tie there in
    4-50 % of rows) and the reference takes whatever its partial sort leaves.  With stable_ties=True equal values are
    taken lowest column first (a stable descending sort) -- the HIP engine's rule -- so that a batch WITH boundary
    ties can be compared at the arithmetic tolerance; on rows without a boundary tie both selections are the same set.

    dp_stats (data parallel, not in the reference): float64 [2 + 2 T d] = (rows, files, column sums, column sums of squares)
    of the WHOLE batch (topk_batch_stats summed over the ranks): total_variance and the mse denominator become the global
    ones, so fvu / auxk / multi / mse are this rank's share of the whole batch's values and the gradients sum exactly.
    """
    B, T, d = x3.shape
    x = x3.reshape(B * T, d).to(torch.float32)
    sae_in = x - b_dec
    if autocast:
        pre = _linear(sae_in.to(BF16), W_enc.to(BF16), b_enc.to(BF16))
        pre = torch.relu(pre)                                            # bf16
    else:
        pre = torch.relu(sae_in @ W_enc.t() + b_enc)
    def select(lat, kk):
        if not stable_ties:
            return lat.topk(kk, dim=-1, sorted=False)
        idx = torch.sort(lat.to(torch.float32), dim=-1, descending=True, stable=True).indices[..., :kk]
        return torch.gather(lat, -1, idx), idx

    top_acts, top_idx = select(pre, k)

    def decode(acts, idx):
        buf = acts.new_zeros(acts.shape[:-1] + (W_dec.shape[0],))
        dense = buf.scatter_(dim=-1, index=idx, src=acts)
        if autocast:
            y = _mm(dense, W_dec.to(BF16)).to(torch.float32)
        else:
            y = dense @ W_dec
        return y + b_dec, dense

    x_hat, dense = decode(top_acts, top_idx)
    e = x_hat - x
    if dp_stats is None:
        xm = x3.to(torch.float32).mean(0)                                # [T, d]
        total_variance = ((x3.to(torch.float32) - xm) ** 2).sum()
        mse_den = float(x.numel())
    else:
        TD = T * d
        s1, s2 = dp_stats[2:2 + TD], dp_stats[2 + TD:2 + 2 * TD]
        total_variance = (s2 - s1 * s1 / dp_stats[1]).sum().to(torch.float32)
        mse_den = float(dp_stats[0]) * d
    if float(total_variance) == 0.0:
        total_variance = torch.tensor(1.0)
    out = {"pre": pre, "top_acts": top_acts, "top_indices": top_idx, "x_hat": x_hat, "e": e,
           "dense": dense, "total_variance": total_variance}
    num_dead = int(dead_mask.sum()) if dead_mask is not None else 0
    if num_dead > 0:
        k_aux = d // 2
        scale = min(num_dead / k_aux, 1.0)
        k_aux = min(k_aux, num_dead)
        neg_inf = torch.tensor(-float("inf"), dtype=pre.dtype)
        aux_lat = torch.where(dead_mask[None], pre, neg_inf)
        aux_acts, aux_idx = select(aux_lat, k_aux)
        e_hat, aux_dense = decode(aux_acts, aux_idx)
        auxk = scale * ((e_hat - e) ** 2).sum() / total_variance
        out.update({"aux_acts": aux_acts, "aux_indices": aux_idx, "e_hat": e_hat,
                    "aux_dense": aux_dense, "aux_scale": torch.tensor(scale)})
    else:
        auxk = torch.tensor(0.0)
    out["fvu"] = (e ** 2).sum() / total_variance
    out["auxk_loss"] = auxk * auxk_alpha
    if multi_topk:
        m_acts, m_idx = select(pre, 4 * k)
        x_hat_m, m_dense = decode(m_acts, m_idx)
        e_m = x_hat_m - x
        out["multi_topk_fvu"] = (e_m ** 2).sum() / total_variance
        out.update({"multi_acts": m_acts, "multi_indices": m_idx, "multi_dense": m_dense, "e_multi": e_m,
                    "fire_indices": m_idx, "sae_out": x_hat_m})
    else:
        out["multi_topk_fvu"] = torch.tensor(0.0)
        out.update({"fire_indices": top_idx, "sae_out": x_hat})
    out["mse"] = (e ** 2).sum() / mse_den
    return out


def topk_batch_stats(x3: torch.Tensor) -> torch.Tensor:
    """(rows, files, per-(t, feature) sums of x and of x^2 over the files) in float64: dp_kernels.h."""
    B, T, d = x3.shape
    xd = x3.to(torch.float32).to(torch.float64).reshape(B, T * d)
    return torch.cat([torch.tensor([float(B * T), float(B)], dtype=torch.float64), xd.sum(0), (xd * xd).sum(0)])


def l1_batch_stats(x: torch.Tensor) -> torch.Tensor:
    """(unmasked entries, rows) in float64: dp_kernels.h."""
    x = x.reshape(-1, x.shape[-1]).to(torch.float32)
    return torch.tensor([float((x != -1.0).sum()), float(x.shape[0])], dtype=torch.float64)


def topk_backward(x3: torch.Tensor, W_enc: torch.Tensor, b_enc: torch.Tensor, W_dec: torch.Tensor,
                  b_dec: torch.Tensor, fwd: Dict[str, torch.Tensor], auxk_alpha: float = 0.0,
                  autocast: bool = True) -> Dict[str, torch.Tensor]:
    """Backward of loss = fvu + auxk_loss + multi_topk_fvu/8 (train_sae.py:442), fp32 or
    bf16-autocast rounding.  Gradients w.r.t. W_enc [n,d], b_enc [n], W_dec [n,d], b_dec [d].

    e is *not* detached in the AuxK term (topkautoencoder.py:127), so
      de = 2 e / tv  -  2 a s (e_hat - e) / tv ,   de_hat = 2 a s (e_hat - e) / tv .
    Only selected latents (top-k, and the aux top-k) carry gradient into ``pre``; relu gates it.
    """
    B, T, d = x3.shape
    x = x3.reshape(B * T, d).to(torch.float32)
    tv = fwd["total_variance"]
    e = fwd["e"]
    de = 2.0 * e / tv
    have_aux = "e_hat" in fwd and auxk_alpha != 0.0
    if have_aux:
        coef = auxk_alpha * float(fwd["aux_scale"]) * 2.0 / tv
        de_hat = coef * (fwd["e_hat"] - e)
        de = de - de_hat
    # x_hat = dense @ W_dec + b_dec ;  e_hat likewise with aux_dense
    def dec_bwd(dy, dense):
        if autocast:
            dyb = dy.to(BF16)
            dW = _mm(dense.to(BF16).t(), dyb).to(torch.float32)          # [n, d]
            ddense = _mm(dyb, W_dec.to(BF16).t()).to(torch.float32)      # [M, n]
        else:
            dW = dense.t() @ dy
            ddense = dy @ W_dec.t()
        return dW, ddense, dy.sum(0)

    # Three decodes can feed W_dec, b_dec and `pre`: autograd runs them in reverse forward order (multi-TopK, AuxK, main)
    # and ACCUMULATES each gradient into the buffer of the shared tensor: in bf16 for the bf16 `pre` (one rounding per
    # addition, in that order); in fp32 for W_dec -- decode() matmuls with the VIEW W_dec.mT.mT, which autocast casts
    # anew for every decode (only leaf parameters are cached), so each bf16 GEMM output is cast to fp32 before the sum.
    terms = []
    if "e_multi" in fwd:                                                 # + multi_topk_fvu / 8 (train_sae.py:442)
        terms.append(((2.0 / 8.0) * fwd["e_multi"] / tv, fwd["multi_dense"], fwd["multi_indices"]))
    if have_aux:
        terms.append((de_hat, fwd["aux_dense"], fwd["aux_indices"]))
    terms.append((de, fwd["dense"], fwd["top_indices"]))
    rnd = _r if autocast else (lambda t: t)
    dW_dec = db_dec = dpre = None
    for dy, dense_t, idx_t in terms:
        dW_t, ddense_t, db_t = dec_bwd(dy, dense_t)
        sel_t = torch.zeros_like(ddense_t, dtype=torch.bool).scatter_(1, idx_t, True)
        dpre_t = torch.where(sel_t, ddense_t, torch.zeros_like(ddense_t))
        dW_dec = dW_t if dW_dec is None else dW_dec + dW_t
        db_dec = db_t if db_dec is None else db_dec + db_t
        dpre = dpre_t if dpre is None else rnd(dpre + dpre_t)
    dpre = dpre * (fwd["pre"].to(torch.float32) > 0)
    sae_in = x - b_dec
    if autocast:
        dpb = dpre.to(BF16)
        dW_enc = _mm(dpb.t(), sae_in.to(BF16)).to(torch.float32)         # [n, d]
        dsae_in = _mm(dpb, W_enc.to(BF16)).to(torch.float32)             # [M, d]
        db_enc = _r(dpb.to(torch.float32).sum(0))                        # Linear's bias is a bf16 cast: bf16 gradient
    else:
        dW_enc = dpre.t() @ sae_in
        dsae_in = dpre @ W_enc
        db_enc = dpre.sum(0)
    db_dec = db_dec - dsae_in.sum(0)                                     # sae_in = x - b_dec (:74)
    return {"W_enc": dW_enc, "b_enc": db_enc, "W_dec": dW_dec, "b_dec": db_dec}


# --------------------------------------------------------------------------------------
# clip, optimizers, schedules
# --------------------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm: float) -> Tuple[torch.Tensor, list]:
    """torch.nn.utils.clip_grad_norm_ (train_sae.py:449): total 2-norm over all grads;
    coef = clamp(max_norm / (norm + 1e-6), max=1); every grad multiplied by coef."""
    if NORM_MODE == "native":
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    else:
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(torch.float32)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [g * coef for g in grads]


@dataclass
class OptState:
    step: int = 0
    exp_avg: Dict[str, torch.Tensor] = field(default_factory=dict)
    exp_avg_sq: Dict[str, torch.Tensor] = field(default_factory=dict)


def adam_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], st: OptState, lr: float,
              betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.Adam single-tensor path as called at train_sae.py:378-379 (weight_decay
    is *not* passed to Adam there).  In-place on ``params``."""
    st.step += 1
    b1, b2 = betas
    bc1 = 1 - b1 ** st.step
    bc2 = 1 - b2 ** st.step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for k, p in params.items():
        g = grads[k]
        m = st.exp_avg.setdefault(k, torch.zeros_like(p))
        v = st.exp_avg_sq.setdefault(k, torch.zeros_like(p))
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / bc2_sqrt).add_(eps)
        p.addcdiv_(m, denom, value=-step_size)


def radam_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], st: OptState, lr: float,
               betas=(0.9, 0.999), eps: float = 1e-5, weight_decay: float = 0.0) -> None:
    """torch.optim.RAdam single-tensor path as called at train_sae.py:374-377 (eps=1e-5,
    non-decoupled weight decay).  In-place on ``params``."""
    st.step += 1
    b1, b2 = betas
    t = st.step
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    rho_inf = 2 / (1 - b2) - 1
    rho_t = rho_inf - 2 * t * (b2 ** t) / bc2
    for k, p in params.items():
        g = grads[k]
        if weight_decay != 0:
            g = g.add(p, alpha=weight_decay)
        m = st.exp_avg.setdefault(k, torch.zeros_like(p))
        v = st.exp_avg_sq.setdefault(k, torch.zeros_like(p))
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        m_hat = m / bc1
        if rho_t > 5.0:
            rect = math.sqrt((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t))
            adaptive = math.sqrt(bc2) / (v.sqrt().add(eps))
            p.add_(m_hat * lr * adaptive * rect, alpha=-1.0)
        else:
            p.add_(m_hat * lr, alpha=-1.0)


def lr_at(step_index: int, base_lr: float, scheduler: str, steps: int, num_warmup_steps: int = 0) -> float:
    """Learning rate used by the optimizer step with 0-based index ``step_index``
    (= number of scheduler.step() calls so far; train_sae.py:383-394,450-451).
    cosine: CosineAnnealingLR(T_max=steps, eta_min=0) closed form;
    linear: transformers.get_linear_schedule_with_warmup lambda."""
    t = step_index
    if scheduler == "cosine":
        return base_lr * (1 + math.cos(math.pi * t / steps)) / 2
    if scheduler == "linear":
        if t < num_warmup_steps:
            return base_lr * float(t) / float(max(1, num_warmup_steps))
        return base_lr * max(0.0, float(steps - t) / float(max(1, steps - num_warmup_steps)))
    raise ValueError(f"Invalid scheduler: {scheduler}, must be 'cosine' or 'linear'")


# --------------------------------------------------------------------------------------
# whole train steps (what bench.py's cpu_baseline times and the parity tests compare against)
# --------------------------------------------------------------------------------------
def l1_train_step(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor, st: OptState, *, recon_alpha: float,
                  lr: float, clip_thresh: float, optimizer: str = "radam", weight_decay: float = 0.0,
                  autocast: bool = True, precision: str = "bf16") -> Dict[str, torch.Tensor]:
    """One iteration of train_sae.py:429-451 for the L1 variant.  W, b updated in place
    (W is first column-normalised in place, as encode() does)."""
    W.copy_(normalize_columns(W))
    fwd = l1_forward(x, W, b, recon_alpha, autocast, "fp8" if precision == "fp8bwd" else precision)
    dW, db = l1_backward(x, W, b, fwd, recon_alpha, autocast, precision)
    gnorm, (db_c, dW_c) = clip_grad_norm([db, dW], clip_thresh)
    params = {"encoder_bias": b, "decoder.weight": W}
    grads = {"encoder_bias": db_c, "decoder.weight": dW_c}
    if optimizer == "radam":
        radam_step(params, grads, st, lr, eps=1e-5, weight_decay=weight_decay)
    elif optimizer == "adam":
        adam_step(params, grads, st, lr)
    else:
        raise ValueError(f"Invalid optimizer: {optimizer}, must be 'radam' or 'adam'")
    return {"l1_loss": fwd["l1_loss"], "reconstruction_loss": fwd["reconstruction_loss"],
            "mse": fwd["mse"], "grad_norm": gnorm, "dW": dW, "db": db}


def topk_train_step(x3: torch.Tensor, P: Dict[str, torch.Tensor], st: OptState, *, k: int, lr: float,
                    clip_thresh: float, dead_mask: Optional[torch.Tensor] = None, auxk_alpha: float = 0.0,
                    optimizer: str = "adam", weight_decay: float = 0.0,
                    autocast: bool = True, multi_topk: bool = False, stable_ties: bool = False) -> Dict[str, torch.Tensor]:
    """One iteration of train_sae.py:429-451 for the TopK variant.  P holds the reference's
    state_dict keys W_dec, b_dec, encoder.weight, encoder.bias (updated in place)."""
    fwd = topk_forward(x3, P["encoder.weight"], P["encoder.bias"], P["W_dec"], P["b_dec"], k,
                       dead_mask, auxk_alpha, autocast, multi_topk, stable_ties)
    g = topk_backward(x3, P["encoder.weight"], P["encoder.bias"], P["W_dec"], P["b_dec"], fwd,
                      auxk_alpha, autocast)
    # nn.Module.parameters() order: the module's own parameters (W_dec, b_dec) come before its children's (encoder.*)
    order = ["W_dec", "b_dec", "encoder.weight", "encoder.bias"]
    raw = [g["W_dec"], g["b_dec"], g["W_enc"], g["b_enc"]]
    gnorm, clipped = clip_grad_norm(raw, clip_thresh)
    params = {kk: P[kk] for kk in order}
    grads = dict(zip(order, clipped))
    if optimizer == "radam":
        radam_step(params, grads, st, lr, eps=1e-5, weight_decay=weight_decay)
    else:
        adam_step(params, grads, st, lr)
    return {"fvu": fwd["fvu"], "auxk_loss": fwd["auxk_loss"], "mse": fwd["mse"], "grad_norm": gnorm,
            "multi_topk_fvu": fwd["multi_topk_fvu"], "fire_indices": fwd["fire_indices"],
            "top_indices": fwd["top_indices"], "top_acts": fwd["top_acts"], "grads": dict(zip(order, raw))}

"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and against the golden
vectors produced by the real reference.  Tolerances (stated, fp): single-step losses rtol 1e-3;
k-step trajectories rtol 1e-2 on losses and rel-L2 <= 1e-3 on weights; raw gradients
rel-Frobenius <= 5e-3 (the oracle rounds both weight-gradient GEMMs to bf16 as CPU autocast does,
the engine keeps them in fp32)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu

L1_CASES = ["l1_radam_cosine_d16", "l1_adam_linear_d48", "l1_radam_wd_d32", "l1_radam_cosine_d384"]


def _engine(**kw):
    from freud_amd.engine import SaeEngine
    return SaeEngine(**kw)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("name", L1_CASES)
def test_l1_steps_match_reference_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    meta = json.loads(str(z["meta"]))
    d, n, M = meta["d"], meta["n"], meta["B"] * meta["T"]
    eng = _engine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer=meta["optimizer"],
                  recon_alpha=meta["recon_alpha"], clip_thresh=meta["clip_thresh"], weight_decay=meta["weight_decay"])
    eng.set_params({"decoder.weight": z["W0"], "encoder_bias": z["b0"]})
    xs = torch.tensor(z["x"]).reshape(meta["steps"], M, d).cuda()
    for i in range(meta["steps"]):
        lr = O.lr_at(i, meta["lr"], meta["scheduler"], meta["total_steps"], meta["num_warmup_steps"])
        eng.forward_backward(xs[i])
        if i == 0:
            g = eng.debug_read(2, d * n + n)
            assert _rel(g[: d * n], z["dW_step1"].ravel()) < 5e-3
            # db sums few rows here; one bf16 flip of x_hat (MFMA vs MKL summation order) moves e = x_hat - x by
            # 2^-8 |x_hat|, so the bias gradient gets a wider band than dW on these tiny batches
            assert _rel(g[d * n:], z["db_step1"]) < 1.5e-2
            c = eng.debug_read(0, M * n).reshape(M, n)
            # latent: engine stores bf16(c); reference c is fp32 relu(bf16(xW)+b)
            cref = z["c_step1"].reshape(M, n)
            np.testing.assert_allclose(c, cref, rtol=1e-2, atol=2e-2 * np.abs(cref).max())
        eng.optimizer_step(lr)
        m = eng.metrics()
        tol = 1e-3 if i == 0 else 1e-2
        assert m[0] == pytest.approx(float(z["recon"][i]), rel=tol)
        assert m[1] == pytest.approx(float(z["l1"][i]), rel=tol)
        assert m[2] == pytest.approx(float(z["mse"][i]), rel=tol)
        assert m[3] == pytest.approx(float(z["gnorm"][i]), rel=tol)
    p = eng.get_params()
    assert _rel(p["decoder.weight"], z["W_final"]) < 1e-3
    assert np.abs(p["encoder_bias"] - z["b_final"]).max() < 1e-3 * max(np.abs(z["b_final"]).max(), 1e-3) + 1e-6
    step, m1, m2 = eng.get_opt_state()
    assert step == meta["steps"]
    assert _rel(m1["decoder.weight"], z["m_W"]) < 1e-2
    assert _rel(m2["decoder.weight"], z["v_W"]) < 2e-2
    assert _rel(m1["encoder_bias"], z["m_b"]) < 3e-2          # (db sums few rows on these tiny batches: see above)
    assert _rel(m2["encoder_bias"], z["v_b"]) < 6e-2
    # validate()-style forward on the last batch (train_sae.py:168-190): the reference's CPU validate runs fp32 without
    # autocast, the engine's eval runs the bf16 kernels -> rtol 1e-2 on the tiny fixtures; the in-place column
    # renormalisation of encode() (l1autoencoder.py:71-73) must leave the same weights
    eng.eval(xs[-1])
    me = eng.metrics()
    assert me[0] == pytest.approx(float(z["eval_recon"]), rel=1e-2)
    assert me[1] == pytest.approx(float(z["eval_l1"]), rel=1e-2)
    assert me[2] == pytest.approx(float(z["eval_mse"]), rel=1e-2)
    assert _rel(eng.get_params()["decoder.weight"], z["W_after_eval"]) < 1e-3
    eng.close()


@pytest.mark.parametrize("d,n,M,generic", [(384, 3072, 1500, False), (384, 200, 1500, True), (1280, 2560, 700, False)])
def test_eval_and_latent_colmax_match_oracle(d, n, M, generic):
    """validate() numerics: sae_eval losses and sae_latent_colmax (torch.max(|latent|, dim=0),
    train_sae.py:176-178) against the oracle's autocast forward on the same weights: losses rtol 2e-3, per-feature
    maxima equal to the maxima of the oracle's bf16-rounded latent up to one bf16 ulp, dead features (max == 0) identical."""
    g = torch.Generator().manual_seed(d + n)
    W = torch.randn(d, n, generator=g)
    b = 0.02 * torch.randn(n, generator=g) - 0.05
    x = ((torch.relu(torch.randn(M, 48, generator=g)) * 0.1) @ torch.randn(48, d, generator=g))
    x.view(-1)[torch.randint(0, x.numel(), (40,), generator=g)] = -1.0
    eng = _engine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=1e4, force_generic=generic)
    eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
    eng.eval(x.cuda())
    m = eng.metrics()
    cm = eng.latent_colmax()
    Wn = O.normalize_columns(W)
    f = O.l1_forward(x, Wn, b, 1e4, autocast=True)
    assert m[0] == pytest.approx(f["reconstruction_loss"].item(), rel=2e-3)
    assert m[1] == pytest.approx(f["l1_loss"].item(), rel=2e-3)
    assert m[2] == pytest.approx(f["mse"].item(), rel=2e-3)
    assert m[4] == float(f["count"])
    ref = f["c"].to(torch.bfloat16).float().max(0).values.numpy()      # the engine's latent is stored as bf16
    assert cm.shape == (n,)
    np.testing.assert_allclose(cm, ref, rtol=2 ** -7, atol=1e-6)
    assert np.array_equal(cm == 0, ref == 0) or (np.abs(cm - ref)[(cm == 0) != (ref == 0)] < 1e-3).all()
    # the synchronisation-free form validate() uses (sae_eval_into): rows on the device, the same numbers (a second forward
    # renormalises the already normalised columns again: last-bit differences in W, hence no bitwise equality)
    met = torch.zeros(2, 8, device="cuda")
    cmx = torch.full((2, n), -1.0, device="cuda")
    eng.eval_into(x.cuda(), met[1], cmx[1])
    torch.cuda.synchronize()
    np.testing.assert_allclose(met[1].cpu().numpy(), m, rtol=1e-4)
    assert float(met[0].abs().sum()) == 0.0 and float(cmx[0].max()) == -1.0          # the neighbouring rows are untouched
    np.testing.assert_allclose(cmx[1].cpu().numpy(), cm, rtol=2 ** -7, atol=1e-6)
    eng.close()


@pytest.mark.parametrize("dtype,where", [(torch.bfloat16, (5 * 128 + 77, 201)), (torch.bfloat16, (0, 0)), (torch.bfloat16, (2047, 383)),
                                         (torch.float16, (9 * 128 + 3, 130))])
def test_one_masked_entry_in_one_block_of_many(dtype, where):
    """The fused forward looks for -1.0 entries while it stages its 128 x 384 block of x and takes the arithmetic without the mask when the
    block holds none (fwd_fused2.h, round 5): ONE masked entry in one block of sixteen must still be found (count = M d - 1 exactly), left out
    of the masked MSE and zeroed in dx_hat.  x is small (1e-3) so that the masked entry's own error (about 1) would double the MSE
    and dominate the gradients if it were not taken out; losses / gradients against the oracle (train_sae.py:421-453, autoencoder.py:8-16)."""
    d, n, M = 384, 3072, 2048
    g = torch.Generator().manual_seed(77)
    W = torch.randn(d, n, generator=g) / d ** 0.5
    b = 0.01 * torch.randn(n, generator=g)
    x = (1e-3 * torch.randn(M, d, generator=g)).to(dtype)
    assert int((x == -1.0).sum()) == 0
    x[where[0], where[1]] = -1.0
    eng = _engine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e4)
    eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
    eng.forward_backward(x.cuda())
    graw = eng.debug_read(2, d * n + n)
    eng.optimizer_step(4e-4)
    m = eng.metrics()
    out = O.l1_train_step(x.float(), W.clone(), b.clone(), O.OptState(), recon_alpha=1e4, lr=4e-4, clip_thresh=1.0, optimizer="adam")
    assert m[4] == float(M * d - 1)
    assert m[0] == pytest.approx(out["reconstruction_loss"].item(), rel=1e-3)
    assert m[2] == pytest.approx(out["mse"].item(), rel=1e-3)
    assert _rel(graw[: d * n], out["dW"].numpy().ravel()) < 5e-3
    assert _rel(graw[d * n:], out["db"].numpy()) < 1e-2
    eng.close()


@pytest.mark.parametrize("d,n,M,dtype,opt,generic", [
    (384, 3072, 1024, torch.float32, "radam", False),
    (384, 3072, 1024, torch.float32, "radam", True),       # generic three-GEMM backward on the same case
    (384, 3072, 1000, torch.bfloat16, "radam", False),     # ragged M (not a tile multiple)
    (384, 200, 1500, torch.float16, "adam", False),        # the stock tiny_l1.json dictionary size
    (384, 200, 1500, torch.float16, "adam", True),
    (384, 3072, 8192 + 96, torch.bfloat16, "adam", False), # several 32-row steps per row range, uneven split
    (768, 1536, 512, torch.float32, "adam", False),
    (1280, 2560, 384, torch.bfloat16, "radam", False),
    (1280, 5120, 1024, torch.bfloat16, "adam", False),     # every GEMM on the 256x256 kernel (configs[3] proportions)
    (384, 12288, 1024, torch.bfloat16, "radam", False),    # the reference's default expansion_factor 32 (config.py:7)
    (1280, 40960, 512, torch.bfloat16, "adam", False),     # BASELINE configs[3] at its real dictionary size
])
def test_l1_step_matches_oracle(d, n, M, dtype, opt, generic):
    g = torch.Generator().manual_seed(d + n + M)
    if n <= 8192:
        W = torch.empty(d, n)
        torch.nn.init.orthogonal_(W, generator=g)
    else:        # (orthogonal_ of a 40 960 x 1280 matrix takes minutes on the host; parity does not need it)
        W = torch.randn(d, n, generator=g) / d ** 0.5
    b = 0.01 * torch.randn(n, generator=g)
    z = torch.relu(torch.randn(M, 64, generator=g)) * 0.1
    x = (z @ torch.randn(64, d, generator=g)).to(dtype)
    x.view(-1)[torch.randint(0, x.numel(), (50,), generator=g)] = -1.0
    alpha, lr = 1e4, 4e-4
    eng = _engine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer=opt, recon_alpha=alpha, force_generic=generic)
    eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
    Wo, bo, st = W.clone(), b.clone(), O.OptState()
    xd = x.cuda()
    for i in range(3):
        eng.forward_backward(xd)
        graw = eng.debug_read(2, d * n + n)
        eng.optimizer_step(lr)
        m = eng.metrics()
        out = O.l1_train_step(x.float(), Wo, bo, st, recon_alpha=alpha, lr=lr, clip_thresh=1.0, optimizer=opt)
        assert _rel(graw[: d * n], out["dW"].numpy().ravel()) < 5e-3
        assert _rel(graw[d * n:], out["db"].numpy()) < 1e-2
        tol = 1e-3 if i == 0 else 1e-2
        assert m[0] == pytest.approx(out["reconstruction_loss"].item(), rel=tol)
        assert m[1] == pytest.approx(out["l1_loss"].item(), rel=tol)
        assert m[3] == pytest.approx(out["grad_norm"].item(), rel=tol)
    p = eng.get_params()
    assert _rel(p["decoder.weight"], Wo.numpy()) < 1e-3
    eng.close()


def test_gemm256_matches_gemm128():
    """The 256x256 LDS-DMA GEMM (taken when both output dimensions are multiples of 256) and the 128x128 kernel compute
    the same step: same K order per output element, only the split-K partition of the weight gradient may differ."""
    d, n, M = 512, 1024, 768
    g = torch.Generator().manual_seed(7)
    W = torch.empty(d, n)
    torch.nn.init.orthogonal_(W, generator=g)
    b = 0.01 * torch.randn(n, generator=g)
    x = (torch.relu(torch.randn(M, 64, generator=g)) * 0.1 @ torch.randn(64, d, generator=g)).to(torch.bfloat16).cuda()
    res = []
    for force128 in (False, True):
        eng = _engine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e4, force_gemm128=force128)
        eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
        eng.forward_backward(x)
        res.append((eng.debug_read(2, d * n + n), eng.debug_read(0, M * n), eng.metrics().copy()))
        eng.close()
    (g256, c256, m256), (g128, c128, m128) = res
    assert np.array_equal(c256, c128)                       # latent: bit-identical
    assert _rel(g256, g128) < 1e-5
    assert np.allclose(m256[:3], m128[:3], rtol=1e-5)


@pytest.mark.parametrize("n", [13312, 16384])
def test_weight_gradient_tail_split_matches_uniform_split_k(n):
    """Generic L1 path, >= 256 output tiles of the weight gradient: whole tiles are written straight into the gradient buffer
    and the tiles left over after whole rounds of 256 workgroups are computed in K pieces (n = 13 312: 260 tiles, 4 tail tiles
    x 4 pieces; n = 16 384: 320 tiles, 64 tail tiles x 4 pieces).  debug_flags 81 = the uniform split-K through slabs: the
    same sums in another association."""
    d, M = 1280, 2048
    g = torch.Generator().manual_seed(9)
    W = torch.empty(d, n)
    torch.nn.init.orthogonal_(W, generator=g)
    b = 0.01 * torch.randn(n, generator=g)
    x = (torch.relu(torch.randn(M, 64, generator=g)) * 0.1 @ torch.randn(64, d, generator=g)).to(torch.bfloat16).cuda()
    res = []
    for dbg in (0, 81):
        eng = _engine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e4, debug_flags=dbg)
        eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
        eng.forward_backward(x)
        res.append(eng.debug_read(2, d * n + n))
        eng.step(x, 1e-3)
        res.append(eng.get_params()["decoder.weight"])
        eng.close()
    assert np.abs(res[0]).max() > 0
    assert _rel(res[0], res[2]) < 1e-5                      # raw gradient
    assert _rel(res[1], res[3]) < 1e-6                      # weights after a step


@pytest.mark.parametrize("variant,d,n,M", [("l1", 384, 3072, 1024), ("l1", 1024, 2048, 512), ("topk", 256, 1024, 512)])
def test_grad_ready_callback_ranges(variant, d, n, M):
    """sae_set_grad_ready_callback: the announced ranges are disjoint and cover the gradient buffer; the chunked
    weight-gradient GEMM of the generic L1 path (d_p >= 1024) yields the same gradients as the single launch."""
    g = torch.Generator().manual_seed(11)
    x = (torch.relu(torch.randn(M, 64, generator=g)) * 0.1 @ torch.randn(64, d, generator=g)).to(torch.bfloat16).cuda()
    kw = dict(variant=variant, d_model=d, n_dict=n, max_rows=M, optimizer="adam")
    if variant == "topk":
        kw.update(k=16, auxk_alpha=0.03125)
    outs = []
    for hook in (False, True):
        eng = _engine(**kw)
        if variant == "l1":
            W = torch.empty(d, n)
            torch.nn.init.orthogonal_(W, generator=torch.Generator().manual_seed(3))
            eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
        else:
            We = (torch.rand(n, d, generator=torch.Generator().manual_seed(3)) * 2 - 1) / d ** 0.5
            eng.set_topk_options(1e9, M)
            eng.set_params({"encoder.weight": We.numpy(), "encoder.bias": np.zeros(n, np.float32),
                            "W_dec": (We / We.norm(dim=1, keepdim=True)).numpy(), "b_dec": np.zeros(d, np.float32)})
        ranges = []
        if hook:
            eng.set_grad_ready_callback(lambda off, cnt: ranges.append((off, cnt)))
        eng.forward_backward(x)
        torch.cuda.synchronize()
        grads = eng.grad_tensor().cpu().numpy().copy()
        if hook:
            total = grads.size
            cover = np.zeros(total, np.int32)
            for off, cnt in ranges:
                assert 0 <= off and off + cnt <= total and cnt > 0
                cover[off:off + cnt] += 1
            assert (cover == 1).all()
            assert len(ranges) == {"l1": 1 if d < 1024 else 3, "topk": 3}[variant]
            eng.set_grad_ready_callback(None)
            ranges.clear()
            eng.forward_backward(x)
            assert ranges == []

            def boom(off, cnt):
                raise RuntimeError("hook failed")
            eng.set_grad_ready_callback(boom)
            with pytest.raises(RuntimeError, match="hook failed"):
                eng.forward_backward(x)
        outs.append(grads)
        eng.close()
    assert _rel(outs[1], outs[0]) < 1e-5


def test_l1_determinism_and_eval():
    """Two identical runs are bitwise equal (fixed-order reductions, no float atomics); eval
    renormalises the decoder columns in place like the reference's encode() (l1autoencoder.py:71-73)."""
    d, n, M = 384, 1024, 2048
    g = torch.Generator().manual_seed(7)
    W = torch.randn(d, n, generator=g)
    x = torch.randn(M, d, generator=g).cuda()
    outs = []
    for _ in range(2):
        eng = _engine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=10.0)
        eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
        for _i in range(3):
            eng.step(x, 1e-3)
        outs.append((eng.get_params()["decoder.weight"].copy(), eng.metrics().copy()))
        eng.eval(x)
        Wn = eng.get_params()["decoder.weight"]
        np.testing.assert_allclose(np.linalg.norm(Wn, axis=0), 1.0, rtol=1e-5)
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def test_full_size_properties():
    """BASELINE config 2 size (M=65 536, d=384, n=3072): size-independent properties instead of
    the oracle -- (i) masked entries contribute nothing: a batch whose second half is entirely
    -1.0 gives the same masked MSE and count as the first half alone; (ii) the loss goes down
    over steps on a learnable batch."""
    d, n, M = 384, 3072, 65536
    g = torch.Generator().manual_seed(0)
    W = torch.empty(d, n)
    torch.nn.init.orthogonal_(W, generator=g)
    x = ((torch.relu(torch.randn(M, 64, generator=g)) * 0.1) @ torch.randn(64, d, generator=g)).to(torch.bfloat16).cuda()
    eng = _engine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=1e4)
    eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
    eng.eval(x)
    full = eng.metrics().copy()
    x2 = x.clone()
    x2[M // 2:] = -1.0
    eng.eval(x2)
    half = eng.metrics().copy()
    eng.eval(x[: M // 2].contiguous())
    first = eng.metrics().copy()
    assert half[4] == pytest.approx(first[4], rel=1e-6)                 # same unmasked count
    assert half[0] == pytest.approx(first[0], rel=1e-4)                 # same masked MSE
    assert full[4] == pytest.approx(float((x != -1.0).sum().item()), rel=1e-6)   # bf16 data does hit -1.0 exactly
    losses = []
    for _ in range(20):
        eng.step(x, 4e-4)
        m = eng.metrics()
        losses.append(m[0] + m[1])
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    eng.close()


def test_c4_full_size_properties():
    """BASELINE configs[3] at its REAL shape on one GPU (d=1280, n=40 960, M=65 536 rows -- VERDICT r3: the one hot shape that
    had no asserted property at size; the oracle case of this shape is M = 512, where the weight-gradient GEMM's K loop is two
    tiles long instead of 2048).  Size-independent properties: (i) a batch whose second half is entirely -1.0 has the unmasked
    count and masked MSE of the first half alone; (ii) the loss goes down on a learnable batch; (iii) two runs are BITWISE
    equal (tail-split weight gradient, k-half ring, persistent K = d GEMMs: nothing may depend on timing); (iv) the tail-split
    weight gradient (whole tiles straight into the gradient + 32 x 8 K pieces) equals the uniform split-K form (debug_flags 81)
    to summation order (5e-5: K = 131 072 fp32 additions in two different orders)."""
    d, n, M = 1280, 40960, 65536
    g = torch.Generator().manual_seed(0)
    W = torch.randn(d, n, generator=g)
    W /= W.norm(dim=0, keepdim=True)
    x = ((torch.relu(torch.randn(M, 64, generator=g)) * 0.1) @ torch.randn(64, d, generator=g)).to(torch.bfloat16).cuda()
    params = {"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)}

    def run(steps, **kw):
        eng = _engine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e4, **kw)
        eng.set_params(params)
        losses = []
        for _ in range(steps):
            eng.step(x, 1e-4)
            m = eng.metrics()
            losses.append(float(m[0] + m[1]))
        return eng, losses

    eng, _ = run(0)
    eng.eval(x)
    full = eng.metrics().copy()
    x2 = x.clone()
    x2[M // 2:] = -1.0
    eng.eval(x2)
    half = eng.metrics().copy()
    eng.eval(x[: M // 2].contiguous())
    first = eng.metrics().copy()
    del x2
    assert half[4] == pytest.approx(first[4], rel=1e-6)
    assert half[0] == pytest.approx(first[0], rel=1e-4)
    assert full[4] == pytest.approx(float((x != -1.0).sum().item()), rel=1e-6)
    eng.forward_backward(x)                                   # raw gradient of step 1, tail-split form
    torch.cuda.synchronize()
    g_tail = eng.grad_tensor()[: d * n].clone()
    eng.close()
    outs = []
    for _ in range(2):
        eng, losses = run(4)
        assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
        outs.append(eng.get_params()["decoder.weight"].copy())
        eng.close()
    assert np.array_equal(outs[0], outs[1])
    eng, _ = run(0, debug_flags=81)                           # uniform split-K through slabs
    eng.forward_backward(x)
    torch.cuda.synchronize()
    g_uni = eng.grad_tensor()[: d * n]
    rel = float((g_tail - g_uni).norm() / g_uni.norm())
    eng.close()
    assert rel < 5e-5, rel       # K = 2 M = 131 072 products per element, fp32 accumulators, two summation orders (measured 1.0e-5)


def test_alternative_launch_forms_are_bitwise_identical(monkeypatch):
    """Round 3 changed HOW the fused d = 384 step is launched, not what it computes: the second forward decomposition
    (fwd_fused2.h; FREUD_FWD=1 selects the first), the loss finalisation folded into reduce_grads (debug_flags 78 = own kernel),
    the optimizer that leaves the column-norm partials (debug_flags 79 = flat optimizer + separate pass) and then the one that
    also writes the next forward's weight copies (debug_flags 80 = round-3-first-half order).  Same arithmetic in
    the same order: weights and optimizer moments after three steps must be BITWISE equal across all of them (the logged loss
    scalars agree to fp32 round-off: their partial sums are taken in another order)."""
    from freud_amd.engine import SaeEngine
    d, n, M = 384, 1024, 1024
    g = torch.Generator().manual_seed(5)
    W = torch.empty(d, n)
    torch.nn.init.orthogonal_(W, generator=g)
    b = 0.01 * torch.randn(n, generator=g)
    x = ((torch.relu(torch.randn(M, 32, generator=g)) * 0.1) @ torch.randn(32, d, generator=g)).to(torch.bfloat16)
    x.view(-1)[::501] = -1.0
    xd = x.cuda()
    results = []
    for fwd, dbg in (("2", 0), ("1", 0), ("2", 78), ("2", 79), ("2", 80)):
        monkeypatch.setenv("FREUD_FWD", fwd)
        eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=1e4, debug_flags=dbg)
        eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
        for i in range(3):
            eng.step(xd, 4e-4)
        _, m1, v1 = eng.get_opt_state()
        results.append((eng.get_params(), m1, v1, eng.metrics().copy()))
        eng.close()
    ref = results[0]
    names = ["fwd1", "dbg78", "dbg79", "dbg80"]
    for name, other in zip(names, results[1:]):
        for k in ref[0]:
            dw = np.abs(ref[0][k] - other[0][k]).max()
            assert np.array_equal(ref[0][k], other[0][k]), (name, k, "params", dw)
            assert np.array_equal(ref[1][k], other[1][k]) and np.array_equal(ref[2][k], other[2][k]), (name, k, "moments")
        # (the loss SCALARS are sums over per-workgroup partials taken in another order: equal to fp32 round-off, not bitwise)
        np.testing.assert_allclose(other[3], ref[3], rtol=2e-6, err_msg=name)


@pytest.mark.parametrize("d,n", [(384, 1024), (64, 256)])
def test_folded_weight_preparation_keeps_the_reference_state_at_every_observation(d, n):
    """L1 with d <= 384: the update also writes the next forward's bf16 weight copies and leaves the fp32 master un-normalised
    (the reference's state after optimizer.step(), train_sae.py:450); the in-place normalisation of the next forward
    (l1autoencoder.py:71-73) happens lazily.  Whatever is observed in between -- parameters right after a step, after an eval
    forward, a second eval, decode, a resumed run -- must be BITWISE what the plain order (debug_flags 80: update, then column
    norms + normalize_cast in the next forward) produces."""
    from freud_amd.engine import SaeEngine
    M = 512
    g = torch.Generator().manual_seed(11)
    W = torch.empty(d, n)
    torch.nn.init.orthogonal_(W, generator=g)
    b = 0.01 * torch.randn(n, generator=g)
    x = ((torch.relu(torch.randn(M, 16, generator=g)) * 0.1) @ torch.randn(16, d, generator=g)).to(torch.bfloat16).cuda()
    lat = torch.relu(torch.randn(8, n, generator=g)).cuda()

    def run(dbg):
        eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e4, debug_flags=dbg)
        eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
        seen = []
        eng.step(x, 1e-3)
        seen.append(eng.get_params()["decoder.weight"].copy())          # after an update: un-normalised
        eng.step(x, 1e-3)
        eng.step(x, 1e-3)
        eng.eval(x)
        seen.append(np.array(eng.metrics()[:2]))
        seen.append(eng.get_params()["decoder.weight"].copy())          # after update + one forward: normalised once
        eng.eval(x)                                                      # the reference normalises again
        eng.step(x, 1e-3)
        eng.eval(x)
        eng.eval(x)
        eng.step(x, 1e-3)
        xh = torch.empty(8, d, device="cuda", dtype=torch.float32)
        eng.decode(lat, xh)                                              # decode() reads the master as is
        torch.cuda.synchronize()
        seen.append(xh.cpu().numpy())
        eng.step(x, 1e-3)
        p = eng.get_params()
        _, m1, v1 = eng.get_opt_state()
        seen += [p["decoder.weight"], p["encoder_bias"], m1["decoder.weight"], v1["decoder.weight"]]
        eng.close()
        return seen

    new, old = run(0), run(80)
    for i, (a, o) in enumerate(zip(new, old)):
        if i == 1:
            np.testing.assert_allclose(a, o, rtol=2e-6)                  # loss scalars: partial sums in another order
        else:
            assert np.array_equal(a, o), (i, np.abs(a - o).max())
    # and the column norms: un-normalised right after an update, 1 after update + forward
    assert np.abs(np.linalg.norm(new[0], axis=0) - 1).max() > 1e-6
    assert np.abs(np.linalg.norm(new[2], axis=0) - 1).max() < 1e-5


@pytest.mark.parametrize("case", ["l1", "l1_fp8", "topk"])
def test_streaming_gemms_equal_the_tile_form(case, monkeypatch):
    """Round 5: the K = d GEMMs of the generic paths run in the streaming form of csrc/gemm256s.h wherever a launch has >= 2048
    tiles of 256x256 (encoder, dpre, TopK encoder, fp8 encoder); FREUD_GEMM_STREAM=0 keeps gemm256.h's tile form.  Same products in
    the same order: the latent (bf16 bits) and the weight gradient must be IDENTICAL between the two, reductions over a tile (L1
    partial sums, db column sums) associate differently and agree to fp32 round-off.  Shape: d = 256, n = 32 768, M = 4096 rows =
    16 x 128 tiles, with a ragged last row block (M = 4000 of M_p = 4096: the PARTIAL instantiation of the streaming epilogue)."""
    from freud_amd.engine import SaeEngine
    d, n, M = 256, 32768, 4000
    g = torch.Generator().manual_seed(5)
    x = ((torch.relu(torch.randn(M, 48, generator=g)) * 0.2) @ torch.randn(48, d, generator=g)).to(torch.bfloat16).cuda()
    outs = []
    for stream in ("1", "0"):
        monkeypatch.setenv("FREUD_GEMM_STREAM", stream)
        if case == "topk":
            eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=32, auxk_alpha=0.0)
            eng.set_topk_options(1e12, M)
            We = (torch.rand(n, d, generator=torch.Generator().manual_seed(2)) * 2 - 1) / d ** 0.5
            eng.set_params({"encoder.weight": We.numpy(), "encoder.bias": (0.01 * torch.randn(n, generator=torch.Generator().manual_seed(3))).numpy(),
                            "W_dec": (We / We.norm(dim=1, keepdim=True)).numpy(), "b_dec": np.zeros(d, np.float32)})
        else:
            eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e3,
                            precision="fp8" if case == "l1_fp8" else "bf16")
            W = torch.randn(d, n, generator=torch.Generator().manual_seed(2))
            W /= W.norm(dim=0, keepdim=True)
            eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": (0.01 * torch.randn(n, generator=torch.Generator().manual_seed(3))).numpy()})
        eng.forward_backward(x)
        torch.cuda.synchronize()
        if case == "topk":
            idx = eng.debug_read(3, M * 32).copy()
            grads = eng.debug_read(2, 2 * n * d + n + d).copy()
            outs.append((idx, grads, eng.metrics().copy()))
        else:
            latent = eng.debug_read(0, M * n).copy()
            grads = eng.debug_read(2, d * n + n).copy()
            outs.append((latent, grads, eng.metrics().copy()))
        eng.close()
    (a0, g0, m0), (a1, g1, m1) = outs
    assert np.array_equal(a0, a1)                                     # latent bits / selected indices
    if case == "topk":
        assert np.array_equal(g0, g1)                                 # the whole sparse backward follows from identical pre-activations
    else:
        assert np.array_equal(g0[: d * n], g1[: d * n])               # dW: same c, same dpre, same GEMM
        np.testing.assert_allclose(g0[d * n:], g1[d * n:], rtol=2e-5, atol=1e-7)       # db: column sums in another order
    np.testing.assert_allclose(m0[:4], m1[:4], rtol=2e-5)

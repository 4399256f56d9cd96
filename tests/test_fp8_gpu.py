"""GPU parity of the fp8 (OCP e4m3) encoder / decoder GEMM path (BASELINE configs[4]: "d=1280 dict 64x fp8 MFMA enc/dec
with bf16 accumulate"; SAE_PREC_FP8, freud_amd/csrc/l1_fp8.h + gemm256_fp8.h) against the oracle's fp8 mode
(oracle/sae_oracle.py l1_forward(precision="fp8"): same power-of-two per-tensor scales, torch's e4m3fn rounding).

The reference has no fp8 mode (its precision is CPU autocast = bf16), so the tolerance is stated in two layers:
(a) engine vs the fp8 oracle -- same quantisation, so the quantised activations must be IDENTICAL bytes, the scales
    identical, the quantised latent identical except where a pre-activation sat on a bf16 rounding boundary (different
    fp32 summation order; <= 0.5 % of the elements, each by at most one e4m3 step, but for < 1e-4 of them in columns where
    a weight sat on an e4m3 rounding boundary), losses rtol 2e-3 at the first step /
    1e-2 along a trajectory, raw gradients rel-Frobenius 1e-2, weights rel-L2 1e-3;
(b) fp8 path vs the bf16 oracle (what the precision change costs): L1 and reconstruction losses within 2e-2, latent
    rel-L2 within 6e-2 on the synthetic batches used here."""
import numpy as np
import pytest
import torch

from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _case(d, n, M, seed, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(d, n, generator=g) / d ** 0.5
    b = 0.01 * torch.randn(n, generator=g)
    x = ((torch.relu(torch.randn(M, 64, generator=g)) * 0.1) @ torch.randn(64, d, generator=g)).to(dtype)
    x.view(-1)[torch.randint(0, x.numel(), (50,), generator=g)] = -1.0
    return W, b, x


@pytest.mark.parametrize("d,n,M,dtype", [
    (1280, 2560, 600, torch.bfloat16),      # ragged M, every padded dimension already a multiple of 256
    (384, 1000, 300, torch.float32),        # d -> 512, n -> 1024, M -> 512 padding
    (1280, 81920, 512, torch.bfloat16),     # BASELINE configs[4] at its real dictionary size
])
def test_fp8_step_matches_fp8_oracle(d, n, M, dtype):
    from freud_amd.engine import SaeEngine
    W, b, x = _case(d, n, M, d + n + M, dtype)
    alpha, lr = 1e4, 4e-4
    eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=alpha, precision="fp8")
    eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
    Wo, bo, st = W.clone(), b.clone(), O.OptState()
    xd = x.cuda()
    for i in range(3 if n <= 4096 else 1):
        eng.forward_backward(xd)
        if i == 0:
            Wn = O.normalize_columns(Wo)
            f8 = O.l1_forward(x.float(), Wn, bo, alpha, True, "fp8")
            f16 = O.l1_forward(x.float(), Wn, bo, alpha, True)
            sc = eng.debug_read(7, 8)
            assert sc[0] == f8["s_x"] and sc[2] == f8["s_c"], (sc, f8["s_x"], f8["s_c"])
            x8 = eng.debug_read(8, M * d).reshape(M, d)
            assert np.array_equal(x8, f8["x8"].numpy()), "activation quantisation differs from torch's e4m3fn rounding"
            c8 = eng.debug_read(9, M * n).reshape(M, n)
            ref8 = f8["c8"].numpy()
            diff = c8 != ref8
            assert diff.mean() < 5e-3, diff.mean()
            if diff.any():
                # where the bf16 pre-activation flipped (one bf16 ulp of |pre| <= c + |b|): the latent moves by that ulp and
                # then by at most one e4m3 step of the LARGER of the two values (2^-3 relative; 2^-9 among subnormals):
                # |c8 - ref8| <= ulp + (ref8 + |c8 - ref8|) / 8
                # A handful of elements (< 1e-4) may differ by more: columns where one 256 w sits on an e4m3 rounding boundary
                # and the engine's column normalisation (fp32 sqrt / divide on the GPU) lands on the other side of it
                ulp = 2.0 ** -7 * (np.abs(ref8[diff]) + 0.06 * f8["s_c"])
                far = np.abs(c8 - ref8)[diff] > 0.15 * np.abs(ref8[diff]) + 2.0 ** -8 + 1.2 * ulp
                assert far.sum() <= 1e-4 * c8.size, far.sum()
            c = eng.debug_read(0, M * n).reshape(M, n)
            assert _rel(c, f8["c"].to(torch.bfloat16).float().numpy()) < 2e-3
            # (b) what fp8 costs against the bf16 arithmetic of the reference
            assert _rel(c, f16["c"].numpy()) < 6e-2
        graw = eng.debug_read(2, d * n + n)
        eng.optimizer_step(lr)
        m = eng.metrics()
        out = O.l1_train_step(x.float(), Wo, bo, st, recon_alpha=alpha, lr=lr, clip_thresh=1.0, optimizer="adam", precision="fp8")
        tol = 2e-3 if i == 0 else 1e-2
        assert m[0] == pytest.approx(out["reconstruction_loss"].item(), rel=tol)
        assert m[1] == pytest.approx(out["l1_loss"].item(), rel=tol)
        assert m[3] == pytest.approx(out["grad_norm"].item(), rel=tol)
        assert _rel(graw[: d * n], out["dW"].numpy().ravel()) < 1e-2
        assert _rel(graw[d * n:], out["db"].numpy()) < 2e-2
        if i == 0:
            assert m[0] == pytest.approx(f16["reconstruction_loss"].item(), rel=2e-2)
            assert m[1] == pytest.approx(f16["l1_loss"].item(), rel=2e-2)
    assert _rel(eng.get_params()["decoder.weight"], Wo.numpy()) < 1e-3
    eng.close()


def test_fp8_rejects_topk_and_bad_precision():
    from freud_amd.engine import SaeEngine, EngineError
    with pytest.raises(EngineError, match="L1 variant only"):
        SaeEngine(variant="topk", d_model=256, n_dict=1024, max_rows=256, optimizer="adam", k=8, precision="fp8")
    with pytest.raises(ValueError, match="Invalid precision"):
        SaeEngine(variant="l1", d_model=256, n_dict=1024, max_rows=256, precision="fp4")


def test_fp8_c5_full_size_properties():
    """BASELINE configs[4] at full size (d=1280, n=81 920, M=65 536 rows): finite, two identical runs bitwise equal,
    the loss decreases, and the eval losses sit within 2e-2 of the bf16 path's on the same weights and batch."""
    from freud_amd.engine import SaeEngine
    d, n, M = 1280, 81920, 65536
    g = torch.Generator().manual_seed(0)
    W = torch.randn(d, n, generator=g) / d ** 0.5
    x = ((torch.relu(torch.randn(M, 64, generator=g)) * 0.1) @ torch.randn(64, d, generator=g)).to(torch.bfloat16).cuda()
    runs = []
    for _ in range(2):
        eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e4, precision="fp8")
        eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
        eng.eval(x)
        ev = eng.metrics().copy()
        losses = []
        for _i in range(3):
            eng.step(x, 1e-4)
            mm = eng.metrics()
            losses.append(float(mm[0] + mm[1]))
        runs.append((ev, losses, eng.get_params()["encoder_bias"].copy()))
        eng.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1] and np.array_equal(runs[0][2], runs[1][2])
    assert np.isfinite(runs[0][1]).all() and runs[0][1][-1] < runs[0][1][0]
    eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e4)
    eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
    eng.eval(x)
    ref = eng.metrics().copy()
    eng.close()
    assert runs[0][0][0] == pytest.approx(ref[0], rel=2e-2)
    assert runs[0][0][1] == pytest.approx(ref[1], rel=2e-2)


@pytest.mark.parametrize("d,n,M", [(1280, 2560, 600), (1280, 81920, 512)])
def test_fp8_dpre_gemm_matches_oracle_and_states_its_cost(d, n, M):
    """SAE_PREC_FP8_BWD (beyond BASELINE configs[4]): the dc = dx_hat W GEMM of the backward on e4m3 operands.
    (a) against the oracle's fp8bwd mode (same scale s_g = 2^floor(log2(448 / max|dx_hat|)), torch's e4m3fn rounding): losses
        as in the fp8 forward test, raw gradients rel-Frobenius 1e-2;
    (b) what it costs: raw gradients within 5e-2 (rel-Frobenius) of the fp8-forward / bf16-backward arithmetic."""
    from freud_amd.engine import SaeEngine
    W, b, x = _case(d, n, M, d + n + M + 1)
    alpha = 1e4
    eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=alpha, precision="fp8bwd")
    eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
    eng.forward_backward(x.cuda())
    g = eng.debug_read(2, d * n + n)
    gW, gb = g[: d * n].reshape(d, n), g[d * n:]
    m = eng.metrics()
    Wn = O.normalize_columns(W.clone())
    f8 = O.l1_forward(x.float(), Wn, b, alpha, True, "fp8")
    assert m[0] == pytest.approx(f8["reconstruction_loss"].item(), rel=2e-3)
    assert m[1] == pytest.approx(f8["l1_loss"].item(), rel=2e-3)
    dW8, db8 = O.l1_backward(x.float(), Wn, b, f8, alpha, True, "fp8bwd")
    dW16, db16 = O.l1_backward(x.float(), Wn, b, f8, alpha, True)
    sg = eng.debug_read(7, 8)[6]
    dxb = ((f8["diff"] * 2.0) * (alpha / f8["count"].to(torch.float32))).to(torch.bfloat16).float()
    assert sg == O._pow2_scale(float(dxb.abs().max()))
    assert _rel(gW, dW8.numpy()) < 1e-2 and _rel(gb, db8.numpy()) < 1e-2                 # (a) same quantisation
    cost_W, cost_b = _rel(gW, dW16.numpy()), _rel(gb, db16.numpy())
    assert cost_W < 5e-2 and cost_b < 5e-2, (cost_W, cost_b)                             # (b) the stated cost of e4m3 there
    eng.close()

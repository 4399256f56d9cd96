"""GPU parity of the TopK engine (encoder MFMA GEMM -> radix top-k select -> sparse decode -> FVU / AuxK ->
backward GEMMs -> Adam) against the CPU oracle and the reference-generated golden vectors.

Top-k ties: `pre` is bf16 under autocast, so the k-th and (k+1)-th largest latents of a row are EQUAL in
4-16 % of rows of the golden batches; the reference takes whichever torch.topk's partial sort leaves
(no rule: sometimes the lower, sometimes the higher column), the engine takes the lower column.  Parity is
therefore stated as: (i) the multiset of selected VALUES is identical on every row; (ii) the index sets
are identical on every row without a boundary tie; (iii) on a batch without boundary ties gradients agree to
rel-Frobenius 1e-2 and losses to rtol 2e-3; (iv) on the golden batches (with ties) losses agree to rtol 3e-3 at
step 1 / 2e-2 along the trajectory and gradients to 0.15 (tie rows pick different decoder rows)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu
KEYS = ["encoder.weight", "encoder.bias", "W_dec", "b_dec"]


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _split(flat, n, d):
    nd = n * d
    return {"encoder.weight": flat[:nd].reshape(n, d), "encoder.bias": flat[nd:nd + n],
            "W_dec": flat[nd + n:2 * nd + n].reshape(n, d), "b_dec": flat[2 * nd + n:2 * nd + n + d]}


def _boundary_ties(pre, k):
    srt = pre.float().sort(dim=1, descending=True).values
    return srt[:, k - 1] == srt[:, k]


def _check_selection(eng, fwd, M, n, k):
    pre = fwd["pre"].reshape(M, n)
    ties = _boundary_ties(pre, k).numpy()
    idx = eng.debug_read(3, M * k).reshape(M, k).astype(np.int64)
    ref = fwd["top_indices"].reshape(M, k).numpy()
    same = (np.sort(idx, 1) == np.sort(ref, 1)).all(1)
    assert same[~ties].all(), "index sets differ on rows without a boundary tie"
    dense = eng.debug_read(0, M * n).reshape(M, n)
    got_vals = np.sort(np.take_along_axis(dense, idx, 1), 1)
    ref_vals = np.sort(fwd["top_acts"].float().reshape(M, k).numpy(), 1)
    assert np.array_equal(got_vals, ref_vals), "selected activation values differ"
    return ties


@pytest.mark.parametrize("name", ["topk_adam_linear_d16", "topk_adam_linear_d64"])
def test_topk_steps_match_reference_golden(golden_dir, name):
    from freud_amd.engine import SaeEngine
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    meta = json.loads(str(z["meta"]))
    d, n, k, B, T = meta["d"], meta["n"], meta["k"], meta["B"], meta["T"]
    M = B * T
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k,
                    auxk_alpha=meta["auxk_alpha"], clip_thresh=1.0)
    eng.set_topk_options(meta["dead_feature_threshold"], T)
    eng.set_params({kk: z["init__" + kk] for kk in KEYS})
    P0 = {kk: torch.tensor(z["init__" + kk]) for kk in KEYS}
    xs = torch.tensor(z["x"])
    xd = xs.cuda()
    for i in range(meta["steps"]):
        lr = O.lr_at(i, meta["lr"], "linear", meta["steps"], meta["num_warmup_steps"])
        eng.forward_backward(xd[i])
        if i == 0:
            f = O.topk_forward(xs[0], P0["encoder.weight"], P0["encoder.bias"], P0["W_dec"], P0["b_dec"], k)
            assert torch.equal(f["top_indices"].reshape(M, k).sort(1).values,
                               torch.tensor(z["first__top_indices"]).reshape(M, k).sort(1).values)   # oracle == reference
            _check_selection(eng, f, M, n, k)
            g = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
            for kk in KEYS:
                assert _rel(g[kk], z["first__" + kk]) < 0.15, kk
        eng.optimizer_step(lr)
        m = eng.metrics()
        tol = 3e-3 if i == 0 else 2e-2
        assert m[0] == pytest.approx(float(z["fvu"][i]), rel=tol)
        assert m[1] == pytest.approx(float(z["auxk"][i]), rel=0.1, abs=1e-6)
        assert m[5] == pytest.approx(float(z["num_dead"][i]) / n, abs=2.0 / n)
    p = eng.get_params()
    for kk in KEYS:   # Adam turns a tie row's different gradient into a full-size step on the few elements it touches
        assert _rel(p[kk], z["final__" + kk]) < (5e-3 if kk in ("encoder.weight", "W_dec") else 0.15), kk
    eng.close()


def _make_case(d, n, k, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    We = torch.randn(n, d, generator=g) / d ** 0.5
    Wd = We.clone()
    Wd /= Wd.norm(dim=1, keepdim=True) + torch.finfo(torch.float32).eps
    P = {"encoder.weight": We, "encoder.bias": torch.zeros(n), "W_dec": Wd, "b_dec": 0.01 * torch.randn(d, generator=g)}
    x = (torch.relu(torch.randn(B * T, 48, generator=g)) @ torch.randn(48, d, generator=g) * 0.2).reshape(B, T, d)
    return P, x


@pytest.mark.parametrize("d,n,k,B,T", [(768, 512, 8, 2, 4), (384, 1024, 8, 2, 6), (1280, 768, 16, 1, 6)])
def test_topk_tie_free_batch_matches_oracle(d, n, k, B, T):
    """A batch without any boundary tie (seed searched on the CPU): selection must match exactly and the
    whole step arithmetic to the stated fp tolerance."""
    from freud_amd.engine import SaeEngine
    for seed in range(200):
        P, x = _make_case(d, n, k, B, T, seed)
        f = O.topk_forward(x, P["encoder.weight"], P["encoder.bias"], P["W_dec"], P["b_dec"], k)
        if not _boundary_ties(f["pre"].reshape(B * T, n), k).any():
            break
    else:
        pytest.skip("no tie-free batch found")
    M = B * T
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=0.0)
    eng.set_topk_options(1e9, T)
    eng.set_params({kk: v.numpy() for kk, v in P.items()})
    eng.forward_backward(x.cuda())
    ties = _check_selection(eng, f, M, n, k)
    assert not ties.any()
    graw = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
    eng.optimizer_step(1e-4)
    m = eng.metrics()
    st = O.OptState()
    out = O.topk_train_step(x, P, st, k=k, lr=1e-4, clip_thresh=1.0, dead_mask=None, auxk_alpha=0.0, optimizer="adam")
    assert m[0] == pytest.approx(out["fvu"].item(), rel=2e-3)
    assert m[2] == pytest.approx(out["mse"].item(), rel=2e-3)
    assert m[3] == pytest.approx(out["grad_norm"].item(), rel=1e-2)
    for kk in KEYS:
        assert _rel(graw[kk], out["grads"][kk].numpy()) < 1e-2, kk
    p = eng.get_params()
    for kk in KEYS:
        assert _rel(p[kk], P[kk].numpy()) < 1e-3
    eng.close()


def test_topk_auxk_and_dead_bookkeeping():
    """Dead-latent bookkeeping (train_sae.py:436-446) and the AuxK branch: with a tiny threshold the latents that
    did not fire in step 1 are dead in step 2 and the AuxK loss switches on, like the oracle's."""
    from freud_amd.engine import SaeEngine
    d, n, k, B, T, aux = 384, 1024, 16, 2, 64, 0.03125
    P, x = _make_case(d, n, k, B, T, 7)
    thr = 0.5 * B * T
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=B * T, optimizer="adam", k=k, auxk_alpha=aux)
    eng.set_topk_options(thr, T)
    eng.set_params({kk: v.numpy() for kk, v in P.items()})
    st, nfsf = O.OptState(), torch.zeros(n, dtype=torch.long)
    xd = x.cuda()
    for i in range(3):
        dead = nfsf > thr
        eng.step(xd, 1e-4)
        m = eng.metrics()
        out = O.topk_train_step(x, P, st, k=k, lr=1e-4, clip_thresh=1.0, dead_mask=dead, auxk_alpha=aux, optimizer="adam")
        did = torch.zeros(n, dtype=torch.bool)
        did[out["top_indices"].flatten()] = True
        nfsf += B * T
        nfsf[did] = 0
        assert m[5] == pytest.approx(float(dead.float().mean()), abs=3.0 / n)      # dead_pct (ties move a few latents)
        assert m[0] == pytest.approx(out["fvu"].item(), rel=2e-2)
        assert m[1] == pytest.approx(out["auxk_loss"].item(), rel=0.1, abs=1e-7)
        if i >= 1:
            assert m[1] > 0
    # sae_get_topk_state / sae_set_topk_state (resume fidelity): the device counters follow the
    # reference's bookkeeping up to boundary ties, and a second context seeded with them reports the same dead fraction
    got = eng.get_topk_state()
    assert got.dtype == np.int64 and got.shape == (n,)
    assert (got != nfsf.numpy()).sum() <= 32           # boundary ties move a few latents per step
    eng2 = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=B * T, optimizer="adam", k=k, auxk_alpha=aux)
    eng2.set_topk_options(thr, T)
    eng2.set_params(eng.get_params())
    eng2.set_topk_state(got)
    assert np.array_equal(eng2.get_topk_state(), got)
    eng2.forward_backward(xd)
    assert eng2.metrics()[5] == pytest.approx(float((got > thr).mean()), abs=1e-6)
    with pytest.raises(Exception):
        eng2.set_topk_state(got[:-1])
    eng2.close()
    eng.close()


def test_sparse_dacts_matches_dense_ddense():
    """TopK backward: the sparse d pre-activation kernel (gathered dot products on the selected latents, exact fixed-point
    d b_enc) and the dense GEMM + mask it replaces give the same gradients, with AuxK active (dead latents) as well."""
    from freud_amd.engine import SaeEngine
    d, n, k, B, T, aux = 384, 1024, 16, 2, 64, 0.03125
    P, x = _make_case(d, n, k, B, T, 9)
    xd = x.cuda()
    res = []
    for dense in (False, True):
        eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=B * T, optimizer="adam", k=k, auxk_alpha=aux,
                        topk_dense_backward=dense)
        eng.set_topk_options(0.5 * B * T, T)
        eng.set_params({kk: v.numpy() for kk, v in P.items()})
        grads = []
        for i in range(2):                      # step 2 has dead latents -> the aux pass runs
            eng.forward_backward(xd)
            grads.append(eng.debug_read(2, 2 * n * d + n + d))
            m = eng.metrics().copy()
            eng.optimizer_step(1e-4)
        assert m[1] > 0                         # AuxK was active
        res.append(grads)
        eng.close()
    for gs, gd in zip(*res):
        s_, d_ = _split(gs, n, d), _split(gd, n, d)
        for key in KEYS:
            assert _rel(s_[key], d_[key]) < 2e-3, key
    # determinism of the fixed-point bias-gradient accumulation: two sparse runs are bitwise equal
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=B * T, optimizer="adam", k=k, auxk_alpha=aux)
    eng.set_topk_options(0.5 * B * T, T)
    eng.set_params({kk: v.numpy() for kk, v in P.items()})
    eng.forward_backward(xd)
    assert np.array_equal(eng.debug_read(2, 2 * n * d + n + d), res[0][0])
    eng.close()

"""GPU parity of the TopK engine (encoder MFMA GEMM -> radix top-k select -> sparse decode -> FVU / AuxK ->
backward GEMMs -> Adam) against the CPU oracle and the reference-generated golden vectors.

Top-k ties: `pre` is bf16 under autocast, so the k-th and (k+1)-th largest latents of a row are EQUAL in
4-50 % of rows; the reference takes whichever torch.topk's partial sort leaves (no rule: sometimes the lower,
sometimes the higher column), the engine takes the lower column.  Parity is therefore stated in two layers:
(a) against the reference's own outputs (golden fixtures): the multiset of selected VALUES is identical on every row,
    the index sets are identical on every row without a boundary tie, and the numbers agree as far as the tie rows
    allow (losses rtol 3e-2, gradients rel-Frobenius 0.25: a tie row decodes through a different W_dec row);
(b) against the oracle evaluated with the engine's tie rule (oracle stable_ties=True: lowest column first, identical
    to the reference on tie-free rows -- tests/test_oracle.py): index sets identical on EVERY row, losses rtol 3e-3
    at step 1 / 2e-2 along a trajectory, gradients rel-Frobenius 1e-2, also on batches full of ties and at the real
    dictionary sizes."""
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


def _check_selection(eng, fwd, M, n, k, exact=False, approx=False):
    """exact: fwd comes from the oracle with the engine's tie rule -> the index sets must agree on every row.
    approx (large d: the MFMA and the host matmul sum the K = d products in different orders, so a few of the M x n
    pre-activations differ by one bf16 ulp and exact set equality is not defined): the engine's set must be a valid top-k
    of the oracle's pre-activations up to one ulp at the boundary, its values must agree to one ulp, and the sets must be
    identical on all but a few rows."""
    pre = fwd["pre"].reshape(M, n)
    ties = _boundary_ties(pre, k).numpy()
    idx = eng.debug_read(3, M * k).reshape(M, k).astype(np.int64)
    ref = fwd["top_indices"].reshape(M, k).numpy()
    same = (np.sort(idx, 1) == np.sort(ref, 1)).all(1)
    dense = eng.debug_read(0, M * n).reshape(M, n)
    if approx:
        pf = pre.float().numpy()
        assert (np.sort(idx, 1)[:, 1:] != np.sort(idx, 1)[:, :-1]).all() and idx.min() >= 0 and idx.max() < n
        sel = np.zeros((M, n), bool)
        np.put_along_axis(sel, idx, True, 1)
        lo = np.where(sel, pf, np.inf).min(1)            # smallest selected / largest unselected oracle pre-activation
        hi = np.where(sel, -np.inf, pf).max(1)
        assert (lo >= hi * (1 - 2.0 ** -7) - 1e-30).all(), "a selected latent is more than one bf16 ulp below an unselected one"
        got = np.take_along_axis(dense, idx, 1)
        np.testing.assert_allclose(got, np.take_along_axis(pf, idx, 1), rtol=2.0 ** -7, atol=1e-30)
        assert (dense != 0).sum() == (got != 0).sum()
        assert same.mean() > 0.9, f"index sets identical on only {same.mean():.3f} of the rows"
        return ties
    assert same[~ties].all(), "index sets differ on rows without a boundary tie"
    if exact:
        assert same.all(), f"index sets differ on {int((~same).sum())} tie rows (tie rule: lowest column first)"
    got_vals = np.sort(np.take_along_axis(dense, idx, 1), 1)
    ref_vals = np.sort(fwd["top_acts"].float().reshape(M, k).numpy(), 1)
    assert np.array_equal(got_vals, ref_vals), "selected activation values differ"
    return ties


def test_topk_tie_free_reference_fixture_tight(golden_dir):
    """topk_tiefree_d64: a reference-generated trajectory (tests/golden/make_golden.py seed-searches it) on which every row
    of every step keeps a relative gap > 2^-6 between its k-th and (k+1)-th pre-activation.  torch.topk's arbitrary choice
    among ties (topkautoencoder.py:79-81) plays no part there, so the engine is held to the REFERENCE'S OWN numbers at the
    arithmetic tolerance: identical index sets, losses 3e-3 / 2e-2 along the trajectory, raw gradients rel-Frobenius 1e-2
    at the first step and 2e-2 at the last (the other reference fixtures allow 0.25 for their tie rows), final weights 2e-3."""
    from freud_amd.engine import SaeEngine
    z = np.load(os.path.join(golden_dir, "topk_tiefree_d64.npz"))
    meta = json.loads(str(z["meta"]))
    d, n, k, B, T, steps = meta["d"], meta["n"], meta["k"], meta["B"], meta["T"], meta["steps"]
    M = B * T
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=meta["auxk_alpha"])
    eng.set_topk_options(meta["dead_feature_threshold"], T)
    eng.set_params({kk: z["init__" + kk] for kk in KEYS})
    xd = torch.tensor(z["x"]).cuda()
    for i in range(steps):
        lr = O.lr_at(i, meta["lr"], "linear", steps, meta["num_warmup_steps"])
        eng.forward_backward(xd[i])
        for tag, ii in (("first", 0), ("last", steps - 1)):
            if i == ii:
                idx = np.sort(eng.debug_read(3, M * k).reshape(M, k).astype(np.int64), 1)
                assert np.array_equal(idx, np.sort(z[f"{tag}__top_indices"].reshape(M, k), 1)), tag
                g = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
                for kk in KEYS:
                    assert _rel(g[kk], z[f"{tag}__{kk}"]) < (1e-2 if tag == "first" else 2e-2), (tag, kk)
        eng.optimizer_step(lr)
        m = eng.metrics()
        assert m[0] == pytest.approx(float(z["fvu"][i]), rel=3e-3 if i == 0 else 2e-2)
        assert m[2] == pytest.approx(float(z["mse"][i]), rel=3e-3 if i == 0 else 2e-2)
        assert m[3] == pytest.approx(float(z["gnorm"][i]), rel=2e-2)
    p = eng.get_params()
    for kk in KEYS:
        assert _rel(p[kk], z["final__" + kk]) < (2e-3 if kk in ("encoder.weight", "W_dec") else 5e-2), kk
    assert np.array_equal(eng.get_topk_state(), z["nfsf_final"])
    eng.close()


@pytest.mark.parametrize("name", ["topk_adam_linear_d16", "topk_adam_linear_d64", "topk_multi_d32"])
def test_topk_steps_match_reference_golden(golden_dir, name):
    from freud_amd.engine import SaeEngine
    z = np.load(os.path.join(golden_dir, f"{name}.npz"))
    meta = json.loads(str(z["meta"]))
    d, n, k, B, T = meta["d"], meta["n"], meta["k"], meta["B"], meta["T"]
    M = B * T
    multi = bool(meta.get("multi_topk", False))      # topkautoencoder.py:134-140; the fixture's top_indices are the 4k set
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k,
                    auxk_alpha=meta["auxk_alpha"], clip_thresh=1.0, multi_topk=multi)
    eng.set_topk_options(meta["dead_feature_threshold"], T)
    eng.set_params({kk: z["init__" + kk] for kk in KEYS})
    P0 = {kk: torch.tensor(z["init__" + kk]) for kk in KEYS}
    Po = {kk: v.clone() for kk, v in P0.items()}          # oracle with the engine's tie rule, stepped alongside
    st, nfsf = O.OptState(), torch.zeros(n, dtype=torch.long)
    xs = torch.tensor(z["x"])
    xd = xs.cuda()
    kf = 4 * k if multi else k
    for i in range(meta["steps"]):
        lr = O.lr_at(i, meta["lr"], "linear", meta["steps"], meta["num_warmup_steps"])
        eng.forward_backward(xd[i])
        if i == 0:
            f = O.topk_forward(xs[0], P0["encoder.weight"], P0["encoder.bias"], P0["W_dec"], P0["b_dec"], k, multi_topk=multi)
            assert torch.equal(f["fire_indices"].reshape(M, kf).sort(1).values,
                               torch.tensor(z["first__top_indices"]).reshape(M, kf).sort(1).values)   # oracle == reference
            _check_selection(eng, f, M, n, k)
            g = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
            for kk in KEYS:
                assert _rel(g[kk], z["first__" + kk]) < 0.25, kk
        dead = nfsf > meta["dead_feature_threshold"]
        out = O.topk_train_step(xs[i], Po, st, k=k, lr=lr, clip_thresh=1.0, dead_mask=dead, auxk_alpha=meta["auxk_alpha"],
                                optimizer="adam", multi_topk=multi, stable_ties=True)
        did = torch.zeros(n, dtype=torch.bool)
        did[out["fire_indices"].flatten()] = True
        nfsf += M
        nfsf[did] = 0
        if i == 0:
            g = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
            for kk in KEYS:
                assert _rel(g[kk], out["grads"][kk].numpy()) < 1e-2, kk
        eng.optimizer_step(lr)
        m = eng.metrics()
        # (b) the oracle under the engine's tie rule: arithmetic tolerance
        tol = 3e-3 if i == 0 else 2e-2
        assert m[0] == pytest.approx(out["fvu"].item(), rel=tol)
        # (the AuxK term rests on a handful of dead latents here and drifts with the trajectory: fp32 vs bf16-rounded
        # weight gradients under Adam)
        assert m[1] == pytest.approx(out["auxk_loss"].item(), rel=0.1, abs=1e-7)
        assert m[3] == pytest.approx(out["grad_norm"].item(), rel=2e-2)
        assert m[5] == pytest.approx(float(dead.float().mean()), abs=1e-7)
        assert m[6] == (pytest.approx(out["multi_topk_fvu"].item(), rel=tol) if multi else 0.0)
        # (a) the reference's own numbers: as far as its arbitrary choice among tied latents allows
        assert m[0] == pytest.approx(float(z["fvu"][i]), rel=3e-2)
        assert m[1] == pytest.approx(float(z["auxk"][i]), rel=0.1, abs=1e-6)
        assert m[5] == pytest.approx(float(z["num_dead"][i]) / n, abs=2.0 / n)
        if multi:
            assert m[6] == pytest.approx(float(z["multi"][i]), rel=3e-2)
    p = eng.get_params()
    for kk in KEYS:
        assert _rel(p[kk], Po[kk].numpy()) < (2e-3 if kk in ("encoder.weight", "W_dec") else 5e-2), kk
        # Adam turns a tie row's different gradient into a full-size step on the few elements it touches
        assert _rel(p[kk], z["final__" + kk]) < (5e-3 if kk in ("encoder.weight", "W_dec") else 0.15), kk
    assert np.array_equal(eng.get_topk_state(), nfsf.numpy())
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
        out = O.topk_train_step(x, P, st, k=k, lr=1e-4, clip_thresh=1.0, dead_mask=dead, auxk_alpha=aux, optimizer="adam",
                                stable_ties=True)
        did = torch.zeros(n, dtype=torch.bool)
        did[out["top_indices"].flatten()] = True
        nfsf += B * T
        nfsf[did] = 0
        assert m[5] == pytest.approx(float(dead.float().mean()), abs=3.0 / n)      # dead_pct
        assert m[0] == pytest.approx(out["fvu"].item(), rel=2e-2)
        assert m[1] == pytest.approx(out["auxk_loss"].item(), rel=5e-2, abs=1e-7)
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
    """TopK backward, three implementations of the same gradients, with AuxK active (dead latents) as well: 0 = the CSC
    sparse backward (selection sorted by latent, gathered row sums: topk_sparse.h), 2 = sparse d pre-activations (gathered
    dot products, exact fixed-point d b_enc) + dense weight-gradient GEMMs, 1 = everything as dense GEMMs + mask."""
    from freud_amd.engine import SaeEngine
    d, n, k, B, T, aux = 384, 1024, 16, 2, 64, 0.03125
    P, x = _make_case(d, n, k, B, T, 9)
    xd = x.cuda()
    res = []
    for dense in (0, 1, 2):
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
    for other in (0, 2):
        for gs, gd in zip(res[other], res[1]):
            s_, d_ = _split(gs, n, d), _split(gd, n, d)
            for key in KEYS:
                assert _rel(s_[key], d_[key]) < 2e-3, (other, key)
    # determinism of the fixed-point bias-gradient accumulation: two sparse runs are bitwise equal
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=B * T, optimizer="adam", k=k, auxk_alpha=aux)
    eng.set_topk_options(0.5 * B * T, T)
    eng.set_params({kk: v.numpy() for kk, v in P.items()})
    eng.forward_backward(xd)
    assert np.array_equal(eng.debug_read(2, 2 * n * d + n + d), res[0][0])
    eng.close()


@pytest.mark.parametrize("d,n,n_dead", [(768, 2048, 100), (768, 2048, 700), (384, 1024, 1024), (1280, 1536, 900),
                                         (384, 1152, 1152),       # n_p = 9 x 128: the compact width must not pass n_p
                                         # dictionaries above 24 576 latents (round 5): the dead columns are COPIED into compact rows and
                                         # the general select runs in place on them (topk_kernels.h: compact_mode) --
                                         (384, 32768, 100),       # ... fewer dead latents than k_aux = 192: the copy is the selection
                                         (384, 32768, 3000),      # ... a real selection among 3000 dead latents: <12> on the compact rows
                                         (384, 40960, 28000)])    # ... more than 24 576 dead latents: <44> on the compact rows
def test_auxk_compact_dead_set_matches_gather_path_and_oracle(d, n, n_dead):
    """AuxK as dense GEMMs over the compacted dead latents (topk_aux.h, the default) against the gather kernels it replaces
    (debug_flags 76 switches the compaction off) and against the oracle under the engine's tie rule: fewer dead latents
    than k_aux = d/2 (everything dead is taken), more (a real selection), and ALL latents dead.  Some dead latents are
    also picked by the main selection of this step (their rows receive both gradients)."""
    from freud_amd.engine import SaeEngine
    k, B, T, aux = 16, 2, 96, 0.03125
    P, x = _make_case(d, n, k, B, T, 11)
    M = B * T
    thr = 100.0
    g = torch.Generator().manual_seed(2)
    nfsf = torch.zeros(n, dtype=torch.long)
    nfsf[torch.randperm(n, generator=g)[:n_dead]] = 1000
    dead = nfsf > thr
    out = []
    for flags in (0, 76):
        eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=aux, debug_flags=flags)
        eng.set_topk_options(thr, T)
        eng.set_params({kk: v.numpy() for kk, v in P.items()})
        eng.set_topk_state(nfsf.numpy())
        eng.forward_backward(x.cuda())
        graw = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
        eng.optimizer_step(1e-4)
        out.append((graw, eng.metrics().copy(), eng.get_topk_state()))
        eng.close()
    (g0, m0, s0), (g1, m1, s1) = out
    assert m0[1] > 0 and m0[5] == pytest.approx(n_dead / n, abs=1e-7)
    assert m0[1] == pytest.approx(m1[1], rel=2e-3) and m0[0] == pytest.approx(m1[0], rel=1e-5)
    assert np.array_equal(s0, s1)
    for kk in KEYS:
        assert _rel(g0[kk], g1[kk]) < 2e-3, kk
    ref = O.topk_train_step(x, P, O.OptState(), k=k, lr=1e-4, clip_thresh=1.0, dead_mask=dead, auxk_alpha=aux, optimizer="adam",
                            stable_ties=True)
    assert m0[0] == pytest.approx(ref["fvu"].item(), rel=5e-3)
    assert m0[1] == pytest.approx(ref["auxk_loss"].item(), rel=2e-2)
    assert m0[3] == pytest.approx(ref["grad_norm"].item(), rel=2e-2)
    for kk in KEYS:
        assert _rel(g0[kk], ref["grads"][kk].numpy()) < 2e-2, kk


def _tie_free_case(d, n, k, B, T, seeds=200, k_also=None):
    for seed in range(seeds):
        P, x = _make_case(d, n, k, B, T, seed)
        f = O.topk_forward(x, P["encoder.weight"], P["encoder.bias"], P["W_dec"], P["b_dec"], k)
        pre = f["pre"].reshape(B * T, n)
        if not _boundary_ties(pre, k).any() and (k_also is None or not _boundary_ties(pre, k_also).any()):
            return P, x, f
    pytest.skip("no tie-free batch found")


def test_topk_auxk_tie_free_matches_oracle():
    """AuxK arithmetic at a tight tolerance: a batch without boundary ties in the main selection and FEWER dead latents
    than k_aux = d/2 (so the aux selection takes every dead latent: no tie possible there either).  AuxK loss to rel 2e-2,
    raw gradients to rel-Frobenius 1e-2, dead_pct exact."""
    from freud_amd.engine import SaeEngine
    d, n, k, B, T, aux = 384, 1024, 8, 2, 6, 0.03125
    P, x, _ = _tie_free_case(d, n, k, B, T)
    M = B * T
    thr = 100.0
    g = torch.Generator().manual_seed(1)
    dead_idx = torch.randperm(n, generator=g)[:120]             # 120 < d/2 = 192
    nfsf = torch.zeros(n, dtype=torch.long)
    nfsf[dead_idx] = 1000
    dead = nfsf > thr
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=aux)
    eng.set_topk_options(thr, T)
    eng.set_params({kk: v.numpy() for kk, v in P.items()})
    eng.set_topk_state(nfsf.numpy())
    eng.forward_backward(x.cuda())
    graw = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
    eng.optimizer_step(1e-4)
    m = eng.metrics()
    st = O.OptState()
    out = O.topk_train_step(x, P, st, k=k, lr=1e-4, clip_thresh=1.0, dead_mask=dead, auxk_alpha=aux, optimizer="adam")
    assert out["auxk_loss"].item() > 0
    assert m[5] == pytest.approx(120.0 / n, abs=1e-7)
    assert m[0] == pytest.approx(out["fvu"].item(), rel=2e-3)
    assert m[1] == pytest.approx(out["auxk_loss"].item(), rel=2e-2)
    assert m[3] == pytest.approx(out["grad_norm"].item(), rel=1e-2)
    for kk in KEYS:
        assert _rel(graw[kk], out["grads"][kk].numpy()) < 1e-2, kk
    eng.close()


def test_topk_multi_topk_sparse_path_matches_oracle():
    """cfg.multi_topk on the sparse-backward path (d_p = 384): the 4k selection, its decode, multi_topk_fvu, the
    loss / 8 gradient and did_fire from the 4k set (train_sae.py:442-446).  No seed gives a batch without a boundary
    tie at BOTH k and 4k (bf16 pre-activations: searched 0..199), so the comparison is against the oracle under the
    engine's tie rule (lowest column first), like the neighbouring tests."""
    from freud_amd.engine import SaeEngine
    d, n, k, B, T = 384, 1024, 8, 2, 6
    P, x = _make_case(d, n, k, B, T, 11)
    M = B * T
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=0.0, multi_topk=True)
    eng.set_topk_options(1e9, T)
    eng.set_params({kk: v.numpy() for kk, v in P.items()})
    eng.forward_backward(x.cuda())
    graw = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
    eng.optimizer_step(1e-4)
    m = eng.metrics()
    st = O.OptState()
    out = O.topk_train_step(x, P, st, k=k, lr=1e-4, clip_thresh=1.0, dead_mask=None, auxk_alpha=0.0, optimizer="adam",
                            multi_topk=True, stable_ties=True)
    assert m[0] == pytest.approx(out["fvu"].item(), rel=2e-3)
    assert m[6] == pytest.approx(out["multi_topk_fvu"].item(), rel=2e-3)
    assert m[3] == pytest.approx(out["grad_norm"].item(), rel=1e-2)
    for kk in KEYS:
        assert _rel(graw[kk], out["grads"][kk].numpy()) < 1e-2, kk
    fired = np.zeros(n, bool)
    fired[out["fire_indices"].numpy().ravel()] = True
    got = eng.get_topk_state()                      # counters: 0 where the 4k set fired, M elsewhere
    assert np.array_equal(got == 0, fired)
    assert set(np.unique(got)) <= {0, M}
    eng.close()


@pytest.mark.parametrize("d,n,k,B,T,kernel", [
    (768, 24576, 64, 2, 256, "reg12"),      # BASELINE configs[2] at its real n and k (register select, 12 vectors / thread)
    (384, 32768, 32, 2, 64, "reg44"),       # n_p > 24 576: topk_select_reg_kernel<44>
    (384, 98304, 32, 1, 64, "generic"),     # n_p > 90 112: the radix-select fallback (topk_select_kernel)
    (1280, 40960, 32, 1, 64, "csc2"),       # large-v3 with the reference's default expansion 32 (config.py:7): n_p > 32 768, the
                                            # sparse backward's counting sort runs in two dictionary segments
])
def test_topk_real_dictionary_sizes_match_oracle(d, n, k, B, T, kernel):
    """The configs[2] shape at full n / k against the oracle (M = 512 rows), and the two select kernels that only large
    dictionaries reach.  Ties at the k-th value are common with bf16 pre-activations (half of the rows here): against the
    oracle under the engine's tie rule the index sets are identical on every row, losses agree to rtol 5e-3 and gradients
    to rel-Frobenius 1e-2."""
    from freud_amd.engine import SaeEngine
    P, x = _make_case(d, n, k, B, T, 3)
    M = B * T
    f = O.topk_forward(x, P["encoder.weight"], P["encoder.bias"], P["W_dec"], P["b_dec"], k, stable_ties=True)
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=0.0)
    eng.set_topk_options(1e12, T)
    eng.set_params({kk: v.numpy() for kk, v in P.items()})
    eng.forward_backward(x.cuda())
    _check_selection(eng, f, M, n, k, exact=d <= 384, approx=d > 384)
    graw = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
    eng.optimizer_step(1e-4)
    m = eng.metrics()
    st = O.OptState()
    out = O.topk_train_step(x, P, st, k=k, lr=1e-4, clip_thresh=1.0, dead_mask=None, auxk_alpha=0.0, optimizer="adam",
                            stable_ties=True)
    assert m[0] == pytest.approx(out["fvu"].item(), rel=5e-3)
    assert m[2] == pytest.approx(out["mse"].item(), rel=5e-3)
    assert m[3] == pytest.approx(out["grad_norm"].item(), rel=1e-2)
    for kk in KEYS:
        assert _rel(graw[kk], out["grads"][kk].numpy()) < 1e-2, kk
    eng.close()


def test_topk_c3_full_size_properties():
    """BASELINE configs[2] at full size (d=768, n=24 576, k=64, M=65 536): size-independent properties -- exactly k
    non-zero latents per row (or fewer only where a row has fewer positive pre-activations), did_fire consistent with the
    index lists, two identical runs bitwise equal, the loss decreases."""
    from freud_amd.engine import SaeEngine
    d, n, k, B, T = 768, 24576, 64, 64, 1024
    M = B * T
    g = torch.Generator().manual_seed(0)
    We = (torch.rand(n, d, generator=g) * 2 - 1) / d ** 0.5
    Wd = We / (We.norm(dim=1, keepdim=True) + torch.finfo(torch.float32).eps)
    x = ((torch.relu(torch.randn(M, 64, generator=g)) * 0.1) @ torch.randn(64, d, generator=g)).to(torch.bfloat16).reshape(B, T, d).cuda()
    runs = []
    for _ in range(2):
        eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=0.03125)
        eng.set_topk_options(1e6, T)
        eng.set_params({"encoder.weight": We.numpy(), "encoder.bias": np.zeros(n, np.float32), "W_dec": Wd.numpy(),
                        "b_dec": np.zeros(d, np.float32)})
        fv = []
        for i in range(4):
            eng.step(x, 1e-4)
            fv.append(float(eng.metrics()[0]))
        idx = eng.topk_indices_tensor(M, "cuda:0")
        ptr, ld = eng.latent_buffer()

        class _Alias:
            __cuda_array_interface__ = {"shape": (M, ld), "typestr": "<i2", "data": (ptr, False), "version": 2}

        dense = torch.as_tensor(_Alias(), device="cuda:0").view(torch.bfloat16)
        nnz = (dense != 0).sum(1)
        assert int(nnz.max()) <= k
        assert ((idx >= 0).sum(1) == k).all()                     # k indices per row, all distinct and in range
        assert int(idx.max()) < n
        srt = idx.sort(1).values
        assert (srt[:, 1:] != srt[:, :-1]).all()
        vals = torch.gather(dense, 1, idx.long())
        assert torch.equal((vals != 0).sum(1), nnz)               # every non-zero of the dense row is a listed index
        fired = torch.zeros(n, dtype=torch.bool, device="cuda:0")
        fired[idx.long().flatten()] = True
        nf = torch.from_numpy(eng.get_topk_state()).cuda()
        assert torch.equal(nf == 0, fired)
        p = eng.get_params()
        runs.append((fv, p["W_dec"].copy(), p["encoder.bias"].copy()))
        eng.close()
    assert runs[0][0] == runs[1][0]
    assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    assert np.isfinite(runs[0][0]).all() and runs[0][0][-1] < runs[0][0][0]



def test_topk_auxk_runs_are_bitwise_reproducible():
    """Two identical runs with a dead set that appears, changes size and crosses k_aux (AuxK through its copy path, the
    full selection and the dynamic-width GEMMs, whose launch sizes follow a stale host-side estimate): every quantity that
    enters the arithmetic is decided on the device, so parameters and metrics are bitwise equal (tools/longrun_topk.py is
    the long form)."""
    from freud_amd.engine import SaeEngine
    M, d, n, k, T = 4096, 768, 4096, 16, 512
    g = torch.Generator().manual_seed(0)
    We = (torch.rand(n, d, generator=g) * 2 - 1) / d ** 0.5
    Wd = We / (We.norm(dim=1, keepdim=True) + torch.finfo(torch.float32).eps)
    xs = [((torch.relu(torch.randn(M, 48, generator=g)) * 0.1) @ torch.randn(48, d, generator=g)).to(torch.bfloat16)
          .reshape(M // T, T, d).cuda() for _ in range(3)]
    runs = []
    for _ in range(2):
        eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=0.03125)
        eng.set_topk_options(2.0 * M, T)
        eng.set_params({"encoder.weight": We.numpy(), "encoder.bias": np.zeros(n, np.float32), "W_dec": Wd.numpy(),
                        "b_dec": np.zeros(d, np.float32)})
        hist = []
        for i in range(40):
            eng.step(xs[i % 3], 3e-4)
            if i % 5 == 4:
                hist.append(eng.metrics().copy())
        runs.append((np.stack(hist), eng.get_params()))
        eng.close()
    (h0, p0), (h1, p1) = runs
    assert h0[:, 5].max() > 0 and h0[:, 1].max() > 0            # latents died and AuxK ran
    assert np.array_equal(h0, h1)
    for key in p0:
        assert np.array_equal(p0[key], p1[key]), key


@pytest.mark.parametrize("n_dead", [100, 400])
def test_topk_multi_topk_with_auxk_matches_oracle(n_dead):
    """cfg.multi_topk AND dead latents in one step: the 4k pass and the main pass through the CSC backward, the AuxK pass over
    the compacted dead set (copy path at 100 dead <= d/2, full selection at 400), did_fire from the 4k set -- against the
    oracle under the engine's tie rule."""
    from freud_amd.engine import SaeEngine
    d, n, k, B, T, aux = 384, 1024, 8, 2, 96, 0.03125
    P, x = _make_case(d, n, k, B, T, 21)
    M = B * T
    thr = 100.0
    g = torch.Generator().manual_seed(5)
    nfsf = torch.zeros(n, dtype=torch.long)
    nfsf[torch.randperm(n, generator=g)[:n_dead]] = 1000
    dead = nfsf > thr
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=aux, multi_topk=True)
    eng.set_topk_options(thr, T)
    eng.set_params({kk: v.numpy() for kk, v in P.items()})
    eng.set_topk_state(nfsf.numpy())
    eng.forward_backward(x.cuda())
    graw = _split(eng.debug_read(2, 2 * n * d + n + d), n, d)
    eng.optimizer_step(1e-4)
    m = eng.metrics()
    ref = O.topk_train_step(x, P, O.OptState(), k=k, lr=1e-4, clip_thresh=1.0, dead_mask=dead, auxk_alpha=aux, optimizer="adam",
                            multi_topk=True, stable_ties=True)
    assert m[1] > 0 and m[6] > 0
    assert m[0] == pytest.approx(ref["fvu"].item(), rel=5e-3)
    assert m[1] == pytest.approx(ref["auxk_loss"].item(), rel=2e-2)
    assert m[6] == pytest.approx(ref["multi_topk_fvu"].item(), rel=5e-3)
    assert m[3] == pytest.approx(ref["grad_norm"].item(), rel=2e-2)
    for kk in KEYS:
        assert _rel(graw[kk], ref["grads"][kk].numpy()) < 2e-2, kk
    fired = np.zeros(n, bool)
    fired[ref["fire_indices"].numpy().ravel()] = True
    got = eng.get_topk_state()
    assert np.mean((got == 0) != fired) < 0.01           # boundary ties in the 4k selection move a few latents
    eng.close()


@pytest.mark.parametrize("B", [5, 40, 70])
def test_total_variance_forms_match_oracle_on_bf16_batches(B):
    """FVU = sum e^2 / total variance, the variance over the FILES of a batch (topkautoencoder.py:104-106).  bf16 activations take the
    one-pass register kernels (up to 32 files: eight columns per thread; up to 64: four), more files the per-column kernel: all three
    against the oracle on the same bf16-rounded batch."""
    from freud_amd.engine import SaeEngine
    d, n, k, T = 384, 1024, 8, 4
    P, x = _make_case(d, n, k, B, T, seed=3)
    xb = x.to(torch.bfloat16)
    M = B * T
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=0.0)
    eng.set_topk_options(1e9, T)
    eng.set_params({kk: v.numpy() for kk, v in P.items()})
    eng.forward_backward(xb.cuda())
    eng.optimizer_step(1e-4)
    m = eng.metrics()
    out = O.topk_train_step(xb.float(), P, O.OptState(), k=k, lr=1e-4, clip_thresh=1.0, dead_mask=None, auxk_alpha=0.0, optimizer="adam")
    tv = ((xb.float() - xb.float().mean(0)) ** 2).sum().item()
    assert np.isfinite(m[0]) and tv > 0
    assert m[0] == pytest.approx(out["fvu"].item(), rel=3e-3)
    assert m[2] == pytest.approx(out["mse"].item(), rel=3e-3)
    eng.close()

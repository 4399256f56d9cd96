"""GPU: data-parallel exactness of the HIP path on ONE GPU.  Two engine contexts play two ranks: each gets half of the
rows, the batch statistics (sae_batch_stats) are summed by the test as an all-reduce would, each context's backward
normalises by the global values (sae_set_dp_world), and the SUM of the two gradient buffers must equal the gradient of
one context that sees the whole batch -- with masked entries spread unevenly over the halves (different unmasked counts)
and, for TopK, total_variance around the mean over ALL files.  Also: the statistics kernels against the oracle's, and the
engine's own RCCL communicator (sae_dist_init) with one rank against the plain single-GPU step."""
import numpy as np
import pytest
import torch

from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("d,n,M,generic", [(384, 1024, 2048, False), (384, 1024, 2048, True), (768, 1024, 1024, False)])
def test_l1_two_halves_sum_to_whole_batch(d, n, M, generic):
    from freud_amd.engine import SaeEngine
    g = torch.Generator().manual_seed(d + M)
    W = torch.empty(d, n)
    torch.nn.init.orthogonal_(W, generator=g)
    b = 0.01 * torch.randn(n, generator=g)
    x = ((torch.relu(torch.randn(M, 64, generator=g)) * 0.1) @ torch.randn(64, d, generator=g)).to(torch.bfloat16)
    x[: M // 8] = torch.where(torch.rand(M // 8, d, generator=g) < 0.5, torch.tensor(-1.0, dtype=torch.bfloat16), x[: M // 8])
    xd = x.cuda()
    halves = [xd[: M // 2].contiguous(), xd[M // 2:].contiguous()]

    def make(rows):
        e = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=rows, optimizer="adam", recon_alpha=1e4, force_generic=generic)
        e.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
        return e

    whole = make(M)
    whole.batch_stats(xd)
    st = whole.stats_tensor().cpu()
    assert torch.equal(st, O.l1_batch_stats(x))                 # (unmasked entries, rows): exact integers
    whole.forward_backward(xd)
    torch.cuda.synchronize()
    ref = whole.grad_tensor().cpu().numpy().copy()
    ranks = [make(M // 2) for _ in halves]
    for e, h in zip(ranks, halves):
        e.batch_stats(h)
    total = ranks[0].stats_tensor().clone() + ranks[1].stats_tensor()
    assert total[0].item() != 2 * ranks[0].stats_tensor()[0].item()     # the halves really have different counts
    acc = None
    for e, h in zip(ranks, halves):
        e.stats_tensor().copy_(total)                            # "all-reduce"
        e.set_dp_world(2)
        e.forward_backward(h)
        torch.cuda.synchronize()
        gbuf = e.grad_tensor().cpu().numpy().copy()
        acc = gbuf if acc is None else acc + gbuf
    nW = acc.size - 8
    assert _rel(acc[:nW], ref[:nW]) < 2e-5                       # gradients: fp32 summation order only
    np.testing.assert_allclose(acc[nW:nW + 3], ref[nW:nW + 3], rtol=2e-5)       # loss shares sum to the whole batch's losses
    assert acc[nW + 4] == ref[nW + 4]                            # unmasked count
    # optimizer on the summed gradient == optimizer of the whole batch
    for e in ranks:
        e.grad_tensor().copy_(torch.from_numpy(acc).cuda())
        e.optimizer_step(1e-3, 1.0)
    whole.optimizer_step(1e-3, 1.0)
    assert _rel(ranks[0].get_params()["decoder.weight"], whole.get_params()["decoder.weight"]) < 1e-6
    assert np.array_equal(ranks[0].get_params()["decoder.weight"], ranks[1].get_params()["decoder.weight"])
    for e in ranks + [whole]:
        e.close()


def test_topk_two_halves_sum_to_whole_batch():
    from freud_amd.engine import SaeEngine
    d, n, k, B, T = 384, 1024, 8, 4, 64
    g = torch.Generator().manual_seed(5)
    We = torch.randn(n, d, generator=g) / d ** 0.5
    Wd = We / (We.norm(dim=1, keepdim=True) + torch.finfo(torch.float32).eps)
    P = {"encoder.weight": We, "encoder.bias": torch.zeros(n), "W_dec": Wd, "b_dec": 0.01 * torch.randn(d, generator=g)}
    x = (torch.relu(torch.randn(B * T, 48, generator=g)) @ torch.randn(48, d, generator=g) * 0.2).reshape(B, T, d)
    x[2:] += 0.3                                               # the halves have different means: a local x.mean(0) would differ
    xd = x.cuda()
    halves = [xd[:2].contiguous(), xd[2:].contiguous()]
    nfsf = np.zeros(n, np.int64)
    nfsf[::7] = 10 ** 6                                        # some dead latents: the AuxK branch runs

    def make(rows):
        e = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=rows, optimizer="adam", k=k, auxk_alpha=0.03125)
        e.set_topk_options(1000.0, T)
        e.set_params({kk: v.numpy() for kk, v in P.items()})
        e.set_topk_state(nfsf)
        return e

    whole = make(B * T)
    whole.batch_stats(xd)
    np.testing.assert_allclose(whole.stats_tensor().cpu().numpy(), O.topk_batch_stats(x).numpy(), rtol=1e-12, atol=1e-12)
    whole.forward_backward(xd)
    torch.cuda.synchronize()
    ref = whole.grad_tensor().cpu().numpy().copy()
    ranks = [make(2 * T) for _ in halves]
    for e, h in zip(ranks, halves):
        e.batch_stats(h)
    total = ranks[0].stats_tensor().clone() + ranks[1].stats_tensor()
    acc = None
    for e, h in zip(ranks, halves):
        e.stats_tensor().copy_(total)
        e.set_dp_world(2)
        e.forward_backward(h)
        torch.cuda.synchronize()
        gbuf = e.grad_tensor().cpu().numpy().copy()
        acc = gbuf if acc is None else acc + gbuf
    np_ = 2 * n * d + n + d
    assert ref[np_ + 1] > 0                                       # AuxK active
    # (per-rank bf16 roundings of de = 2 e / tv and of the decoder-gradient operands differ from the whole batch's only
    # through the fp32 value of tv: identical here, so the gradients agree to summation order)
    assert _rel(acc[:np_], ref[:np_]) < 5e-4
    np.testing.assert_allclose(acc[np_:np_ + 3], ref[np_:np_ + 3], rtol=1e-4)      # fvu, auxk, mse shares sum up
    assert acc[np_ + 5] == pytest.approx(ref[np_ + 5], rel=1e-6)                   # dead_pct: 1/world from each rank
    assert np.array_equal(acc[np_ + 8:] > 0, ref[np_ + 8:] > 0)                    # did_fire: OR == sum > 0
    for e in ranks:
        e.grad_tensor().copy_(torch.from_numpy(acc).cuda())
        e.optimizer_step(1e-4, 1.0)
    whole.optimizer_step(1e-4, 1.0)
    assert np.array_equal(ranks[0].get_topk_state(), whole.get_topk_state())       # frames of ALL ranks are counted
    for e in ranks + [whole]:
        e.close()


_DIST_CHILD = r"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.environ["FREUD_ROOT"])
from freud_amd.engine import SaeEngine
variant = sys.argv[1]
carrier = sys.argv[2] if len(sys.argv) > 2 else "rccl"
torch.cuda.set_device(0)
g = torch.Generator().manual_seed(1)
d, n, M = {"l1": (384, 1024, 2048), "l1_d1280": (1280, 1024, 1024), "l1_d1280_wide": (1280, 16384, 1024),
           "topk": (768, 2048, 1024)}[variant]
x = ((torch.relu(torch.randn(M, 64, generator=g)) * 0.1) @ torch.randn(64, d, generator=g)).to(torch.bfloat16).cuda()
outs = []
for dist_mode in (False, True):
    if variant.startswith("l1"):
        eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=1e4)
        W = torch.empty(d, n); torch.nn.init.orthogonal_(W, generator=torch.Generator().manual_seed(2))
        eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
        key = "decoder.weight"
    else:
        eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=16, auxk_alpha=0.03125)
        eng.set_topk_options(1e9, M)
        We = (torch.rand(n, d, generator=torch.Generator().manual_seed(2)) * 2 - 1) / d ** 0.5
        eng.set_params({"encoder.weight": We.numpy(), "encoder.bias": np.zeros(n, np.float32),
                        "W_dec": (We / We.norm(dim=1, keepdim=True)).numpy(), "b_dec": np.zeros(d, np.float32)})
        key = "W_dec"
    if dist_mode and carrier == "rccl":
        eng.dist_init(SaeEngine.dist_unique_id(), 0, 1)
        assert eng.dist_world() == 1
    elif dist_mode:            # the peer exchange with one rank: export, "map the peers" (nobody), self-test, in-engine protocol
        eng.p2p_init([eng.p2p_export()], 0, 1)
        assert eng.dist_world() == 1
    for i in range(4):
        eng.step(x, 1e-3)
    torch.cuda.synchronize()
    outs.append((eng.get_params()[key].copy(), eng.metrics().copy()))
    eng.close()
rel = float(np.linalg.norm(outs[0][0] - outs[1][0]) / np.linalg.norm(outs[0][0]))
print(json.dumps({"rel": rel, "m0": outs[0][1].tolist(), "m1": outs[1][1].tolist()}))
"""


@pytest.mark.parametrize("variant,carrier", [("l1", "rccl"), ("l1_d1280", "rccl"), ("topk", "rccl"),
                                              ("l1_d1280_wide", "rccl"), ("l1_d1280_wide", "p2p")])
def test_in_engine_rccl_single_rank_equals_plain_step(tmp_path, variant, carrier):
    """sae_dist_init with a communicator of ONE rank: statistics all-reduce on the communication stream, gradient ranges
    all-reduced as they become final, stream joins -- the whole in-engine protocol -- must reproduce the plain step
    (global statistics == local ones).  Child process: the communicator must not leak into the other tests.
    l1_d1280: the generic three-GEMM path, whose weight gradient travels as COLUMN chunks through a contiguous staging buffer
    (summed there by RCCL, copied back into the strided block of the gradient buffer).  l1_d1280_wide (n = 16 384: 64 tile
    columns): the ROUND-sized chunks -- 51 tile columns as one round of whole tiles written straight into the gradient (p2p) or
    the staging block (RCCL), the 13 left over as K pieces through the overflow buffer -- on both in-engine carriers."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(str(tmp_path), "child.py")
    open(script, "w").write(_DIST_CHILD)
    env = dict(os.environ, FREUD_ROOT=root, NCCL_DEBUG_FILE="/tmp/rccl_debug_%h_%p.log")
    out = subprocess.run([sys.executable, script, variant, carrier], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["rel"] < 1e-6, res
    np.testing.assert_allclose(res["m1"][:4], res["m0"][:4], rtol=1e-5)


# ------------------------------------------------------------------------------------------------------------------
# Two REAL processes on one GPU: the in-engine protocol with the peer exchange (csrc/p2p_exchange.h) end to end
# ------------------------------------------------------------------------------------------------------------------
_TRAIN_CHILD = r"""
import json, os, sys
sys.path.insert(0, os.environ["FREUD_ROOT"])
from freud_amd.train_sae import train
cfg = json.load(open(sys.argv[1]))
world = int(os.environ.get("WORLD_SIZE", "1"))
state = train(**cfg, dist_backend="gloo" if world > 1 else None)     # gloo: host channel only (two NCCL ranks cannot share a GPU)
print("AUDITS_PASSED", state.get("exchange_audits_passed", -1), flush=True)
if world > 1:
    import torch.distributed as dist
    dist.destroy_process_group()
"""


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_children_once(script, cfg_path, world, extra_env, timeout=420):
    """Start `world` ranks of `script` on GPU 0 and wait for them.  Output goes to files (no pipe to fill up while nobody reads); the
    wait polls ALL ranks: when one exits with an error the others get 30 s to follow and are then stopped, and when the time is up
    every rank still alive is made to dump its Python stacks (faulthandler, SIGABRT) -- a failure reports every rank's tail, because
    the rank that reports is rarely the one that failed, and a hang says where each rank stood."""
    import os
    import signal
    import subprocess
    import sys
    import tempfile
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    logdir = tempfile.mkdtemp(prefix="freud_dp_children_")
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ, FREUD_ROOT=root, PYTHONFAULTHANDLER="1", **extra_env)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        if world > 1:      # all ranks on GPU 0: the exchange runs between processes that share the device
            env.update(RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        fo, fe = open(os.path.join(logdir, f"rank{r}.out"), "w"), open(os.path.join(logdir, f"rank{r}.err"), "w")
        files.append((fo, fe))
        procs.append(subprocess.Popen([sys.executable, script, cfg_path], env=env, stdout=fo, stderr=fe, text=True))
    deadline = time.time() + timeout
    why = None
    try:
        while True:
            rcs = [pr.poll() for pr in procs]
            if all(rc is not None for rc in rcs):
                break
            now = time.time()
            if any(rc not in (None, 0) for rc in rcs) and why is None:
                why = "a rank exited with an error; the others were given 30 s"
                deadline = min(deadline, now + 30)
            if now > deadline:
                why = why or f"still running after {timeout} s"
                for pr in procs:
                    if pr.poll() is None:
                        pr.send_signal(signal.SIGABRT)          # the exact children this test started
                time.sleep(5)
                break
            time.sleep(0.1)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()           # the exact children this test started
        for pr in procs:
            pr.wait()
        for fo, fe in files:
            fo.close()
            fe.close()
    outs = []
    for r in range(world):
        with open(os.path.join(logdir, f"rank{r}.out")) as f1, open(os.path.join(logdir, f"rank{r}.err")) as f2:
            outs.append((f1.read(), f2.read()))
    if why is not None or any(pr.returncode != 0 for pr in procs):
        raise AssertionError((why or "a rank failed") + "\n" +
                             "\n".join(f"--- rank {r}: rc {pr.returncode}\n{so[-600:]}\n{se[-3000:]}" for r, (pr, (so, se)) in enumerate(zip(procs, outs))))
    return outs


# what a start-up failure of the BOX looks like (several processes initialising one GPU / one rendezvous port at once), as opposed to a
# failure of the code under test: such a run is repeated once, with the first attempt's report kept on stderr
_INFRA_PATTERNS = ("Address already in use", "Connection refused", "Connection reset", "hipErrorNoDevice", "hipErrorOutOfMemory",
                   "hipErrorInvalidDevice", "no ROCm-capable device", "HSA_STATUS_ERROR_OUT_OF_RESOURCES", "failed to initialize",
                   "DistNetworkError", "DistStoreError")


def _run_children(script, cfg_path, world, extra_env, timeout=420):
    import sys
    try:
        return _run_children_once(script, cfg_path, world, extra_env, timeout)
    except AssertionError as e:
        if not any(p in str(e) for p in _INFRA_PATTERNS):
            raise
        print("first attempt failed at start-up, repeating once:\n" + str(e)[-4000:], file=sys.stderr)
        return _run_children_once(script, cfg_path, world, extra_env, timeout)


@pytest.mark.parametrize("case,payload,overlap", [
    ("l1_fused", "float32", 1),        # d = 384: fused forward / backward, whole gradient exchanged in line
    ("l1_fused", "bfloat16", 1),       # ... as a bf16 copy (rounded once by the owner of each shard)
    ("l1_fused", "float32", 2),        # ... backward in two column-tile ranges, range 0 exchanged under range 1's backward
    ("l1_generic", "float32", 1),      # d = 1280: three-GEMM backward, dW column chunks (2-D segments) exchanged under the remaining ones
    ("topk", "float32", 1),            # TopK with AuxK: statistics (column sums), did_fire OR, dW_dec before dW_enc
])
def test_two_processes_one_gpu_train_like_one_process(tmp_path, case, payload, overlap):
    _ranks_vs_single_process(tmp_path, case, payload, overlap, world=2)


def test_four_processes_one_gpu_train_like_one_process(tmp_path):
    """The same with FOUR ranks (three peers per exchange workgroup, four shards): L1 fused path, fp32 payload."""
    _ranks_vs_single_process(tmp_path, "l1_fused", "float32", 1, world=4)


@pytest.mark.parametrize("case", ["l1_fused", "l1_generic"])
def test_eight_processes_one_gpu_train_like_one_process(tmp_path, case):
    """EIGHT ranks -- the world the peer exchange is built for and the size of the driver's scaling run -- as eight processes on
    one GPU (8 x <= 48 exchange workgroups are co-resident on 256 CUs): seven peers per exchange workgroup, eight shards, every
    `q < world` index path of p2p_allreduce_kernel (only 2 and 4 ranks had ever executed).  l1_fused: the whole gradient
    in line (segments of 98 304, 256 and TWO vectors: the last one is smaller than the world, so six ranks own an empty shard);
    l1_generic (d = 1280): strided 2-D column blocks on the communication stream.  The host-side model of the same index functions
    for worlds 1-8 is tests/test_p2p_index.py."""
    _ranks_vs_single_process(tmp_path, case, "float32", 1, world=8)


def test_two_processes_coarse_grained_buffers(tmp_path):
    """Multi-rank runs put the peer-read buffers in fine-grained memory by default (train_sae.py); FREUD_P2P_FINEGRAINED=0 keeps
    them coarse-grained -- the form whose cross-device visibility rests on the fences alone -- and must train the same."""
    _ranks_vs_single_process(tmp_path, "l1_fused", "float32", 1, world=2, extra_env={"FREUD_P2P_FINEGRAINED": "0"})


@pytest.mark.parametrize("world", [2, 4])
def test_statistics_on_the_communication_stream_train_the_same(tmp_path, world):
    """FREUD_DP_STATS=stream (round 6): the fused path's batch statistics -- a pass over x and their exchange -- on the communication
    stream under the weight preparation and the forward, instead of pushed from inside the loss finalisation between forward and
    backward: one cross-GPU round trip less on the critical path, the same sums, so R ranks must still train like one process on R
    times the batch (planted -1.0 entries make the exchanged count matter)."""
    _ranks_vs_single_process(tmp_path, "l1_fused", "float32", 1, world=world, extra_env={"FREUD_DP_STATS": "stream"})


def _ranks_vs_single_process(tmp_path, case, payload, overlap, world, extra_env=None):
    """R x B == 1 x RB on REAL kernels with the REAL exchange: two freshly spawned processes (ranks 0 and 1, both on GPU 0)
    run train() with the in-engine protocol over hipIpc peer mappings -- handles through a gloo group, batch statistics
    summed before the backward, every gradient range summed by the engine's exchange kernels, self-test at start-up -- and
    must reproduce a single process that trains on twice the batch: weights / optimizer moments to fp32 summation order
    (bf16 payload: to the bf16 rounding of the summed gradient), logged losses alike.  Files carry very different numbers of
    masked (-1) entries, so per-rank means would NOT average to the whole batch's (l1autoencoder.py:29-36)."""
    import copy
    import json
    import os
    from freud_amd.loader import write_shards
    d, n, T = {"l1_fused": (384, 1024, 64), "l1_generic": (1280, 512, 32), "topk": (384, 1024, 32)}[case]
    n_files, B, steps = (16 if world <= 4 else 32), (2 if world == 2 else 1), 4
    g = torch.Generator().manual_seed(11)
    rows = ((torch.relu(torch.randn(n_files * T, 16, generator=g)) * 0.2) @ torch.randn(16, d, generator=g)).reshape(n_files, T * d)
    for f, frac in ((0, 0.5), (3, 0.3), (5, 0.6), (10, 0.2)):
        idx = torch.randperm(T * d, generator=g)[: int(frac * T * d)]
        rows[f, idx] = -1.0
    folder = os.path.join(str(tmp_path), "train")
    write_shards(folder, "enc", rows.numpy(), [T, d])
    base = {
        "whisper_config": {"model": "tiny", "layer_name": "enc"}, "seed": 0, "train_folder": folder, "val_folder": folder,
        "device": "cuda", "lr": 1e-3, "weight_decay": 0.0, "steps": steps, "clip_thresh": 1.0, "dl_max_workers": 0,
        "log_tb_every": 1, "save_every": 2, "val_every": 1000, "scheduler_params": {}, "start_checkpoint": None, "from_disk": True,
    }
    if case == "topk":
        base.update(autoencoder_variant="topk", optimizer="adam", scheduler="cosine",
                    autoencoder_config={"n_dict_components": n, "k": 8, "auxk_alpha": 0.03125, "normalize_decoder": True,
                                        "multi_topk": False, "dead_feature_threshold": 100.0})
    else:
        base.update(autoencoder_variant="l1", optimizer="radam", scheduler="cosine",
                    autoencoder_config={"n_dict_components": n, "recon_alpha": 100.0})
    script = os.path.join(str(tmp_path), "child.py")
    open(script, "w").write(_TRAIN_CHILD)
    cfg2 = dict(copy.deepcopy(base), batch_size=B, run_dir=os.path.join(str(tmp_path), "dp2"))
    cfg1 = dict(copy.deepcopy(base), batch_size=world * B, run_dir=os.path.join(str(tmp_path), "dp1"))
    for name, cfg in (("cfg2.json", cfg2), ("cfg1.json", cfg1)):
        json.dump(cfg, open(os.path.join(str(tmp_path), name), "w"))
    env = {"FREUD_DP": "p2p", "FREUD_DP_PAYLOAD": payload, "FREUD_DP_OVERLAP": str(overlap), "FREUD_P2P_TIMEOUT_MS": "20000"}
    env.update(extra_env or {})
    outs = _run_children(script, os.path.join(str(tmp_path), "cfg2.json"), world, env)
    assert "exchange = p2p" in outs[0][0], outs[0][0][-1000:]
    # every step of these runs is a logging step: each one's exchanged gradient was checked against a gloo all-reduce of the
    # ranks' own contributions (dp.Auditor), and the replicas' checksums were compared before every checkpoint
    for so, _ in outs:
        assert f"AUDITS_PASSED {steps}" in so, so[-500:]
    _run_children(script, os.path.join(str(tmp_path), "cfg1.json"), 1, {})
    a = torch.load(os.path.join(cfg2["run_dir"], "checkpoints", f"step{steps}.pth"), map_location="cpu")
    b = torch.load(os.path.join(cfg1["run_dir"], "checkpoints", f"step{steps}.pth"), map_location="cpu")
    sc = lambda run: {(json.loads(l)["tag"], json.loads(l)["step"]): json.loads(l)["value"]
                      for l in open(os.path.join(run, "metrics.jsonl"))}
    s2, s1 = sc(cfg2["run_dir"]), sc(cfg1["run_dir"])
    wtol, ltol = (5e-3, 1e-2) if payload == "bfloat16" else (1e-4, 1e-4)
    for k in a["model"]:
        # (the bias vectors are tiny and move by ~lr per Adam step: the same absolute differences weigh ten times more)
        assert _rel(a["model"][k].numpy(), b["model"][k].numpy()) < (wtol if a["model"][k].dim() == 2 else 10 * wtol), k
    tags = ("train/fvu", "train/auxk_loss", "train/grad_norm") if case == "topk" else ("train/loss_recon", "train/loss_l1", "train/grad_norm")
    for step in range(1, steps + 1):
        for tag in tags:
            assert s2[(tag, step)] == pytest.approx(s1[(tag, step)], rel=ltol, abs=1e-7), (tag, step)


_ABSENT_PEER_CHILD = r"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["FREUD_ROOT"])
from freud_amd.engine import SaeEngine, EngineError
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
d, n, M = 384, 1024, 512
eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e2)
W = torch.empty(d, n); torch.nn.init.orthogonal_(W, generator=torch.Generator().manual_seed(0))
eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
blobs = [None, None]
dist.all_gather_object(blobs, eng.p2p_export())
eng.p2p_init(blobs, rank, 2)                      # collective self-test exchange: both ranks alive here
dist.barrier()
if rank == 1:
    os._exit(0)                                   # the peer disappears without a word
x = torch.randn(M, d, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).cuda()
t0 = time.time()
eng.step(x, 1e-3)                                 # the statistics push / exchange kernel wait for a rank that never comes
try:
    eng.dist_check()
    print("NO_ERROR", flush=True)
except EngineError as e:
    print("ERROR_AFTER %.1f s: %s" % (time.time() - t0, e), flush=True)
os._exit(0)
"""


def test_absent_peer_times_out_instead_of_hanging(tmp_path):
    """Failure detection of the in-engine exchange: rank 1 exits after the start-up self-test; rank 0's next step must NOT hang
    the GPU -- the polling kernels give up after FREUD_P2P_TIMEOUT_MS and sae_dist_check reports the absent peer."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(str(tmp_path), "absent.py")
    open(script, "w").write(_ABSENT_PEER_CHILD)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, FREUD_ROOT=root, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FREUD_P2P_TIMEOUT_MS="400")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.time()
    try:
        outs = [pr.communicate(timeout=180) for pr in procs]
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    assert procs[0].returncode == 0, outs[0][1][-3000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith(("ERROR_AFTER", "NO_ERROR"))][-1]
    assert line.startswith("ERROR_AFTER") and "did not arrive" in line, line
    assert time.time() - t0 < 120


# ------------------------------------------------------------------------------------------------------------------
# Round 4: the guards of the peer exchange must CATCH a wrong exchange
# ------------------------------------------------------------------------------------------------------------------
def _small_l1_config(tmp_path, steps=4, **over):
    import json
    import os
    from freud_amd.loader import write_shards
    d, n, T, n_files = 384, 1024, 64, 16
    g = torch.Generator().manual_seed(3)
    rows = ((torch.relu(torch.randn(n_files * T, 16, generator=g)) * 0.2) @ torch.randn(16, d, generator=g)).reshape(n_files, T * d)
    folder = os.path.join(str(tmp_path), "train")
    write_shards(folder, "enc", rows.numpy(), [T, d])
    cfg = {
        "whisper_config": {"model": "tiny", "layer_name": "enc"}, "seed": 0, "train_folder": folder, "val_folder": folder,
        "device": "cuda", "lr": 1e-3, "weight_decay": 0.0, "steps": steps, "clip_thresh": 1.0, "dl_max_workers": 0,
        "log_tb_every": 2, "save_every": 2, "val_every": 1000, "scheduler_params": {}, "start_checkpoint": None, "from_disk": True,
        "autoencoder_variant": "l1", "optimizer": "radam", "scheduler": "cosine",
        "autoencoder_config": {"n_dict_components": n, "recon_alpha": 100.0}, "batch_size": 2,
        "run_dir": os.path.join(str(tmp_path), "run"),
    }
    cfg.update(over)
    path = os.path.join(str(tmp_path), "cfg.json")
    json.dump(cfg, open(path, "w"))
    return cfg, path


def _run_cli(cfg_path, world, extra_env, timeout=600):
    """python -m freud_amd.train_sae --config ... as `world` processes on GPU 0 (gloo host channel).  Returns [(rc, out, err)]."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), FREUD_DIST_BACKEND="gloo",
                   RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **extra_env)
        procs.append(subprocess.Popen([sys.executable, "-m", "freud_amd.train_sae", "--config", cfg_path], env=env, cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for pr in procs:
            so, se = pr.communicate(timeout=timeout)
            res.append((pr.returncode, so, se))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return res


def test_injected_fault_is_caught_by_the_startup_selftest(tmp_path):
    """FREUD_P2P_FAULT=skip_phase2:1 -- rank 1 never copies one shard from its owner, i.e. it keeps what a stale read would give.
    The start-up self-test (patterns that change between exchanges over the same addresses) must see wrong sums on rank 1:
    FREUD_DP=p2p then refuses to start on EVERY rank, FREUD_DP=auto falls back together (host protocol over gloo here) and
    trains."""
    cfg, path = _small_l1_config(tmp_path)
    env = {"FREUD_P2P_FAULT": "skip_phase2:1", "FREUD_P2P_TIMEOUT_MS": "20000"}
    res = _run_cli(path, 2, dict(env, FREUD_DP="p2p"))
    assert all(rc != 0 for rc, _, _ in res), [r[0] for r in res]
    assert "self-test" in res[1][2] and "wrong values" in res[1][2], res[1][2][-2000:]
    assert "could not be set up on every rank" in res[0][2], res[0][2][-2000:]
    res = _run_cli(path, 2, dict(env, FREUD_DP="auto"))
    assert all(rc == 0 for rc, _, _ in res), [r[2][-1500:] for r in res]
    assert "exchange = host" in res[0][1], res[0][1][-500:]


@pytest.mark.parametrize("audit", ["1", "0"])
def test_injected_fault_after_selftest_stops_the_run_before_anything_is_written(tmp_path, audit):
    """The same fault switched on AFTER the self-test (its 12 gradient-channel exchanges): the run starts, rank 1's first exchange
    is wrong.  With the audit (default) the first step's sum check against the gloo all-reduce of the same inputs fails on rank
    1 and every rank stops; without it (FREUD_DP_AUDIT=0) the replica checksums differ at the first logging step.  Either way:
    exit code 3 on every rank, the last good checkpoint named, and NO checkpoint of the diverged run on disk."""
    import os
    cfg, path = _small_l1_config(tmp_path)
    env = {"FREUD_P2P_FAULT": "skip_phase2:1:12", "FREUD_P2P_TIMEOUT_MS": "20000", "FREUD_DP": "p2p", "FREUD_DP_AUDIT": audit}
    res = _run_cli(path, 2, env)
    for rc, so, se in res:
        assert rc == 3, (rc, se[-2000:])
        assert "FATAL: data-parallel exchange failed" in se and "last good checkpoint" in se, se[-2000:]
    assert ("differs from the gloo all-reduce" if audit == "1" else "replicas diverged") in res[0][2], res[0][2][-2000:]
    ck = os.path.join(cfg["run_dir"], "checkpoints")
    assert not os.path.isdir(ck) or not [f for f in os.listdir(ck) if f.endswith(".pth")], os.listdir(ck)


_LATE_PEER_CHILD = r"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["FREUD_ROOT"])
from freud_amd.engine import SaeEngine, EngineError
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
d, n, M = 384, 1024, 512
eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e2)
W = torch.empty(d, n); torch.nn.init.orthogonal_(W, generator=torch.Generator().manual_seed(0))
eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
blobs = [None, None]
dist.all_gather_object(blobs, eng.p2p_export())
eng.p2p_init(blobs, rank, 2)
dist.barrier()
x = torch.randn(M, d, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).cuda()
if rank == 1:
    time.sleep(4.0)                               # far beyond rank 0's timeout: rank 0 has given up when rank 1 arrives
eng.step(x, 1e-3)
polled = "clean"
torch.cuda.synchronize()
try:
    eng.dist_poll()
except EngineError as e:
    polled = "poll:" + str(e)
try:
    eng.dist_check()
    print("NO_ERROR", polled, flush=True)
except EngineError as e:
    print("ERROR: %s | %s" % (e, polled), flush=True)
dist.barrier()
os._exit(0)
"""


def test_late_peer_is_poisoned_not_served(tmp_path):
    """rank 0 times out waiting for rank 1 AFTER having published its own arrival flags.  When rank 1 arrives, those
    flags must not let it through: rank 0 has overwritten them with the poison value, rank 1's polls read it, rank 1 fails too
    (and says why), and both ranks' host-mapped failure words show it without a synchronisation."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(str(tmp_path), "late.py")
    open(script, "w").write(_LATE_PEER_CHILD)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, FREUD_ROOT=root, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FREUD_P2P_TIMEOUT_MS="500")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        outs = [pr.communicate(timeout=180) for pr in procs]
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    lines = []
    for pr, (so, se) in zip(procs, outs):
        assert pr.returncode == 0, se[-3000:]
        lines.append([l for l in so.splitlines() if l.startswith(("ERROR", "NO_ERROR"))][-1])
    assert lines[0].startswith("ERROR") and "did not arrive" in lines[0], lines
    assert lines[1].startswith("ERROR") and "left the protocol" in lines[1], lines
    assert "poll:" in lines[0] and "poll:" in lines[1], lines


def test_param_checksum_is_deterministic_and_sensitive():
    """sae_param_checksum: the same state gives the same words (order-independent sum), one flipped bit changes them."""
    from freud_amd.engine import SaeEngine
    d, n, M = 384, 1024, 512
    g = torch.Generator().manual_seed(0)
    W = torch.empty(d, n)
    torch.nn.init.orthogonal_(W, generator=g)
    x = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
    sums = []
    for flip in (False, False, True):
        eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e2)
        Wn = W.numpy().copy()
        if flip:
            Wn.view(np.uint32)[7, 11] ^= 1
        eng.set_params({"decoder.weight": Wn, "encoder_bias": np.zeros(n, np.float32)})
        for _ in range(2):
            eng.step(x, 1e-3)
        sums.append(eng.param_checksum())
        assert eng.param_checksum() == sums[-1]
        eng.close()
    assert sums[0] == sums[1] and sums[0][3] == 2
    assert sums[2][0] != sums[0][0]



def _run_bench_ranks(world, extra_args, extra_env=None, timeout=900):
    """bench.py as `world` processes on GPU 0, launched the way the driver launches it (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    in the environment) with FREUD_BENCH_SHARE_GPU=1: every rank on device 0, gloo as the host channel (two RCCL ranks cannot share
    a device; the engine's own peer exchange can).  Returns [(rc, stdout, stderr)] by rank."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, FREUD_BENCH_SHARE_GPU="1", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world)] + extra_args, env=env,
                                      cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for pr in procs:
            so, se = pr.communicate(timeout=timeout)
            res.append((pr.returncode, so, se))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return res


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_with_several_ranks_prints_one_line_with_the_guards_verdict(world):
    """The N > 1 flow of bench.py -- the one the driver's scaling run takes -- end to end on one GPU: set-up of the peer exchange
    (start-up self-test), audited warm-up steps (the exchanged gradient against a torch.distributed all-reduce of the same
    contributions), barrier-bracketed timed region, max over ranks, replica checksums after the run, ONE JSON line from rank 0
    whose value is the whole job's rows over that time."""
    import json
    res = _run_bench_ranks(world, ["--steps", "6", "--warmup", "4", "--rows", "8192", "--no-cpu-baseline", "--spinup", "0"])
    for rc, so, se in res:
        assert rc == 0, se[-3000:]
    lines = [[l for l in so.splitlines() if l.startswith("{")] for _, so, _ in res]
    assert len(lines[0]) == 1 and all(len(l) == 0 for l in lines[1:]), lines
    d = json.loads(lines[0][0])
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["parallelism"] == f"dp{world}"
    assert d["value"] == pytest.approx(8192 * world / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert d["config"]["dp"].startswith("in-engine peer exchange"), d["config"]["dp"]
    g = d["config"]["dp_guards"]
    assert g["audited_steps"] == 3 and g["max_rel_diff_vs_torch_distributed"] < 1e-5, g
    assert g["replica_checksums"] == "identical" and g["replica_checksums_after_run"] == "identical", g
    assert np.isfinite(d["loss"]["recon"]) and d["roofline"]["kernel_launches"] >= 6
    # the line explains its own efficiency: the exchange's duration, and the step without any exchange timed in
    # the same process -- exposed = ms_per_step - plain
    t = d["dp_timing"]
    assert t["carrier"] == "p2p" and t["exchange_launches_per_step"] == 1.0, t
    assert 0 < t["exchange_ms_min_over_ranks"] <= t["exchange_ms"] and t["stats_exchange_ms"] > 0, t
    assert t["plain_ms_per_step"] > 0 and t["exposed_exchange_ms"] == pytest.approx(d["ms_per_step"] - t["plain_ms_per_step"], abs=1e-9), t


def test_bench_with_a_broken_peer_exchange_falls_back_and_says_so():
    """Same launch with a fault injected into the peer exchange AFTER its start-up self-test (one shard of the all-gather is not
    copied on rank 1): the audit of the first warm-up step catches it on every rank, the measurement is repeated from scratch on
    the next carrier (host-driven here: two RCCL ranks cannot share this GPU), and the line carries the reason."""
    import json
    res = _run_bench_ranks(2, ["--steps", "4", "--warmup", "3", "--rows", "8192", "--no-cpu-baseline", "--spinup", "0"],
                           {"FREUD_P2P_FAULT": "skip_phase2:1:12", "FREUD_P2P_TIMEOUT_MS": "20000"})
    for rc, so, se in res:
        assert rc == 0, se[-3000:]
    d = json.loads([l for l in res[0][1].splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["dp"].startswith("host-driven"), d["config"]["dp"]
    assert "fallback" in d["config"]["dp"] and "differs from the gloo all-reduce" in d["config"]["dp"], d["config"]["dp"]
    assert np.isfinite(d["loss"]["recon"])


def test_bench_gpus_n_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO rank environment: the process becomes the launcher -- it touches no
    GPU, starts two child ranks with the launcher's variables, relays rank 0's single line -- so that the command can never print an
    `n_gpus: 1` line for `--gpus 2`.  (Both ranks on GPU 0 here: FREUD_BENCH_SHARE_GPU=1.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["FREUD_BENCH_SHARE_GPU"] = "1"
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "4", "--rows", "8192",
                         "--no-cpu-baseline", "--spinup", "0"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and "self-launched" in d["config"]["launcher"]
    assert d["config"]["dp"].startswith("in-engine peer exchange"), d["config"]["dp"]
    t = d["dp_timing"]
    assert t["ranks_counted_by_collective"] == 2 and t["collective_backend"] == "gloo", t
    assert d["value"] == pytest.approx(8192 * 2 / (d["ms_per_step"] * 1e-3), rel=1e-6)


def test_bench_reports_the_other_carrier_next_to_the_headline():
    """Every data-parallel bench line carries a short leg on the OTHER in-engine carrier (north_star names RCCL; the peer exchange is
    auto's first choice), run on a fresh context under a watchdog after everything else is in the line.  With the nccl backend the
    leg needs real RCCL, which two ranks cannot share on one device -- `--force-dist` runs the same code path with ONE rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FREUD_BENCH_SHARE_GPU")}
    env["MASTER_PORT"] = str(_free_port())
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "6", "--warmup", "3", "--rows", "8192",
                         "--no-cpu-baseline", "--spinup", "0"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-3000:]
    d = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
    t = d["dp_timing"]
    assert t["carrier"] == "p2p" and t["other_carrier"]["carrier"] == "rccl", t
    assert t["other_carrier"]["healthy"] and t["other_carrier"]["ms_per_step"] > 0 and t["other_carrier"]["ranks_in_engine_communicator"] == 1, t

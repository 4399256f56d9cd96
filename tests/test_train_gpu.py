"""GPU: the whole drop-in path (shards on disk -> pinned/HBM streamer -> HIP engine -> checkpoints)
against the reference's own train() run stored in tests/golden/trainloop_l1.npz.
Tolerance: losses rtol 1e-2 over the 7-step trajectory, final weights rel-L2 <= 1e-3."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from freud_amd.loader import MemoryMappedActivationDataLoader, write_shards

pytestmark = pytest.mark.gpu


def test_train_cli_path_matches_reference(tmp_path, golden_dir):
    from freud_amd.train_sae import main
    z = np.load(os.path.join(golden_dir, "trainloop_l1.npz"))
    meta = json.loads(str(z["meta"]))
    folder = os.path.join(str(tmp_path), "train")
    write_shards(folder, meta["layer"], z["shard"], [meta["T"], meta["d"]],
                 [f"/data/audio/file_{i:04d}.flac" for i in range(meta["n_files"])])
    cfg = copy.deepcopy(meta["config"])
    cfg.update(train_folder=folder, val_folder=folder, run_dir=os.path.join(str(tmp_path), "run"), device="cuda")
    cfg_path = os.path.join(str(tmp_path), "cfg.json")
    json.dump(cfg, open(cfg_path, "w"))
    main(["--config", cfg_path])
    ck_dir = os.path.join(cfg["run_dir"], "checkpoints")
    assert sorted(os.listdir(ck_dir)) == sorted(meta["checkpoint_files"])
    got = {}
    for line in open(os.path.join(cfg["run_dir"], "metrics.jsonl")):
        s = json.loads(line)
        got[(s["tag"], s["step"])] = s["value"]
    for tag, val, step in meta["scalars"]:
        assert got[(tag, step)] == pytest.approx(val, rel=1e-2), (tag, step)
    ck = torch.load(os.path.join(ck_dir, "step7.pth"), map_location="cpu", weights_only=True)
    assert sorted(ck.keys()) == meta["checkpoint_keys"]
    W, Wref = ck["model"]["decoder.weight"].numpy(), z["model__decoder.weight"]
    assert np.linalg.norm(W - Wref) / np.linalg.norm(Wref) < 1e-3
    # second run with validation switched on (it draws from the global RNG like the reference's val
    # DataLoader does, so later batches differ from the fixture: only check what it produces)
    cfg.update(run_dir=os.path.join(str(tmp_path), "run_val"), val_every=5)
    json.dump(cfg, open(cfg_path, "w"))
    main(["--config", cfg_path])
    tags = {(json.loads(l)["tag"], json.loads(l)["step"]) for l in open(os.path.join(cfg["run_dir"], "metrics.jsonl"))}
    assert ("val/loss_recon", 5) in tags and ("val/encoded/num_dead", 5) in tags and ("val/mse", 5) in tags
    assert os.path.exists(os.path.join(cfg["run_dir"], "checkpoints", "bestval.pth"))


@pytest.mark.parametrize("direct", ["1", "0"])
def test_cuda_streamer_delivers_exact_rows(tmp_path, direct, monkeypatch):
    """Both streaming modes: rows by DMA straight from the host-registered mapping, and the staged path (threaded gather ->
    pinned ring -> copy stream) that is the fallback when registration is not possible."""
    monkeypatch.setenv("FREUD_LOADER_DIRECT", direct)
    n_files, T, d = 23, 5, 384
    rows = np.random.default_rng(0).standard_normal((n_files, T * d)).astype(np.float16)
    write_shards(str(tmp_path), "L", rows, [T, d])
    dl = MemoryMappedActivationDataLoader(str(tmp_path), "L", 4, 0, None, {"shuffle": True, "drop_last": False},
                                          device="cuda")
    assert dl._direct == (direct == "1")
    seen = set()
    for x, names in dl:
        assert x.is_cuda and x.dtype == torch.float16
        xc = x.cpu().numpy()          # sync: the batch must be complete when handed over
        for xi, nm in zip(xc, names):
            i = int(nm[5:11])
            np.testing.assert_array_equal(xi.reshape(-1), rows[i])
            seen.add(i)
    assert seen == set(range(n_files))


def test_topk_train_path_matches_reference(tmp_path, golden_dir):
    """TopK end to end (shards -> engine -> checkpoint) against the reference's own train() run.  Boundary ties in
    the bf16 top-k (tests/test_topk_gpu.py) make trajectories drift slightly: losses rtol 3e-2."""
    from freud_amd.train_sae import main
    z = np.load(os.path.join(golden_dir, "trainloop_topk.npz"))
    meta = json.loads(str(z["meta"]))
    folder = os.path.join(str(tmp_path), "train")
    write_shards(folder, meta["layer"], z["shard"], [meta["T"], meta["d"]],
                 [f"/data/audio/file_{i:04d}.flac" for i in range(meta["n_files"])])
    cfg = copy.deepcopy(meta["config"])
    cfg.update(train_folder=folder, val_folder=folder, run_dir=os.path.join(str(tmp_path), "run"), device="cuda")
    cfg_path = os.path.join(str(tmp_path), "cfg.json")
    json.dump(cfg, open(cfg_path, "w"))
    main(["--config", cfg_path])
    ck_dir = os.path.join(cfg["run_dir"], "checkpoints")
    assert sorted(os.listdir(ck_dir)) == sorted(meta["checkpoint_files"])
    got = {}
    for line in open(os.path.join(cfg["run_dir"], "metrics.jsonl")):
        s = json.loads(line)
        got[(s["tag"], s["step"])] = s["value"]
    for tag, val, step in meta["scalars"]:
        if tag in ("train/fvu", "train/loss", "train/lr"):
            assert got[(tag, step)] == pytest.approx(val, rel=3e-2), (tag, step)
    ck = torch.load(os.path.join(ck_dir, "step7.pth"), map_location="cpu", weights_only=True)
    assert list(ck["model"].keys()) == meta["model_keys"]
    for k in ("W_dec", "encoder.weight"):
        W, Wref = ck["model"][k].numpy(), z[f"model__{k}"]
        assert np.linalg.norm(W - Wref) / np.linalg.norm(Wref) < 5e-3


def test_bench_data_parallel_path_single_rank():
    """The DP code path of bench.py (RCCL all-reduce of the engine's gradient buffer aliased as a torch tensor,
    1/R scaling, separate optimizer call) on one rank: same throughput contract, finite losses."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                          "--rows", "8192", "--force-dist", "--no-cpu-baseline", "--spinup", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and np.isfinite(res["loss"]["recon"])
    single = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                             "--rows", "8192", "--no-cpu-baseline", "--spinup", "0"], capture_output=True, text=True, timeout=600)
    ref = json.loads([l for l in single.stdout.splitlines() if l.startswith("{")][-1])
    assert res["loss"]["recon"] == pytest.approx(ref["loss"]["recon"], rel=1e-5)      # same arithmetic either way
    assert res["loss"]["grad_norm"] == pytest.approx(ref["loss"]["grad_norm"], rel=1e-4)


@pytest.mark.parametrize("host_driven", ["0", "1"])
@pytest.mark.parametrize("fixture", ["trainloop_l1", "trainloop_topk"])
def test_train_data_parallel_path_single_rank(tmp_path, golden_dir, fixture, host_driven):
    """train() through its data-parallel code paths with one rank (FREUD_FORCE_DIST=1) ends with the same weights as the
    single-process path: host_driven 0 = the engine's own RCCL communicator (sae_dist_init; statistics and gradient
    ranges all-reduced on its communication stream, no Python in the step), 1 = the same protocol from Python through
    torch.distributed (FREUD_DP_HOST=1: statistics all-reduce, gradient-ready callback, asynchronous all-reduce of every
    announced range, separate optimizer call).  Run in a child process: the process group must not leak into other tests."""
    import subprocess
    import sys
    z = np.load(os.path.join(golden_dir, f"{fixture}.npz"))
    meta = json.loads(str(z["meta"]))
    folder = os.path.join(str(tmp_path), "train")
    write_shards(folder, meta["layer"], z["shard"], [meta["T"], meta["d"]],
                 [f"/data/audio/file_{i:04d}.flac" for i in range(meta["n_files"])])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    finals = []
    for force in ("0", "1"):
        cfg = copy.deepcopy(meta["config"])
        cfg.update(train_folder=folder, val_folder=folder, run_dir=os.path.join(str(tmp_path), "run" + force), device="cuda")
        cfg_path = os.path.join(str(tmp_path), f"cfg{force}.json")
        json.dump(cfg, open(cfg_path, "w"))
        env = dict(os.environ, FREUD_FORCE_DIST=force, FREUD_DP_HOST=host_driven, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
        out = subprocess.run([sys.executable, "-m", "src.scripts.train_sae", "--config", cfg_path], cwd=root, env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        ck = torch.load(os.path.join(cfg["run_dir"], "checkpoints", "step7.pth"), map_location="cpu", weights_only=True)
        finals.append(ck["model"])
    for k in finals[0]:
        a, b = finals[0][k].numpy(), finals[1][k].numpy()
        assert np.linalg.norm(a - b) <= 1e-6 * max(np.linalg.norm(a), 1e-12), k

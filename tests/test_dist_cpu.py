"""CPU, world_size 2 over gloo: the data-parallel path of train() (sharded sampler, summed
[grads | loss scalars] buffer, 1/R scaling, rank-0-only checkpoints) with the oracle-backed
stand-in engine.  Two ranks with per-rank batch B must reproduce one process with batch 2B on the
same epoch permutation (rows are independent; the L1 mean and the MSE mean both average)."""
import copy
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from freud_amd.loader import write_shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cfg, autocast):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import functools
    from freud_amd.train_sae import train
    from tests.fake_engine import OracleEngine
    train(**cfg, engine_factory=functools.partial(OracleEngine, autocast=autocast), dist_backend="gloo")
    import torch.distributed as dist
    dist.destroy_process_group()


def _scalars(run_dir):
    return {(json.loads(l)["tag"], json.loads(l)["step"]): json.loads(l)["value"] for l in open(os.path.join(run_dir, "metrics.jsonl"))}


def _run_pair(tmp_path, variant, world, autocast, per_rank_batch=2):
    """world ranks x per_rank_batch files against one process x world * per_rank_batch files, same shard, same seed."""
    import functools
    T, d, n_files = 6, 16, 16
    g = torch.Generator().manual_seed(3)
    rows = (torch.relu(torch.randn(n_files * T, 4, generator=g)) @ torch.randn(4, d, generator=g)).reshape(n_files, T * d)
    # exact -1.0 entries, many in a few files: the ranks see DIFFERENT unmasked counts, so averaging per-rank means would
    # not be the whole batch's mean (mse_loss's ignored_index, l1autoencoder.py:29-36)
    for f, frac in ((0, 0.5), (3, 0.3), (5, 0.6), (10, 0.2)):
        idx = torch.randperm(T * d, generator=g)[: int(frac * T * d)]
        rows[f, idx] = -1.0
    folder = os.path.join(str(tmp_path), "train")
    write_shards(folder, "enc", rows.numpy(), [T, d])
    base = {
        "whisper_config": {"model": "tiny", "layer_name": "enc"}, "seed": 0, "train_folder": folder,
        "val_folder": folder, "device": "cpu", "lr": 1e-3, "weight_decay": 0.0, "steps": 4, "clip_thresh": 1.0,
        "dl_max_workers": 0, "log_tb_every": 1, "save_every": 2, "val_every": 1000,
        "scheduler_params": {}, "start_checkpoint": None, "from_disk": True,
    }
    if variant == "l1":
        base.update(autoencoder_variant="l1", autoencoder_config={"n_dict_components": 32, "recon_alpha": 100.0},
                    optimizer="radam", scheduler="cosine")
    else:
        base.update(autoencoder_variant="topk", optimizer="adam", scheduler="cosine",
                    autoencoder_config={"n_dict_components": 48, "k": 4, "auxk_alpha": 0.03125, "normalize_decoder": True,
                                        "multi_topk": False, "dead_feature_threshold": 20.0})
    cfgR = dict(copy.deepcopy(base), batch_size=per_rank_batch, run_dir=os.path.join(str(tmp_path), "dpR"))
    mp.spawn(_worker, args=(world, _free_port(), cfgR, autocast), nprocs=world, join=True)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    from freud_amd.train_sae import train
    from tests.fake_engine import OracleEngine
    cfg1 = dict(copy.deepcopy(base), batch_size=world * per_rank_batch, run_dir=os.path.join(str(tmp_path), "dp1"))
    train(**cfg1, engine_factory=functools.partial(OracleEngine, autocast=autocast))
    a = torch.load(os.path.join(cfgR["run_dir"], "checkpoints", "step4.pth"), map_location="cpu")
    b = torch.load(os.path.join(cfg1["run_dir"], "checkpoints", "step4.pth"), map_location="cpu")
    assert sorted(os.listdir(os.path.join(cfgR["run_dir"], "checkpoints"))) == \
        sorted(os.listdir(os.path.join(cfg1["run_dir"], "checkpoints")))              # rank 0 alone writes
    return a, b, _scalars(cfgR["run_dir"]), _scalars(cfg1["run_dir"])


@pytest.mark.parametrize("world", [2, 4])
def test_l1_ranks_match_single_process_exactly(tmp_path, world):
    """fp32 math, masked entries unevenly spread over the ranks: R ranks x B files train exactly like one process x R B
    files -- weights to 1e-5 (fp32 summation order), logged losses to 1e-5.  This is what summing the unmasked count and
    the row number over the ranks BEFORE the backward buys; with per-rank means the runs differ at the 1e-2 level here."""
    a, b, sR, s1 = _run_pair(tmp_path, "l1", world, autocast=False, per_rank_batch=2 if world == 2 else 1)
    for k in a["model"]:
        assert torch.linalg.norm(a["model"][k] - b["model"][k]) <= 1e-5 * torch.linalg.norm(b["model"][k]), k
    for pid in (0, 1):
        for mk in ("exp_avg", "exp_avg_sq"):
            u, v = a["optimizer"]["state"][pid][mk], b["optimizer"]["state"][pid][mk]
            assert torch.linalg.norm(u - v) <= 1e-4 * torch.linalg.norm(v)
    for step in range(1, 5):
        for tag in ("train/loss_recon", "train/loss_l1", "train/grad_norm"):
            assert sR[(tag, step)] == pytest.approx(s1[(tag, step)], rel=1e-5), (tag, step)
        assert sR[("train/lr", step)] == s1[("train/lr", step)]


def test_l1_two_ranks_bf16_autocast_bound(tmp_path):
    """The same under bf16 autocast: per-rank GEMM outputs are rounded to bf16 before they are summed, so the runs agree
    to the bf16 re-ordering bound (2^-9 per rounded element: weights rel-L2 3e-3, losses 1e-2), not bitwise."""
    a, b, sR, s1 = _run_pair(tmp_path, "l1", 2, autocast=True)
    W2, W1 = a["model"]["decoder.weight"], b["model"]["decoder.weight"]
    assert torch.linalg.norm(W2 - W1) / torch.linalg.norm(W1) < 3e-3
    for step in range(1, 5):
        for tag in ("train/loss_recon", "train/loss_l1"):
            assert sR[(tag, step)] == pytest.approx(s1[(tag, step)], rel=1e-2), (tag, step)


def test_topk_two_ranks_match_single_process_exactly(tmp_path):
    """TopK, fp32 math: total_variance around x.mean(0) over ALL files (topkautoencoder.py:104-106), the did_fire OR and the
    frame counter over all ranks' rows (train_sae.py:443-446): two ranks x 2 files == one process x 4 files."""
    a, b, sR, s1 = _run_pair(tmp_path, "topk", 2, autocast=False)
    for k in a["model"]:
        assert torch.linalg.norm(a["model"][k] - b["model"][k]) <= 2e-5 * max(torch.linalg.norm(b["model"][k]), 1e-3), k
    for step in range(1, 5):
        for tag in ("train/fvu", "train/auxk_loss", "train/dead_pct", "train/grad_norm"):
            assert sR[(tag, step)] == pytest.approx(s1[(tag, step)], rel=2e-5, abs=1e-7), (tag, step)


def _diverge_worker(rank, world, port, cfg):
    """train() on `world` gloo ranks with an engine whose rank 1 silently corrupts one weight in the third step: the replica
    guard must stop EVERY rank before the next checkpoint is written."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import functools
    from freud_amd import dp
    from freud_amd.train_sae import train
    from tests.fake_engine import OracleEngine

    class Corrupting(OracleEngine):
        def optimizer_step(self, lr, grad_scale=1.0, stream=None):
            super().optimizer_step(lr, grad_scale, stream)
            if rank == 1 and self.st.step == 3:
                k = sorted(self.P)[0]
                self.P[k].view(-1)[3] += 1e-3
    try:
        train(**cfg, engine_factory=functools.partial(Corrupting, autocast=False), dist_backend="gloo")
        outcome = "no error"
    except dp.ExchangeError as e:
        outcome = "ExchangeError: " + str(e)
    open(os.path.join(cfg["run_dir"], f"outcome_rank{rank}.txt"), "w").write(outcome)
    import torch.distributed as dist
    dist.destroy_process_group()


def test_replica_guard_stops_every_rank_before_the_next_write(tmp_path):
    """freud_amd/dp.py: check_replicas (round 4).  Replicas are bit-identical by construction; here rank 1's parameters change
    behind the protocol's back during step 3.  The guard of that step's logging raises dp.ExchangeError on BOTH ranks -- naming
    the diverged rank -- and the checkpoint of step 4 is never written (the one of step 2, checked and written before the
    corruption, stays the last good one)."""
    T, d, n_files = 6, 16, 16
    g = torch.Generator().manual_seed(3)
    rows = (torch.relu(torch.randn(n_files * T, 4, generator=g)) @ torch.randn(4, d, generator=g)).reshape(n_files, T * d)
    folder = os.path.join(str(tmp_path), "train")
    write_shards(folder, "enc", rows.numpy(), [T, d])
    cfg = {
        "whisper_config": {"model": "tiny", "layer_name": "enc"}, "seed": 0, "train_folder": folder, "val_folder": folder,
        "device": "cpu", "lr": 1e-3, "weight_decay": 0.0, "steps": 6, "clip_thresh": 1.0, "dl_max_workers": 0, "log_tb_every": 1,
        "save_every": 2, "val_every": 1000, "scheduler_params": {}, "start_checkpoint": None, "from_disk": True,
        "autoencoder_variant": "l1", "autoencoder_config": {"n_dict_components": 32, "recon_alpha": 100.0},
        "optimizer": "radam", "scheduler": "cosine", "batch_size": 2, "run_dir": os.path.join(str(tmp_path), "run"),
    }
    os.makedirs(cfg["run_dir"], exist_ok=True)
    mp.spawn(_diverge_worker, args=(2, _free_port(), cfg), nprocs=2, join=True)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    outs = [open(os.path.join(cfg["run_dir"], f"outcome_rank{r}.txt")).read() for r in range(2)]
    for o in outs:
        assert o.startswith("ExchangeError") and "replicas diverged" in o and "rank 1 differs from rank 0 in parameters" in o, o
    assert sorted(os.listdir(os.path.join(cfg["run_dir"], "checkpoints"))) == ["step2.pth"]


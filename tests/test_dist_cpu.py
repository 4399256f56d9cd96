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


def _worker(rank, world, port, cfg):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from freud_amd.train_sae import train
    from tests.fake_engine import OracleEngine
    train(**cfg, engine_factory=OracleEngine, dist_backend="gloo")
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_ranks_match_single_process(tmp_path):
    T, d, n_files = 6, 16, 16
    g = torch.Generator().manual_seed(3)
    rows = (torch.relu(torch.randn(n_files * T, 4, generator=g)) @ torch.randn(4, d, generator=g)).reshape(n_files, T * d)
    folder = os.path.join(str(tmp_path), "train")
    write_shards(folder, "enc", rows.numpy(), [T, d])
    base = {
        "whisper_config": {"model": "tiny", "layer_name": "enc"}, "autoencoder_variant": "l1",
        "autoencoder_config": {"n_dict_components": 32, "recon_alpha": 100.0}, "seed": 0, "train_folder": folder,
        "val_folder": folder, "device": "cpu", "lr": 1e-3, "weight_decay": 0.0, "steps": 4, "clip_thresh": 1.0,
        "dl_max_workers": 0, "log_tb_every": 1, "save_every": 2, "val_every": 1000, "optimizer": "radam",
        "scheduler": "cosine", "scheduler_params": {}, "start_checkpoint": None, "from_disk": True,
    }
    cfg2 = dict(copy.deepcopy(base), batch_size=2, run_dir=os.path.join(str(tmp_path), "dp2"))
    mp.spawn(_worker, args=(2, _free_port(), cfg2), nprocs=2, join=True)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    from freud_amd.train_sae import train
    from tests.fake_engine import OracleEngine
    cfg1 = dict(copy.deepcopy(base), batch_size=4, run_dir=os.path.join(str(tmp_path), "dp1"))
    train(**cfg1, engine_factory=OracleEngine)

    a = torch.load(os.path.join(cfg2["run_dir"], "checkpoints", "step4.pth"), map_location="cpu")
    b = torch.load(os.path.join(cfg1["run_dir"], "checkpoints", "step4.pth"), map_location="cpu")
    assert sorted(os.listdir(os.path.join(cfg2["run_dir"], "checkpoints"))) == \
        sorted(os.listdir(os.path.join(cfg1["run_dir"], "checkpoints")))              # rank 0 alone writes
    W2, W1 = a["model"]["decoder.weight"], b["model"]["decoder.weight"]
    W0 = torch.nn.functional.normalize(W1, dim=0)
    # both runs moved the weights the same way (bf16 rounding of per-rank vs whole-batch GEMMs differs)
    assert torch.linalg.norm(W2 - W1) / torch.linalg.norm(W1) < 2e-3
    s2 = {(json.loads(l)["tag"], json.loads(l)["step"]): json.loads(l)["value"]
          for l in open(os.path.join(cfg2["run_dir"], "metrics.jsonl"))}
    s1 = {(json.loads(l)["tag"], json.loads(l)["step"]): json.loads(l)["value"]
          for l in open(os.path.join(cfg1["run_dir"], "metrics.jsonl"))}
    for step in range(1, 5):
        for tag in ("train/loss_recon", "train/loss_l1"):
            assert s2[(tag, step)] == pytest.approx(s1[(tag, step)], rel=2e-2), (tag, step)
        assert s2[("train/lr", step)] == s1[("train/lr", step)]

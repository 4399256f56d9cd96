#!/usr/bin/env python3
"""Host side of the path measured on the GPU box: shard streaming (mmap -> pinned -> HBM double buffer) alone, and the
whole train() loop (loader + engine) on synthetic shards in the collector's format.  Prints one JSON line.

  python tools/bench_loader.py [--files 400] [--d 384] [--dtype float16] [--batch-size 40] [--steps 150]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from freud_amd.loader import MemoryMappedActivationDataLoader, write_shards
from freud_amd import train_sae
from freud_amd.train_sae import train


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=400)
    ap.add_argument("--d", type=int, default=384)
    ap.add_argument("--T", type=int, default=1500)
    ap.add_argument("--dtype", default="float16", choices=["float16", "float32"])
    ap.add_argument("--batch-size", type=int, default=40)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--expansion", type=int, default=8)
    ap.add_argument("--sweep", action="store_true", help="loader-only rate over gather threads x pipeline depth, and raw H2D")
    ap.add_argument("--deliver", default="native", choices=["native", "bfloat16"],
                    help="bfloat16: fp32 shard rows are down-converted in the gather threads (half the PCIe bytes)")
    ap.add_argument("--direct", default="auto", choices=["auto", "0", "1"], help="FREUD_LOADER_DIRECT")
    args = ap.parse_args()
    os.environ["FREUD_LOADER_DELIVER"] = args.deliver
    os.environ["FREUD_LOADER_DIRECT"] = args.direct
    tmp = tempfile.mkdtemp(prefix="freud_loader_", dir="/tmp")
    try:
        layer = "encoder.blocks.2"
        rng = np.random.default_rng(0)
        z = np.maximum(rng.standard_normal((args.files * args.T, 32), dtype=np.float32), 0) * 0.1
        rows = (z @ rng.standard_normal((32, args.d), dtype=np.float32)).astype(args.dtype).reshape(args.files, args.T * args.d)
        folder = os.path.join(tmp, "train")
        write_shards(folder, layer, rows, [args.T, args.d], [f"/data/f{i}.flac" for i in range(args.files)])
        nbytes = rows.nbytes
        del rows, z
        out = {"files": args.files, "T": args.T, "d": args.d, "dtype": args.dtype, "deliver": args.deliver, "direct": args.direct,
               "shard_GB": nbytes / 1e9, "batch_size_files": args.batch_size, "host_cores": os.cpu_count()}

        dl = MemoryMappedActivationDataLoader(folder, layer, args.batch_size, 0, None, {"shuffle": True, "drop_last": True},
                                              device="cuda")
        out["mode"] = "direct (registered mapping, one DMA per file)" if dl._direct else (
            "staged (gather threads -> pinned ring)" + (" + fp32->bf16 conversion" if dl._convert else ""))
        for epoch in range(3):           # epoch 0 warms the page cache; report the best of the rest
            t0 = time.perf_counter()
            nb = 0
            for xb, _ in dl:
                nb += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if epoch:
                rate = nb * args.batch_size * args.T / dt
                out["loader_only_act_per_s"] = max(out.get("loader_only_act_per_s", 0), rate)
                out["loader_only_GB_per_s"] = max(out.get("loader_only_GB_per_s", 0),
                                                  nb * args.batch_size * args.T * args.d * np.dtype(args.dtype).itemsize / dt / 1e9)

        if args.sweep:
            nb_bytes = args.batch_size * args.T * args.d * np.dtype(args.dtype).itemsize
            pin = torch.empty(nb_bytes, dtype=torch.uint8, pin_memory=True)
            dev = torch.empty(nb_bytes, dtype=torch.uint8, device="cuda")
            dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                dev.copy_(pin, non_blocking=True)
            torch.cuda.synchronize()
            out["h2d_GB_per_s"] = 20 * nb_bytes / (time.perf_counter() - t0) / 1e9
            sweep = {}
            for threads in (4, 8, 16, 32):
                for depth in (2, 3, 4):
                    dl2 = MemoryMappedActivationDataLoader(folder, layer, args.batch_size, threads, None,
                                                           {"shuffle": True, "drop_last": True}, device="cuda", depth=depth)
                    best = 0
                    for _ in range(2):
                        t0 = time.perf_counter()
                        nb = sum(1 for _ in dl2)
                        torch.cuda.synchronize()
                        best = max(best, nb * nb_bytes / (time.perf_counter() - t0) / 1e9)
                    sweep[f"t{threads}_d{depth}"] = round(best, 1)
            out["loader_GB_per_s_sweep"] = sweep
        # engine alone on one resident batch of the loader's shape and dtype
        from freud_amd.engine import SaeEngine
        xb = next(iter(dl))[0].clone()
        del dl                      # releases its host registration of the shard mapping
        import gc
        gc.collect()
        eng = SaeEngine(variant="l1", d_model=args.d, n_dict=args.d * args.expansion, max_rows=xb.shape[0] * xb.shape[1],
                        optimizer="radam", recon_alpha=1e4)
        W = torch.empty(args.d, args.d * args.expansion)
        torch.nn.init.orthogonal_(W)
        eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(args.d * args.expansion, np.float32)})
        for _ in range(50):
            eng.step(xb, 1e-4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            eng.step(xb, 1e-4)
        torch.cuda.synchronize()
        out["engine_only_ms_per_step"] = (time.perf_counter() - t0) / 100 * 1e3
        eng.close()

        cfg = {"whisper_config": {"model": "tiny", "layer_name": layer}, "autoencoder_variant": "l1",
               "autoencoder_config": {"expansion_factor": args.expansion, "recon_alpha": 1e4}, "seed": 0,
               "train_folder": folder, "val_folder": folder, "device": "cuda", "run_dir": os.path.join(tmp, "run"),
               "lr": 4e-4, "weight_decay": 0.0, "steps": args.steps, "clip_thresh": 1.0, "batch_size": args.batch_size,
               "dl_max_workers": 0, "log_tb_every": 10 ** 9, "save_every": 10 ** 9, "val_every": 10 ** 9,
               "optimizer": "radam", "scheduler": "cosine", "scheduler_params": {}, "start_checkpoint": None,
               "from_disk": True}
        # the synthetic set is small, so an epoch is a handful of steps and the reference's epoch-end checkpoint
        # (train_sae.py:600-602) would dominate: time the loop with checkpoint writing stubbed out
        train_sae.save_checkpoint = lambda *a, **k: None
        train(**dict(cfg, steps=20, run_dir=os.path.join(tmp, "warm")))      # warm-up: library load, clocks
        # steady-state step time = slope between two run lengths: a run also pays for creating the context, host-
        # registering the shard (~30 ms) and unregistering it at the end (~140 ms for 2.3 GB), which a real run
        # amortises over thousands of steps
        times = []
        for k, st in enumerate((args.steps, 3 * args.steps)):
            t0 = time.perf_counter()
            train(**dict(cfg, steps=st, run_dir=os.path.join(tmp, f"run{k}")))
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        per_step = (times[1] - times[0]) / (2 * args.steps)
        out["train_loop_act_per_s"] = args.batch_size * args.T / per_step
        out["train_loop_ms_per_step"] = per_step * 1e3
        out["train_run_fixed_cost_ms"] = (times[0] - per_step * args.steps) * 1e3
        out["rows_per_step"] = args.batch_size * args.T
        print(json.dumps(out), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where does a loader + engine loop spend its time?  Host time inside next(loader) and inside eng.step(), and wall time
per step, for the direct loader mode on synthetic fp16 shards."""
import os, sys, time, tempfile, shutil
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freud_amd.engine import SaeEngine
from freud_amd.loader import MemoryMappedActivationDataLoader, write_shards

tmp = tempfile.mkdtemp(prefix="freud_lp_", dir="/tmp")
try:
    files, T, d, n, B = 2000, 1500, 384, 3072, 40
    rows = (np.random.default_rng(0).standard_normal((files, T * d), dtype=np.float32) * 0.1).astype(np.float16)
    write_shards(os.path.join(tmp, "s"), "l", rows, [T, d], [f"f{i}" for i in range(files)])
    del rows
    dl = MemoryMappedActivationDataLoader(os.path.join(tmp, "s"), "l", B, 0, None, {"shuffle": True, "drop_last": True}, device="cuda")
    eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=B * T, optimizer="radam", recon_alpha=1e4)
    W = torch.empty(d, n); torch.nn.init.orthogonal_(W)
    eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
    for epoch in range(3):
        torch.cuda.synchronize(); t_load = t_step = 0.0; nb = 0; t0 = time.perf_counter(); it = iter(dl)
        while True:
            a = time.perf_counter()
            try: xb, _ = next(it)
            except StopIteration: break
            b = time.perf_counter(); eng.step(xb, 1e-4); c = time.perf_counter()
            t_load += b - a; t_step += c - b; nb += 1
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        print("epoch %d: direct=%s wall %.3f ms/step | host in next(loader) %.3f | host in eng.step %.3f" %
              (epoch, dl._direct, wall / nb * 1e3, t_load / nb * 1e3, t_step / nb * 1e3))
finally:
    shutil.rmtree(tmp, ignore_errors=True)

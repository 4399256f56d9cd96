#!/usr/bin/env python3
"""Run ON THE GPU BOX: train two tiny SAEs (L1, TopK) for a few steps with the real HIP engine through the drop-in
train() and leave what the reference-side consumption check (tools/reference_consumes_checkpoint.py, build container)
needs under gpurun_out/engine_ckpt/: the engine-written checkpoints, the batch they are evaluated on and the engine's own
eval losses / latents on that batch.  The files are copied to tests/golden/engine_ckpt/ and committed (data only)."""
import json
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freud_amd.loader import write_shards
from freud_amd.train_sae import train
from freud_amd.engine import SaeEngine

OUT = os.path.join(ROOT, "gpurun_out", "engine_ckpt")
os.makedirs(OUT, exist_ok=True)
T, d, n_files, layer = 16, 32, 12, "encoder.blocks.2"
g = torch.Generator().manual_seed(11)
rows = (torch.relu(torch.randn(n_files * T, 6, generator=g)) * 0.5) @ torch.randn(6, d, generator=g) + 0.05 * torch.randn(n_files * T, d, generator=g)
folder = os.path.join(OUT, "shards")
write_shards(folder, layer, rows.reshape(n_files, T * d).numpy().astype(np.float32), [T, d])
x_eval = rows[: 2 * T].reshape(2, T, d).contiguous()
np.save(os.path.join(OUT, "x_eval.npy"), x_eval.numpy())
base = {"whisper_config": {"model": "tiny", "layer_name": layer}, "seed": 0, "train_folder": folder, "val_folder": folder,
        "device": "cuda", "weight_decay": 0.0, "steps": 6, "clip_thresh": 1.0, "batch_size": 3, "dl_max_workers": 0,
        "log_tb_every": 1, "save_every": 3, "val_every": 1000, "start_checkpoint": None, "from_disk": True}
cases = {
    "l1": dict(autoencoder_variant="l1", autoencoder_config={"n_dict_components": 96, "recon_alpha": 1e2}, lr=1e-3,
               optimizer="radam", scheduler="cosine", scheduler_params={}),
    "topk": dict(autoencoder_variant="topk", lr=1e-3, optimizer="adam", scheduler="linear", scheduler_params={"num_warmup_steps": 2},
                 autoencoder_config={"expansion_factor": 4, "normalize_decoder": True, "k": 8, "multi_topk": False,
                                     "auxk_alpha": 0.03125, "dead_feature_threshold": 100.0}),
}
summary = {}
for name, extra in cases.items():
    run_dir = os.path.join(OUT, "run_" + name)
    shutil.rmtree(run_dir, ignore_errors=True)
    st = train(**dict(base, **extra, run_dir=run_dir))
    eng = st["engine"]
    eng.eval(x_eval.cuda())
    m = eng.metrics()
    lat = eng.debug_read(0, 2 * T * eng.n).reshape(2 * T, eng.n)
    np.save(os.path.join(OUT, f"latent_{name}.npy"), lat)
    for ck in ("step3.pth", "step6.pth"):
        shutil.copy(os.path.join(run_dir, "checkpoints", ck), os.path.join(OUT, f"{name}_{ck}"))
    summary[name] = {"eval_metrics": [float(v) for v in m], "n_dict": eng.n, "d": d, "T": T}
    eng.close()
json.dump(summary, open(os.path.join(OUT, "summary.json"), "w"), indent=1)
shutil.rmtree(folder, ignore_errors=True)
for name in cases:
    shutil.rmtree(os.path.join(OUT, "run_" + name), ignore_errors=True)
print(json.dumps(summary))

#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): prove that the reference itself consumes checkpoints WRITTEN BY THE HIP
ENGINE (tests/golden/engine_ckpt/*.pth, produced on an MI355X by tools/make_engine_checkpoints.py):

  * src.dataset.activations.init_sae_from_checkpoint(path)  (dataset/activations.py:16-31) rebuilds the SAE and its
    CPU fp32 forward on the stored batch reproduces the losses / latents the engine reported for the same weights;
  * src.scripts.train_sae.load_checkpoint(state, path, device) (train_sae.py:265-294) pushes model / optimizer /
    scheduler state_dicts into live reference objects (torch's own load_state_dict validates every key and shape),
    and one more reference optimizer step on them runs.

Writes a log (profiles/r02_reference_consumes_checkpoint.log).  The reference's absent third-party imports are stubbed
exactly as in tests/golden/make_golden.py; no reference source is copied."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402

CK = os.path.join(ROOT, "tests", "golden", "engine_ckpt")


def main(log_path=None):
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    from src.dataset.activations import init_sae_from_checkpoint
    from src.scripts.train_sae import load_checkpoint
    from src.models.l1autoencoder import L1AutoEncoder
    from src.models.topkautoencoder import TopKAutoEncoder
    from src.models.config import L1AutoEncoderConfig, TopKAutoEncoderConfig
    from transformers import get_linear_schedule_with_warmup

    summary = json.load(open(os.path.join(CK, "summary.json")))
    x = torch.tensor(np.load(os.path.join(CK, "x_eval.npy")))
    lines = []

    def log(s):
        print(s)
        lines.append(s)

    ok = True
    for name in ("l1", "topk"):
        path = os.path.join(CK, f"{name}_step6.pth")
        model = init_sae_from_checkpoint(path, device="cpu")
        m = summary[name]["eval_metrics"]
        lat_eng = np.load(os.path.join(CK, f"latent_{name}.npy"))
        with torch.no_grad():
            out, mse = model(x, return_mse=True)
        if name == "l1":
            assert isinstance(model, L1AutoEncoder)
            got = (out.reconstruction_loss.item(), out.l1_loss.item(), mse.item())
            lat_ref = out.encoded.latent.reshape(-1, lat_eng.shape[1]).numpy()
        else:
            assert isinstance(model, TopKAutoEncoder)
            got = (out.fvu.item(), out.auxk_loss.item(), mse.item())
            lat_ref = torch.zeros(lat_eng.shape).scatter_(1, out.encoded.top_indices.reshape(-1, model.cfg.k),
                                                          out.encoded.top_acts.reshape(-1, model.cfg.k)).numpy()
        rel = [abs(a - b) / max(abs(b), 1e-12) for a, b in zip(got, m[:3])]
        lat_rel = float(np.linalg.norm(lat_ref - lat_eng) / max(np.linalg.norm(lat_ref), 1e-12))
        log(f"[{name}] init_sae_from_checkpoint OK: {type(model).__name__}; reference fp32 forward vs engine (bf16) eval on the "
            f"same weights: losses {got} vs {tuple(m[:3])} (rel {['%.1e' % r for r in rel]}), latent rel-L2 {lat_rel:.2e}")
        # fp32 reference vs bf16 engine arithmetic: 2e-2 on these tiny batches (TopK: boundary ties pick other latents)
        ok &= all(r < (2e-2 if name == "l1" else 6e-2) or (b == 0 and a == 0) for r, (a, b) in zip(rel, zip(got, m[:3])))
        ok &= lat_rel < (2e-2 if name == "l1" else 0.3)

        # resume path: the reference's own objects, load_checkpoint, one more step
        ck = torch.load(path, map_location="cpu")
        hp = ck["hparams"]
        if name == "l1":
            fresh = L1AutoEncoder(hp["activation_size"], L1AutoEncoderConfig.from_dict(hp["autoencoder_config"]))
            opt = torch.optim.RAdam(fresh.parameters(), eps=1e-5, lr=hp["lr"], weight_decay=hp["weight_decay"])
            sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=hp["steps"], eta_min=0)
        else:
            fresh = TopKAutoEncoder(hp["activation_size"], TopKAutoEncoderConfig.from_dict(hp["autoencoder_config"]))
            opt = torch.optim.Adam(fresh.parameters(), lr=hp["lr"])
            sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=hp["scheduler_params"]["num_warmup_steps"],
                                                  num_training_steps=hp["steps"])
        state = {"model": fresh, "optimizer": opt, "scheduler": sch, "step": 0, "best_val_loss": float("inf"), "hparams": {}}
        load_checkpoint(state, os.path.join(CK, f"{name}_step3.pth"), torch.device("cpu"))
        assert state["step"] == 3 and state["hparams"]["autoencoder_variant"] == name
        shapes_ok = all(opt.state[p]["exp_avg"].shape == p.shape for p in fresh.parameters())
        from torch.amp import autocast
        opt.zero_grad()
        with autocast("cpu"):
            o2 = fresh(x)
            loss = (o2.reconstruction_loss + o2.l1_loss) if name == "l1" else (o2.fvu + o2.auxk_loss)
        loss.backward()
        opt.step()
        sch.step()
        step_now = int(next(iter(opt.state.values()))["step"])
        log(f"[{name}] load_checkpoint OK: step {state['step']}, optimizer moments on the right parameters: {shapes_ok}, "
            f"one more reference step ran (optimizer step counter {step_now}, lr {sch.get_last_lr()[0]:.3e}, loss {loss.item():.4f})")
        ok &= shapes_ok and step_now == 4 and bool(torch.isfinite(loss))
    log("RESULT: " + ("PASS" if ok else "FAIL"))
    if log_path:
        open(log_path, "w").write("\n".join(lines) + "\n")
    return ok


if __name__ == "__main__":
    sys.exit(0 if main(os.path.join(ROOT, "profiles", "r02_reference_consumes_checkpoint.log")) else 1)

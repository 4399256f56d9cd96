"""Generate golden vectors by running the REAL reference (ksadov/FREUD) on CPU.

Run in the build container only (needs /root/reference; nothing here travels to the GPU box
except the .npz/.json it writes):

    python tests/golden/make_golden.py

What it does (SURVEY.md section 8c recipe):
  * stubs the reference's absent third-party imports (jaxtyping, simple_parsing, whisper,
    torchaudio, tensorboard) in sys.modules, puts /root/reference on sys.path and imports
    src.models.{l1autoencoder,topkautoencoder}, src.scripts.train_sae, src.dataset.activations;
  * step fixtures: instantiates the reference model under a seed and runs the literal
    step sequence of src/scripts/train_sae.py:429-451 (zero_grad -> autocast('cpu') forward ->
    backward -> clip_grad_norm_ -> optimizer.step -> scheduler.step) for a few steps on seeded
    batches (with planted -1.0 entries to pin the mse_loss mask), recording inputs, per-step
    losses / grad norms / lrs, first-step raw gradients, final parameters + optimizer moments;
  * train-loop fixture: writes a tiny synthetic shard directory in the collector's format
    (collect_activations.py:12-63), runs the reference's own train() on it and records the
    logged scalars, the checkpoint key structure and the final weights.

The outputs are data only (inputs + expected outputs).  No reference source text is stored.
"""
import dataclasses
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    import transformers  # noqa: F401  (must be imported before torchaudio is stubbed)

    jt = types.ModuleType("jaxtyping")

    class _Sub:
        def __class_getitem__(cls, item):
            return cls

    jt.Float = _Sub
    sys.modules["jaxtyping"] = jt

    sp = types.ModuleType("simple_parsing")

    class Serializable:
        @classmethod
        def from_dict(cls, d, drop_extra_fields=True):
            names = {f.name for f in dataclasses.fields(cls)}
            return cls(**{k: v for k, v in d.items() if k in names})

        def to_dict(self):
            return dataclasses.asdict(self)

    sp.Serializable = Serializable
    sys.modules["simple_parsing"] = sp

    wh = types.ModuleType("whisper")
    wh.load_model = lambda *a, **k: None
    wh.DecodingOptions = object
    wh.Whisper = object
    sys.modules["whisper"] = wh
    sys.modules["torchaudio"] = types.ModuleType("torchaudio")

    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        scalars = []
        texts = []

        def __init__(self, *a, **k):
            pass

        def add_scalar(self, tag, value, step):
            SummaryWriter.scalars.append((tag, float(value), int(step)))

        def add_text(self, tag, text, step=None):
            SummaryWriter.texts.append(tag)

        def add_histogram(self, *a, **k):
            pass

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    return SummaryWriter


def make_batches(seed, steps, B, T, d, plant=True):
    g = torch.Generator().manual_seed(seed)
    xs = []
    for _ in range(steps):
        z = torch.relu(torch.randn(B * T, 8, generator=g)) * 0.5
        basis = torch.randn(8, d, generator=g)
        x = (z @ basis + 0.05 * torch.randn(B * T, d, generator=g)).reshape(B, T, d)
        if plant:  # exact -1.0 entries exercise mse_loss's ignored_index mask
            idx = torch.randint(0, x.numel(), (max(3, x.numel() // 97),), generator=g)
            x.view(-1)[idx] = -1.0
        xs.append(x.contiguous())
    return xs


def run_l1_case(name, d, n, B, T, steps, optimizer, scheduler, lr, recon_alpha, seed, sched_params=None,
                weight_decay=0.0, total_steps=None):
    from src.models.config import L1AutoEncoderConfig
    from src.models.l1autoencoder import L1AutoEncoder
    from torch.amp import autocast
    from torch.optim import RAdam, Adam
    from torch.optim.lr_scheduler import CosineAnnealingLR
    from transformers import get_linear_schedule_with_warmup

    total_steps = total_steps or steps
    torch.manual_seed(seed)
    cfg = L1AutoEncoderConfig.from_dict({"n_dict_components": n, "recon_alpha": recon_alpha})
    model = L1AutoEncoder(activation_size=d, cfg=cfg)
    W0 = model.decoder.weight.detach().clone()
    b0 = model.encoder_bias.detach().clone()
    if optimizer == "radam":
        opt = RAdam(model.parameters(), eps=1e-5, lr=lr, weight_decay=weight_decay)
    else:
        opt = Adam(model.parameters(), lr=lr)
    if scheduler == "cosine":
        sch = CosineAnnealingLR(opt, T_max=total_steps, eta_min=0)
    else:
        sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=sched_params["num_warmup_steps"],
                                              num_training_steps=total_steps)
    xs = make_batches(seed + 1, steps, B, T, d)
    rec = {"l1": [], "recon": [], "gnorm": [], "lr_used": [], "mse": []}
    first = {}
    for i, x in enumerate(xs):
        opt.zero_grad()
        rec["lr_used"].append(opt.param_groups[0]["lr"])
        with autocast("cpu"):
            out, mse = model(x, return_mse=True)
            loss = out.reconstruction_loss + out.l1_loss
        loss.backward()
        if i == 0:
            first = {"dW": model.decoder.weight.grad.detach().clone(),
                     "db": model.encoder_bias.grad.detach().clone(),
                     "c": out.encoded.latent.detach().float().clone(),
                     "x_hat": out.sae_out.detach().float().clone()}
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        sch.step()
        rec["l1"].append(out.l1_loss.item())
        rec["recon"].append(out.reconstruction_loss.item())
        rec["mse"].append(mse.item())
        rec["gnorm"].append(gn.item())
    W_final = model.decoder.weight.detach().clone()      # before eval: encode() renormalises in place
    b_final = model.encoder_bias.detach().clone()
    # eval-mode forward on the last batch in fp32 (validate() on cpu uses no autocast, :162-166)
    with torch.no_grad():
        eo, emse = model(xs[-1], return_mse=True)
    osd = opt.state_dict()["state"]
    np.savez_compressed(
        os.path.join(OUT, f"{name}.npz"),
        meta=json.dumps({"variant": "l1", "d": d, "n": n, "B": B, "T": T, "steps": steps, "optimizer": optimizer,
                         "scheduler": scheduler, "lr": lr, "recon_alpha": recon_alpha, "seed": seed,
                         "total_steps": total_steps, "weight_decay": weight_decay, "clip_thresh": 1.0,
                         "num_warmup_steps": (sched_params or {}).get("num_warmup_steps", 0)}),
        x=torch.stack(xs).numpy(), W0=W0.numpy(), b0=b0.numpy(),
        l1=np.array(rec["l1"]), recon=np.array(rec["recon"]), mse=np.array(rec["mse"]),
        gnorm=np.array(rec["gnorm"]), lr_used=np.array(rec["lr_used"]),
        dW_step1=first["dW"].numpy(), db_step1=first["db"].numpy(),
        c_step1=first["c"].numpy().astype(np.float32), x_hat_step1=first["x_hat"].numpy(),
        W_final=W_final.numpy(), b_final=b_final.numpy(),
        m_b=osd[0]["exp_avg"].numpy(), v_b=osd[0]["exp_avg_sq"].numpy(),
        m_W=osd[1]["exp_avg"].numpy(), v_W=osd[1]["exp_avg_sq"].numpy(),
        eval_l1=np.array(eo.l1_loss.item()), eval_recon=np.array(eo.reconstruction_loss.item()),
        eval_mse=np.array(emse.item()), W_after_eval=model.decoder.weight.detach().numpy(),
    )
    print(f"[{name}] l1={rec['l1']} recon={rec['recon']} gnorm={rec['gnorm']}")


def run_topk_case(name, d, n, k, B, T, steps, lr, seed, auxk_alpha, dead_threshold, warmup, multi_topk=False,
                  tie_free_margin=None):
    """tie_free_margin: give up (return False, write nothing) as soon as a row of any step has its k-th and (k+1)-th
    largest pre-activation closer than this RELATIVE gap -- torch.topk has no rule for ties (topkautoencoder.py:79-81), so
    only a batch sequence without near-ties pins the selection itself (main() searches seeds for one)."""
    from src.models.config import TopKAutoEncoderConfig
    from src.models.topkautoencoder import TopKAutoEncoder
    from torch.amp import autocast
    from torch.optim import Adam
    from transformers import get_linear_schedule_with_warmup

    torch.manual_seed(seed)
    cfg = TopKAutoEncoderConfig.from_dict({"n_dict_components": n, "k": k, "auxk_alpha": auxk_alpha,
                                           "normalize_decoder": True, "multi_topk": multi_topk})
    model = TopKAutoEncoder(activation_size=d, cfg=cfg)
    sd0 = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    opt = Adam(model.parameters(), lr=lr)
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=warmup, num_training_steps=steps)
    xs = make_batches(seed + 1, steps, B, T, d, plant=False)
    nfsf = torch.zeros(n, dtype=torch.long)
    rec = {"fvu": [], "auxk": [], "gnorm": [], "lr_used": [], "mse": [], "num_dead": [], "multi": []}
    first = {}
    for i, x in enumerate(xs):
        did_fire = torch.zeros(n, dtype=torch.bool)
        opt.zero_grad()
        rec["lr_used"].append(opt.param_groups[0]["lr"])
        if tie_free_margin is not None:
            # (its own autocast region: the weight casts a no_grad call leaves in the autocast cache would cut the real
            # forward below off from the parameters' gradients)
            with autocast("cpu"), torch.no_grad():
                srt = model.pre_acts(x).float().reshape(-1, n).sort(dim=1, descending=True).values
            hi, lo = srt[:, k - 1], srt[:, k]
            if bool(((hi - lo) <= tie_free_margin * hi.abs()).any()):
                return False
        with autocast("cpu"):
            dead_mask = nfsf > dead_threshold
            out, mse = model(x, dead_mask=dead_mask, return_mse=True)
            loss = out.fvu + out.auxk_loss + out.multi_topk_fvu / 8
            did_fire[out.encoded.top_indices.flatten()] = True
            nfsf += x.shape[0] * x.shape[1]
            nfsf[did_fire] = 0
        loss.backward()
        if i == 0 or (i == steps - 1):
            tag = "first" if i == 0 else "last"
            first[tag] = {kk: p.grad.detach().clone() for kk, p in model.named_parameters()}
            first[tag]["top_indices"] = out.encoded.top_indices.detach().clone()
            first[tag]["top_acts"] = out.encoded.top_acts.detach().float().clone()
            first[tag]["dead_mask"] = dead_mask.clone()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        sch.step()
        rec["fvu"].append(out.fvu.item())
        rec["auxk"].append(out.auxk_loss.item())
        rec["multi"].append(out.multi_topk_fvu.item())
        rec["mse"].append(mse.item())
        rec["gnorm"].append(gn.item())
        rec["num_dead"].append(int(dead_mask.sum()))
    arrays = {}
    for tag, dct in first.items():
        for kk, v in dct.items():
            arrays[f"{tag}__{kk}"] = v.numpy()
    for kk, v in sd0.items():
        arrays[f"init__{kk}"] = v.numpy()
    for kk, v in model.state_dict().items():
        arrays[f"final__{kk}"] = v.detach().numpy()
    np.savez_compressed(
        os.path.join(OUT, f"{name}.npz"),
        meta=json.dumps({"variant": "topk", "d": d, "n": n, "k": k, "B": B, "T": T, "steps": steps, "lr": lr,
                         "seed": seed, "auxk_alpha": auxk_alpha, "dead_feature_threshold": dead_threshold,
                         "num_warmup_steps": warmup, "clip_thresh": 1.0, "optimizer": "adam",
                         "scheduler": "linear", "multi_topk": multi_topk}),
        x=torch.stack(xs).numpy(), fvu=np.array(rec["fvu"]), auxk=np.array(rec["auxk"]), mse=np.array(rec["mse"]),
        multi=np.array(rec["multi"]),
        gnorm=np.array(rec["gnorm"]), lr_used=np.array(rec["lr_used"]), num_dead=np.array(rec["num_dead"]),
        nfsf_final=nfsf.numpy(), **arrays)
    print(f"[{name}] fvu={rec['fvu']} auxk={rec['auxk']} dead={rec['num_dead']}")
    return True


def run_tie_free_topk_case():
    """Seed search (deterministic: first hit from seed 100 up) for a TopK trajectory on which every row of every step has
    a relative gap of > 2^-6 (two bf16 ulps at least) between its k-th and (k+1)-th pre-activation: the reference's selection is then unambiguous
    (and stays so under the engine's summation order), and the engine can be held to the reference's OWN gradients."""
    for seed in range(100, 20000):
        if run_topk_case("topk_tiefree_d64", d=64, n=128, k=4, B=2, T=4, steps=3, lr=1e-3, seed=seed, auxk_alpha=0.0,
                         dead_threshold=1e6, warmup=1, tie_free_margin=2.0 ** -6):
            print(f"[topk_tiefree_d64] seed {seed}")
            return
    raise SystemExit("no tie-free trajectory found")


def write_shards(folder, layer, n_files, T, d, seed, dtype=np.float32):
    """Same on-disk format as collect_activations.py:12-63 (one [1, T*d] row appended per file)."""
    os.makedirs(folder, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    z = torch.relu(torch.randn(n_files * T, 8, generator=g)) * 0.5
    basis = torch.randn(8, d, generator=g)
    x = (z @ basis + 0.05 * torch.randn(n_files * T, d, generator=g)).reshape(n_files, T * d)
    np.save(os.path.join(folder, f"{layer}_tensors.npy"), x.numpy().astype(dtype))
    meta = {"tensor_shape": [T, d], "activation_shape": [T, d],
            "filenames": [f"/data/audio/file_{i:04d}.flac" for i in range(n_files)]}
    with open(os.path.join(folder, f"{layer}_metadata.json"), "w") as f:
        json.dump(meta, f)
    return x.numpy().astype(dtype)


def run_train_loop_case(SummaryWriter, name, variant):
    from src.scripts.train_sae import train

    tmp = tempfile.mkdtemp(prefix="freud_golden_")
    try:
        layer = "encoder.blocks.2"
        T, d, n_files = 12, 16, 10
        data = write_shards(os.path.join(tmp, "train"), layer, n_files, T, d, seed=5)
        if variant == "l1":
            ae = {"n_dict_components": 48, "recon_alpha": 1e4}
            opt, sch, sp, lr = "radam", "cosine", {}, 4e-4
        else:
            ae = {"expansion_factor": 4, "normalize_decoder": True, "k": 4, "multi_topk": False,
                  "auxk_alpha": 0.03125, "dead_feature_threshold": 100.0}
            opt, sch, sp, lr = "adam", "linear", {"num_warmup_steps": 2}, 1e-3
        config = {
            "whisper_config": {"model": "tiny", "layer_name": layer},
            "autoencoder_variant": variant, "autoencoder_config": ae, "seed": 0,
            "train_folder": os.path.join(tmp, "train"), "val_folder": os.path.join(tmp, "train"),
            "device": torch.device("cpu"), "run_dir": os.path.join(tmp, "run"), "lr": lr, "weight_decay": 0.0,
            "steps": 7, "clip_thresh": 1.0, "batch_size": 3, "dl_max_workers": 0, "log_tb_every": 1,
            "save_every": 4, "val_every": 1000, "optimizer": opt, "scheduler": sch, "scheduler_params": sp,
            "start_checkpoint": None, "from_disk": True,
        }
        SummaryWriter.scalars.clear()
        train(**config)
        ck_dir = os.path.join(tmp, "run", "checkpoints")
        files = sorted(os.listdir(ck_dir))
        ck = torch.load(os.path.join(ck_dir, "step7.pth"), map_location="cpu", weights_only=True)
        cfg_json = dict(config)
        cfg_json["device"] = "cpu"
        cfg_json["train_folder"] = "train"
        cfg_json["val_folder"] = "train"
        cfg_json["run_dir"] = "run"
        arrays = {f"model__{k}": v.numpy() for k, v in ck["model"].items()}
        for pid, st in ck["optimizer"]["state"].items():
            for kk, v in st.items():
                arrays[f"opt__{pid}__{kk}"] = v.numpy() if torch.is_tensor(v) else np.array(v)
        np.savez_compressed(
            os.path.join(OUT, f"{name}.npz"),
            meta=json.dumps({"config": cfg_json, "checkpoint_files": files,
                             "checkpoint_keys": sorted(ck.keys()),
                             "model_keys": list(ck["model"].keys()),
                             "hparams": ck["hparams"], "step": ck["step"],
                             "best_val_loss": ck["best_val_loss"],
                             "opt_param_groups": [{k: v for k, v in g.items()} for g in ck["optimizer"]["param_groups"]],
                             "scheduler_keys": sorted(ck["scheduler"].keys()),
                             "scalars": SummaryWriter.scalars, "T": T, "d": d, "n_files": n_files,
                             "layer": layer}),
            shard=data, **arrays)
        print(f"[{name}] checkpoints={files} scalars={SummaryWriter.scalars[:6]} ...")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_sampler_case():
    """Pin the batch order of DataLoader(shuffle=True, drop_last=True) under set_seeds()."""
    from src.scripts.train_sae import set_seeds
    from src.dataset.activations import MemoryMappedActivationDataLoader

    tmp = tempfile.mkdtemp(prefix="freud_golden_")
    try:
        layer = "L"
        write_shards(tmp, layer, 11, 2, 4, seed=9)
        set_seeds(3)
        _ = torch.randn(5)  # some RNG consumption between seeding and iteration, as model init does
        dl = MemoryMappedActivationDataLoader(tmp, layer, 3, 0, None, {"shuffle": True, "drop_last": True})
        epochs = []
        for _e in range(3):
            epochs.append([[os.path.basename(f) for f in names] for (_x, names) in dl])
        with open(os.path.join(OUT, "sampler_order.json"), "w") as f:
            json.dump({"seed": 3, "n_files": 11, "batch_size": 3, "pre_draw": 5, "len": len(dl), "epochs": epochs}, f)
        print("[sampler_order]", epochs[0])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    SummaryWriter = install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])          # fixture names to (re)generate; none = all
    cases = {
        "l1_radam_cosine_d16": lambda: run_l1_case("l1_radam_cosine_d16", d=16, n=64, B=4, T=8, steps=5, optimizer="radam",
                                                   scheduler="cosine", lr=4e-4, recon_alpha=1e4, seed=0, total_steps=100),
        "l1_adam_linear_d48": lambda: run_l1_case("l1_adam_linear_d48", d=48, n=200, B=3, T=20, steps=6, optimizer="adam",
                                                  scheduler="linear", lr=1e-3, recon_alpha=1.0, seed=1,
                                                  sched_params={"num_warmup_steps": 3}, total_steps=20),
        "l1_radam_wd_d32": lambda: run_l1_case("l1_radam_wd_d32", d=32, n=96, B=2, T=16, steps=8, optimizer="radam",
                                               scheduler="cosine", lr=1e-3, recon_alpha=1e2, seed=2, weight_decay=0.01,
                                               total_steps=8),
        "l1_radam_cosine_d384": lambda: run_l1_case("l1_radam_cosine_d384", d=384, n=256, B=2, T=64, steps=3,
                                                    optimizer="radam", scheduler="cosine", lr=4e-4, recon_alpha=1e4, seed=3,
                                                    total_steps=100),
        "topk_adam_linear_d16": lambda: run_topk_case("topk_adam_linear_d16", d=16, n=64, k=4, B=3, T=8, steps=6, lr=1e-3,
                                                      seed=4, auxk_alpha=0.03125, dead_threshold=40.0, warmup=2),
        "topk_adam_linear_d64": lambda: run_topk_case("topk_adam_linear_d64", d=64, n=512, k=16, B=2, T=32, steps=4, lr=1e-4,
                                                      seed=5, auxk_alpha=0.0, dead_threshold=1e6, warmup=2),
        "topk_multi_d32": lambda: run_topk_case("topk_multi_d32", d=32, n=256, k=8, B=2, T=16, steps=5, lr=1e-3, seed=6,
                                                auxk_alpha=0.03125, dead_threshold=40.0, warmup=2, multi_topk=True),
        "topk_tiefree_d64": run_tie_free_topk_case,
        "trainloop_l1": lambda: run_train_loop_case(SummaryWriter, "trainloop_l1", "l1"),
        "trainloop_topk": lambda: run_train_loop_case(SummaryWriter, "trainloop_topk", "topk"),
        "sampler_order": run_sampler_case,
    }
    unknown = only - set(cases)
    if unknown:
        raise SystemExit(f"unknown fixture(s) {sorted(unknown)}; known: {sorted(cases)}")
    for name, fn in cases.items():
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()

"""Drop-in for the reference's `python -m src.scripts.train_sae --config <json>`
(src/scripts/train_sae.py:297-615): same JSON schema (keys == train() kwargs), same run_dir layout
(run_dir/checkpoints/step{N}.pth, bestval.pth), same checkpoint keys, same error conventions.
The step body (train_sae.py:429-451) runs in the HIP engine (freud_amd/engine.py -> libfreud_sae.so);
this file is orchestration only: seeds, loader, LR schedule, logging, checkpoints, data-parallel
all-reduce.  There is no CPU implementation of the step here: without the engine it fails loudly.

Data parallel: launch one process per GPU with torch.distributed.run; rank r trains on
perm[r::R] of every epoch permutation with the per-GPU batch_size of the config.  The step is exact:
the batch statistics the losses normalise by (unmasked-entry count and rows; for TopK the column sums
behind total_variance) are summed over the ranks first, every rank's backward normalises by the
global values, and the summed gradients are the whole batch's (include/freud_sae.h).  On GPUs the
engine does all of it itself (no Python in the step): by default with its own exchange kernels over hipIpc
peer mappings (freud_amd/dp.py, csrc/p2p_exchange.h), optionally on its own RCCL communicator
(FREUD_DP=rccl); a host-driven variant of the same protocol over torch.distributed serves CPU tests
(gloo), FREUD_DP=host and every case where the peers cannot be mapped.
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import random
import sys
import time
from typing import Callable, Dict, Optional

import numpy as np
import torch

from freud_amd.config import L1AutoEncoderConfig, TopKAutoEncoderConfig, get_n_dict_components
from freud_amd.loader import MemoryMappedActivationDataLoader

EXIT_EXCHANGE_FAILED = 3      # main(): the data-parallel exchange failed mid-run; restart fresh processes from the last good checkpoint


# ------------------------------------------------------------------------------------------------
# pieces with the reference's names
# ------------------------------------------------------------------------------------------------
def set_seeds(seed=42):
    """train_sae.py:224-229."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def init_dataloader(from_disk: bool, data_path: str, whisper_model: str, sae_checkpoint: Optional[str], layer_name: str,
                    device, batch_size: int, dl_max_workers: int, subset_size: Optional[int], dl_kwargs: dict,
                    rank: int = 0, world_size: int = 1):
    """train_sae.py:32-67.  Only the from_disk=true branch is in scope (the on-the-fly branch needs
    Whisper itself)."""
    if not from_disk:
        raise NotImplementedError("from_disk=false runs Whisper on the fly (FlyActivationDataLoader); this engine "
                                  "trains from pre-collected activation shards only")
    loader = MemoryMappedActivationDataLoader(data_path=data_path, layer_name=layer_name, batch_size=batch_size,
                                              dl_max_workers=dl_max_workers, subset_size=subset_size,
                                              dl_kwargs=dl_kwargs, device=device, rank=rank, world_size=world_size)
    return loader, loader.activation_shape[-1], loader.dataset_length


def lr_at(step_index: int, base_lr: float, scheduler: str, steps: int, scheduler_params: dict) -> float:
    """Learning rate of the optimizer step with 0-based index `step_index` (= number of
    scheduler.step() calls so far): CosineAnnealingLR(T_max=steps, eta_min=0) closed form
    (train_sae.py:383-384) or get_linear_schedule_with_warmup (train_sae.py:385-390)."""
    t = step_index
    if scheduler == "cosine":
        return base_lr * (1 + math.cos(math.pi * t / steps)) / 2
    if scheduler == "linear":
        w = scheduler_params["num_warmup_steps"]          # KeyError if missing, like the reference (:388)
        if t < w:
            return base_lr * float(t) / float(max(1, w))
        return base_lr * max(0.0, float(steps - t) / float(max(1, steps - w)))
    raise ValueError(f"Invalid scheduler: {scheduler}, must be 'cosine' or 'linear'")


def _torch_state_dicts(param_shapes: Dict[str, tuple], order, optimizer: str, base_lr: float, weight_decay: float,
                       scheduler: str, steps: int, scheduler_params: dict, step: int, exp_avg, exp_avg_sq):
    """optimizer.state_dict() / scheduler.state_dict() exactly as torch would write them
    (train_sae.py:232-251), built from real torch objects over CPU tensors so the reference's
    load_checkpoint (:265-294) accepts them."""
    params = [torch.nn.Parameter(torch.zeros(param_shapes[k]), requires_grad=True) for k in order]
    if optimizer == "radam":
        opt = torch.optim.RAdam(params, eps=1e-5, lr=base_lr, weight_decay=weight_decay)
    elif optimizer == "adam":
        opt = torch.optim.Adam(params, lr=base_lr)
    else:
        raise ValueError(f"Invalid optimizer: {optimizer}, must be 'radam' or 'adam'")
    if scheduler == "cosine":
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps, eta_min=0)
    elif scheduler == "linear":
        w = scheduler_params["num_warmup_steps"]
        sch = torch.optim.lr_scheduler.LambdaLR(
            opt, lambda t: float(t) / float(max(1, w)) if t < w else max(0.0, float(steps - t) / float(max(1, steps - w))))
    else:
        raise ValueError(f"Invalid scheduler: {scheduler}, must be 'cosine' or 'linear'")
    cur_lr = lr_at(step, base_lr, scheduler, steps, scheduler_params)
    if step > 0:
        for p, k in zip(params, order):
            opt.state[p] = {"step": torch.tensor(float(step)),
                            "exp_avg": torch.from_numpy(np.ascontiguousarray(exp_avg[k])).clone(),
                            "exp_avg_sq": torch.from_numpy(np.ascontiguousarray(exp_avg_sq[k])).clone()}
    sch.last_epoch = step
    sch._step_count = step + 1
    sch._last_lr = [cur_lr]
    opt.param_groups[0]["lr"] = cur_lr
    return opt.state_dict(), sch.state_dict()


def _resume_path(checkpoint_path: str) -> str:
    """<run_dir>/checkpoints/<name> -> <run_dir>/resume/<name>"""
    ck_dir, name = os.path.split(os.path.abspath(checkpoint_path))
    return os.path.join(os.path.dirname(ck_dir), "resume", name)


def save_checkpoint(state: dict, save_path: str) -> None:
    """train_sae.py:232-251: torch.save of {model, optimizer, scheduler, step, best_val_loss, hparams};
    tensors / plain containers / scalars only (loadable with weights_only=True)."""
    eng = state["engine"]
    order = state["param_order"]
    params = eng.get_params()
    step, m1, m2 = eng.get_opt_state()
    opt_sd, sch_sd = _torch_state_dicts(eng.param_shapes(), order, state["optimizer"], state["lr"], state["weight_decay"],
                                        state["scheduler"], state["steps"], state["scheduler_params"], state["step"], m1, m2)
    checkpoint = {
        "model": {k: torch.from_numpy(params[k].copy()) for k in state["state_dict_order"]},
        "optimizer": opt_sd, "scheduler": sch_sd, "step": state["step"],
        "best_val_loss": state["best_val_loss"], "hparams": state["hparams"],
    }
    torch.save(checkpoint, save_path)
    # Extension: what the reference loses on resume (train_sae.py:396-415) - the data
    # order position and the TopK dead-latent counters - goes to a side file <run_dir>/resume/<same name>, so that the
    # checkpoint itself keeps exactly the reference's keys.  Tensors and ints only (weights_only loading works).
    resume = {"step": int(state["step"]), "epoch_rng_state": state["epoch_rng_state"].clone(),
              "epoch_batches_done": int(state["epoch_batches_done"]), "world_size": int(state.get("world_size", 1))}
    if hasattr(eng, "get_topk_state") and state["hparams"]["autoencoder_variant"] == "topk":
        resume["num_frames_since_fired"] = torch.from_numpy(np.asarray(eng.get_topk_state(), dtype=np.int64).copy())
    side = _resume_path(save_path)
    os.makedirs(os.path.dirname(side), exist_ok=True)
    torch.save(resume, side)


def load_checkpoint(state: dict, load_path: str, device=None) -> None:
    """train_sae.py:265-294: push model / optimizer / step / best_val_loss / hparams of a checkpoint
    (ours or the reference's own) into the live engine."""
    checkpoint = torch.load(load_path, map_location="cpu")
    eng = state["engine"]
    order = state["param_order"]
    eng.set_params({k: v.detach().cpu().numpy() for k, v in checkpoint["model"].items()})
    ost = checkpoint["optimizer"]["state"]
    if len(ost) > 0:
        shapes = eng.param_shapes()
        for i, k in enumerate(order):           # a mis-indexed state (wrong parameter order) must not be copied blindly
            for mk in ("exp_avg", "exp_avg_sq"):
                if tuple(ost[i][mk].shape) != tuple(shapes[k]):
                    raise ValueError(f"checkpoint optimizer state[{i}][{mk}] has shape {tuple(ost[i][mk].shape)}, "
                                     f"parameter {k} has {tuple(shapes[k])}")
        m1 = {k: ost[i]["exp_avg"].numpy() for i, k in enumerate(order)}
        m2 = {k: ost[i]["exp_avg_sq"].numpy() for i, k in enumerate(order)}
        eng.set_opt_state(int(float(ost[0]["step"])), m1, m2)
    for k in ("step", "best_val_loss", "hparams"):
        state[k] = checkpoint[k]
    # a side file written by save_checkpoint carries the data position and the TopK counters; without it (the
    # reference's own checkpoints) the run restarts its data order from the seed and the counters from zero, exactly
    # as the reference does
    side = _resume_path(load_path)
    if os.path.exists(side):
        resume = torch.load(side, map_location="cpu")
        if int(resume.get("step", -1)) == int(state["step"]) and int(resume.get("world_size", 1)) == int(state.get("world_size", 1)):
            state["resume_rng_state"] = resume["epoch_rng_state"]
            state["resume_skip"] = int(resume["epoch_batches_done"])
            if "num_frames_since_fired" in resume and hasattr(eng, "set_topk_state"):
                eng.set_topk_state(resume["num_frames_since_fired"].numpy())
    del checkpoint
    gc.collect()


class _Logger:
    """TensorBoard scalars with the reference's tags (train_sae.py:465-489, 524-583) when the
    tensorboard package is importable; always a metrics.jsonl next to them."""

    def __init__(self, run_dir: str, enabled: bool):
        self.enabled = enabled
        self.tb = None
        self.f = None
        if not enabled:
            return
        os.makedirs(run_dir, exist_ok=True)
        self.f = open(os.path.join(run_dir, "metrics.jsonl"), "a")
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.tb = SummaryWriter(run_dir, flush_secs=10)
        except Exception:
            self.tb = None

    def add_scalar(self, tag: str, value, step: int) -> None:
        if not self.enabled:
            return
        self.f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        self.f.flush()
        if self.tb is not None:
            self.tb.add_scalar(tag, float(value), step)

    def add_text(self, tag: str, text: str, step: int = 0) -> None:
        if self.enabled and self.tb is not None:
            self.tb.add_text(tag, text, step)

    def add_histogram(self, tag: str, values, step: int) -> None:
        if self.enabled and self.tb is not None:
            self.tb.add_histogram(tag, values, step)

    def close(self) -> None:
        if self.f:
            self.f.close()
        if self.tb is not None:
            self.tb.close()


def validate(engine, val_folder: str, device, layer_name: str, from_disk: bool, variant: str):
    """Numeric part of train_sae.py:121-221: per-file (batch 1, unshuffled) forward with return_mse,
    means of the per-file losses, per-feature max |latent| over time -> maxes / stds over files.
    (The Whisper-transcript part of the reference's validate needs Whisper weights: out of scope.)"""
    loader, _, _ = init_dataloader(from_disk, val_folder, "", None, layer_name, device, 1, 1, None, {"shuffle": False})
    recon, l1, mses, maxes, multi = [], [], [], [], []
    if hasattr(engine, "eval_into"):
        # every file's loss scalars and per-feature maxima stay in device rows; ONE read-back at the end (the reference
        # synchronises four times per file with .item(), train_sae.py:173-190)
        n_files = len(loader)
        met = torch.zeros((max(n_files, 1), 8), dtype=torch.float32, device=device)
        cmx = torch.zeros((max(n_files, 1), engine.n), dtype=torch.float32, device=device)
        i = 0
        for acts, _names in loader:
            engine.eval_into(acts, met[i], cmx[i])
            i += 1
        m = met[:i].cpu().numpy()          # the one synchronisation
        recon, l1, mses, multi = list(m[:, 0]), list(m[:, 1]), list(m[:, 2]), list(m[:, 6])
        maxes = list(cmx[:i].cpu().numpy())
    else:
        for acts, _names in loader:
            engine.eval(acts)
            m = engine.metrics()
            recon.append(float(m[0]))
            l1.append(float(m[1]))
            mses.append(float(m[2]))
            multi.append(float(m[6]))
            maxes.append(engine.latent_colmax())
    mag = np.stack(maxes) if maxes else np.zeros((0, engine.n), np.float32)
    losses = {"l1": float(np.mean(l1)) if variant == "l1" and l1 else None,
              "recon": float(np.mean(recon)) if variant == "l1" and recon else None,
              "fvu": float(np.mean(recon)) if variant == "topk" and recon else None,
              "auxk_loss": float(np.mean(l1)) if variant == "topk" and l1 else None,
              "multi_topk_fvu": (float(np.mean(multi)) if multi else 0.0) if variant == "topk" else None,
              "mse": float(np.mean(mses)) if mses else float("nan")}
    mag_max = mag.max(axis=0) if len(mag) else np.zeros(engine.n, np.float32)
    mag_std = mag.std(axis=0, ddof=1) if len(mag) > 1 else np.zeros(engine.n, np.float32)
    return losses, mag_max, mag_std


def _default_engine_factory(**kw):
    from freud_amd.engine import SaeEngine
    return SaeEngine(**kw)


def _dist_env():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world


# ------------------------------------------------------------------------------------------------
# train()
# ------------------------------------------------------------------------------------------------
def train(seed: int, train_folder: str, val_folder: str, device, run_dir: str, lr: float, weight_decay: float,
          steps: int, clip_thresh: float, batch_size: int, dl_max_workers: int, log_tb_every: int, save_every: int,
          val_every: int, start_checkpoint: Optional[str], whisper_config: dict, optimizer: str, scheduler: str,
          scheduler_params: dict, from_disk: bool, autoencoder_variant: str, autoencoder_config: dict, *,
          eval_precision: Optional[str] = None, engine_factory: Optional[Callable] = None, dist_backend: Optional[str] = None):
    """Same keyword arguments as the reference's train() (train_sae.py:297-320).  One OPTIONAL key beyond them: "eval_precision":
    "fp32" makes validate() -- and with it the choice of bestval.pth -- compute what the reference's validate() computes on
    device='cpu' (no autocast: fp32 end to end, train_sae.py:162-166); "bf16" (default; also FREUD_EVAL_PRECISION) keeps the training
    kernels' arithmetic, which agrees with that to ~1e-2.  A config file without the key is a reference config file.  The two trailing
    keyword-only arguments are test hooks and are never present in a config file."""
    eval_precision = eval_precision or os.environ.get("FREUD_EVAL_PRECISION") or "bf16"
    if eval_precision not in ("bf16", "fp32"):
        raise ValueError(f"Invalid eval_precision: {eval_precision}, must be 'bf16' or 'fp32'")
    device = torch.device(device)
    rank, local_rank, world = _dist_env()
    # FREUD_FORCE_DIST=1 (test hook, like bench.py --force-dist): take the data-parallel code path - process group,
    # gradient-ready callback, asynchronous all-reduce, separate optimizer call - with a single rank
    use_dist = world > 1 or os.environ.get("FREUD_FORCE_DIST") == "1"
    if engine_factory is None:
        if device.type != "cuda":
            raise RuntimeError(f"device={device}: the train step exists only as HIP kernels for MI355X; "
                               "set \"device\": \"cuda\" (no CPU fallback)")
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        engine_factory = _default_engine_factory
    dist = None
    if world > 1:
        # the buffers the peers of the in-engine exchange read (gradient, its bf16 copy, statistics) in FINE-GRAINED device
        # memory: a peer then never holds their lines non-coherently in its L2, so the exchange does not depend on cache
        # maintenance for correctness (it keeps its fences all the same).  Free on one GPU (C2 and C4 measured,
        # profiles/r04_ab_finegrained_*.txt); read by sae_create, hence set before the engine exists.
        os.environ.setdefault("FREUD_P2P_FINEGRAINED", "1")
    if use_dist:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", str(rank))
            os.environ.setdefault("WORLD_SIZE", str(world))
            os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl_debug_%h_%p.log")   # RCCL logs to stdout otherwise
            os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")           # gradient buffer = engine memory
            # (FREUD_DIST_BACKEND=gloo: host channel only -- the peer exchange needs nothing else, and two RCCL ranks cannot
            # share one GPU: how tests/test_dp_gpu.py runs the CLI with two processes on a 1-GPU box)
            backend = dist_backend or os.environ.get("FREUD_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
            kw = {"device_id": device} if backend == "nccl" else {}
            dist.init_process_group(backend, **kw)

    set_seeds(seed)
    dl_kwargs = {"shuffle": True, "drop_last": True}
    train_loader, feat_dim, dset_len = init_dataloader(from_disk, train_folder, whisper_config["model"], None,
                                                       whisper_config["layer_name"], device, batch_size, dl_max_workers,
                                                       None, dl_kwargs, rank=rank, world_size=world)
    hparam_dict = {
        "autoencoder_variant": autoencoder_variant, "autoencoder_config": autoencoder_config, "lr": lr,
        "weight_decay": weight_decay, "steps": steps, "clip_thresh": clip_thresh, "batch_size": batch_size,
        "whisper_config": whisper_config, "activation_size": feat_dim, "train_folder": train_folder,
        "val_folder": val_folder, "optimizer": optimizer, "scheduler": scheduler, "scheduler_params": scheduler_params,
    }
    assert autoencoder_variant in ["l1", "topk"], \
        f"Invalid autoencoder variant: {autoencoder_variant}, must be 'l1' or 'topk'"
    if optimizer not in ("radam", "adam"):
        raise ValueError(f"Invalid optimizer: {optimizer}, must be 'radam' or 'adam'")
    if scheduler not in ("cosine", "linear"):
        raise ValueError(f"Invalid scheduler: {scheduler}, must be 'cosine' or 'linear'")
    if scheduler == "linear":
        scheduler_params["num_warmup_steps"]        # KeyError like train_sae.py:388

    T = int(train_loader.dataset.tensor_shape[-2])
    max_rows = batch_size * T
    try:
        val_ds_T = int(MemoryMappedActivationDataLoader(val_folder, whisper_config["layer_name"], 1).dataset.tensor_shape[-2])
        max_rows = max(max_rows, val_ds_T)
    except (OSError, KeyError, IndexError, ValueError):
        pass        # no (readable) validation shard: validate() will say so itself when it is reached
    if autoencoder_variant == "l1":
        cfg = L1AutoEncoderConfig.from_dict(autoencoder_config)
        n_dict = get_n_dict_components(feat_dim, cfg.expansion_factor, cfg.n_dict_components)
        eng = engine_factory(variant="l1", d_model=feat_dim, n_dict=n_dict, max_rows=max_rows, optimizer=optimizer,
                             recon_alpha=cfg.recon_alpha, clip_thresh=clip_thresh, weight_decay=weight_decay,
                             device_id=(device.index or 0) if device.type == "cuda" else 0)
        # model init exactly as L1AutoEncoder.__init__ (l1autoencoder.py:56-63): consumes the global RNG
        # like nn.Linear(n, d, bias=False) (kaiming init draw) followed by orthogonal_.
        lin = torch.nn.Linear(n_dict, feat_dim, bias=False)
        torch.nn.init.orthogonal_(lin.weight)
        init = {"decoder.weight": lin.weight.detach().numpy(), "encoder_bias": np.zeros(n_dict, np.float32)}
        param_order = ["encoder_bias", "decoder.weight"]          # model.parameters() order -> optimizer indices
        state_dict_order = ["encoder_bias", "decoder.weight"]     # state_dict key order
    else:
        cfg = TopKAutoEncoderConfig.from_dict(autoencoder_config)
        n_dict = get_n_dict_components(feat_dim, cfg.expansion_factor, cfg.n_dict_components)
        eng = engine_factory(variant="topk", d_model=feat_dim, n_dict=n_dict, max_rows=max_rows, optimizer=optimizer,
                             k=cfg.k, auxk_alpha=cfg.auxk_alpha, clip_thresh=clip_thresh, weight_decay=weight_decay,
                             multi_topk=bool(cfg.multi_topk),
                             device_id=(device.index or 0) if device.type == "cuda" else 0)
        enc = torch.nn.Linear(feat_dim, n_dict)                   # topkautoencoder.py:62-70
        enc.bias.data.zero_()
        W_dec = enc.weight.data.clone()
        if cfg.normalize_decoder:
            eps = torch.finfo(W_dec.dtype).eps
            W_dec /= torch.norm(W_dec, dim=1, keepdim=True) + eps
        init = {"encoder.weight": enc.weight.detach().numpy(), "encoder.bias": enc.bias.detach().numpy(),
                "W_dec": W_dec.numpy(), "b_dec": np.zeros(feat_dim, np.float32)}
        # nn.Module.parameters() yields the module's own parameters (W_dec, b_dec) before its children's (encoder.*):
        # that order is the optimizer's parameter indexing (state[0] = W_dec ... state[3] = encoder.bias)
        param_order = ["W_dec", "b_dec", "encoder.weight", "encoder.bias"]
        state_dict_order = ["W_dec", "b_dec", "encoder.weight", "encoder.bias"]
        eng.set_topk_options(float(autoencoder_config["dead_feature_threshold"]), T)   # raw key (:438); T for x.mean(0)
    eng.set_params(init)
    if eval_precision == "fp32":
        if not hasattr(eng, "set_eval_precision"):
            raise RuntimeError("eval_precision='fp32' needs the HIP engine (sae_set_eval_precision)")
        eng.set_eval_precision("fp32")

    is_main = rank == 0
    checkpoint_out_dir = run_dir + "/checkpoints"
    if is_main:
        os.makedirs(run_dir, exist_ok=True)
        os.makedirs(checkpoint_out_dir, exist_ok=True)
    logger = _Logger(run_dir, is_main)
    logger.add_text("hparams", json.dumps(hparam_dict, indent=4))
    n_params = sum(int(np.prod(s)) for s in eng.param_shapes().values())
    if is_main:
        print("Model: %.2fM" % (n_params / 1.0e6))

    state = {"engine": eng, "param_order": param_order, "state_dict_order": state_dict_order, "optimizer": optimizer,
             "scheduler": scheduler, "lr": lr, "weight_decay": weight_decay, "steps": steps,
             "scheduler_params": scheduler_params, "step": 0, "best_val_loss": float("inf"), "hparams": hparam_dict,
             "world_size": world, "epoch_rng_state": torch.get_rng_state(), "epoch_batches_done": 0}
    if start_checkpoint is not None:
        if is_main:
            print(f"Checkpoint: {start_checkpoint}")
        load_checkpoint(state, start_checkpoint, device=device)

    # Data parallel: (a) in-engine -- the context gets its own RCCL communicator and step() runs statistics, gradient
    # all-reduces (overlapped with the backward) and the optimizer without returning to Python; (b) host-driven -- the
    # same protocol through torch.distributed: statistics all-reduce, forward_backward, gradient all-reduce (ranges as
    # the engine announces them when it can), optimizer.
    grads, works, overlap, in_engine = None, [], False, False
    auditor, audit_first, last_good = None, 0, start_checkpoint

    def guard(what):
        """Before anything is persisted or reported: the in-engine exchange has not failed (local, synchronising) and -- with
        more than one rank -- the replicas are bit-identical (collective).  Raises dp.ExchangeError; ADVICE r3: a rank that
        sailed on after a peer gave up must never reach save_checkpoint."""
        if in_engine:
            try:
                eng.dist_check()
            except Exception as e:          # noqa: BLE001 -- EngineError text -> one exception type for the caller
                raise dp.ExchangeError(f"[rank {rank}] step {state['step']} ({what}): {e}") from e
        if use_dist and world > 1 and hasattr(eng, "param_checksum"):
            dp.check_replicas(eng, dist, rank, world, state["step"])

    if use_dist:
        from freud_amd import dp
        dp_mode = dp.setup(eng, dist, rank, world, device, mode=dp.requested_mode(),
                           payload=os.environ.get("FREUD_DP_PAYLOAD", "float32"),
                           overlap=int(os.environ.get("FREUD_DP_OVERLAP", "1")))
        in_engine = dp_mode in ("p2p", "rccl")
        if is_main:
            print(f"data parallel: {world} ranks, exchange = {dp_mode}")
        if dp_mode == "p2p" and world > 1 and os.environ.get("FREUD_DP_AUDIT", "1") != "0":
            # sum check of the peer exchange against torch.distributed on real gradients (freud_amd/dp.py, level 2): the first
            # FREUD_DP_AUDIT_STEPS steps and every logging step
            auditor = dp.Auditor(eng, dist, rank, world, payload=os.environ.get("FREUD_DP_PAYLOAD", "float32"))
            audit_first = int(os.environ.get("FREUD_DP_AUDIT_STEPS", "3"))
        if not in_engine:
            grads = eng.grad_tensor()
            overlap = hasattr(eng, "set_grad_ready_callback") and dist.get_backend() == "nccl"
            if overlap:
                # every range of the gradient buffer is all-reduced (communication stream) as soon as the engine
                # reports it final, i.e. under the backward kernels that are still to run
                eng.set_grad_ready_callback(lambda off, cnt: works.append(dist.all_reduce(grads[off:off + cnt], async_op=True)))
    t_start, rows_done = time.time(), 0
    while state["step"] < steps:
        n_batches = 0
        if "resume_rng_state" in state:       # continue the interrupted epoch: same permutation, seen batches dropped
            torch.set_rng_state(state.pop("resume_rng_state"))
            train_loader.skip_next = n_batches = state.pop("resume_skip")
        state["epoch_rng_state"] = torch.get_rng_state()      # the state the epoch's permutation is drawn from
        state["epoch_batches_done"] = n_batches
        for activations, _ in train_loader:
            n_batches += 1
            state["epoch_batches_done"] = n_batches
            step_lr = lr_at(state["step"], lr, scheduler, steps, scheduler_params)
            if use_dist and not in_engine:
                eng.batch_stats(activations)
                dist.all_reduce(eng.stats_tensor())         # what the losses normalise by, over the whole batch
                eng.forward_backward(activations)
                if overlap:
                    for w in works:                         # the compute stream waits for the communication stream
                        w.wait()
                    works.clear()
                else:
                    dist.all_reduce(grads)                  # sum of [grads | loss shares (| did_fire)] over ranks
                eng.optimizer_step(step_lr, 1.0)            # the summed gradient IS the whole batch's: no 1/R
            else:
                audit_now = auditor is not None and (state["step"] < audit_first or (state["step"] + 1) % log_tb_every == 0)
                if audit_now:
                    auditor.arm()
                eng.step(activations, step_lr)
                if audit_now:
                    auditor.verify(state["step"] + 1)       # raises dp.ExchangeError on a wrong sum
                elif in_engine and hasattr(eng, "dist_poll"):
                    try:
                        eng.dist_poll()                     # host-mapped failure word: no sync, no launch
                    except Exception as e:                  # noqa: BLE001
                        raise dp.ExchangeError(f"[rank {rank}] step {state['step'] + 1}: {e}") from e
            state["step"] += 1
            rows_done += activations.shape[0] * activations.shape[1] * world

            if state["step"] % log_tb_every == 0:           # the only device->host sync of the loop
                guard("logging")                            # a peer that never arrived / diverged replicas: stop here
                m = eng.metrics()
                if autoencoder_variant == "l1":
                    logger.add_scalar("train/loss", float(m[0]) + float(m[1]), state["step"])
                    logger.add_scalar("train/loss_recon", m[0], state["step"])
                    logger.add_scalar("train/loss_l1", m[1], state["step"])
                else:
                    # loss = out.fvu + out.auxk_loss + out.multi_topk_fvu / 8  (train_sae.py:442)
                    logger.add_scalar("train/loss", float(m[0]) + float(m[1]) + float(m[6]) / 8, state["step"])
                    logger.add_scalar("train/fvu", m[0], state["step"])
                    logger.add_scalar("train/auxk_loss", m[1], state["step"])
                    logger.add_scalar("train/multi_topk_fvu", m[6], state["step"])
                    logger.add_scalar("train/dead_pct", m[5], state["step"])
                logger.add_scalar("train/lr", lr_at(state["step"], lr, scheduler, steps, scheduler_params), state["step"])
                logger.add_scalar("train/grad_norm", m[3], state["step"])
                logger.add_scalar("train/activations_per_sec", rows_done / max(time.time() - t_start, 1e-9), state["step"])

            if state["step"] % save_every == 0:
                guard("checkpoint")                         # nothing is written from a failed or diverged run
                if is_main:
                    save_checkpoint(state, checkpoint_out_dir + "/step" + str(state["step"]) + ".pth")
                last_good = checkpoint_out_dir + "/step" + str(state["step"]) + ".pth"
                state["last_good_checkpoint"] = last_good
                if use_dist:
                    dist.barrier()      # rank 0 wrote for a while: re-align before the next step's in-engine exchange (it times out)

            if state["step"] % val_every == 0:
                guard("validation")
                if is_main:
                    print("Validating...")
                losses_dict, mag_max, mag_std = validate(eng, val_folder, device, whisper_config["layer_name"],
                                                         from_disk, autoencoder_variant)
                if autoencoder_variant == "l1":
                    if is_main:
                        print(f"{state['step']} validation, loss_recon={losses_dict['recon']}, "
                              f"loss_l1={losses_dict['l1']}, mse={losses_dict['mse']}")
                    logger.add_scalar("val/loss_recon", losses_dict["recon"], state["step"])
                    logger.add_scalar("val/loss_l1", losses_dict["l1"], state["step"])
                else:
                    if is_main:
                        print(f"{state['step']} validation, fvu={losses_dict['fvu']}, "
                              f"auxk_loss={losses_dict['auxk_loss']}, mse={losses_dict['mse']}")
                    logger.add_scalar("val/fvu", losses_dict["fvu"], state["step"])
                    logger.add_scalar("val/auxk_loss", losses_dict["auxk_loss"], state["step"])
                    logger.add_scalar("val/multi_topk_fvu", losses_dict["multi_topk_fvu"], state["step"])
                logger.add_scalar("val/mse", losses_dict["mse"], state["step"])
                logger.add_histogram("val/encoded/magnitude_maxes", np.array(mag_max), state["step"])
                logger.add_histogram("val/encoded/magnitude_stds", np.array(mag_std), state["step"])
                num_dead = int(np.count_nonzero(mag_max <= 0))
                logger.add_scalar("val/encoded/num_dead", num_dead, state["step"])
                logger.add_scalar("val/encoded/percent_dead", num_dead / mag_max.shape[-1], state["step"])
                save_loss = losses_dict["recon"] if autoencoder_variant == "l1" else losses_dict["fvu"]
                if save_loss < state["best_val_loss"]:
                    state["best_val_loss"] = save_loss
                    if is_main:
                        print("Saving new best validation")
                        save_checkpoint(state, checkpoint_out_dir + "/bestval.pth")
                        # the reference also pickles the whole nn.Module to run_dir + "/mo.bestval"
                        # (model_out[:-3] + ".bestval", train_sae.py:370,594-595; nothing reads it):
                        # we write the state_dict there instead of a pickled module.
                        params = eng.get_params()
                        torch.save({k: torch.from_numpy(params[k].copy()) for k in state_dict_order},
                                   (run_dir + "/model")[:-3] + ".bestval")
                if use_dist:
                    dist.barrier()      # (rank 0 may have written bestval.pth)

            if state["step"] >= steps:
                break
        if n_batches == 0:
            raise RuntimeError(f"train loader yields no batches: {dset_len} files, batch_size {batch_size}, world {world}")
        guard("epoch end")
        if is_main:    # epoch-end checkpoint (train_sae.py:600-602)
            save_checkpoint(state, checkpoint_out_dir + "/step" + str(state["step"]) + ".pth")
        last_good = checkpoint_out_dir + "/step" + str(state["step"]) + ".pth"
        state["last_good_checkpoint"] = last_good
        if use_dist:
            dist.barrier()
    logger.close()
    if use_dist:
        guard("end of run")
        dist.barrier()
    if auditor is not None:
        state["exchange_audits_passed"] = auditor.passed
    return state


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, required=True, help="Path to train configuration file")
    args = parser.parse_args(argv)
    with open(args.config, "r") as f:
        config = json.load(f)
    config["device"] = torch.device(config["device"])
    try:
        train(**config)
    except Exception as e:
        from freud_amd import dp
        if not isinstance(e, dp.ExchangeError):
            raise
        # Mid-run failure of the data-parallel exchange (a peer died or timed out, replicas diverged, an audit mismatch): nothing
        # was written after the failure (every write is behind guard()).  This process has touched the GPU and must not re-exec
        # itself: exit non-zero and let the launcher start FRESH processes from the last good checkpoint with another carrier.
        run_dir = config.get("run_dir", "")
        ckpts = []
        if os.path.isdir(os.path.join(run_dir, "checkpoints")):
            ckpts = sorted((f for f in os.listdir(os.path.join(run_dir, "checkpoints")) if f.startswith("step") and f.endswith(".pth")),
                           key=lambda f: int(f[4:-4]))
        last = os.path.join(run_dir, "checkpoints", ckpts[-1]) if ckpts else config.get("start_checkpoint")
        print(f"FATAL: data-parallel exchange failed: {e}\n"
              f"last good checkpoint: {last}\n"
              f"restart fresh processes with FREUD_DP=host (or rccl) and \"start_checkpoint\": {json.dumps(last)}", file=sys.stderr)
        sys.exit(EXIT_EXCHANGE_FAILED)


if __name__ == "__main__":
    main()

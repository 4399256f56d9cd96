"""Data-parallel set-up shared by train() and bench.py: which exchange a run uses, and the collective hand-shake that makes
every rank take the same one.

The reference is single-process (src/scripts/train_sae.py:421-453 never leaves one device); data parallelism is what
BASELINE.json's north_star adds around `loss.backward()` / `clip_grad_norm_` (train_sae.py:448-449).  Three interchangeable
forms of the same exact protocol (include/freud_sae.h, "data-parallel exactness"):

  "p2p"   in-engine, the engine's own exchange kernels over hipIpc peer mappings (csrc/p2p_exchange.h): direct reduce-scatter
          + all-gather on the xGMI mesh, <= 8 ranks of one node.  Executed with 2 real processes in tests/test_dp_gpu.py.
  "rccl"  in-engine, the context's own RCCL communicator (sae_dist_init).  Opt-in: RCCL refuses two ranks on one GPU, so no
          1-GPU box can execute it with more than one rank.
  "host"  torch.distributed from Python (RCCL under the "nccl" backend, gloo on CPU): statistics all-reduce, forward_backward,
          gradient ranges all-reduced as the engine announces them, optimizer_step.

FREUD_DP = auto (default) | p2p | rccl | host.  auto: p2p where every rank can map its peers and the start-up self-test
exchange passes, else host.  (FREUD_DP_HOST=1 is the older spelling of FREUD_DP=host.)
"""
from __future__ import annotations

import os
import sys


def requested_mode() -> str:
    if os.environ.get("FREUD_DP_HOST") == "1":
        return "host"
    mode = os.environ.get("FREUD_DP", "auto").lower()
    if mode not in ("auto", "p2p", "rccl", "host"):
        raise ValueError(f"FREUD_DP={mode!r}: must be auto, p2p, rccl or host")
    return mode


def _all_agree(dist, ok: bool, device) -> bool:
    import torch
    flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def setup(eng, dist, rank: int, world: int, device, mode: str = "auto", payload: str = "float32", overlap: int = 1) -> str:
    """Give engine context `eng` its exchange.  Collective over the (already initialised) process group `dist`.
    Returns the mode in force: "p2p" / "rccl" (step() runs the whole protocol) or "host" (the caller drives it; the
    engine is told the world size).  payload / overlap apply to the in-engine forms of the fused d=384 path."""
    hip_engine = hasattr(eng, "p2p_export")          # (the oracle-backed stand-in of the CPU tests has no exchange of its own)
    nccl = dist.get_backend() == "nccl"
    if (mode == "p2p" and not hip_engine) or (mode == "rccl" and not (hip_engine and nccl)):
        raise RuntimeError(f"FREUD_DP={mode} needs the HIP engine" + (" and the nccl backend" if mode == "rccl" else ""))
    if not nccl:
        device = "cpu"                               # the hand-shake tensors travel over the host backend
    chosen = "host"
    # the peer exchange only uses the process group as a host channel for the handles: any backend will do
    if hip_engine and mode in ("auto", "p2p") and world <= 8:
        ok, err = True, None
        try:
            blob = eng.p2p_export()
        except Exception as e:              # noqa: BLE001 -- any failure means "not here"; the ranks then agree on a fallback
            ok, err, blob = False, e, b""
        blobs = [None] * world
        dist.all_gather_object(blobs, blob)
        if ok and all(len(b) == len(blob) and len(b) > 0 for b in blobs):
            try:
                eng.p2p_init(blobs, rank, world)     # maps the peers + self-test exchange (collective, times out instead of hanging)
            except Exception as e:          # noqa: BLE001
                ok, err = False, e
        else:
            ok = False
        if _all_agree(dist, ok, device):
            chosen = "p2p"
        else:
            if err is not None:
                print(f"[rank {rank}] peer exchange unavailable ({err})", file=sys.stderr)
            if mode == "p2p":
                raise RuntimeError("FREUD_DP=p2p: the peer exchange could not be set up on every rank")
            if ok:
                # this rank holds working mappings but a peer does not: a context cannot leave the protocol again
                raise RuntimeError("peer exchange came up on some ranks only; restart with FREUD_DP=host")
    elif mode == "rccl":
        ids = [eng.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ok = True
        try:
            eng.dist_init(ids[0], rank, world)
        except Exception as e:              # noqa: BLE001
            print(f"[rank {rank}] in-engine RCCL unavailable ({e})", file=sys.stderr)
            ok = False
        if not _all_agree(dist, ok, device):
            raise RuntimeError("FREUD_DP=rccl: the engine's communicator could not be created on every rank")
        chosen = "rccl"
    if chosen in ("p2p", "rccl"):
        if payload == "bfloat16":
            eng.dist_set_payload("bfloat16")
        if overlap > 1:
            if chosen != "p2p":
                raise RuntimeError("column-range overlap needs the peer exchange")
            eng.dist_set_overlap(overlap)
    else:
        eng.set_dp_world(world)
    return chosen

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

FREUD_DP = auto (default) | p2p | rccl | host.  auto: p2p where every rank can map its peers AND the start-up self-test passes
(four exchanges of every payload form over the same addresses with changing patterns: a stale cached peer line gives wrong
sums there), else the in-engine RCCL form, else host.  (FREUD_DP_HOST=1 is the older spelling of FREUD_DP=host.)

Why p2p may be the default although no two-GPU box has ever been available to this build: its failure is made LOUD,
not silent, at three levels, all of which run on whatever hardware the job lands on --
  1. the self-test above (sae_p2p_init), on every rank, before the first step;
  2. audit(): real gradients -- this rank's contribution is snapshotted before an exchange, the snapshots are summed by an
     INDEPENDENT carrier (torch.distributed: RCCL / gloo) and compared with what the peer exchange produced.  A wrong sum that
     is identical on every rank, which no replica comparison can see, shows here.  Runs on the first FREUD_DP_AUDIT_STEPS (3)
     steps and at every logging step of train(), and in bench.py's warm-up (result in the JSON line);
  3. check_replicas(): 64-bit checksums of parameters and optimizer moments compared across the ranks at every logging step and
     before every checkpoint write: replicas are bit-identical by construction, any difference is an exchange bug.
A failed audit or replica check raises ExchangeError naming rank and step; train() then writes nothing and exits non-zero.
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


class ExchangeError(RuntimeError):
    """The data-parallel exchange produced (or would produce) wrong numbers: replicas diverged, an audit mismatch, a peer that
    left the protocol.  train() turns it into a non-zero exit that names the last good checkpoint."""


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
    if mode == "p2p" and world > 8:
        raise RuntimeError(f"FREUD_DP=p2p serves up to 8 ranks of one node (world = {world}); use FREUD_DP=rccl or host")
    # the peer exchange only uses the process group as a host channel for the handles: any backend will do
    if hip_engine and mode in ("auto", "p2p") and world <= 8:
        ok, err = True, None
        try:
            blob = eng.p2p_export()
        except Exception as e:              # noqa: BLE001 -- any failure means "not here"; the ranks then agree on a fallback
            ok, err, blob = False, e, b""
        blobs = [None] * world
        dist.all_gather_object(blobs, blob)
        # The self-test inside p2p_init waits a bounded time for its peers.  Ranks that load the engine's code object for the first
        # time can be seconds apart (round 5: one of the first eight-process runs on a cold box failed here): every rank first runs
        # a kernel of the library (the parameter checksum) and THEN meets the others at a barrier, so that they enter together.
        try:
            if hasattr(eng, "param_checksum"):
                eng.param_checksum()
        except Exception:                   # noqa: BLE001 -- a warm-up only
            pass
        dist.barrier()
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
                eng.p2p_leave()             # this rank's self-test passed, a peer's did not: fall back TOGETHER
    if chosen == "host" and hip_engine and nccl and (mode == "rccl" or mode == "auto"):
        # in-engine RCCL: the explicit choice, and auto's second option (nodes above 8 ranks, peers that cannot be mapped,
        # a failed self-test) before the Python-driven protocol
        ids = [eng.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ok = True
        try:
            eng.dist_init(ids[0], rank, world)
        except Exception as e:              # noqa: BLE001
            print(f"[rank {rank}] in-engine RCCL unavailable ({e})", file=sys.stderr)
            ok = False
        if _all_agree(dist, ok, device):
            chosen = "rccl"
        elif mode == "rccl":
            raise RuntimeError("FREUD_DP=rccl: the engine's communicator could not be created on every rank")
        elif ok:
            raise RuntimeError("in-engine RCCL came up on some ranks only; restart with FREUD_DP=host")
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


# ------------------------------------------------------------------------------------------------------------------------
# run-time guards of the in-engine exchange (module docstring, levels 2 and 3)
# ------------------------------------------------------------------------------------------------------------------------
class Auditor:
    """Sum check of the peer exchange against an independent carrier.

        aud = Auditor(eng, dist, rank, world)       # allocates the snapshot buffer (same size as the gradient buffer)
        aud.arm(); eng.step(x, lr); aud.verify(step)    # one audited step

    arm() makes every gradient exchange of the next step copy the segments it is about to sum into the snapshot; verify()
    all-reduces the snapshot over `dist` (nccl: on the device; gloo: through host memory) and compares with the gradient
    buffer the engine holds after the step.  Only the exchanged segments are compared (the snapshot is NaN elsewhere).
    fp32 payload: the two sums differ by summation order only -> rel. error <= 1e-5; bf16 payload: one bf16 rounding of
    the inputs and one of the sum -> <= 2e-2."""

    def __init__(self, eng, dist, rank: int, world: int, payload: str = "float32"):
        import torch
        self.eng, self.dist, self.rank, self.world = eng, dist, rank, world
        self.grads = eng.grad_tensor()
        self.snap = torch.empty_like(self.grads)
        # the loss scalars ride in the exchanged buffer, but some of them are written AFTER the exchange (the optimizer leaves the
        # clipped gradient norm there): the comparison covers the parameter gradients and the did_fire flags
        n_par, n_met, _ = eng.grad_layout()
        self.cmp = torch.ones(self.grads.numel(), dtype=torch.bool, device=self.grads.device)
        self.cmp[n_par:n_par + n_met] = False
        self.tol = 2e-2 if payload == "bfloat16" else 1e-5
        self.armed = False
        self.passed = 0

    # what arm() fills the snapshot with: a quiet NaN with a payload no arithmetic produces.  "Exchanged" is decided on these BITS,
    # not on isnan(): a NaN in one rank's contribution then stays inside the comparison (and fails it) instead of giving that rank a
    # different mask from its peers
    _SENTINEL = 0x7FC0BEEF - (1 << 32) if 0x7FC0BEEF >= (1 << 31) else 0x7FC0BEEF
    _BLOCK = 4096

    def arm(self) -> None:
        import torch
        self.snap.view(torch.int32).fill_(self._SENTINEL)
        self.eng.dist_audit(self.snap)
        self.armed = True

    def verify(self, step: int) -> float:
        import torch
        assert self.armed
        self.eng.dist_audit(None)
        self.armed = False
        torch.cuda.synchronize()
        mask = (self.snap.view(torch.int32) != self._SENTINEL) & self.cmp      # the same on every rank: the segments of the step
        ref = torch.where(mask, self.snap, torch.zeros_like(self.snap))
        if self.dist.get_backend() == "nccl":
            self.dist.all_reduce(ref)
        else:
            host = ref.cpu()
            self.dist.all_reduce(host)
            ref = host.to(ref.device)
        got = torch.where(mask, self.grads, torch.zeros_like(self.grads))
        n_cmp = int(mask.sum().item())
        # The error is judged BLOCK by block (4096 consecutive floats) against that block's own magnitude: a stale read in a
        # small-magnitude region (bias gradients, did_fire flags, rarely firing latents) is not hidden by the largest weight
        # gradient of the buffer.  A block of zeros (padding) is held to 1e-3 of the global scale.
        n, blk = ref.numel(), self._BLOCK
        pad = (-n) % blk
        diff = torch.nn.functional.pad((got - ref).abs(), (0, pad)).view(-1, blk)
        mag = torch.nn.functional.pad(ref.abs(), (0, pad)).view(-1, blk).amax(dim=1)
        scale = float(mag.max().item()) if n else 0.0
        allowed = self.tol * torch.clamp(mag, min=1e-3 * max(scale, 1e-30))
        worst = diff.amax(dim=1) / torch.clamp(mag, min=1e-3 * max(scale, 1e-30))
        rel = float(worst.max().item()) if n else 0.0
        mine_bad = n_cmp == 0 or not (rel <= self.tol)        # (NaN compares false: a NaN contribution fails the audit)
        # the verdict is COLLECTIVE: a rank whose own sums are right must stop together with the one whose sums are wrong -- and a
        # rank that saw no exchanged segment at all says so HERE, not by raising before its peers' all_reduce (they would hang)
        flags = torch.zeros(self.world, dtype=torch.int32)
        flags[self.rank] = (2 if n_cmp == 0 else 1) if mine_bad else 0
        if self.dist.get_backend() == "nccl":
            flags = flags.cuda()
        self.dist.all_reduce(flags)
        failed = [r for r in range(self.world) if int(flags[r].item()) != 0]
        if failed:
            if any(int(flags[r].item()) == 2 for r in failed):
                raise ExchangeError(f"[rank {self.rank}] step {step}: exchange audit saw no exchanged segment on rank(s) "
                                    f"{[r for r in failed if int(flags[r].item()) == 2]} (is the peer exchange in force?)")
            detail = ""
            if mine_bad:
                bad = int((diff > allowed[:, None]).sum().item())
                detail = (f": worst block-relative |diff| {rel:.2e} > {self.tol:.0e} (buffer scale {scale:.3e}), {bad} of {n_cmp} elements here"
                          + (" -- NaN in a contribution" if rel != rel else ""))
            raise ExchangeError(f"[rank {self.rank}] step {step}: the peer exchange's sum differs from the {self.dist.get_backend()} "
                                f"all-reduce of the same inputs on rank(s) {failed}{detail} -- stale or misdirected peer reads; "
                                "restart with FREUD_DP=rccl or host")
        self.passed += 1
        return rel


def check_replicas(eng, dist, rank: int, world: int, step: int) -> None:
    """Raise ExchangeError unless every rank holds bit-identical parameters and optimizer moments (64-bit checksums over the
    host channel).  Collective.  Cheap: three passes over the parameters on the device + one small all-gather."""
    import torch
    mine = eng.param_checksum()
    # int64 carries the 64 checksum bits (two's complement): gloo and nccl both move int64
    t = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in mine], dtype=torch.int64)
    nccl = dist.get_backend() == "nccl"
    if nccl:
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    rows = [tuple(int(v) for v in o.cpu().tolist()) for o in out]
    if any(r != rows[0] for r in rows):
        names = ("parameters", "exp_avg", "exp_avg_sq", "step count")
        diff = [(r, [names[k] for k in range(4) if rows[r][k] != rows[0][k]]) for r in range(1, world) if rows[r] != rows[0]]
        raise ExchangeError(f"[rank {rank}] step {step}: replicas diverged (bit-identical by construction, so this is an exchange bug): "
                            + "; ".join(f"rank {r} differs from rank 0 in {', '.join(w)}" for r, w in diff))


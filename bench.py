#!/usr/bin/env python3
"""Benchmark of the SAE train step (BASELINE.json metric: SAE train activations/sec).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Whisper-tiny L1 SAE, d=384, dict 8x
(n=3072), M=65 536 activation rows per GPU per step, RAdam + cosine schedule, recon_alpha=1e4,
clip 1.0; synthetic low-rank activations resident in HBM as bf16 before the timed region;
random-init (orthogonal) weights.  One step = renormalise decoder columns, encoder GEMM+ReLU,
decoder GEMM, masked MSE + L1, full backward, (N>1: RCCL all-reduce of the gradients,) clip,
RAdam.  One process per GPU; per-GPU work is fixed as N grows (weak scaling).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the tied-weight gradient
GEMM pair), timed with HIP events on the launch stream inside the timed region; `cpu_baseline` is
the CPU oracle (oracle/sae_oracle.py, a port of the reference's step) timed on this box's host
cores on a bounded sample (rank 0, N=1 only).

Before the W warm-up steps the same step is run untimed for --spinup seconds (default 1 s): an idle MI355X starts in a
low power state and needs some tens of milliseconds of load to reach its sustained clocks; without it a 10 + 50 step
run (45 ms) is timed on the ramp and reads ~7 % slower than any run of a second or more.  The spin-up must be the
workload itself: a second of a bare MFMA loop (tried: more power than the step draws) leaves the chip throttled and the
following steps 10 % SLOWER (0.66 vs 0.60 ms) -- the steady state of a training run is the one its own steps settle into.
The timed region is still exactly K steps after W warm-up steps.

N > 1: the exchange is set up before the warm-up (peer exchange -> in-engine RCCL -> host-driven, whichever passes its checks; the
peer exchange's first warm-up steps are audited against a torch.distributed all-reduce and the replicas' checksums compared after
the run); `config.dp` / `config.dp_guards` of the line say which carrier ran and what was checked.  FREUD_BENCH_SHARE_GPU=1 (tests)
puts every rank on GPU 0 with gloo as the host channel, so that this flow can run on a one-GPU box.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0    # dense MFMA fp8 (v_mfma_f32_32x32x64_f8f6f4), same guide
# what a bare v_mfma_f32_32x32x16_bf16 loop at 100 % issue sustains on random bf16 operands on this chip (power-managed
# clock 1.68 GHz; tools/mfma_clock_probe.hip, profiles/r01_mfma_clock_probe.jsonl): reported next to the nominal peak
MEASURED_MFMA_LOOP_TFLOPS = 1700.0


def to_activation_dtype(x32, dtype, guard=True):
    """fp32 -> the dtype the activations reach the engine in, the way the loader delivers fp32 shards (freud_amd/csrc/host_convert.c:5-7):
    round to nearest even, and a value that is not -1.0 but would ROUND to it goes to the neighbouring 16-bit value instead -- the
    reference takes its padding mask on the fp32 values (x != -1.0, train_sae.py:431), so only real padding may read -1.0.  Without the
    guard N(0,1)-like data rounded to bf16 hits -1.0 in about 0.1 % of its entries and every block of the batch takes the engine's
    masked-entry path, which a batch from the loader never does (`--raw-cast` = that, as a diagnostic)."""
    x = x32.to(dtype)
    if guard and dtype in (torch.bfloat16, torch.float16):
        hit = (x == -1.0) & (x32 != -1.0)
        if bool(hit.any()):
            ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11          # spacing just below 1.0 (twice that above)
            x = torch.where(hit, torch.where(x32 > -1.0, torch.full_like(x, -1.0 + ulp), torch.full_like(x, -1.0 - 2 * ulp)), x)
    return x


def make_inputs(M, d, n, seed, dtype, kind="lowrank", guard=True):
    g = torch.Generator().manual_seed(seed)
    if kind == "lowrank":      # low-rank, learnable
        z = torch.relu(torch.randn(M, 64, generator=g)) * 0.1
        x = to_activation_dtype(z @ torch.randn(64, d, generator=g), dtype, guard)
    elif kind == "normal":
        x = to_activation_dtype(torch.randn(M, d, generator=g), dtype, guard)
    else:                      # "zeros": clock diagnostic only (DVFS: the chip holds a higher clock on trivial operands)
        x = torch.zeros(M, d, dtype=dtype)
    torch.manual_seed(0)
    W = torch.empty(d, n)
    torch.nn.init.orthogonal_(W)
    return x, W, torch.zeros(n)


def cpu_baseline(x, W, b, steps, lr):
    """The oracle's full train step on the host cores (same batch, same hyper-parameters)."""
    from oracle import sae_oracle as O
    cores = min(os.cpu_count() or 1, 64)     # more threads than this only adds barrier overhead at these sizes
    torch.set_num_threads(cores)
    Wc, bc, st = W.clone(), b.clone(), O.OptState()
    xf = x.float()
    O.l1_train_step(xf, Wc, bc, st, recon_alpha=1e4, lr=lr, clip_thresh=1.0, optimizer="radam")  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.l1_train_step(xf, Wc, bc, st, recon_alpha=1e4, lr=lr, clip_thresh=1.0, optimizer="radam")
    dt = (time.perf_counter() - t0) / steps
    return {"value": x.shape[0] / dt, "unit": "activations/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} full train steps (+1 warm-up) of the same M={x.shape[0]} d={x.shape[1]} n={W.shape[1]} "
                      f"batch, torch-CPU bf16-autocast restatement of train_sae.py:429-451, {dt:.3f} s/step"}


def latent_nonzero_frac(eng, M, n, device):
    """Fraction of non-zero entries in the bf16 latent the last forward left in HBM (the operand content the power-managed clock
    responds to: an over-fitted single batch ends with a much sparser latent than the first steps of a run)."""
    ptr, ld = eng.latent_buffer()

    class _Alias:
        __cuda_array_interface__ = {"shape": (M, int(ld)), "typestr": "<i2", "data": (int(ptr), False), "version": 2}

    c = torch.as_tensor(_Alias(), device=device)[:, :n]
    return float(torch.count_nonzero(c).item()) / (M * n)


def data_sensitivity(M, d, n, dtype, W, b, device_id, steps=20, warmup=5, spinup=0.5):
    """How much of the headline is its data (SURVEY 8d asks for the N(0,1) line next to the low-rank one).  Each
    leg is a FRESH engine context timed the driver's way (spin-up on its own step, `warmup` untimed steps, `steps` timed), outside the
    headline's timed region: `lowrank` = the headline's batch again (the like-for-like reference of the other legs), `normal` = pure
    N(0,1) activations, `rotate4` = four resident low-rank batches in rotation (no batch is over-fitted), `raw_cast` = the low-rank
    batch cast to bf16 WITHOUT the loader's -1.0 guard (0.1 % of its entries then read as padding and every block takes the masked
    path), `fresh_weights` = the clocks are spun up on a SEPARATE context and the timed steps are steps 3.. of a run from the
    orthogonal initialisation (dense latent, nothing fitted yet)."""
    from freud_amd.engine import SaeEngine
    dev = torch.device("cuda", device_id)
    lr = 4e-4

    def fresh_engine():
        e = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=1e4, clip_thresh=1.0,
                      device_id=device_id)
        e.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
        return e

    def spin(e, xs, seconds):
        t0, i = time.perf_counter(), 0
        while True:
            for _ in range(20):
                e.step(xs[i % len(xs)], lr)
                i += 1
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= seconds:
                return

    def leg(xs, fresh=False):
        xs = [x.to(dev) for x in xs]
        e = fresh_engine()
        if fresh:
            other = fresh_engine()
            spin(other, xs, spinup)
            other.close()
            wu = 2
        else:
            spin(e, xs, spinup)
            wu = warmup
        for i in range(wu):
            e.step(xs[i % len(xs)], lr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            e.step(xs[(wu + i) % len(xs)], lr)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        m = e.metrics()
        out = {"ms_per_step": ms, "step_mfma_frac": 10.0 * M * d * n / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
               "latent_nonzero_frac": latent_nonzero_frac(e, M, n, dev), "recon": float(m[0]), "l1": float(m[1])}
        e.close()
        return out

    low = [make_inputs(M, d, n, 1000 + 7919 * j, dtype, "lowrank")[0] for j in range(4)]
    res = {"lowrank": leg(low[:1]),
           "normal": leg([make_inputs(M, d, n, 1000, dtype, "normal")[0]]),
           "rotate4": leg(low),
           "raw_cast": leg([make_inputs(M, d, n, 1000, dtype, "lowrank", guard=False)[0]]),
           "fresh_weights": leg(low[:1], fresh=True),
           "how": f"each leg: fresh context, {spinup} s spin-up on its own step, {warmup} untimed + {steps} timed steps, outside the "
                  "headline's timed region; fresh_weights: spin-up on a separate context, 2 untimed + timed steps from the orthogonal "
                  "initialisation; latent_nonzero_frac of the leg's last forward"}
    return res


def pcie_inclusive_sample(d, n, files=320, batch_files=40, T=1500, epochs=6):
    """The same train step fed by the activation loader from fp32 shards in the collector's format (host page cache -> gather
    threads, fp32 -> bf16 -> pinned ring -> HBM): what a real `--config` run gets when the batch is NOT resident.  A bounded
    sample (a few hundred MB of synthetic shard in /tmp, a few dozen steps); reported next to `value`, never as `value`."""
    import shutil
    import tempfile
    from freud_amd.engine import SaeEngine
    from freud_amd.loader import MemoryMappedActivationDataLoader, write_shards
    tmp = tempfile.mkdtemp(prefix="freud_bench_loader_", dir="/tmp")
    try:
        rng = np.random.default_rng(0)
        z = np.maximum(rng.standard_normal((files * T, 32), dtype=np.float32), 0) * 0.1
        rows = (z @ rng.standard_normal((32, d), dtype=np.float32)).reshape(files, T * d)
        write_shards(tmp, "enc", rows, [T, d], [f"/data/f{i}.flac" for i in range(files)])
        shard_gb = rows.nbytes / 1e9
        del rows, z
        dl = MemoryMappedActivationDataLoader(tmp, "enc", batch_files, 0, None, {"shuffle": True, "drop_last": True},
                                              device="cuda", deliver_dtype="bfloat16")
        eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=batch_files * T, optimizer="radam", recon_alpha=1e4)
        W = torch.empty(d, n)
        torch.nn.init.orthogonal_(W)
        eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
        per_epoch, skip = [], 3
        for epoch in range(epochs):          # the first two epochs warm the page cache, the pinned ring and the clocks
            t0, steps = None, 0
            for xb, _ in dl:
                eng.step(xb, 1e-4)
                steps += 1
                if steps == skip:            # an epoch starts by refilling the loader's pipeline (a real epoch has thousands
                    torch.cuda.synchronize() # of steps, this one eight): time its steady part only
                    t0 = time.perf_counter()
            torch.cuda.synchronize()
            if epoch >= 2 and t0 is not None and steps > skip:
                per_epoch.append((time.perf_counter() - t0) / (steps - skip))
        eng.close()
        nthreads, impl = dl._gather_threads, None
        try:
            from freud_amd.loader import _host_lib
            impl = {0: "portable (AVX2 auto-vectorised)", 1: "AVX-512, streaming stores"}.get(int(_host_lib().freud_host_impl()))
        except Exception:       # noqa: BLE001
            pass
        del dl
        dt = sorted(per_epoch)[len(per_epoch) // 2]          # median epoch
        # the host -> HBM link of THIS box, measured the plain way (pinned buffer, one stream, 256 MB copies): the loader cannot
        # deliver more than this; boxes of the pool differ (Gen5 x16 boxes copy ~55 GB/s, others ~31)
        pin = torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True)
        dev = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        link = 8 * (256 << 20) / (time.perf_counter() - t0) / 1e9
        del pin, dev
        delivered = batch_files * T * d * 2 / dt / 1e9
        return {"value": batch_files * T / dt, "unit": "activations/s", "ms_per_step": dt * 1e3,
                "rows_per_step": batch_files * T, "steps": (steps - skip) * len(per_epoch),
                "delivered_GBps": delivered, "link_h2d_GBps_measured": link, "frac_of_link": delivered / link,
                "gather_threads": nthreads, "converter": impl,
                "sample": f"{shard_gb:.2f} GB fp32 shard ({files} files x {T} x {d}) in the page cache, delivered as bf16 by the "
                          f"gather threads (bit-identical training: the engine rounds x to bf16 first), loader + engine step"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher's environment: spawn the N ranks as CHILD processes (one per GPU, the
    launcher's variables set the way torch.distributed.run sets them), relay rank 0's single JSON line, exit non-zero if any rank
    does.  This parent makes no HIP call (torch.cuda.device_count() does not initialise the GPU on this image) and execs nothing."""
    import socket
    import subprocess
    share = os.environ.get("FREUD_BENCH_SHARE_GPU", "0") == "1"
    have = torch.cuda.device_count()
    if have < n and not share:
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); nothing was measured.  (Tests of the N > 1 control flow on a "
              "one-GPU box: FREUD_BENCH_SHARE_GPU=1.)", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile(mode="w+") as cap:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FREUD_BENCH_LAUNCHER="bench.py (self-launched ranks)")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=cap if r == 0 else sys.stderr))
        rcs = [None] * n
        while any(rc is None for rc in rcs):
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = pr.poll()
            if any(rc not in (None, 0) for rc in rcs):      # a rank died: the others would wait in a collective for ever --
                for r, pr in enumerate(procs):              # stop exactly the processes started here
                    if rcs[r] is None:
                        pr.terminate()
                        try:
                            rcs[r] = pr.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            pr.kill()
                            rcs[r] = pr.wait()
                break
            time.sleep(0.05)
        cap.seek(0)
        out0 = cap.read()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if bad or len(lines) != 1:
        sys.stderr.write(out0 or "")
        print(f"bench.py: self-launched run failed: exit codes by rank {rcs}, {len(lines)} JSON line(s) from rank 0", file=sys.stderr)
        return next((rc for _, rc in bad), 1) or 1
    line = json.loads(lines[0])
    if line.get("n_gpus") != n:
        print(f"bench.py: rank 0 reported n_gpus={line.get('n_gpus')} for --gpus {n}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup", type=float, default=1.0, help="seconds of untimed steps before the warm-up (clock ramp)")
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--d", type=int, default=384)
    ap.add_argument("--n", type=int, default=3072)
    ap.add_argument("--x-dtype", default="bfloat16", choices=["bfloat16", "float16", "float32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie-sample", action="store_true", help="skip the loader-fed (PCIe-inclusive) sample of the same step")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--variant", default="l1", choices=["l1", "topk"], help="topk = BASELINE configs[2] style run")
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--dead-threshold", type=float, default=1e6,
                    help="TopK dead_feature_threshold in frames (configs use 1e6: AuxK switches on after ~16 steps of 65536 "
                         "rows once latents stay silent; 1e15 keeps AuxK off)")
    ap.add_argument("--dead-latents", type=int, default=0,
                    help="TopK: start with this many latents marked dead (counters far beyond --dead-threshold), e.g. with "
                         "--dead-threshold 1e15 to time the AuxK branch at a chosen number of dead latents")
    ap.add_argument("--force-dist", action="store_true", help="take the data-parallel code path even with one rank (test hook)")
    ap.add_argument("--dp-payload", default="float32", choices=["float32", "bfloat16"],
                    help="gradient payload of the in-engine all-reduce (bfloat16: fused d=384 path only, half the bytes)")
    ap.add_argument("--dp", default=None, choices=["auto", "p2p", "rccl", "host"],
                    help="data-parallel exchange (freud_amd/dp.py): p2p = the engine's own kernels over hipIpc peer mappings, rccl = "
                         "the engine's own RCCL communicator, host = torch.distributed from Python; default FREUD_DP or auto")
    ap.add_argument("--dp-host", action="store_true", help="same as --dp host")
    ap.add_argument("--dp-overlap", type=int, default=1,
                    help="fused d=384 backward in this many column-tile ranges, each exchanged under the next one's backward (p2p)")
    ap.add_argument("--gemm128", action="store_true", help="A/B timing: keep the generic GEMMs on the 128x128 kernel")
    ap.add_argument("--data", default="lowrank", choices=["lowrank", "normal", "zeros"],
                    help="synthetic activation distribution (lowrank = the reported workload; zeros = clock diagnostic)")
    ap.add_argument("--rotate", type=int, default=1,
                    help="this many resident synthetic batches (different seeds) taken in rotation; 1 = the headline's single batch")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the data_sensitivity legs of the default N=1 line")
    ap.add_argument("--raw-cast", action="store_true",
                    help="diagnostic: synthetic fp32 cast to the activation dtype WITHOUT the loader's -1.0 guard (see to_activation_dtype)")
    ap.add_argument("--dbg", type=int, default=0, help="kernel timing-experiment flags (invalidates results)")
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel HIP-event breakdown to stderr")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp8", "fp8bwd"],
                    help="fp8 = BASELINE configs[4]: e4m3 encoder / decoder GEMMs (L1 only); fp8bwd: the dpre GEMM of the backward too")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around this process: become one (the parent touches no GPU) -- `--gpus N` can never print an n_gpus: 1 line
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`, or unset WORLD_SIZE and "
                         "let bench.py spawn the ranks itself")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the train step)"
    # FREUD_BENCH_SHARE_GPU=1 (tests): every rank on GPU 0 with gloo as the host channel -- two RCCL ranks cannot share a device,
    # the engine's own peer exchange can (tests/test_dp_gpu.py), so the N > 1 control flow of this file runs on a one-GPU box
    share_gpu = os.environ.get("FREUD_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    ctrl_dev = "cpu" if share_gpu else "cuda"             # where the control tensors of the collectives below live
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # (freud_amd/train_sae.py: peer-read buffers in fine-grained memory; --force-dist too, so that the one-rank overhead figures
        # are measured in the configuration multi-GPU runs actually use -- ADVICE r4)
        os.environ.setdefault("FREUD_P2P_FINEGRAINED", "1")
    if use_dist:
        import torch.distributed as dist
        # RCCL writes its NCCL_DEBUG output (version banner, warnings) to STDOUT: send it to a file instead so that
        # stdout carries exactly one JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl_debug_%h_%p.log")
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")   # the gradient buffer is engine memory, not torch's
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":     # the version banner ignores NCCL_DEBUG_FILE
            os.environ.pop("NCCL_DEBUG")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    from freud_amd.engine import SaeEngine

    M, d, n = args.rows, args.d, args.n
    dtype = getattr(torch, args.x_dtype)
    x_cpu, W, b = make_inputs(M, d, n, seed=1000 + rank, dtype=dtype, kind=args.data, guard=not args.raw_cast)
    x = x_cpu.cuda()
    xs = [x] + [make_inputs(M, d, n, seed=1000 + rank + 7919 * j, dtype=dtype, kind=args.data, guard=not args.raw_cast)[0].cuda()
                for j in range(1, max(args.rotate, 1))]
    total_steps, base_lr = 100000, 4e-4

    def attempt(mode_override=None):
        """engine + data-parallel set-up + spin-up + warm-up + the timed region.  Returns (engine, step function, seconds of the
        timed region, exchange in force, healthy).  (Reads `use_dist` of the enclosing scope at call time: the plain reference pass
        of an N > 1 run switches it off.)"""
        grads, works = None, []
        args.dp_host = args.dp_host_flag or mode_override == "host"
        if args.variant == "topk":
            eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=args.k, auxk_alpha=0.03125,
                            clip_thresh=1.0, device_id=local_rank, force_gemm128=args.gemm128, debug_flags=args.dbg)
            eng.set_topk_options(args.dead_threshold, 1024)
            g = torch.Generator().manual_seed(0)
            We = (torch.rand(n, d, generator=g) * 2 - 1) / d ** 0.5
            Wd = We / (We.norm(dim=1, keepdim=True) + torch.finfo(torch.float32).eps)
            eng.set_params({"encoder.weight": We.numpy(), "encoder.bias": np.zeros(n, np.float32), "W_dec": Wd.numpy(),
                            "b_dec": np.zeros(d, np.float32)})
            if args.dead_latents > 0:
                nf = np.zeros(n, np.int64)
                nf[np.random.default_rng(0).permutation(n)[:args.dead_latents]] = np.int64(4e18)
                eng.set_topk_state(nf)
        else:
            eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=1e4,
                            clip_thresh=1.0, device_id=local_rank, debug_flags=args.dbg, force_gemm128=args.gemm128,
                            precision=args.precision)
            eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
        # Data parallel: the engine's own RCCL communicator (sae_dist_init) runs the whole protocol -- batch statistics and
        # gradient ranges all-reduced on a communication stream under the forward / backward kernels -- inside step(); --dp-host
        # drives the same protocol from Python through torch.distributed (the host-side variant train() keeps for CPU tests).
        dp_mode = "none"
        if use_dist:
            from freud_amd import dp
            mode = mode_override or ("host" if args.dp_host else (args.dp or dp.requested_mode()))
            if mode_override == "rccl" and dist.get_backend() != "nccl":
                mode = "host"
            dp_mode = dp.setup(eng, dist, rank, world, torch.device("cuda", local_rank), mode=mode, payload=args.dp_payload,
                               overlap=args.dp_overlap)
            args.dp_host = dp_mode == "host"
        elif args.dp_overlap > 1:
            eng.dist_set_overlap(args.dp_overlap)          # timing the range-split backward by itself
        if use_dist and args.dp_host:
            grads = eng.grad_tensor()
            eng.set_grad_ready_callback(lambda off, cnt: works.append(dist.all_reduce(grads[off:off + cnt], async_op=True)))
        lr_of = lambda i: base_lr * (1 + math.cos(math.pi * i / total_steps)) / 2

        def one_step(i):
            x = xs[i % len(xs)]
            if use_dist and args.dp_host:
                eng.batch_stats(x)
                dist.all_reduce(eng.stats_tensor())
                eng.forward_backward(x)
                for w in works:
                    w.wait()
                works.clear()
                eng.optimizer_step(lr_of(i), 1.0)
            else:
                eng.step(x, lr_of(i))

        if args.spinup > 0:                      # clock spin-up (see the module docstring); not part of W or K
            t_spin = time.perf_counter()
            while True:
                for j in range(20):
                    one_step(j)                  # (with --rotate the spin-up rotates too; the index only chooses batch and LR)
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t_spin
                if use_dist:                     # every rank must leave after the same number of (collective) steps
                    te = torch.tensor([elapsed], device=ctrl_dev, dtype=torch.float64)
                    dist.all_reduce(te, op=dist.ReduceOp.MAX)
                    elapsed = float(te.item())
                if elapsed >= args.spinup:
                    break
        # Peer exchange with real peers: the first warm-up steps are AUDITED -- this rank's gradient contribution is snapshotted
        # before the exchange, the snapshots are summed by torch.distributed (RCCL) and compared with what the engine's own
        # exchange produced (freud_amd/dp.py: Auditor).  The verdict is collective; a mismatch sends every rank to in-engine RCCL.
        audit_info = None
        if dp_mode == "p2p" and world > 1 and os.environ.get("FREUD_DP_AUDIT", "1") != "0":
            from freud_amd import dp
            aud = dp.Auditor(eng, dist, rank, world, payload=args.dp_payload)
            n_audit, worst = min(3, max(args.warmup, 1)), 0.0
            try:
                for i in range(n_audit):
                    aud.arm()
                    one_step(i)
                    worst = max(worst, aud.verify(i + 1))
                dp.check_replicas(eng, dist, rank, world, n_audit)
                audit_info = {"audited_steps": n_audit, "max_rel_diff_vs_torch_distributed": worst, "replica_checksums": "identical"}
            except dp.ExchangeError as e:
                print(f"[rank {rank}] {e}", file=sys.stderr)
                return eng, None, 0.0, dp_mode, False, {"failed": str(e)[:300]}
        for i in range(args.warmup):
            one_step(i)
        torch.cuda.synchronize()
        # the dominant kernel is bracketed with HIP events: on every 8th step of a long run (an event pair costs a few us of idle
        # GPU between dependent kernels), on EVERY step of a short one (the driver's 20-step run: 20 samples; three interleaved
        # 20-step pairs on one box: 593.3 / 593.9 us sampled every step against 594.0 / 594.2 every second -- no measurable price)
        eng.profile(1, period=8 if args.steps > 64 else 1)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=ctrl_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        healthy = True
        if dp_mode in ("p2p", "rccl"):                       # a timed-out exchange must not produce a number
            try:
                eng.dist_check()
            except Exception as e:          # noqa: BLE001
                print(f"[rank {rank}] in-engine exchange failed during the run: {e}", file=sys.stderr)
                healthy = False
            flag = torch.tensor([1 if healthy else 0], device=ctrl_dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # every rank takes the same decision
            healthy = int(flag.item()) == 1
            if healthy and world > 1:                        # replicas are bit-identical by construction: anything else is an exchange bug
                from freud_amd import dp
                try:
                    dp.check_replicas(eng, dist, rank, world, args.warmup + args.steps)
                    if audit_info is not None:
                        audit_info["replica_checksums_after_run"] = "identical"
                except dp.ExchangeError as e:
                    print(f"[rank {rank}] {e}", file=sys.stderr)
                    healthy = False
                    audit_info = {"failed": str(e)[:300]}
        return eng, one_step, dt, dp_mode, healthy, audit_info

    args.dp_host_flag = args.dp_host
    # N > 1 (or --force-dist): the SAME step without any exchange, on this rank's GPU, in this process, right before the data-parallel
    # run -- K steps after W warm-up steps, max over the ranks like the headline.  ms_per_step minus this is what the exchange costs
    # on the critical path (dp_timing.exposed_exchange_ms): the first multi-GPU run explains its own efficiency.
    plain_ms = None
    if use_dist and os.environ.get("FREUD_BENCH_PLAIN_REF", "1") != "0":
        saved = (use_dist, args.dp_overlap)
        use_dist, args.dp_overlap = False, 1
        try:
            eng0, _, dt0, _, _, _ = attempt()
            eng0.close()
            del eng0
            plain_ms = dt0 / args.steps * 1e3
        finally:
            use_dist, args.dp_overlap = saved
        t = torch.tensor([plain_ms], device=ctrl_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        plain_ms = float(t.item())
    eng, one_step, dt, dp_mode, healthy, audit_info = attempt()
    dp_fallback = None
    if not healthy:
        # the in-engine exchange failed (peers not reached in the middle of the run -- the exchange kernels time out instead of
        # hanging --, an audit mismatch, diverged replicas): the measurement is repeated from scratch on a fresh context with the
        # next carrier (peer exchange -> in-engine RCCL -> host-driven), and config.dp says so
        first_failure = audit_info
        for nxt in (["rccl", "host"] if dp_mode == "p2p" else ["host"]):
            eng.close()
            eng, one_step, dt, dp_mode, healthy, audit_info = attempt(nxt)
            if healthy:
                break
        dp_fallback = first_failure or {"failed": "exchange failed during the first attempt"}
        if not healthy:
            raise SystemExit("bench.py: no data-parallel exchange completed the run")
    times = eng.kernel_times()
    metrics = eng.metrics()
    eng.profile(0)

    ms_per_step = dt / args.steps * 1e3
    value = M * world * args.steps / dt
    dom = eng.dominant_kernel()
    dom_ms, dom_cnt = times[dom]
    dom_avg_ms = dom_ms / max(dom_cnt, 1)
    # algorithmic FLOPs of the dominant kernel: fused backward = dc, dW(dec), dW(enc) GEMMs = 3 x 2*M*d*n;
    # generic path's dw_gemm = the two weight-gradient GEMMs = 2 x 2*M*d*n
    dom_flops = (6.0 if dom == "bwd_fused_gemm" else 4.0) * M * d * n
    if args.variant == "topk":
        dom_flops = 2.0 * M * d * n                  # the dense encoder GEMM
    achieved = dom_flops / (dom_avg_ms * 1e-3) / 1e12 if dom_avg_ms > 0 else 0.0
    step_flops = 10.0 * M * d * n                    # algorithmic FLOPs per activation = 10 d n
    if args.variant == "topk":
        step_flops = (2.0 * d * n + 10.0 * args.k * d) * M   # 2 d n + 10 k d per activation
    fb_ms, fb_cnt = times["fwd_bwd_total"]
    f8 = {"bf16": 0.0, "fp8": 0.4, "fp8bwd": 0.6}[args.precision] if args.variant == "l1" else 0.0
    step_peak = 1.0 / (f8 / PEAK_FP8_TFLOPS + (1.0 - f8) / PEAK_BF16_TFLOPS)

    breakdown, dp_timing = None, None
    if args.breakdown or args.precision != "bf16" or use_dist:   # every rank runs the extra steps (they contain collectives); rank 0 reports
        n_diag = 16 if use_dist else 10
        eng.profile(2)
        # harvested every second step: the engine keeps the last 64 event pairs per kernel id, and a configuration with several
        # exchange launches per step (--dp-overlap > 4, the chunked exchanges of the d >= 1024 paths) would overflow that ring
        # over 16 steps and under-report the exchange
        kt = {}
        for i in range(n_diag):
            one_step(args.warmup + args.steps + i)
            if i % 2 == 1 or i == n_diag - 1:
                for k, v in eng.kernel_times().items():
                    kt[k] = (kt.get(k, (0.0, 0))[0] + v[0], kt.get(k, (0.0, 0))[1] + v[1])
        breakdown = {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()}
        eng.profile(0)
        if use_dist:
            # the exchange's own time per step, from HIP events on the stream it runs on (level-2 profile of n_diag extra steps after
            # the timed region; a peer-exchange kernel's time INCLUDES its wait for the slowest rank): max and min over the ranks
            ex_ms, xn = kt.get("dp_exchange", (0.0, 0))
            ss, sn = kt.get("dp_stats_exchange", (0.0, 0))
            compute = sum(v[0] for k, v in kt.items() if k not in ("dp_exchange", "dp_stats_exchange", "fwd_bwd_total"))
            loc = torch.tensor([ex_ms / n_diag, ss / n_diag, compute / n_diag], device=ctrl_dev, dtype=torch.float64)
            hi, lo = loc.clone(), loc.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            in_engine = dp_mode in ("p2p", "rccl")
            dp_timing = {
                "carrier": dp_mode,
                "exchange_ms": float(hi[0]) if in_engine else None,
                "exchange_ms_min_over_ranks": float(lo[0]) if in_engine else None,
                "exchange_launches_per_step": xn / n_diag if in_engine else None,
                "stats_exchange_ms": float(hi[1]) if in_engine else None,
                "compute_kernels_ms": float(hi[2]),
                "plain_ms_per_step": plain_ms,
                "exposed_exchange_ms": (ms_per_step - plain_ms) if plain_ms is not None else None,
                "how": f"exchange_ms / stats_exchange_ms: HIP events around the exchange launches of {n_diag} extra steps after the timed "
                       "region (sum per step, max over ranks; includes the wait for the slowest rank); plain_ms_per_step: the same "
                       "step WITHOUT any exchange, same process and GPU, K steps after W warm-up (max over ranks); "
                       "exposed_exchange_ms = ms_per_step - plain_ms_per_step",
            }
        if use_dist and world > 1:
            # the rank count as RCCL itself reports it (a one-element sum over torch.distributed's nccl backend = RCCL), so that a
            # line can never claim more ranks than exchanged data -- north_star names RCCL; with gloo as host channel (the shared-GPU
            # test mode) the count is gloo's and says so
            one = torch.ones(1, device=ctrl_dev, dtype=torch.int32)
            dist.all_reduce(one)
            dp_timing["ranks_counted_by_collective"] = int(one.item())
            dp_timing["collective_backend"] = "rccl (torch.distributed nccl backend)" if dist.get_backend() == "nccl" else dist.get_backend()
        if rank == 0 and (args.breakdown or args.precision != "bf16"):
            print("per-kernel ms (HIP events, level-2 profile):", json.dumps(breakdown), file=sys.stderr)
        if rank != 0 or not (args.breakdown or args.precision != "bf16"):
            breakdown = None

    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the value
    # measured with rocprofv3 (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH_SIZE doubled
    # as MI355X_MICROARCH.md prescribes for gfx950) is kept under profiles/ and quoted when the workload matches
    traffic, traffic_source = None, None
    try:
        if args.variant == "l1" and args.precision == "bf16" and (M, d, n) == (65536, 384, 3072):
            import glob
            latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")))[-1]   # newest round's pass
            with open(latest) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
            traffic_source = (f"profiles/{os.path.basename(latest)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              "command (FETCH_SIZE doubled per the gfx950 correction), recorded earlier -- not measured in this run")
    except Exception:
        traffic, traffic_source = None, None

    # what the exchange really carries: the bf16 payload and the column ranges exist on the fused d=384 L1 path only
    fused = args.variant == "l1" and args.precision == "bf16" and (d + 127) // 128 * 128 == 384
    eff_payload = args.dp_payload if (fused and dp_mode in ("p2p", "rccl")) else "float32"
    dp_desc = {"none": "none", "host": "host-driven (torch.distributed)",
               "p2p": f"in-engine peer exchange over hipIpc mappings, {eff_payload} gradients"
                      + (f", backward in {args.dp_overlap} column ranges" if fused and args.dp_overlap > 1 else ""),
               "rccl": f"in-engine RCCL, {eff_payload} gradients"}[dp_mode]
    if dp_fallback:
        dp_desc += " (fallback: the first choice failed: " + str(dp_fallback.get("failed", ""))[:160] + ")"
    model_name = {384: "tiny", 512: "base", 768: "small", 1024: "medium", 1280: "large"}.get(d, f"d={d}")
    out = {
        "metric": f"SAE train activations/sec (d={d} dict {n // d}x)", "value": value, "unit": "activations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_s": args.spinup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": ("synthetic" if args.data == "lowrank" else f"synthetic ({args.data}; diagnostic)") + (" (raw cast; diagnostic)" if args.raw_cast else "")
                + (f" ({len(xs)} resident batches in rotation)" if len(xs) > 1 else ""),
        "config": {"workload": f"Whisper-{model_name} d={d} dict {n // d}x (n={n}) L1 SAE train step, M={M} rows/GPU/step, "
                               f"RAdam+cosine, x {args.x_dtype} resident in HBM"
                               + (" (BASELINE configs[1])" if (d, n, M) == (384, 3072, 65536) else ""),
                   "rows_per_gpu": M, "d_model": d, "n_dict": n, "parallelism": f"dp{world}",
                   "dp": dp_desc, "dp_guards": audit_info},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_source, "kernel": dom,
                     "kernel_avg_ms": dom_avg_ms, "kernel_launches": dom_cnt,
                     "flops_per_launch": dom_flops, "peak_measured_bare_mfma_loop": MEASURED_MFMA_LOOP_TFLOPS,
                     "frac_of_measured": achieved / MEASURED_MFMA_LOOP_TFLOPS},
        # whole step against the MFMA roof; with fp8 GEMMs in the step the roof is FLOP-weighted (time at peak = fp8 FLOPs / fp8
        # peak + bf16 FLOPs / bf16 peak): 4 of the 10 M d n FLOPs run on e4m3 with --precision fp8, 6 with fp8bwd
        "step_mfma_frac": (step_flops / (ms_per_step * 1e-3) / 1e12) / step_peak,
        "step_mfma_peak": step_peak,
        "fwd_bwd_ms": fb_ms / max(fb_cnt, 1),
        "loss": {"recon": float(metrics[0]), "l1": float(metrics[1]), "grad_norm": float(metrics[3])},
    }
    if os.environ.get("FREUD_BENCH_LAUNCHER"):
        out["config"]["launcher"] = os.environ["FREUD_BENCH_LAUNCHER"]
    if args.variant == "l1" and args.precision == "bf16":
        try:
            out["latent_nonzero_frac"] = latent_nonzero_frac(eng, M, n, torch.device("cuda", local_rank))
        except Exception as e:              # noqa: BLE001 -- an extra
            out["latent_nonzero_frac"] = None
    if breakdown:
        out["kernel_ms"] = breakdown
    if dp_timing:
        out["dp_timing"] = dp_timing
    if args.precision != "bf16" and breakdown:
        # the two fp8 GEMMs against the dense fp8 MFMA peak (2 M d n FLOPs each), next to the (bf16) dominant kernel above
        out["dtype"] = ("fp8 (e4m3 encoder/decoder GEMMs, fp32 accumulate) + bf16 backward" if args.precision == "fp8" else
                        "fp8 (e4m3 encoder/decoder/dpre GEMMs, fp32 accumulate) + bf16 weight-gradient GEMMs")
        out["fp8_gemms"] = {k: {"ms": breakdown[k], "tflops": 2.0 * M * d * n / (breakdown[k] * 1e-3) / 1e12,
                                "frac_of_fp8_peak": 2.0 * M * d * n / (breakdown[k] * 1e-3) / 1e12 / PEAK_FP8_TFLOPS}
                            for k in ("enc_fwd_gemm", "dec_fwd_gemm") + (("dpre_gemm",) if args.precision == "fp8bwd" else ()) if breakdown.get(k)}
        out["config"]["workload"] = out["config"]["workload"].replace("L1 SAE train step", "L1 SAE train step, fp8 enc/dec GEMMs")
    if args.dbg == 65 and rank == 0:     # diagnostic build: cycle shares of one fused-forward iteration + in-kernel clock
        st = eng.debug_read(5, (M // 128) * 32).reshape(-1, 8)
        st = st[st[:, 3] > 0]
        per = st[:, :3] / st[:, 3:4]
        if per.sum() > 0 and os.environ.get("FREUD_FWD", "2") == "1":      # (the per-phase stamps exist in fwd_fused.h only)
            print("fwd stamps (cycles/iteration, median over waves): decoder gaps 0-11 %.0f | decoder gaps 12-23 (+barrier, DMA) %.0f | "
                  "encoder gaps 24-47 %.0f | total %.0f" % (*np.median(per, 0), np.median(per.sum(1))), file=sys.stderr)
        print("fwd in-kernel clock %.0f MHz (median), loop cycles/iteration %.0f" %
              (np.median(st[:, 4] / st[:, 5]) * 100.0, np.median(st[:, 4] / st[:, 3])), file=sys.stderr)
        print("fwd per-workgroup cycles (median): prologue %.0f | loop %.0f | epilogue %.0f | whole %.0f" %
              (np.median(st[:, 6]), np.median(st[:, 4]), np.median(st[:, 7] - st[:, 6] - st[:, 4]), np.median(st[:, 7])),
              file=sys.stderr)
        if len(st) == (M // 128) * 4 and M // 128 > 256:      # workgroups 0-255 start together (one per CU), the rest as CUs come free
            r1, r2 = st[: 256 * 4], st[256 * 4:]
            print("fwd prologue / whole, first 256 workgroups: %.0f / %.0f | the others: %.0f / %.0f" %
                  (np.median(r1[:, 6]), np.median(r1[:, 7]), np.median(r2[:, 6]), np.median(r2[:, 7])), file=sys.stderr)
        if os.environ.get("FREUD_FF2_WAITSTAMP") == "1":      # -DFF2_WAITSTAMP builds: the hand-over of every iteration, per wave
            it = st[:, 3:4]
            wv = st[:, :3] / it
            print("fwd hand-over per iteration (cycles, median over waves | p90 | max): own DMA + staging write waited for %.0f | %.0f | %.0f ; "
                  "barrier (the other waves) %.0f | %.0f | %.0f ; barrier exit -> next hand-over %.0f | %.0f | %.0f" %
                  (np.median(wv[:, 0]), np.percentile(wv[:, 0], 90), wv[:, 0].max(), np.median(wv[:, 1]), np.percentile(wv[:, 1], 90), wv[:, 1].max(),
                   np.median(wv[:, 2]), np.percentile(wv[:, 2], 90), wv[:, 2].max()), file=sys.stderr)
            byw = wv.reshape(-1, 4, 3)
            print("  by wave 0..3 (median): DMA wait %s ; barrier %s" %
                  (np.median(byw[:, :, 0], 0).round().tolist(), np.median(byw[:, :, 1], 0).round().tolist()), file=sys.stderr)
            # the slowest wave gates the CU: per workgroup, the spread of the waves' arrival = max over waves of the barrier wait
            print("  per workgroup: max over its waves of the barrier wait %.0f (median), min %.0f" %
                  (np.median(byw[:, :, 1].max(1)), np.median(byw[:, :, 1].min(1))), file=sys.stderr)
        elif os.environ.get("FREUD_FWD", "2") != "1":      # fwd_fused2.h keeps epilogue phase stamps in slots 0-2
            ep = st[:, 7] - st[:, 6] - st[:, 4]
            print("fwd epilogue phases (median cycles): last half iteration + latent drain %.0f | x staged %.0f | residual arithmetic %.0f | "
                  "dx_hat publication + stores + sums %.0f" % (np.median(st[:, 0]), np.median(st[:, 1]), np.median(st[:, 2]),
                                                                np.median(ep - st[:, 0] - st[:, 1] - st[:, 2])), file=sys.stderr)
    if args.dbg == 66 and rank == 0:     # clock stamps of the fused backward
        full = None
        for grid in ((n // 128 + (1 if n % 128 else 0)) * 10, 256):      # uniform at C2: 24 x 10 (too small a buffer for the balanced form: refused); balanced: 256
            try:
                full = eng.debug_read(6, grid * 8).reshape(2, grid, 4)
                break
            except Exception:
                continue
        st, whole = full[0], full[1][:, 0]
        ok = st[:, 2] > 0
        st, whole = st[ok], whole[ok]
        print("bwd in-kernel clock %.0f MHz (median), loop cycles/step %.0f (ideal 72 MFMA x 32 = 2304); %d workgroups, steps per "
              "workgroup %.0f-%.0f, whole-kernel cycles per workgroup: median %.0f, max %.0f, outside the loop (median) %.0f" %
              (np.median(st[:, 0] / st[:, 1]) * 100.0, np.median(st[:, 0] / st[:, 2]), len(st), st[:, 2].min(), st[:, 2].max(),
               np.median(whole), whole.max(), np.median(whole - st[:, 0])), file=sys.stderr)
    if args.dbg == 67 and rank == 0 and args.variant == "topk":     # -DSEL_STAMP builds: phase stamps of the compact AuxK select
        st = eng.debug_read(11, 4096 * 8).reshape(-1, 8)
        slow = st[st[:, 7] >= 1000]
        if len(slow):
            last = np.array([r[int(r[7]) - 1001] for r in slow])
            print("AuxK compact select: %d of %d rows took the SLOW path (block-wide counting probes): median %.0f cycles per row, marks %s" %
                  (len(slow), len(st), np.median(last), np.median(slow[:, :7], axis=0).round().tolist()), file=sys.stderr)
        st = st[(st[:, 5] > 0) & (st[:, 7] < 1000)]
        if len(st):
            names = ["row loaded + masked", "candidate count (block scan)", "lower bound L", "candidates appended", "histogram select + stores"]
            d_ = np.diff(st[:, :6], axis=1)
            print("AuxK compact select, cycles per row (median over %d rows): %s | total %.0f" %
                  (len(st), ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, np.median(d_, axis=0))), np.median(st[:, 5])), file=sys.stderr)
    if args.dbg == 69 and rank == 0 and args.variant == "topk":     # -DCSCF_STAMP builds: phase stamps of csc_fill
        st = eng.debug_read(13, 1024 * 8).reshape(-1, 8)
        st = st[st[:, 6] > 0]
        if len(st):
            names = ["positions initialised", "indices loaded", "LDS slots", "activations loaded", "stores issued", "stores acknowledged"]
            d_ = np.diff(st[:, :7], axis=1)
            print("csc_fill, shader cycles per workgroup (median over %d): %s | total %.0f" %
                  (len(st), ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, np.median(d_, axis=0))), np.median(st[:, 6])), file=sys.stderr)
    if args.dbg == 68 and rank == 0 and args.variant == "topk":     # -DTSEL_STAMP builds: phase stamps of the tile-driven main select
        st = eng.debug_read(12, 4096 * 8).reshape(-1, 8)
        st = st[st[:, 6] > 0]
        if len(st):
            names = ["tile maxima in registers", "L + candidate tile list", "candidate tiles in registers", "candidates appended",
                     "block barrier", "ranked + stored"]
            d_ = np.diff(st[:, :7], axis=1)
            print("tile-driven select, shader cycles per row (median over %d rows): %s | total %.0f | "
                  "candidates %.0f, candidate tiles %.0f (median)" %
                  (len(st), ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, np.median(d_, axis=0))), np.median(st[:, 6]),
                   np.median(st[:, 7] // 1000), np.median(st[:, 7] % 1000)), file=sys.stderr)
    if args.variant == "topk":
        out["metric"] = f"SAE train activations/sec (TopK d={d} n={n} k={args.k})"
        out["config"]["workload"] = (f"TopK SAE d={d} n={n} k={args.k} train step, M={M} rows/GPU/step, Adam, x {args.x_dtype} "
                                     "resident in HBM (BASELINE configs[2] shape)")
        out["loss"] = {"fvu": float(metrics[0]), "auxk": float(metrics[1]), "grad_norm": float(metrics[3]), "dead_frac": float(metrics[5])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.variant == "l1":
        out["cpu_baseline"] = cpu_baseline(x_cpu, W, b, args.cpu_steps, base_lr)
    elif rank == 0:
        out["cpu_baseline"] = None
    eng.close()
    if (rank == 0 and world == 1 and not use_dist and not args.no_sensitivity and not args.no_cpu_baseline and args.variant == "l1"
            and args.precision == "bf16" and args.dbg == 0):
        try:
            out["data_sensitivity"] = data_sensitivity(M, d, n, dtype, W, b, local_rank)
        except Exception as e:              # noqa: BLE001 -- an extra; must never cost the bench line
            out["data_sensitivity"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and not args.no_pcie_sample and not args.no_cpu_baseline and args.variant == "l1" and args.precision == "bf16":
        try:
            out["pcie_inclusive"] = pcie_inclusive_sample(d, n)
        except Exception as e:              # noqa: BLE001 -- an extra; must never cost the bench line
            out["pcie_inclusive"] = {"error": str(e)[:200]}
    # N > 1: the OTHER carrier's number next to the headline's (north_star names RCCL; the peer exchange is auto's first choice): a short
    # leg on a fresh context after everything else is in `out`.  It runs under a watchdog thread: a carrier that hangs (RCCL can; the
    # engine's own exchange times out by itself) costs this leg, never the line -- rank 0 prints the line without it and every rank leaves.
    # (--force-dist runs the leg with one rank as well: the only way this code path executes on a one-GPU box with the nccl backend)
    other = {"p2p": "rccl", "rccl": "p2p"}.get(dp_mode) if (use_dist and (world > 1 or args.force_dist) and dp_timing is not None) else None
    if (other and dp_fallback is None and os.environ.get("FREUD_BENCH_OTHER_CARRIER", "1") != "0"
            and (other != "rccl" or dist.get_backend() == "nccl")):
        import threading
        eng.close()
        limit = float(os.environ.get("FREUD_BENCH_OTHER_CARRIER_TIMEOUT", "180"))

        def give_up():
            if rank == 0:
                dp_timing["other_carrier"] = {"carrier": other, "error": f"no result within {limit:.0f} s; leg abandoned"}
                out["dp_timing"] = dp_timing
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(limit, give_up)
        dog.daemon = True
        dog.start()
        saved_leg = (args.steps, args.warmup, args.spinup)
        args.steps, args.warmup, args.spinup = min(args.steps, 20), min(args.warmup, 5), min(args.spinup, 0.5)
        try:
            e2, _, dt2, mode2, ok2, _ = attempt(other)
            good = bool(ok2) and mode2 == other and dt2 > 0
            dp_timing["other_carrier"] = {"carrier": mode2, "healthy": bool(ok2), "steps": args.steps,
                                          "ms_per_step": dt2 / args.steps * 1e3 if good else None,
                                          "value": M * world * args.steps / dt2 if good else None}
            if mode2 == "rccl":
                dp_timing["other_carrier"]["ranks_in_engine_communicator"] = e2.dist_world()
            e2.close()
        except BaseException as e:          # noqa: BLE001 -- an extra leg; the headline stands without it
            dp_timing["other_carrier"] = {"carrier": other, "error": str(e)[:200]}
        finally:
            args.steps, args.warmup, args.spinup = saved_leg
            dog.cancel()
        out["dp_timing"] = dp_timing
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

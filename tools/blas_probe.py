#!/usr/bin/env python3
"""Measuring stick for the generic GEMMs: what the vendor library (hipBLASLt / rocBLAS through torch.matmul) reaches on this
box for the four GEMM shapes of an L1 step at d=1280 (C4) and the TopK encoder at d=768 (C3).  Not part of the product:
the engine never calls a BLAS.  Prints one JSON line; run under `rocprofv3 --kernel-trace --stats` to see which tile
configuration the library picked.

  python tools/blas_probe.py [--M 65536] [--d 1280] [--n 40960]
"""
import argparse
import json
import time

import torch


def bench(f, iters=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=65536)
    ap.add_argument("--d", type=int, default=1280)
    ap.add_argument("--n", type=int, default=40960)
    a = ap.parse_args()
    M, d, n = a.M, a.d, a.n
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, d, device=dev, dtype=torch.bfloat16, generator=g)
    W = torch.randn(d, n, device=dev, dtype=torch.bfloat16, generator=g) * 0.03       # [d][n]
    Wt = W.t().contiguous()                                                             # [n][d]
    c = torch.relu(torch.randn(M, n, device=dev, dtype=torch.bfloat16, generator=g))
    # spin the clocks up with the work itself
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        torch.matmul(x, W)
    torch.cuda.synchronize()
    out = {"M": M, "d": d, "n": n}
    fl = 2.0 * M * d * n
    for name, f, flops in (
        ("enc  x[M,d] @ W[d,n]", lambda: torch.matmul(x, W), fl),
        ("enc  x[M,d] @ Wt[n,d]^T", lambda: torch.matmul(x, Wt.t()), fl),
        ("dec  c[M,n] @ Wt[n,d]", lambda: torch.matmul(c, Wt), fl),
        ("dec  c[M,n] @ W[d,n]^T", lambda: torch.matmul(c, W.t()), fl),
        ("dW   x[M,d]^T @ c[M,n]", lambda: torch.matmul(x.t(), c), fl),
        ("dWt  c[M,n]^T @ x[M,d]", lambda: torch.matmul(c.t(), x), fl),
    ):
        dt = bench(f)
        out[name] = {"ms": round(dt * 1e3, 3), "PFLOPs": round(flops / dt / 1e15, 3)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

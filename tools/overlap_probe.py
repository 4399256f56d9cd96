#!/usr/bin/env python3
"""Do host->HBM copies overlap with the train step's kernels?  Times N engine steps on a resident batch, N pinned->HBM
copies of one batch on a copy stream, and both enqueued together."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freud_amd.engine import SaeEngine

M, d, n, N = 60000, 384, 3072, 200
x = (torch.randn(M, d) * 0.1).to(torch.float16).cuda()
eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=1e4)
W = torch.empty(d, n); torch.nn.init.orthogonal_(W)
eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": np.zeros(n, np.float32)})
pin = torch.empty(M * d, dtype=torch.float16, pin_memory=True)
dev = torch.empty(M * d, dtype=torch.float16, device="cuda")
cs = torch.cuda.Stream()
for _ in range(300): eng.step(x, 1e-4)
torch.cuda.synchronize()

def run(steps, copies):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        if copies:
            with torch.cuda.stream(cs): dev.copy_(pin, non_blocking=True)
        if steps: eng.step(x, 1e-4)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / N * 1e3

a, b, c = run(True, False), run(False, True), run(True, True)
print("step alone %.3f ms | copy alone %.3f ms (%.1f GB/s) | both %.3f ms  (perfect overlap = %.3f, serial = %.3f)" %
      (a, b, M * d * 2 / b / 1e6, c, max(a, b), a + b))

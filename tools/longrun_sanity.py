"""3000 train steps at the C2 shape on rotating synthetic batches: losses fall, everything stays finite, the decoder
columns stay unit-norm (python tools/longrun_sanity.py, ~3 s on an MI355X)."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from freud_amd.engine import SaeEngine
M, d, n = 65536, 384, 3072
x_cpu, W, b = make_inputs(M, d, n, seed=1000, dtype=torch.bfloat16)
xs = [x_cpu.cuda()] + [make_inputs(M, d, n, seed=2000 + i, dtype=torch.bfloat16)[0].cuda() for i in range(3)]
eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=1e4, clip_thresh=1.0)
eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
steps = 3000
for i in range(steps):
    lr = 4e-4 * (1 + math.cos(math.pi * i / steps)) / 2
    eng.step(xs[i % 4], lr)
    if i in (0, 10, 100, 500, 1000, 2000, 2999):
        m = eng.metrics()
        print(i, "recon %.2f l1 %.2f gnorm %.1f finite %s" % (m[0], m[1], m[3], np.isfinite(m).all()))
p = eng.get_params()
print("W finite:", np.isfinite(p["decoder.weight"]).all(), "col norms ~", np.linalg.norm(p["decoder.weight"], axis=0)[:3])

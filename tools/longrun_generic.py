"""Soak of the generic (d != 384) L1 path at the C4 shape, bf16 and fp8: 150 train steps each on four rotating synthetic batches --
losses fall, everything stays finite, the decoder columns stay unit-norm (python tools/longrun_generic.py, ~15 s on an MI355X)."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from freud_amd.engine import SaeEngine

M, d, n = 65536, 1280, 40960
x0, W, b = make_inputs(M, d, n, seed=1000, dtype=torch.bfloat16)
xs = [x0.cuda()] + [make_inputs(M, d, n, seed=2000 + i, dtype=torch.bfloat16)[0].cuda() for i in range(3)]
ok = True
for prec in ("bf16", "fp8"):
    eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="radam", recon_alpha=1e4, clip_thresh=1.0, precision=prec)
    eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
    steps, first = 150, None
    for i in range(steps):
        eng.step(xs[i % 4], 4e-4 * (1 + math.cos(math.pi * i / steps)) / 2)
        if i in (0, 10, 50, 100, 149):
            m = eng.metrics()
            first = m[0] if first is None else first
            print(prec, i, "recon %.2f l1 %.2f gnorm %.1f finite %s" % (m[0], m[1], m[3], np.isfinite(m).all()))
            ok = ok and bool(np.isfinite(m).all())
    p = eng.get_params()
    norms = np.linalg.norm(p["decoder.weight"], axis=0)
    print(prec, "W finite:", np.isfinite(p["decoder.weight"]).all(), "col norms in [%.6f, %.6f]" % (norms.min(), norms.max()), "recon fell:", m[0] < first)
    ok = ok and bool(np.isfinite(p["decoder.weight"]).all()) and m[0] < first
    eng.close()
print("OK" if ok else "FAILED")

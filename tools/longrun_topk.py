"""TopK long run with a dead set that comes and goes (AuxK switching between none / the copy path / the full selection, the
dynamic GEMMs between their persistent and plain instantiations): 400 steps at d=768, n=8192 on rotating synthetic batches,
twice -- the two runs must be bitwise equal -- and once more with the AuxK branch on the gather kernels (debug_flags 76),
whose trajectory must stay close.  python tools/longrun_topk.py  (~1 min on an MI355X)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freud_amd.engine import SaeEngine

M, d, n, k, T = 16384, 768, 8192, 32, 1024


def run(flags, steps=400):
    g = torch.Generator().manual_seed(0)
    We = (torch.rand(n, d, generator=g) * 2 - 1) / d ** 0.5
    Wd = We / (We.norm(dim=1, keepdim=True) + torch.finfo(torch.float32).eps)
    xs = [((torch.relu(torch.randn(M, 48, generator=g)) * 0.1) @ torch.randn(48, d, generator=g)).to(torch.bfloat16)
          .reshape(M // T, T, d).cuda() for _ in range(4)]
    eng = SaeEngine(variant="topk", d_model=d, n_dict=n, max_rows=M, optimizer="adam", k=k, auxk_alpha=0.03125, debug_flags=flags)
    eng.set_topk_options(3.0 * M, T)              # dead after 3 silent steps: the dead set appears early and shrinks as AuxK revives
    eng.set_params({"encoder.weight": We.numpy(), "encoder.bias": np.zeros(n, np.float32), "W_dec": Wd.numpy(),
                    "b_dec": np.zeros(d, np.float32)})
    hist = []
    for i in range(steps):
        eng.step(xs[i % 4], 2e-4)
        if i % 20 == 19:
            m = eng.metrics().copy()
            hist.append((i, float(m[0]), float(m[1]), float(m[5])))
    p = eng.get_params()
    eng.close()
    return hist, p


if __name__ == "__main__":
    h1, p1 = run(0)
    h2, p2 = run(0)
    h3, p3 = run(76)
    for a in h1[::4]:
        print("step %4d fvu %.4f auxk %.5f dead %.4f" % a)
    ok = all(np.isfinite(v).all() for v in p1.values()) and np.isfinite(np.array(h1)).all()
    same = h1 == h2 and all(np.array_equal(p1[key], p2[key]) for key in p1)
    dead_seen = sorted({round(a[3], 3) for a in h1})
    print("finite:", ok, " two runs bitwise equal:", same, " dead fractions seen:", dead_seen[:3], "...", dead_seen[-3:])
    print("compact vs gather AuxK, final fvu %.4f vs %.4f, dead %.4f vs %.4f" % (h1[-1][1], h3[-1][1], h1[-1][3], h3[-1][3]))
    assert ok and same and h1[-1][1] < h1[0][1]
    assert abs(h1[-1][1] - h3[-1][1]) < 0.2 * h3[-1][1]
    print("OK")

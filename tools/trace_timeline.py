#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: for the last full step (from one normalize_cast
launch to the next) print every dispatch's start offset, duration and the idle gap before it; and the timed-region
averages (last K launches of every kernel) that the bench line's HIP-event figures are compared with.

  python tools/trace_timeline.py <kernel_trace.csv> [K]
"""
import csv
import sys
from collections import defaultdict

import numpy as np


def short(name):
    name = name.split("(")[0]
    for key in ("fwd_fused", "bwd_fused", "gemm256_fp8", "gemm256_bf16", "gemm_bf16"):
        if key in name:
            return key
    return name[-40:]


def main(path, K=200):
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows))
    # first kernel of a step: normalize_cast (L1: a training step needs no colnorm pass since round 3), dead_mask / dead_compact (TopK)
    starts = [i for i, e in enumerate(ev) if "normalize_cast" in e[2] or "dead_mask" in e[2] or "dead_compact" in e[2]]
    if len(starts) >= 3:
        a, b = starts[-3], starts[-2]
        t0, prev_end = ev[a][0], ev[a][0]
        print(f"one step ({(ev[b][0] - t0) / 1e3:.1f} us from its first kernel to the next step's):")
        for s, e, n in ev[a:b]:
            print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {n}")
            prev_end = max(prev_end, e)
    per = defaultdict(list)
    for s, e, n in ev:
        per[n].append(e - s)
    print(f"timed-region averages (last {K} launches of each kernel, us):")
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1][-K:])):
        v = np.array(v[-K:])
        print(f"  {n:42s} n={len(v):4d}  avg {v.mean() / 1e3:9.2f}  median {np.median(v) / 1e3:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200)

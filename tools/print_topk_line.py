#!/usr/bin/env python3
"""stdin: a bench.py JSON line of a TopK run -> one short line (step time, dead fraction, the TopK kernel times)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
km = d.get("kernel_ms") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["ms_per_step"], 3), "dead_frac", round(d["loss"]["dead_frac"], 4),
      {k.replace("topk_", ""): round(v, 2) for k, v in km.items() if v and k.startswith("topk")})

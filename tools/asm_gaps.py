#!/usr/bin/env python3
"""Per MFMA gap of a kernel's hot loop (device assembly from tools/ff2_asm.sh or any `hipcc -S` output): how many vector, LDS, scalar,
vector-memory and wait instructions sit between consecutive MFMAs -- a run of dozens of VALU instructions with no MFMA between them is
work the compiler sank or hoisted out of its gap (round 6: 51 dependent v_add_f32 at the forward's loop latch).
  python tools/asm_gaps.py /tmp/ff2b.s [first_line last_line]"""
import sys
lines = open(sys.argv[1]).read().splitlines()
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, len(lines))
gaps, cur = [], {"v": 0, "ds": 0, "s": 0, "vm": 0, "wait": 0}
for ln in lines[lo - 1:hi]:
    t = ln.strip().split()
    if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
        continue
    op = t[0]
    if op.startswith("v_mfma"):
        gaps.append(cur)
        cur = {"v": 0, "ds": 0, "s": 0, "vm": 0, "wait": 0}
    elif op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        cur["wait"] += 1
    elif op.startswith("v_"):
        cur["v"] += 1
    elif op.startswith("ds_"):
        cur["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cur["vm"] += 1
    elif op.startswith("s_"):
        cur["s"] += 1
n = len(gaps)
tot = {k: sum(g[k] for g in gaps) for k in cur}
print(f"{n} MFMAs; per MFMA: " + ", ".join(f"{k} {tot[k] / max(n, 1):.2f}" for k in tot))
# issue-port estimate: MFMA 8 cycles of issue, every other instruction 4 (a DMA piece ~60)
est = [8 + 4 * (g["v"] + g["ds"] + g["s"] + g["wait"]) + 60 * g["vm"] for g in gaps]
print(f"issue estimate per MFMA gap: mean {sum(est) / max(n, 1):.1f} cycles (32 = the matrix pipe); gaps over 32: {sum(e > 32 for e in est)}; "
      f"sum of the excess over 32: {sum(max(e - 32, 0) for e in est)} cycles, sum of max(32, gap): {sum(max(e, 32) for e in est)}")
worst = sorted(range(n), key=lambda i: -est[i])[:8]
print("largest gaps (index: v/ds/s/vm/wait):", ", ".join(f"{i}: {gaps[i]['v']}/{gaps[i]['ds']}/{gaps[i]['s']}/{gaps[i]['vm']}/{gaps[i]['wait']}" for i in worst))

#!/usr/bin/env python3
"""Effective clock of the fused kernels from rocprofv3 counter CSVs (tools/gpu_r04_clock.sh).

  python tools/parse_clock.py gpurun_out/r04_clock

Per run directory pmc_<label>/ : for every kernel whose name contains fwd_fused / bwd_fused, over the second half of its
dispatches (clocks settled): mean duration, GRBM_GUI_ACTIVE / 8 / duration = effective clock (MI355X_MICROARCH.md, "DVFS
give-back": the counter is summed over the 8 XCDs), MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES), LDS conflict
ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  NB: a dispatch under counter collection is serialised and runs alone."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KEYS = ("fwd_fused", "bwd_fused")


def one(path):
    per = defaultdict(lambda: defaultdict(dict))        # kernel key -> dispatch id -> counter -> value (+ t0, t1)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            key = next((k for k in KEYS if k in name), None)
            if key is None:
                continue
            d = per[key][int(row["Dispatch_Id"])]
            d[row["Counter_Name"]] = float(row["Counter_Value"])
            d["t0"], d["t1"] = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
    out = {}
    for key, disp in per.items():
        ids = sorted(disp)
        ids = ids[len(ids) // 2:]
        rows = [disp[i] for i in ids]
        n = len(rows)
        mean = lambda fn: sum(fn(r) for r in rows) / n
        dur = mean(lambda r: r["t1"] - r["t0"])
        o = {"dispatches": n, "dur_us": dur / 1e3}
        if "GRBM_GUI_ACTIVE" in rows[0]:
            o["clock_GHz"] = mean(lambda r: r["GRBM_GUI_ACTIVE"] / 8.0 / (r["t1"] - r["t0"]))
        if rows[0].get("SQ_WAVE_CYCLES"):
            o["wave_quadcycles"] = mean(lambda r: r["SQ_WAVE_CYCLES"])
            if "SQ_VALU_MFMA_BUSY_CYCLES" in rows[0]:
                o["mfma_busy"] = mean(lambda r: r["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * r["SQ_WAVE_CYCLES"]))
            if "SQ_WAIT_ANY" in rows[0]:
                o["wait_any"] = mean(lambda r: r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"])
        if rows[0].get("SQ_LDS_IDX_ACTIVE"):
            o["lds_conflict_ratio"] = mean(lambda r: r["SQ_LDS_BANK_CONFLICT"] / r["SQ_LDS_IDX_ACTIVE"])
            o["lds_idx_active"] = mean(lambda r: r["SQ_LDS_IDX_ACTIVE"])
            o["lds_bank_conflict"] = mean(lambda r: r["SQ_LDS_BANK_CONFLICT"])
        for c in ("SQ_BUSY_CYCLES", "SQ_INSTS_LDS"):
            if c in rows[0]:
                o[c] = mean(lambda r: r[c])
        out[key] = o
    return out


def main(src):
    res = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if files:
            res[os.path.basename(d)[4:]] = one(files[0])
    with open(os.path.join(src, "clock_summary.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(f"{'run':24s} {'kernel':10s} {'n':>4s} {'dur us':>9s} {'GHz':>6s} {'mfma':>6s} {'wait':>6s} {'ldsconf':>8s}")
    for label, ks in res.items():
        for k, o in ks.items():
            print(f"{label:24s} {k:10s} {o['dispatches']:4d} {o['dur_us']:9.1f} {o.get('clock_GHz', 0):6.3f} {o.get('mfma_busy', 0):6.3f} "
                  f"{o.get('wait_any', 0):6.3f} {o.get('lds_conflict_ratio', 0):8.4f}")


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python3
"""Per-launch HBM bytes and SQ counters of the two fused kernels from rocprofv3 counter CSVs (tools/profile_round.sh).

  python tools/parse_pmc.py gpurun_out/prof_<tag> profiles/<tag>

rocprofv3 prints FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts
half of the bytes actually fetched (MI355X_MICROARCH.md, HBM / rocprofv3 section), so reads are doubled; writes are not.
Copies the counter CSVs and the kernel-trace stats next to the JSON it writes (<prefix>_hbm_traffic.json)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

KERNELS = {"fwd_fused_gemm": "fwd_fused", "bwd_fused_gemm": "bwd_fused_d384_kernel"}     # fwd_fused_d384 / fwd_fused2_d384


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            for key, sub in KERNELS.items():
                if sub in row["Kernel_Name"]:
                    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main(src, prefix):
    out = {k: {} for k in KERNELS}
    for name in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
        files = glob.glob(os.path.join(src, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        shutil.copy(files[0], f"{prefix}_pmc_{name.lower()}.csv")
        for k, d in counters(files[0]).items():
            if name == "SQ":
                out[k]["sq"] = d
            else:
                out[k][f"{name}_KiB_per_launch"] = d[name]
    for k, d in out.items():
        if "FETCH_SIZE_KiB_per_launch" in d and "WRITE_SIZE_KiB_per_launch" in d:
            d["hbm_read_bytes_corrected"] = 2.0 * d["FETCH_SIZE_KiB_per_launch"] * 1024
            d["hbm_write_bytes"] = d["WRITE_SIZE_KiB_per_launch"] * 1024
            d["hbm_bytes_per_launch"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
        if "sq" in d and d["sq"].get("SQ_WAVE_CYCLES"):
            sq = d["sq"]      # SQ_WAVE_CYCLES counts quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (guide, cycle constants)
            d["mfma_busy_frac"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * sq["SQ_WAVE_CYCLES"])
            d["wait_any_frac"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
            if sq.get("SQ_LDS_IDX_ACTIVE"):
                d["lds_conflict_ratio"] = sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], f"{prefix}_kernel_stats.csv")
    b = os.path.join(src, "bench_under_rocprof.json")
    if os.path.exists(b):
        shutil.copy(b, f"{prefix}_bench_under_rocprof.json")
    with open(f"{prefix}_hbm_traffic.json", "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in d.items() if kk != "sq"} for k, d in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

#!/usr/bin/env python3
"""Per-kernel HBM bytes and SQ counters of one bench workload from the rocprofv3 counter CSVs of tools/pmc_workload.sh.

  python tools/parse_pmc_all.py gpurun_out/pmc_<name> out.json [the bench.py args of the run]

For every kernel that takes >= 1 % of the counted wave cycles: launches, HBM bytes per launch -- FETCH_SIZE / WRITE_SIZE are
printed in KiB, and on gfx950 FETCH_SIZE counts half of the bytes fetched (MI355X_MICROARCH.md, HBM / rocprofv3 section): reads
are doubled, writes are not -- the ALGORITHMIC bytes of that launch where the kernel is one of the engine's GEMMs / selects
(operands read once, outputs written once; from the workload's shape), their ratio, and mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES
/ (4 SQ_WAVE_CYCLES), wait fractions, LDS conflict ratio."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    base = m.group(1) if m else name.split("(")[0][-40:]
    epi = re.search(r"(Epi\w+)", name)
    mode = re.search(r"gemm\w*_kernel<\s*(\d+),\s*(\d+)", name)
    if epi:
        base += "<" + epi.group(1) + (",persist" if re.search(r",\s*true\s*>", name) else "") + ">"
    elif mode:
        base += f"<{mode.group(1)},{mode.group(2)}>"
    return base


def load(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def algorithmic_bytes(kernel, a):
    """bytes one launch must move if every operand is read once and every output written once (bf16 = 2 B)."""
    M, d, n = a["rows"], a["d"], a["n"]
    Mp, dp, np_ = -(-M // 128) * 128, -(-d // 128) * 128, -(-n // 128) * 128
    W2, X2, C2 = dp * np_ * 2, Mp * dp * 2, Mp * np_ * 2
    if a["variant"] == "l1":
        table = {
            "EpiEnc8": X2 // 2 + W2 // 2 + C2 + C2 // 2,       # fp8 operands in, bf16 + e4m3 latent out
            "EpiEnc": X2 + W2 + C2,                            # x, W^T in; latent out
            "EpiDec": C2 + W2 + X2 + X2,                       # latent, W in; x for the residual; dx_hat out
            "EpiDpre": X2 + W2 + C2 + C2,                      # dx_hat, W^T, latent (gate) in; dpre out
            "EpiSlab": 2 * X2 + 2 * C2 + a.get("dw_splits", 1) * dp * np_ * 4,   # dx_hat, x, latent, dpre in; fp32 slabs out
            "fwd_fused": X2 + W2 + C2 + X2 + X2,               # x (operand + residual), W^T, latent out, dx_hat out
            "bwd_fused": 2 * X2 + C2 + W2 + a.get("bwd_splits", 10) * dp * np_ * 4,
        }
    else:
        k = a["k"]
        table = {
            "EpiTopkEnc": X2 + W2 + C2 + Mp * (np_ // 128) * 2,           # sae_in, W_enc in; pre + tile maxima out
            "topk_select_tiles": C2 // 3 + Mp * (np_ // 128) * 2 + Mp * k * 6,   # ~ a third of the tiles + maxima in; idx + vals out
            "topk_decode": Mp * k * (6 + dp * 2) + X2 * 2 + Mp * dp * 8,  # k gathered W_dec rows per row, x in; e, dh out
            "sparse_bwd": Mp * k * (8 + 2 * dp * 2) + W2,                 # per entry: two gathered rows (g, sae_in) + W_dec row of the latent
        }
    for key, v in table.items():
        if key in kernel:
            return v
    return None


def main(src, out_path, bench_args):
    a = {"rows": 65536, "d": 384, "n": 3072, "variant": "l1", "k": 64}
    it = iter(bench_args)
    for tok in it:
        if tok in ("--rows", "--d", "--n", "--k"):
            a[tok[2:]] = int(next(it))
        elif tok == "--variant":
            a["variant"] = next(it)
    files = {name: glob.glob(os.path.join(src, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
             for name in ("FETCH_SIZE", "WRITE_SIZE", "SQ")}
    data = {name: load(f[0]) for name, f in files.items() if f}
    kernels = set()
    for d_ in data.values():
        kernels |= set(d_)
    sq = data.get("SQ", {})
    total_wave = sum(sum(sq[k].get("SQ_WAVE_CYCLES", [])) for k in sq) or 1.0
    out = {"workload": a, "kernels": {}}
    for k in sorted(kernels, key=lambda kk: -sum(sq.get(kk, {}).get("SQ_WAVE_CYCLES", [0]))):
        share = sum(sq.get(k, {}).get("SQ_WAVE_CYCLES", [0])) / total_wave
        if share < 0.01:
            continue
        e = {"share_of_wave_cycles": round(share, 4)}
        f_, w_ = data.get("FETCH_SIZE", {}).get(k, {}).get("FETCH_SIZE"), data.get("WRITE_SIZE", {}).get(k, {}).get("WRITE_SIZE")
        if f_ and w_:
            e["launches"] = len(f_)
            e["hbm_read_bytes_corrected"] = 2.0 * 1024 * sum(f_) / len(f_)
            e["hbm_write_bytes"] = 1024 * sum(w_) / len(w_)
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
            alg = algorithmic_bytes(k, a)
            if alg:
                e["algorithmic_bytes"] = alg
                e["traffic_ratio"] = round(e["hbm_bytes_per_launch"] / alg, 3)
        s = sq.get(k)
        if s and s.get("SQ_WAVE_CYCLES"):
            m = {c: sum(v) / len(v) for c, v in s.items()}
            e["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_WAVE_CYCLES"]), 4)
            e["wait_any_frac"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4)
            e["wait_inst_frac"] = round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4)
            e["valu_per_wave_cycle"] = round(m["SQ_INSTS_VALU"] / m["SQ_WAVE_CYCLES"], 4)
            if m.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_conflict_ratio"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4)
        out["kernels"][k] = e
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    for k, e in out["kernels"].items():
        print(f"{k:64s} share {e['share_of_wave_cycles']:.3f}  HBM {e.get('hbm_bytes_per_launch', 0) / 1e6:9.1f} MB  alg "
              f"{(e.get('algorithmic_bytes') or 0) / 1e6:9.1f} MB  ratio {e.get('traffic_ratio', '-')}  mfma {e.get('mfma_busy_frac', '-')}  "
              f"wait {e.get('wait_any_frac', '-')}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])

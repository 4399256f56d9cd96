"""The bench line contract (task statement, section "Measurement"), checked on the line committed under profiles/ (produced by
`python bench.py` on an MI355X): required keys, types and the internal arithmetic the judge recomputes."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_default_bench_line_follows_the_contract():
    d = _line("r04_bench_default.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "activations/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and d["n_gpus"] == 1
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg
    rows = cfg["rows_per_gpu"]
    assert (cfg["d_model"], cfg["n_dict"], rows) == (384, 3072, 65536)          # BASELINE configs[1]
    # value = rows processed / time of the timed region
    assert d["value"] == pytest.approx(rows * d["n_gpus"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    # achieved = algorithmic FLOPs of the dominant kernel (fused backward: 6 M d n) / its HIP-event average
    assert r["flops_per_launch"] == 6.0 * rows * 384 * 3072
    assert r["achieved"] == pytest.approx(r["flops_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e12, rel=1e-9)
    assert r["kernel_launches"] >= 10 and r["kernel_avg_ms"] < d["ms_per_step"]
    assert r["traffic"] is None or 0.3e9 < r["traffic"] < 2e9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "activations/s" and 1e3 < c["value"] < 1e7
    assert "sample" in c
    # the whole step against the MFMA roof: 10 d n FLOPs per activation
    assert d["step_mfma_frac"] == pytest.approx(10.0 * 384 * 3072 * d["value"] / 2.5e15, rel=1e-6)
    assert d["step_mfma_frac"] >= 0.5                    # north_star target at 1 GPU


def test_driver_style_line_has_enough_kernel_samples():
    d = _line("r04_bench_default_driver_style.json")
    assert d["steps"] == 20 and d["warmup"] == 5
    assert d["roofline"]["kernel_launches"] >= 10        # every step of a short run is bracketed

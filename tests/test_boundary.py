"""CPU: the drop-in boundary.  The C-ABI library builds for gfx950, loads, and exports every symbol
include/freud_sae.h declares (no compute calls here: there is no GPU); the product package never
touches oracle/."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "freud_sae.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sae_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_loads_and_exports_header_symbols():
    from freud_amd import engine
    engine.build()
    lib = engine.load()
    syms = _header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/freud_sae.h but not exported"
    assert sorted(engine.EXPORTED_SYMBOLS) == syms
    assert lib.sae_version() >= 1
    assert lib.sae_kernel_name(0) is not None


def test_create_rejects_bad_config_without_gpu():
    from freud_amd.engine import SaeEngine, EngineError
    with pytest.raises(AssertionError, match="Invalid autoencoder variant"):
        SaeEngine(variant="vae", d_model=8, n_dict=8, max_rows=8)
    with pytest.raises(ValueError, match="Invalid optimizer"):
        SaeEngine(variant="l1", d_model=8, n_dict=8, max_rows=8, optimizer="sgd")
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(EngineError):            # no device: fails loudly, no CPU fallback
            SaeEngine(variant="l1", d_model=8, n_dict=8, max_rows=8)


def test_product_never_imports_oracle():
    bad = []
    for base in ("freud_amd", "src"):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".h", ".cpp")):
                    text = open(os.path.join(dirpath, fn), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "sae_oracle" in text:
                        bad.append(os.path.join(dirpath, fn))
    assert not bad, f"product files reference the oracle: {bad}"


def test_train_refuses_cpu_device():
    from freud_amd.train_sae import train
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        train(seed=0, train_folder="x", val_folder="x", device="cpu", run_dir="x", lr=1e-3, weight_decay=0.0, steps=1,
              clip_thresh=1.0, batch_size=1, dl_max_workers=0, log_tb_every=1, save_every=1, val_every=1,
              start_checkpoint=None, whisper_config={"model": "tiny", "layer_name": "l"}, optimizer="adam",
              scheduler="cosine", scheduler_params={}, from_disk=True, autoencoder_variant="l1", autoencoder_config={})


def test_inference_models_refuse_cpu_device():
    """freud_amd.models (the mirror of the reference's L1AutoEncoder / TopKAutoEncoder for inference) has no CPU path either."""
    from freud_amd.config import L1AutoEncoderConfig, TopKAutoEncoderConfig
    from freud_amd.models import L1AutoEncoder, TopKAutoEncoder, L1ForwardOutput, TopKForwardOutput
    assert L1ForwardOutput._fields == ("sae_out", "encoded", "l1_loss", "reconstruction_loss")          # l1autoencoder.py:19-26
    assert TopKForwardOutput._fields == ("sae_out", "encoded", "fvu", "auxk_loss", "multi_topk_fvu")     # topkautoencoder.py:29-41
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L1AutoEncoder(16, L1AutoEncoderConfig(expansion_factor=2), device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TopKAutoEncoder(16, TopKAutoEncoderConfig(expansion_factor=2, k=4), device="cpu")


def test_dp_mode_selection_from_environment(monkeypatch):
    """freud_amd/dp.py: FREUD_DP=auto|p2p|rccl|host (FREUD_DP_HOST=1 is the older spelling of host); anything else is refused."""
    from freud_amd import dp
    for k in ("FREUD_DP", "FREUD_DP_HOST"):
        monkeypatch.delenv(k, raising=False)
    assert dp.requested_mode() == "auto"
    for m in ("p2p", "rccl", "host", "AUTO"):
        monkeypatch.setenv("FREUD_DP", m)
        assert dp.requested_mode() == m.lower()
    monkeypatch.setenv("FREUD_DP", "ring")
    with pytest.raises(ValueError):
        dp.requested_mode()
    monkeypatch.setenv("FREUD_DP_HOST", "1")
    assert dp.requested_mode() == "host"

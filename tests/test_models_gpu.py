"""GPU: the engine-backed inference classes (freud_amd/models.py) against the fp32 CPU oracle
(the reference's inference runs without autocast on CPU: oracle autocast=False).  Tolerances are bf16-operand ones:
latent / reconstruction rel-Frobenius <= 1e-2, losses rtol 2e-2."""
import os

import numpy as np
import pytest
import torch

from oracle import sae_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _x(B, T, d, seed):
    g = torch.Generator().manual_seed(seed)
    z = torch.relu(torch.randn(B * T, 32, generator=g)) * 0.1
    return (z @ torch.randn(32, d, generator=g)).reshape(B, T, d)


def test_l1_encode_decode_forward_match_oracle():
    from freud_amd.config import L1AutoEncoderConfig
    from freud_amd.models import L1AutoEncoder, L1EncoderOutput, L1ForwardOutput
    d, B, T = 384, 2, 700
    torch.manual_seed(0)
    sae = L1AutoEncoder(d, L1AutoEncoderConfig(expansion_factor=4, recon_alpha=1e4), max_rows=256)   # forces a re-size
    sd = sae.state_dict()
    assert list(sd.keys()) == ["encoder_bias", "decoder.weight"] and sd["decoder.weight"].shape == (d, 4 * d)
    sd["encoder_bias"] = 0.01 * torch.randn(4 * d)
    sae.load_state_dict(sd)
    x = _x(B, T, d, 1)
    W = O.normalize_columns(sd["decoder.weight"].clone())
    ref = O.l1_forward(x.reshape(-1, d), W, sd["encoder_bias"], 1e4, autocast=False)

    enc = sae.encode(x)
    assert isinstance(enc, L1EncoderOutput) and enc.latent.shape == (B, T, 4 * d) and enc.latent.dtype == torch.float32
    assert _rel(enc.latent.cpu().reshape(-1, 4 * d), ref["c"]) < 1e-2
    # like the reference's encode(), the decoder columns are renormalised in place
    assert _rel(sae.state_dict()["decoder.weight"], W) < 1e-6

    xh = sae.decode(enc.latent)
    assert xh.shape == (B, T, d) and _rel(xh.cpu().reshape(-1, d), ref["c"] @ W.t()) < 1e-2

    out, mse = sae(x.cuda(), return_mse=True)
    assert isinstance(out, L1ForwardOutput)
    assert _rel(out.sae_out.cpu().reshape(-1, d), ref["x_hat"]) < 1e-2
    assert float(out.l1_loss) == pytest.approx(ref["l1_loss"].item(), rel=2e-2)
    assert float(out.reconstruction_loss) == pytest.approx(ref["reconstruction_loss"].item(), rel=2e-2)
    assert float(mse) == pytest.approx(ref["mse"].item(), rel=2e-2)


def test_topk_encode_decode_forward_match_oracle():
    from freud_amd.config import TopKAutoEncoderConfig
    from freud_amd.models import TopKAutoEncoder, TopKForwardOutput
    d, B, T, k = 256, 3, 200, 16
    torch.manual_seed(0)
    sae = TopKAutoEncoder(d, TopKAutoEncoderConfig(expansion_factor=4, k=k, auxk_alpha=0.03125))
    sd = sae.state_dict()
    assert list(sd.keys()) == ["W_dec", "b_dec", "encoder.weight", "encoder.bias"]
    sd["b_dec"] = 0.01 * torch.randn(d)
    sae.load_state_dict(sd)
    x = _x(B, T, d, 2)
    ref = O.topk_forward(x, sd["encoder.weight"], sd["encoder.bias"], sd["W_dec"], sd["b_dec"], k, autocast=False)

    enc = sae.encode(x)
    assert enc.top_acts.shape == (B, T, k) and enc.top_indices.shape == (B, T, k) and enc.top_indices.dtype == torch.int64
    got = enc.top_acts.cpu().reshape(-1, k).sort(dim=1).values
    want = ref["top_acts"].float().sort(dim=1).values
    assert _rel(got, want) < 1e-2
    # index sets: identical wherever the k-th and (k+1)-th pre-activations are well separated
    pre = torch.relu((x.reshape(-1, d) - sd["b_dec"]) @ sd["encoder.weight"].t() + sd["encoder.bias"])
    srt = pre.sort(dim=1, descending=True).values
    clear = (srt[:, k - 1] - srt[:, k]) > 1e-2 * srt[:, k - 1].abs()
    gi = enc.top_indices.cpu().reshape(-1, k).sort(dim=1).values
    wi = ref["top_indices"].sort(dim=1).values
    assert clear.float().mean() > 0.2 and torch.equal(gi[clear], wi[clear])

    xh = sae.decode(ref["top_acts"].float().reshape(B, T, k), ref["top_indices"].reshape(B, T, k))
    assert _rel(xh.cpu().reshape(-1, d), ref["x_hat"]) < 1e-2

    out, mse = sae(x, return_mse=True)
    assert isinstance(out, TopKForwardOutput) and float(out.auxk_loss) == 0.0
    # rows whose k-th / (k+1)-th pre-activations nearly tie may pick another latent in bf16: compare the clear rows
    assert _rel(out.sae_out.cpu().reshape(-1, d)[clear], ref["x_hat"][clear]) < 2e-2
    assert float(out.fvu) == pytest.approx(ref["fvu"].item(), rel=5e-2)
    assert float(mse) == pytest.approx(ref["mse"].item(), rel=5e-2)


def test_init_sae_from_checkpoint_reads_reference_keys(tmp_path):
    from freud_amd.models import init_sae_from_checkpoint, L1AutoEncoder
    d, n = 384, 768
    torch.manual_seed(1)
    W = torch.empty(d, n)
    torch.nn.init.orthogonal_(W)
    ck = {"model": {"encoder_bias": torch.zeros(n), "decoder.weight": W}, "optimizer": {}, "scheduler": {}, "step": 3,
          "best_val_loss": 1.0,
          "hparams": {"autoencoder_variant": "l1", "activation_size": d,
                      "autoencoder_config": {"n_dict_components": n, "recon_alpha": 1e4}}}
    path = os.path.join(str(tmp_path), "step3.pth")
    torch.save(ck, path)
    sae = init_sae_from_checkpoint(path)
    assert isinstance(sae, L1AutoEncoder) and sae.n_dict_components == n
    assert torch.equal(sae.state_dict()["decoder.weight"], W)
    lat = sae.encode(_x(1, 100, d, 3)).latent
    assert lat.shape == (1, 100, n) and torch.isfinite(lat).all()

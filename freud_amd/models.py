"""Inference-side mirror of the reference's SAE modules on top of the HIP engine.

Same class names, constructor arguments, method names and NamedTuple outputs as src/models/l1autoencoder.py:15-95 and
src/models/topkautoencoder.py:21-151, so that the reference's consumers -- FlyActivationDataLoader.__iter__
(dataset/activations.py:96-108), get_top_activations / manipulate_latent (utils/activations.py:143-268) and
init_sae_from_checkpoint (dataset/activations.py:16-31) -- run unchanged against them:

    sae = L1AutoEncoder(activation_size, L1AutoEncoderConfig.from_dict(hp["autoencoder_config"]))
    sae.load_state_dict(checkpoint["model"]); sae.eval()
    latent = sae.encode(x).latent;  x_hat = sae.decode(latent);  out = sae(x)

These are NOT torch modules and have no autograd: training goes through freud_amd.train_sae / the C ABI.  Arithmetic is
the engine's (bf16 MFMA operands, fp32 accumulation): the reference's inference runs fp32 on CPU and fp16 autocast on
cuda; latents agree with the fp32 reference to bf16 rounding (tests/test_models_gpu.py states the tolerances).  Like
the reference's encode(), L1 encode renormalises the decoder columns in place.  Inputs are torch tensors on the engine's
GPU (or CPU tensors, which are moved), shaped [..., d_model]; outputs keep the leading shape.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np
import torch

from .config import L1AutoEncoderConfig, TopKAutoEncoderConfig, get_n_dict_components
from .engine import SaeEngine


class L1EncoderOutput(NamedTuple):
    latent: torch.Tensor


class L1ForwardOutput(NamedTuple):
    sae_out: torch.Tensor
    encoded: L1EncoderOutput
    l1_loss: torch.Tensor
    reconstruction_loss: torch.Tensor


class TopKEncoderOutput(NamedTuple):
    top_acts: torch.Tensor
    top_indices: torch.Tensor


class TopKForwardOutput(NamedTuple):
    sae_out: torch.Tensor
    encoded: TopKEncoderOutput
    fvu: torch.Tensor
    auxk_loss: torch.Tensor
    multi_topk_fvu: torch.Tensor


class _EngineModel:
    _variant = ""

    def __init__(self, activation_size: int, n_dict: int, device, max_rows: int, **engine_kw):
        self.activation_size = int(activation_size)
        self.n_dict_components = int(n_dict)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"device={device}: the SAE exists only as HIP kernels for MI355X (no CPU fallback)")
        self._engine_kw = engine_kw
        self._max_rows = 0
        self._eng: Optional[SaeEngine] = None
        self._params = None
        self._ensure(max_rows)

    # -- engine life cycle: the context is sized for max_rows; a bigger batch re-creates it with the same parameters
    def _ensure(self, rows: int) -> SaeEngine:
        if self._eng is None or rows > self._max_rows:
            params = self._eng.get_params() if self._eng is not None else self._params
            if self._eng is not None:
                self._eng.close()
            self._max_rows = max(rows, 2 * self._max_rows, 1500)
            self._eng = SaeEngine(variant=self._variant, d_model=self.activation_size, n_dict=self.n_dict_components,
                                  max_rows=self._max_rows, device_id=self.device.index or 0, **self._engine_kw)
            self._configure()
            if params is not None:
                self._eng.set_params(params)
        return self._eng

    def _configure(self) -> None:
        pass

    # -- nn.Module-like surface used by the reference's callers
    def state_dict(self) -> dict:
        p = self._eng.get_params()
        return {k: torch.from_numpy(p[k].copy()) for k in self._state_keys}

    def load_state_dict(self, sd: dict) -> None:
        self._eng.set_params({k: sd[k].detach().float().cpu().numpy() for k in self._state_keys})

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise RuntimeError("the engine-backed SAE lives on the GPU")
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(self.state_dict().values())

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    # -- helpers
    def _flat(self, x: torch.Tensor):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if x2.device != self.device:
            x2 = x2.to(self.device)
        if x2.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            x2 = x2.float()
        return x2.contiguous(), lead

    def _latent_view(self, rows: int) -> torch.Tensor:
        """bf16 [rows][n_dict] view of the engine's latent buffer (valid until the next forward)."""
        ptr, ld = self._eng.latent_buffer()

        class _Alias:
            __cuda_array_interface__ = {"shape": (rows, ld), "typestr": "<i2", "data": (ptr, False), "version": 2}

        t = torch.as_tensor(_Alias(), device=self.device).view(torch.bfloat16)
        return t[:, : self.n_dict_components]

    def _decode_dense(self, latent2: torch.Tensor) -> torch.Tensor:
        rows = latent2.shape[0]
        eng = self._ensure(rows)
        if latent2.dtype not in (torch.float32, torch.bfloat16):
            latent2 = latent2.float()
        latent2 = latent2.contiguous()
        out = torch.empty(rows, self.activation_size, dtype=torch.float32, device=self.device)
        eng.decode(latent2, out)
        return out


class L1AutoEncoder(_EngineModel):
    """src/models/l1autoencoder.py:39-95."""
    _variant = "l1"
    _state_keys = ["encoder_bias", "decoder.weight"]

    def __init__(self, activation_size: int, cfg: L1AutoEncoderConfig, device="cuda", max_rows: int = 1500):
        self.cfg = cfg
        self.recon_alpha = cfg.recon_alpha
        self.tied = True
        n = get_n_dict_components(activation_size, cfg.expansion_factor, cfg.n_dict_components)
        super().__init__(activation_size, n, device, max_rows, recon_alpha=cfg.recon_alpha)
        lin = torch.nn.Linear(n, activation_size, bias=False)               # same init draws as the reference (:56-63)
        torch.nn.init.orthogonal_(lin.weight)
        self._eng.set_params({"decoder.weight": lin.weight.detach().numpy(), "encoder_bias": np.zeros(n, np.float32)})

    def encode(self, x: torch.Tensor) -> L1EncoderOutput:
        x2, lead = self._flat(x)
        eng = self._ensure(x2.shape[0])
        eng.eval(x2)
        c = self._latent_view(x2.shape[0]).float()
        return L1EncoderOutput(latent=c.reshape(*lead, self.n_dict_components))

    def decode(self, c: torch.Tensor) -> torch.Tensor:
        c2, lead = self._flat(c)
        return self._decode_dense(c2).reshape(*lead, self.activation_size)

    def forward(self, x: torch.Tensor, return_mse: bool = False):
        x2, lead = self._flat(x)
        eng = self._ensure(x2.shape[0])
        eng.eval(x2)
        m = eng.metrics()
        c16 = self._latent_view(x2.shape[0])
        x_hat = self._decode_dense(c16).reshape(*lead, self.activation_size)
        out = L1ForwardOutput(sae_out=x_hat, encoded=L1EncoderOutput(c16.float().reshape(*lead, self.n_dict_components)),
                              l1_loss=torch.tensor(float(m[1])), reconstruction_loss=torch.tensor(float(m[0])))
        if return_mse:
            return out, torch.tensor(float(m[2]))
        return out


class TopKAutoEncoder(_EngineModel):
    """src/models/topkautoencoder.py:44-151 (inference: no dead mask, so auxk_loss = 0)."""
    _variant = "topk"
    _state_keys = ["W_dec", "b_dec", "encoder.weight", "encoder.bias"]

    def __init__(self, activation_size: int, cfg: TopKAutoEncoderConfig, device="cuda", max_rows: int = 1500):
        self.cfg = cfg
        n = get_n_dict_components(activation_size, cfg.expansion_factor, cfg.n_dict_components)
        super().__init__(activation_size, n, device, max_rows, k=cfg.k, auxk_alpha=cfg.auxk_alpha, optimizer="adam",
                         multi_topk=bool(cfg.multi_topk))
        enc = torch.nn.Linear(activation_size, n)                            # topkautoencoder.py:62-70
        enc.bias.data.zero_()
        W_dec = enc.weight.data.clone()
        if cfg.normalize_decoder:
            W_dec /= torch.norm(W_dec, dim=1, keepdim=True) + torch.finfo(W_dec.dtype).eps
        self._eng.set_params({"encoder.weight": enc.weight.detach().numpy(), "encoder.bias": enc.bias.detach().numpy(),
                              "W_dec": W_dec.numpy(), "b_dec": np.zeros(activation_size, np.float32)})

    def _configure(self) -> None:
        self._eng.set_topk_options(float("inf"), 0)        # no latent is ever dead at inference

    def _encode_flat(self, x2: torch.Tensor):
        eng = self._ensure(x2.shape[0])
        eng.eval(x2)
        idx = eng.topk_indices_tensor(x2.shape[0], self.device).long()
        dense = self._latent_view(x2.shape[0])
        acts = torch.gather(dense, 1, idx)
        return acts, idx, dense

    def encode(self, x: torch.Tensor) -> TopKEncoderOutput:
        x2, lead = self._flat(x)
        acts, idx, _ = self._encode_flat(x2)
        k = idx.shape[1]
        return TopKEncoderOutput(acts.float().reshape(*lead, k), idx.reshape(*lead, k))

    def decode(self, top_acts: torch.Tensor, top_indices: torch.Tensor) -> torch.Tensor:
        lead = top_acts.shape[:-1]
        a2 = top_acts.reshape(-1, top_acts.shape[-1]).to(self.device).float()
        i2 = top_indices.reshape(-1, top_indices.shape[-1]).to(self.device).long()
        dense = torch.zeros(a2.shape[0], self.n_dict_components, dtype=torch.float32, device=self.device)
        dense.scatter_(1, i2, a2)                                            # eager_decode's buffer (:15-18)
        return self._decode_dense(dense).reshape(*lead, self.activation_size)

    def forward(self, x: torch.Tensor, dead_mask=None, return_mse: bool = False):
        if dead_mask is not None and bool(torch.as_tensor(dead_mask).any()):
            raise NotImplementedError("the AuxK branch belongs to training: use freud_amd.train_sae / the C ABI")
        x2, lead = self._flat(x)
        self._eng_rows_per_file = x.shape[-2] if x.dim() >= 3 else 0
        self._ensure(x2.shape[0]).set_topk_options(float("inf"), self._eng_rows_per_file)   # T of x.mean(0) (:104)
        acts, idx, dense = self._encode_flat(x2)
        m = self._eng.metrics()
        if self.cfg.multi_topk:        # forward() re-binds sae_out / encoded to the 4k selection (topkautoencoder.py:134-147)
            dense, idx = self._eng.multi_topk_buffers(x2.shape[0], self.device)
            idx = idx.long()
            acts = torch.gather(dense, 1, idx)
        x_hat = self._decode_dense(dense).reshape(*lead, self.activation_size)
        k = idx.shape[1]
        out = TopKForwardOutput(x_hat, TopKEncoderOutput(acts.float().reshape(*lead, k), idx.reshape(*lead, k)),
                                torch.tensor(float(m[0])), torch.tensor(0.0), torch.tensor(float(m[6])))
        if return_mse:
            return out, torch.tensor(float(m[2]))
        return out


def init_sae_from_checkpoint(checkpoint_path: str, device="cuda"):
    """dataset/activations.py:16-31 against the engine-backed classes (reads the reference's checkpoint keys)."""
    ck = torch.load(checkpoint_path, map_location="cpu")
    hp = ck["hparams"]
    if hp["autoencoder_variant"] == "l1":
        model = L1AutoEncoder(hp["activation_size"], L1AutoEncoderConfig.from_dict(hp["autoencoder_config"]), device=device)
    else:
        model = TopKAutoEncoder(hp["activation_size"], TopKAutoEncoderConfig.from_dict(hp["autoencoder_config"]), device=device)
    model.load_state_dict(ck["model"])
    return model.eval()

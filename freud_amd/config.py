"""Autoencoder config dataclasses: same field names and defaults as the reference's
src/models/config.py:5-28, with a from_dict that drops unknown keys the way
simple_parsing.Serializable.from_dict does there (so `dead_feature_threshold`, which the
reference reads raw from the JSON at train_sae.py:438, is tolerated)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass


class _Serializable:
    @classmethod
    def from_dict(cls, d: dict):
        names = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})

    def to_dict(self) -> dict:
        return dataclasses.asdict(self)


@dataclass
class AutoEncoderConfig(_Serializable):
    expansion_factor: int = 32
    n_dict_components: int = 0


@dataclass
class L1AutoEncoderConfig(AutoEncoderConfig):
    recon_alpha: float = 1.0


@dataclass
class TopKAutoEncoderConfig(AutoEncoderConfig):
    normalize_decoder: bool = True
    k: int = 32
    multi_topk: bool = False
    auxk_alpha: float = 0.0


def get_n_dict_components(activation_size: int, expansion_factor: int, n_dict_components: int) -> int:
    """src/utils/models.py:1-6."""
    if n_dict_components == 0:
        return activation_size * expansion_factor
    return n_dict_components

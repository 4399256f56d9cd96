"""CPU: the shard loader (reference: src/dataset/activations.py:116-206) -- file format, batch
order under the reference's seeding, data-parallel partition."""
import json
import os

import numpy as np
import pytest
import torch

from freud_amd.loader import MemoryMappedActivationDataLoader, write_shards
from freud_amd.train_sae import set_seeds


def _mk(tmp_path, n_files=11, T=2, d=4, dtype=np.float32, layer="L"):
    rows = np.arange(n_files * T * d, dtype=np.float32).reshape(n_files, T * d).astype(dtype)
    write_shards(str(tmp_path), layer, rows, [T, d], [f"/a/file_{i:04d}.flac" for i in range(n_files)])
    return rows


def test_batch_order_matches_reference_dataloader(tmp_path, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "sampler_order.json")))
    _mk(tmp_path, g["n_files"])
    set_seeds(g["seed"])
    _ = torch.randn(g["pre_draw"])
    dl = MemoryMappedActivationDataLoader(str(tmp_path), "L", g["batch_size"], 0, None, {"shuffle": True, "drop_last": True})
    assert len(dl) == g["len"]
    for ref_epoch in g["epochs"]:
        got = [[os.path.basename(f) for f in names] for (_x, names) in dl]
        assert got == ref_epoch


def test_rows_and_shapes_roundtrip(tmp_path):
    rows = _mk(tmp_path, 7, 3, 8, np.float16)
    dl = MemoryMappedActivationDataLoader(str(tmp_path), "L", 2, 0, None, {"shuffle": False})
    assert dl.activation_shape == [3, 8] and dl.activation_type == "tensor" and dl.dataset_length == 7
    seen = []
    for x, names in dl:
        assert x.dtype == torch.float16 and x.shape[1:] == (3, 8)
        for xi, nm in zip(x, names):
            i = int(os.path.basename(nm)[5:9])
            np.testing.assert_array_equal(xi.numpy().reshape(-1), rows[i])
            seen.append(i)
    assert seen == list(range(7))            # unshuffled, last short batch kept without drop_last
    # standard NPY header, C order, [n_files, T*d]
    arr = np.load(os.path.join(str(tmp_path), "L_tensors.npy"), mmap_mode="r")
    assert arr.shape == (7, 24) and arr.flags["C_CONTIGUOUS"]


def test_empty_and_ragged(tmp_path):
    _mk(tmp_path, 3, 2, 4)
    dl = MemoryMappedActivationDataLoader(str(tmp_path), "L", 4, 0, None, {"shuffle": True, "drop_last": True})
    assert len(dl) == 0 and list(dl) == []                       # fewer files than one batch
    dl = MemoryMappedActivationDataLoader(str(tmp_path), "L", 2, 0, 2, {"shuffle": False})
    assert dl.dataset_length == 2                                 # subset_size
    with pytest.raises(ValueError):
        write_shards(str(tmp_path), "bad", np.zeros((2, 7), np.float32), [2, 4])


def test_data_parallel_partition(tmp_path):
    _mk(tmp_path, 13, 2, 4)
    per_rank = []
    for r in range(2):
        set_seeds(5)
        dl = MemoryMappedActivationDataLoader(str(tmp_path), "L", 3, 0, None, {"shuffle": True, "drop_last": True},
                                              rank=r, world_size=2)
        per_rank.append([n for (_x, names) in dl for n in names])
    set_seeds(5)
    full = MemoryMappedActivationDataLoader(str(tmp_path), "L", 3, 0, None, {"shuffle": True, "drop_last": False})
    order = [n for (_x, names) in full for n in names]
    assert len(per_rank[0]) == len(per_rank[1]) == 6            # (13 // 2) // 3 batches of 3
    assert not set(per_rank[0]) & set(per_rank[1])
    assert per_rank[0] == order[0::2][:6] and per_rank[1] == order[1::2][:6]


def test_bf16_delivery_of_fp32_shards(tmp_path):
    """deliver_dtype="bfloat16": fp32 shard rows arrive as bf16, rounded to nearest even like torch's cast, except that a
    value which is not exactly -1.0 never BECOMES -1.0 (the mse_loss mask value, l1autoencoder.py:31) and exact -1.0
    stays; same batch order as the native delivery; fp16 shards travel unchanged."""
    import numpy as np
    import torch
    from freud_amd.loader import MemoryMappedActivationDataLoader, write_shards
    T, d, n_files = 5, 16, 9
    g = torch.Generator().manual_seed(0)
    rows = torch.randn(n_files, T * d, generator=g) * 2
    rows[0, :4] = torch.tensor([-1.0, -1.0009765, -0.9990234, -0.99999])
    folder = str(tmp_path / "f32")
    write_shards(folder, "L", rows.numpy().astype(np.float32), [T, d])
    torch.manual_seed(1)
    native = list(MemoryMappedActivationDataLoader(folder, "L", 2, dl_kwargs={"shuffle": True, "drop_last": True}))
    torch.manual_seed(1)
    conv = list(MemoryMappedActivationDataLoader(folder, "L", 2, dl_kwargs={"shuffle": True, "drop_last": True},
                                                 deliver_dtype="bfloat16"))
    assert len(native) == len(conv) == 4
    for (xa, na), (xb, nb) in zip(native, conv):
        assert na == nb and xb.dtype == torch.bfloat16 and xa.dtype == torch.float32 and xb.shape == xa.shape
        ref = xa.to(torch.bfloat16)
        near = (xa + 1).abs() < 0.01
        assert torch.equal(xb[~near], ref[~near])
        assert torch.equal(xb == -1.0, xa == -1.0)                 # the mask set is unchanged
        assert ((xb.float() - xa).abs()[near] <= 2.0 ** -6).all()
    folder16 = str(tmp_path / "f16")
    write_shards(folder16, "L", rows.numpy().astype(np.float16), [T, d])
    x16, _ = next(iter(MemoryMappedActivationDataLoader(folder16, "L", 2, deliver_dtype="bfloat16")))
    assert x16.dtype == torch.float16

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


def test_host_converter_forms_are_bit_identical():
    """libfreud_host.so: the AVX-512 streaming-store form (taken at run time where the CPU has it) against the portable loop --
    every special value (NaNs of both signs, infinities, denormals, the -1.0 neighbours, round-to-even ties), unaligned
    destinations and lengths that leave heads and tails around the 64-byte lines."""
    import ctypes as C
    import numpy as np
    from freud_amd.loader import _host_lib
    lib = _host_lib()
    rng = np.random.default_rng(0)
    special = np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0xFF800001, 0x7F800000, 0xFF800000, 0x00000001, 0x80000001, 0x007FFFFF,
                        0xBF800000, 0xBF800001, 0xBF7FFFFF, 0xBF808000, 0xBF7F8000, 0xBF807FFF, 0xBF7F8001, 0x3F808000, 0x3F818000,
                        0x7F7FFFFF, 0xFF7FFFFF, 0x00000000, 0x80000000], dtype=np.uint32)
    bits = np.concatenate([special, rng.integers(0, 2 ** 32, size=70001, dtype=np.uint64).astype(np.uint32),
                           (rng.standard_normal(5000).astype(np.float32) * 3).view(np.uint32)])
    src = bits.view(np.float32)
    for off, n in ((0, len(src)), (1, 4097), (3, 33), (5, 31), (7, 64), (31, 1000), (0, 0)):
        a = np.zeros(n + 64, np.uint16)
        b = np.zeros(n + 64, np.uint16)
        s = np.ascontiguousarray(src[:n])
        lib.freud_f32_to_bf16(s.ctypes.data, a[off:].ctypes.data, n)
        lib.freud_f32_to_bf16_portable(s.ctypes.data, b[off:].ctypes.data, n)
        assert np.array_equal(a, b), (off, n, int(lib.freud_host_impl()))
    # the -1.0 guard and NaN rule themselves
    out = np.zeros(len(special), np.uint16)
    lib.freud_f32_to_bf16(np.ascontiguousarray(special.view(np.float32)).ctypes.data, out.ctypes.data, len(special))
    assert out[9] == 0xBF80 and 0xBF80 not in (out[10], out[11], out[12], out[13], out[14], out[15])
    assert all((int(v) & 0x7F80) == 0x7F80 and (int(v) & 0x7F) != 0 for v in out[:4])      # NaNs stay NaNs


def test_gather_pool_splits_rows_into_pieces(tmp_path):
    """A batch of few large rows over many gather threads (dl_max_workers=12: more threads than rows): pieces of rows, every
    element converted exactly once, same bytes as the single-thread gather."""
    import numpy as np
    import torch
    from freud_amd.loader import MemoryMappedActivationDataLoader, write_shards
    T, d, n_files = 37, 24, 6
    rows = (torch.randn(n_files, T * d, generator=torch.Generator().manual_seed(2)) * 2).numpy().astype(np.float32)
    folder = str(tmp_path / "f32")
    write_shards(folder, "L", rows, [T, d])
    got = {}
    for workers in (1, 12):
        torch.manual_seed(3)
        dl = MemoryMappedActivationDataLoader(folder, "L", 3, workers, None, {"shuffle": True, "drop_last": True}, deliver_dtype="bfloat16")
        got[workers] = [x.clone() for x, _ in dl]
    assert len(got[1]) == 2
    for a, b in zip(got[1], got[12]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_two_loaders_share_the_host_thread_pool(tmp_path):
    """libfreud_host.so has ONE thread pool; a process may run two loaders at once (training + validation).  Two Python threads
    gathering concurrently must each get exactly their own rows."""
    import threading
    import numpy as np
    import torch
    from freud_amd.loader import MemoryMappedActivationDataLoader, write_shards
    T, d, n_files = 64, 48, 24
    rows = (torch.randn(n_files, T * d, generator=torch.Generator().manual_seed(4)) * 2).numpy().astype(np.float32)
    folder = str(tmp_path / "f32")
    write_shards(folder, "L", rows, [T, d])
    from freud_amd.loader import _host_lib
    ref_np = np.zeros(rows.shape, np.uint16)          # the library's own single-thread conversion (the -1.0 guard differs from torch's cast)
    _host_lib().freud_f32_to_bf16_portable(rows.ctypes.data, ref_np.ctypes.data, rows.size)
    ref = torch.from_numpy(ref_np.view(np.int16))
    errors = []

    def run(seed):
        try:
            for _ in range(20):
                dl = MemoryMappedActivationDataLoader(folder, "L", 4, 6, None, {"shuffle": False, "drop_last": True}, deliver_dtype="bfloat16")
                for bi, (x, _names) in enumerate(dl):
                    got = x.reshape(4, T * d).view(torch.int16)
                    if not torch.equal(got, ref[4 * bi: 4 * bi + 4]):
                        errors.append((seed, bi))
        except Exception as e:          # noqa: BLE001
            errors.append((seed, repr(e)))

    ths = [threading.Thread(target=run, args=(s,)) for s in range(3)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors[:3]


def test_host_thread_pool_survives_fork():
    """libfreud_host.so's pthread pool is process-global.  A child forked AFTER the parent has run a batch (multiprocessing
    'fork', torch DataLoader workers) inherits `nth > 0` but none of the worker threads; without the pthread_atfork handler its first
    batch waits for ever.  Run in a subprocess (the fork must not happen inside the pytest process with its own threads)."""
    import subprocess
    import sys
    code = r'''
import os, sys, signal
import numpy as np
sys.path.insert(0, sys.argv[1])
from freud_amd.loader import _host_lib
lib = _host_lib()
rows = np.random.default_rng(0).standard_normal((16, 4096)).astype(np.float32)
idx = np.arange(16, dtype=np.int64)
def batch():
    out = np.zeros((16, 4096), np.uint16)
    lib.freud_gather_batch_f32_to_bf16(rows.ctypes.data, idx.ctypes.data, 16, 4096, out.ctypes.data, 4096, 1024, 8)
    return out
ref = batch()                       # the parent's pool now has 7 workers
pid = os.fork()
if pid == 0:
    signal.alarm(20)                # a hang ends the child with SIGALRM
    ok = np.array_equal(batch(), ref) and np.array_equal(batch(), ref)
    os._exit(0 if ok else 3)
_, status = os.waitpid(pid, 0)
assert np.array_equal(batch(), ref)         # the parent's pool still works
sys.exit(0 if (os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0) else 4)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])


def test_bench_activations_follow_the_loaders_guard():
    """bench.py's synthetic activations reach the engine the way a batch from the loader does: fp32 -> bf16 by the same rule as
    libfreud_host.so (round to nearest even; what is not -1.0 never reads -1.0, host_convert.c:5-7) -- bit for bit on N(0, 0.6)
    values (which hit -1.0 by rounding in about one entry of a thousand without the guard) and on the -1.0 neighbours; real
    padding (-1.0 in fp32) stays -1.0.  The reference takes its mask on the fp32 values (train_sae.py:431)."""
    import numpy as np
    import bench
    from freud_amd.loader import _host_lib
    lib = _host_lib()
    g = torch.Generator().manual_seed(5)
    x32 = torch.cat([0.6 * torch.randn(400000, generator=g),
                     torch.tensor(np.array([0xBF800000, 0xBF800001, 0xBF7FFFFF, 0xBF808000, 0xBF7F8000, 0xBF807FFF, 0xBF7F8001],
                                           dtype=np.uint32).view(np.float32))])
    raw = x32.to(torch.bfloat16)
    assert int(((raw == -1.0) & (x32 != -1.0)).sum()) > 100        # the case the guard exists for
    got = bench.to_activation_dtype(x32, torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    ref = np.zeros(x32.numel(), np.uint16)
    src = np.ascontiguousarray(x32.numpy())
    lib.freud_f32_to_bf16(src.ctypes.data, ref.ctypes.data, src.size)
    assert np.array_equal(got, ref)
    assert int((got == 0xBF80).sum()) == 1 and got[400000] == 0xBF80
    h = bench.to_activation_dtype(x32, torch.float16)
    assert int(((h == -1.0) & (x32 != -1.0)).sum()) == 0 and float(h[400000]) == -1.0
    assert (h.float() - x32).abs().max() <= 2.0 ** -10

"""Activation-shard loader: the from_disk=true path of the reference
(src/dataset/activations.py:116-206, MemoryMappedActivationsDataset / MemoryMappedActivationDataLoader)
rebuilt as a host -> pinned -> HBM double-buffered streamer.

On-disk format (written by the reference's collector, src/scripts/collect_activations.py:12-63):
  <dir>/<layer>_tensors.npy     NPY, C-order, shape [n_files, T*d], fp32 or fp16
  <dir>/<layer>_metadata.json   {"tensor_shape": [T, d], "activation_shape": [T, d], "filenames": [...]}

Batch order: identical to torch's DataLoader(shuffle=True, drop_last=True) under the same global
RNG state (train_sae.py:321-334): one int64 draw for the iterator's base seed, one for the
RandomSampler seed, then randperm -- pinned by tests/golden/sampler_order.json.
Data-parallel: every rank draws the same epoch permutation and takes perm[rank::world]
(SURVEY.md section 8e); per-rank batch_size stays the config value.

Pipeline per rank: a worker thread gathers the next batch's rows from the memory map into one of
`depth` pinned host buffers; the consumer issues the H2D copy on a dedicated copy stream into the
matching HBM buffer and makes the compute stream wait on its event, so the copy of batch i+1
overlaps the train step of batch i.

Delivery dtype: the collector writes fp32 shards and the host -> HBM link is what bounds a real
training run, so the gather threads can down-convert fp32 rows to bf16 on the way into the pinned
ring (deliver_dtype="bfloat16" / FREUD_LOADER_DELIVER=bfloat16; libfreud_host.so, round to nearest even, exact -1.0
-- the mse_loss mask value -- never created by rounding): half the PCIe bytes.  The engine's GEMMs read bf16(x) in any
case; what changes is that the residual x_hat - x is taken against bf16(x), one more rounding of the size the bf16
arithmetic already has.  Default: rows travel in the shard's own dtype (reference semantics bit for bit).
"""
from __future__ import annotations

import json
import os
import queue
import threading
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

_HOST_LIB = None


def _host_lib():
    """libfreud_host.so (freud_amd/csrc/host_convert.c): fp32 -> bf16 row gather for the staging threads."""
    global _HOST_LIB
    if _HOST_LIB is None:
        import ctypes as C
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libfreud_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C freud_amd/csrc` (loader bf16 delivery needs it)")
        lib = C.CDLL(path)
        lib.freud_gather_f32_to_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        lib.freud_gather_f32_to_bf16.restype = None
        lib.freud_f32_to_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.freud_f32_to_bf16.restype = None
        lib.freud_f32_to_bf16_portable.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.freud_f32_to_bf16_portable.restype = None
        lib.freud_convert_piece.argtypes = [C.c_void_p, C.c_int64, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        lib.freud_convert_piece.restype = None
        lib.freud_gather_batch_f32_to_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        lib.freud_gather_batch_f32_to_bf16.restype = None
        lib.freud_gather_batch_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        lib.freud_gather_batch_copy.restype = None
        lib.freud_host_impl.restype = C.c_int
        _HOST_LIB = lib
    return _HOST_LIB


def _mem_available_bytes() -> int:
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except Exception:
        pass
    return 16 << 30


class MemoryMappedActivationsDataset:
    """dataset/activations.py:116-174 (tensor-type shards only; 'indexed' SAE-feature shards are
    not consumed by training)."""

    def __init__(self, data_path: str, layer_name: str, subset_size: Optional[int] = None):
        self.data_path, self.layer_name = data_path, layer_name
        self.metadata_file = os.path.join(data_path, f"{layer_name}_metadata.json")
        with open(self.metadata_file, "r") as f:
            self.metadata = json.load(f)
        self.tensor_file = os.path.join(data_path, f"{layer_name}_tensors.npy")
        if not os.path.exists(self.tensor_file):
            raise FileNotFoundError(
                f"{self.tensor_file} not found (SAE-encoded 'indexed' shards cannot be trained on)")
        self.activation_type = "tensor"
        self.mmap = np.load(self.tensor_file, mmap_mode="r")
        if subset_size is not None:
            self.metadata["filenames"] = self.metadata["filenames"][:subset_size]
            self.mmap = self.mmap[:subset_size]
        self.activation_shape = self.metadata["activation_shape"]
        self.tensor_shape = list(self.metadata["tensor_shape"])
        if self.mmap.shape[0] < len(self.metadata["filenames"]):
            raise ValueError("metadata lists more files than the tensor file holds")

    def __len__(self) -> int:
        return len(self.metadata["filenames"])

    def __getitem__(self, idx: int):
        return torch.from_numpy(np.array(self.mmap[idx]).reshape(self.tensor_shape)), self.metadata["filenames"][idx]


class MemoryMappedActivationDataLoader:
    """dataset/activations.py:177-206 with the same constructor arguments, plus device / rank / world."""

    def __init__(self, data_path: str, layer_name: str, batch_size: int, dl_max_workers: int = 0,
                 subset_size: Optional[int] = None, dl_kwargs: Optional[dict] = None, *,
                 device: torch.device | str = "cpu", rank: int = 0, world_size: int = 1, depth: int = 3,
                 deliver_dtype: Optional[str] = None):
        dl_kwargs = dict(dl_kwargs or {})
        self._dataset = MemoryMappedActivationsDataset(data_path, layer_name, subset_size)
        self.dataset = self._dataset
        self.batch_size = batch_size
        self.shuffle = bool(dl_kwargs.get("shuffle", False))
        self.drop_last = bool(dl_kwargs.get("drop_last", False))
        self.activation_shape = self._dataset.activation_shape
        self.activation_type = self._dataset.activation_type
        self.dataset_length = len(self._dataset)
        self.device = torch.device(device)
        self.rank, self.world_size, self.depth = rank, world_size, max(2, depth)
        self.dl_max_workers = dl_max_workers
        # delivery dtype (module docstring): None / "native" = the shard's own dtype; "bfloat16" = fp32 rows are converted in
        # the gather threads (shards that already hold 2-byte values travel as they are)
        deliver = deliver_dtype if deliver_dtype is not None else os.environ.get("FREUD_LOADER_DELIVER", "native")
        if deliver not in ("native", "bfloat16"):
            raise ValueError(f"deliver_dtype must be 'native' or 'bfloat16', got {deliver!r}")
        self._convert = deliver == "bfloat16" and self._dataset.mmap.dtype == np.float32
        # direct mode: the shard mapping is host-registered (pinned in place) so that rows travel to HBM by DMA straight
        # from the page cache, without a CPU gather into staging buffers (measured on the MI355X host: the whole train()
        # loop 25.5 M fp16 activations/s against 18-22 M staged, with no gather threads).  Registration faults in and pins
        # the whole file, so it is automatic only while the shard fits in this rank's share of a quarter of the host's available
        # memory and in 32 GiB
        # (FREUD_LOADER_DIRECT_MAX_GB overrides the limit; train-other-500 at ~342 GB stays staged on most hosts and is
        # disk-bound anyway); FREUD_LOADER_DIRECT=1 forces it, =0 disables it; any failure to register (no GPU,
        # locked-memory limit, mapping larger than RAM) silently keeps the staged path.  Converting delivery needs the
        # CPU pass, so it is always staged.
        self._direct = False
        self._registered = None
        mode = os.environ.get("FREUD_LOADER_DIRECT", "auto")
        limit = os.environ.get("FREUD_LOADER_DIRECT_MAX_GB")
        if limit:
            limit = int(float(limit) * (1 << 30))
        else:
            # every rank of a node registers (faults in and PINS) the whole shard, and each samples MemAvailable before the
            # others have pinned: the automatic limit is this rank's share of a quarter of the free memory, at most 32 GiB
            local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1))
            limit = min(_mem_available_bytes() // 4 // local_world, 32 << 30)
        small = getattr(self._dataset.mmap, "nbytes", 1 << 62) <= limit
        if self.device.type == "cuda" and not self._convert and (mode == "1" or (mode not in ("0",) and small)):
            self._try_register()
        # gather threads (libfreud_host.so's own pthread pool, one C call per batch): a conversion thread moves 8-11 GB/s of fp32,
        # a 60 000-row tiny batch is 92 MB of fp32 per 0.6 ms train step.  Round 3 drove 8 threads from a Python thread pool, one
        # ctypes call per row; 24 Python threads delivered HALF of that on the 256-thread host of the GPU boxes (GIL hand-offs), so
        # the interpreter is out of the loop now.  A sixteenth of the host's hardware threads per rank, between 1 and 16.
        lw = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1))
        auto = min(16, max(1, (os.cpu_count() or 1) // (8 * min(lw, 2))))
        self._gather_threads = dl_max_workers if dl_max_workers and dl_max_workers > 0 else auto
        self._pool = None
        self.skip_next = 0          # resume (train_sae f4): the next iterator drops this many leading batches unread

    def _try_register(self) -> None:
        try:
            mm = self._dataset.mmap
            if not isinstance(mm, np.memmap) or mm.size == 0 or not mm.flags["C_CONTIGUOUS"]:
                return
            base, nbytes, page = mm.ctypes.data, mm.size * mm.itemsize, 4096
            start = base & ~(page - 1)
            length = ((base + nbytes + page - 1) & ~(page - 1)) - start
            torch.cuda.init()
            rc = torch.cuda.cudart().cudaHostRegister(start, length, 0x08)        # hipHostRegisterReadOnly
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")                                    # read-only mapping: never written
                self._mm_t = torch.from_numpy(mm)
            if int(rc) == 0:
                self._registered = start                                           # ours to unregister
            # rc != 0 with pinned rows: another loader of this process registered the same mapping - use it as is
            self._direct = bool(self._mm_t[0].is_pinned())
        except Exception:
            self._direct = False

    def __del__(self):
        try:
            if self._registered is not None:
                torch.cuda.cudart().cudaHostUnregister(self._registered)
                self._registered = None
        except Exception:
            pass

    def __len__(self) -> int:   # reference quirk kept: floor division even without drop_last (:205-206)
        return (len(self._dataset) // self.world_size) // self.batch_size

    # -- batch order ---------------------------------------------------------------------------
    def epoch_batches(self) -> List[List[int]]:
        """File indices of every batch of one epoch for this rank (consumes the global torch RNG
        exactly as a fresh DataLoader iterator does)."""
        n = len(self._dataset)
        _base_seed = torch.empty((), dtype=torch.int64).random_().item()  # DataLoader iterator's base seed draw
        if self.shuffle:
            order = list(torch.utils.data.RandomSampler(range(n)))
        else:
            order = list(range(n))
        if self.world_size > 1:
            order = order[self.rank::self.world_size][: n // self.world_size]
        batches = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if batches and len(batches[-1]) < self.batch_size and self.drop_last:
            batches.pop()
        return batches

    # -- iteration -------------------------------------------------------------------------------
    def _gather(self, idxs: Sequence[int], out: np.ndarray) -> None:
        """Rows idxs of the shard -> out[0 .. len(idxs)).  One row is ~1-8 MB, so this is memory-bandwidth work; it runs as ONE
        call into libfreud_host.so, which cuts the rows into pieces and spreads them over its own threads (no Python per row, the
        GIL released for the whole batch).  dl_max_workers > 0 sets the thread count (the reference hands that key to torch's
        DataLoader as num_workers, train_sae.py:51-63); 0 = automatic.  Without the library (never on a built tree) a plain loop."""
        mm = self._dataset.mmap
        nthreads = self._gather_threads
        idx64 = np.ascontiguousarray(np.asarray(idxs, dtype=np.int64))
        if len(idx64) == 0:
            return
        row = mm.shape[1]
        per = max(1, -(-2 * nthreads // len(idx64)))            # pieces per row: every thread gets the same bytes
        if self._convert:        # fp32 rows -> bf16 bit patterns (out is a uint16 / bf16-viewed buffer)
            lib = _host_lib()
            out16 = out.view(np.uint16) if out.dtype != np.uint16 else out
            piece = ((row + per - 1) // per + 31) // 32 * 32     # whole 64-byte lines of the destination
            lib.freud_gather_batch_f32_to_bf16(mm.ctypes.data, idx64.ctypes.data, len(idx64), row, out16.ctypes.data,
                                               out16.strides[0] // 2, piece, nthreads)
            return
        try:
            lib = _host_lib()
        except RuntimeError:
            lib = None
        if lib is None or out.strides[-1] != out.itemsize or mm.strides[-1] != mm.itemsize:
            for j, i in enumerate(idxs):
                out[j] = mm[i]
            return
        row_bytes = row * mm.itemsize
        piece = ((row_bytes + per - 1) // per + 63) // 64 * 64
        lib.freud_gather_batch_copy(mm.ctypes.data, idx64.ctypes.data, len(idx64), row_bytes, out.ctypes.data, out.strides[0], piece, nthreads)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, List[str]]]:
        batches = self.epoch_batches()
        if self.skip_next:          # mid-epoch resume: same permutation (the caller restored the RNG), batches already seen dropped
            batches = batches[self.skip_next:]
            self.skip_next = 0
        T, d = self._dataset.tensor_shape[-2], self._dataset.tensor_shape[-1]
        names = self._dataset.metadata["filenames"]
        np_dtype = self._dataset.mmap.dtype
        if self.device.type != "cuda":
            for idxs in batches:
                if self._convert:
                    t = torch.empty((len(idxs), T * d), dtype=torch.bfloat16)
                    self._gather(idxs, t.view(torch.int16).numpy().view(np.uint16))
                    yield t.reshape(len(idxs), T, d), [names[i] for i in idxs]
                    continue
                buf = np.empty((len(idxs), T * d), dtype=np_dtype)
                self._gather(idxs, buf)
                yield torch.from_numpy(buf).reshape(len(idxs), T, d), [names[i] for i in idxs]
            return
        yield from self._iter_cuda(batches, T, d, names, np_dtype)

    def _iter_cuda_direct(self, batches, T, d, names):
        """Rows -> HBM by asynchronous copies straight from the registered mapping (one DMA per file), into a ring of
        `depth` device buffers; the compute stream waits for a batch's copies, the copy stream for the compute work that
        last used the buffer."""
        B, depth, dev = self.batch_size, self.depth, self.device
        hbm = [torch.empty((B, T * d), dtype=self._mm_t.dtype, device=dev) for _ in range(depth)]
        copy_stream = torch.cuda.Stream(device=dev)
        consumed = [None] * depth
        import ctypes
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemcpyAsync.restype = ctypes.c_int
        except OSError:
            hip = None
        row_bytes = T * d * self._mm_t.element_size()
        src0 = self._mm_t.data_ptr()
        nstreams = int(os.environ.get("FREUD_LOADER_STREAMS", "2"))      # copy engines fed in parallel (rows alternate)
        streams = [copy_stream] + [torch.cuda.Stream(device=dev) for _ in range(max(nstreams, 1) - 1)]
        for bi, idxs in enumerate(batches):
            slot = bi % depth
            compute = torch.cuda.current_stream(dev)
            for st_ in streams:
                if consumed[slot] is not None:
                    st_.wait_event(consumed[slot])
            if hip is not None:          # one hipMemcpyAsync per file, issued without the per-call torch overhead
                dst0 = hbm[slot].data_ptr()
                sts = [ctypes.c_void_p(st_.cuda_stream) for st_ in streams]
                for j, i in enumerate(idxs):
                    rc = hip.hipMemcpyAsync(ctypes.c_void_p(dst0 + j * row_bytes), ctypes.c_void_p(src0 + i * row_bytes),
                                            ctypes.c_size_t(row_bytes), 1, sts[j % len(sts)])
                    if rc != 0:
                        raise RuntimeError(f"hipMemcpyAsync failed with {rc}")
            else:
                for j, i in enumerate(idxs):
                    with torch.cuda.stream(streams[j % len(streams)]):
                        hbm[slot][j].copy_(self._mm_t[i], non_blocking=True)
            for st_ in streams:
                landed = torch.cuda.Event()
                landed.record(st_)
                compute.wait_event(landed)
            yield hbm[slot][: len(idxs)].view(len(idxs), T, d), [names[i] for i in idxs]
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            consumed[slot] = ev

    def _iter_cuda(self, batches, T, d, names, np_dtype):
        if self._direct:
            yield from self._iter_cuda_direct(batches, T, d, names)
            return
        B, depth, dev = self.batch_size, self.depth, self.device
        tdtype = torch.bfloat16 if self._convert else torch.from_numpy(np.empty(0, dtype=np_dtype)).dtype
        # the pinned ring and its HBM twin are allocated ONCE per loader and reused by every epoch (pinning 3 x 46 MB per epoch
        # cost several milliseconds of every epoch start; all work on them is finished when an epoch's iterator ends)
        key = (B, T * d, tdtype, depth, str(dev))
        new_ring = lambda: ([torch.empty((B, T * d), dtype=tdtype, pin_memory=True) for _ in range(depth)],
                            [torch.empty((B, T * d), dtype=tdtype, device=dev) for _ in range(depth)])
        # the shared ring belongs to ONE live iterator at a time: a second iterator on the same loader (or an epoch whose
        # predecessor's gather worker has not ended, below) gets buffers of its own
        shared = not getattr(self, "_ring_busy", False)
        if shared:
            if getattr(self, "_ring_key", None) != key:
                self._ring = new_ring()
                self._ring_key = key
            pinned, hbm = self._ring
            self._ring_busy = True
        else:
            pinned, hbm = new_ring()
        free = [threading.Semaphore(1) for _ in range(depth)]
        copied = [None] * depth     # copy-stream event of the H2D copy that last read pinned[slot]
        ready: "queue.Queue" = queue.Queue()
        stop = threading.Event()

        def worker():
            try:
                for bi, idxs in enumerate(batches):
                    slot = bi % depth
                    while not free[slot].acquire(timeout=0.1):
                        if stop.is_set():
                            return
                    if copied[slot] is not None:
                        copied[slot].synchronize()      # the slot's previous batch has left for the GPU
                    if self._convert:
                        self._gather(idxs, pinned[slot].view(torch.int16).numpy().view(np.uint16)[: len(idxs)])
                    else:
                        self._gather(idxs, pinned[slot].numpy()[: len(idxs)])
                    ready.put((slot, idxs))
                ready.put(None)
            except BaseException as e:  # surface loader errors in the training thread
                ready.put(e)

        th = threading.Thread(target=worker, name="shard-gather", daemon=True)
        copy_stream = None
        consumed = [None] * depth   # compute-stream event: batch in hbm[slot] fully enqueued
        try:
            # (inside the try: a failure to start the thread or to create the stream must still give the shared ring back -- with the
            # flag stuck every later epoch pinned a fresh ring)
            th.start()
            copy_stream = torch.cuda.Stream(device=dev)
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                slot, idxs = item
                compute = torch.cuda.current_stream(dev)
                with torch.cuda.stream(copy_stream):
                    if consumed[slot] is not None:
                        copy_stream.wait_event(consumed[slot])
                    hbm[slot][: len(idxs)].copy_(pinned[slot][: len(idxs)], non_blocking=True)
                    landed = torch.cuda.Event()
                    landed.record(copy_stream)
                compute.wait_event(landed)
                # pinned[slot] may be refilled once the DMA has read it: the GATHER worker waits for that (copied[slot]), not
                # this thread -- round 3 synchronised here, so the training thread sat ~0.8 ms per batch in a host wait instead
                # of enqueueing the step
                copied[slot] = landed
                free[slot].release()
                yield hbm[slot][: len(idxs)].view(len(idxs), T, d), [names[i] for i in idxs]
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                consumed[slot] = ev
        finally:
            stop.set()
            if th.ident is not None:
                th.join(timeout=60)     # (the worker looks at `stop` between batches; a gather stalls at worst for one batch's page faults)
            torch.cuda.current_stream(dev).synchronize()     # the ring is reused by the next epoch: nothing of this one is in flight
            if copy_stream is not None:
                copy_stream.synchronize()
            if shared:
                if th.is_alive():       # a straggler may still write into pinned[slot]: the ring is abandoned to it, the next epoch
                    self._ring_key = None   # allocates a fresh one (round 4 ignored the join result: silent batch corruption)
                    self._ring = None
                self._ring_busy = False


def write_shards(folder: str, layer_name: str, rows: np.ndarray, tensor_shape: Sequence[int],
                 filenames: Optional[Sequence[str]] = None) -> None:
    """Write a shard directory byte-compatible with what collect_activations.py:12-63 produces
    (one [1, T*d] row per file, standard NPY header, C order) -- SURVEY.md section 8f row f2."""
    os.makedirs(folder, exist_ok=True)
    rows = np.ascontiguousarray(rows)
    n_files = rows.shape[0]
    T, d = int(tensor_shape[-2]), int(tensor_shape[-1])
    if rows.reshape(n_files, -1).shape[1] != T * d:
        raise ValueError(f"All tensors must have the same shape as the first tensor. Expected {[T, d]}")
    np.save(os.path.join(folder, f"{layer_name}_tensors.npy"), rows.reshape(n_files, T * d))
    meta = {"tensor_shape": [T, d], "activation_shape": [T, d],
            "filenames": list(filenames) if filenames is not None else [f"file_{i:06d}.flac" for i in range(n_files)]}
    with open(os.path.join(folder, f"{layer_name}_metadata.json"), "w") as f:
        json.dump(meta, f)

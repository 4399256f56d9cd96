"""ctypes binding of libfreud_sae.so (C ABI: include/freud_sae.h).

This is the only door between the Python host (orchestration, torch for device memory,
streams and torch.distributed) and the HIP engine that runs the SAE train step
(reference hot path: src/scripts/train_sae.py:421-453).  There is no CPU fallback: if the
shared library is missing the import of this module's `load()` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from typing import Dict, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FREUD_SAE_LIB: development override (A/B timing of two builds of the engine on one GPU box)
LIB_PATH = os.environ.get("FREUD_SAE_LIB") or os.path.join(_HERE, "lib", "libfreud_sae.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

VARIANT = {"l1": 0, "topk": 1}
OPTIMIZER = {"radam": 0, "adam": 1}
DTYPE = {"float32": 0, "float16": 1, "bfloat16": 2}
PRECISION = {"bf16": 0, "fp8": 1, "fp8bwd": 2}      # fp8bwd: fp8 encoder / decoder AND dpre GEMMs (include/freud_sae.h)
NUM_METRICS = 8
M_LOSS_RECON, M_LOSS_L1, M_MSE, M_GRAD_NORM, M_COUNT, M_DEAD_PCT, M_MULTI_TOPK_FVU = 0, 1, 2, 3, 4, 5, 6


class SaeConfig(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("d_model", C.c_int32), ("n_dict", C.c_int32), ("k", C.c_int32),
        ("optimizer", C.c_int32), ("device_id", C.c_int32), ("max_rows", C.c_int64),
        ("recon_alpha", C.c_double), ("auxk_alpha", C.c_double), ("clip_thresh", C.c_double),
        ("weight_decay", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
        ("force_generic", C.c_int32), ("debug_flags", C.c_int32), ("force_gemm128", C.c_int32),
        ("topk_dense_backward", C.c_int32), ("multi_topk", C.c_int32), ("precision", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


class EngineError(RuntimeError):
    pass


_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP engine for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-C", CSRC_DIR], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if not os.path.exists(LIB_PATH):
        raise EngineError(f"build did not produce {LIB_PATH}")
    return LIB_PATH


GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)   # sae_grad_ready_fn


def load() -> C.CDLL:
    """dlopen the engine and declare every symbol of include/freud_sae.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"{LIB_PATH} is missing: the HIP engine has not been built "
            f"(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C {CSRC_DIR}`). "
            "There is no CPU fallback for the train step.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    fptr = C.POINTER(C.c_float)
    sig = {
        "sae_last_error": (C.c_char_p, []),
        "sae_version": (C.c_int, []),
        "sae_create": (C.c_int, [C.POINTER(SaeConfig), C.POINTER(vp)]),
        "sae_destroy": (None, [vp]),
        "sae_set_params": (C.c_int, [vp, vp, vp, vp, vp, C.c_int]),
        "sae_get_params": (C.c_int, [vp, vp, vp, vp, vp, C.c_int]),
        "sae_set_opt_state": (C.c_int, [vp, i64, C.POINTER(vp), C.POINTER(vp), C.c_int]),
        "sae_get_opt_state": (C.c_int, [vp, C.POINTER(i64), C.POINTER(vp), C.POINTER(vp), C.c_int]),
        "sae_forward_backward": (C.c_int, [vp, vp, i64, C.c_int, vp]),
        "sae_grad_buffer": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i64)]),
        "sae_optimizer_step": (C.c_int, [vp, dbl, dbl, vp]),
        "sae_batch_stats": (C.c_int, [vp, vp, i64, C.c_int, vp]),
        "sae_stats_buffer": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i64)]),
        "sae_set_dp_world": (C.c_int, [vp, C.c_int]),
        "sae_dist_unique_id": (C.c_int, [vp, i64]),
        "sae_dist_init": (C.c_int, [vp, vp, i64, C.c_int, C.c_int]),
        "sae_dist_world": (C.c_int, [vp]),
        "sae_dist_set_payload": (C.c_int, [vp, C.c_int]),
        "sae_p2p_blob_bytes": (C.c_int, []),
        "sae_p2p_export": (C.c_int, [vp, vp, i64]),
        "sae_p2p_init": (C.c_int, [vp, vp, i64, C.c_int, C.c_int]),
        "sae_p2p_leave": (C.c_int, [vp]),
        "sae_dist_set_overlap": (C.c_int, [vp, C.c_int]),
        "sae_dist_check": (C.c_int, [vp]),
        "sae_dist_poll": (C.c_int, [vp]),
        "sae_dist_audit": (C.c_int, [vp, vp]),
        "sae_grad_layout": (C.c_int, [vp, C.POINTER(i64)]),
        "sae_param_checksum": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "sae_set_grad_ready_callback": (C.c_int, [vp, GRAD_READY_FN, vp]),
        "sae_get_topk_state": (C.c_int, [vp, C.POINTER(C.c_int64), i64]),
        "sae_set_topk_state": (C.c_int, [vp, C.POINTER(C.c_int64), i64]),
        "sae_set_topk_options": (C.c_int, [vp, dbl, i64]),
        "sae_step": (C.c_int, [vp, vp, i64, C.c_int, dbl, vp]),
        "sae_eval": (C.c_int, [vp, vp, i64, C.c_int, vp]),
        "sae_eval_into": (C.c_int, [vp, vp, i64, C.c_int, vp, vp, vp]),
        "sae_set_eval_precision": (C.c_int, [vp, C.c_int]),
        "sae_latent_buffer": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i64)]),
        "sae_topk_indices": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_int)]),
        "sae_decode": (C.c_int, [vp, vp, C.c_int, i64, i64, vp, vp]),
        "sae_multi_topk_buffers": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(C.c_int)]),
        "sae_read_metrics": (C.c_int, [vp, fptr, vp]),
        "sae_latent_colmax": (C.c_int, [vp, fptr, i64, vp]),
        "sae_debug_read": (C.c_int, [vp, C.c_int, fptr, i64]),
        "sae_profile": (C.c_int, [vp, C.c_int]),
        "sae_profile_period": (C.c_int, [vp, C.c_int]),
        "sae_kernel_times": (C.c_int, [vp, fptr, C.POINTER(i32), C.c_int]),
        "sae_kernel_name": (C.c_char_p, [C.c_int]),
        "sae_dominant_kernel": (C.c_int, [vp]),
    }
    skipped = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)      # AttributeError here = header / library mismatch
        except AttributeError:
            # FREUD_SAE_ALLOW_OLD_LIB=1 (set by the A/B tools that time an OLDER build against this one, tools/ab_bench.sh): the
            # missing entry points are named once and fail when called.  A user-supplied FREUD_SAE_LIB alone does NOT switch the
            # header / library check off (a stale library would otherwise fail at an arbitrary call site mid-training).
            if os.environ.get("FREUD_SAE_ALLOW_OLD_LIB") == "1":
                skipped.append(name)
                continue
            raise EngineError(f"{LIB_PATH} does not export {name}: the library is older than include/freud_sae.h "
                              "(rebuild it, or set FREUD_SAE_ALLOW_OLD_LIB=1 for an A/B timing of an old build)") from None
        fn.restype = res
        fn.argtypes = args
    if skipped:
        print(f"freud_amd.engine: {LIB_PATH} lacks {', '.join(skipped)} (FREUD_SAE_ALLOW_OLD_LIB=1)", file=sys.stderr)
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "sae_last_error", "sae_version", "sae_create", "sae_destroy", "sae_set_params", "sae_get_params",
    "sae_set_opt_state", "sae_get_opt_state", "sae_forward_backward", "sae_grad_buffer", "sae_optimizer_step",
    "sae_set_grad_ready_callback", "sae_batch_stats", "sae_stats_buffer", "sae_set_dp_world", "sae_dist_unique_id",
    "sae_dist_init", "sae_dist_world", "sae_dist_set_payload", "sae_p2p_blob_bytes", "sae_p2p_export", "sae_p2p_init", "sae_p2p_leave",
    "sae_dist_set_overlap", "sae_dist_check", "sae_dist_poll", "sae_dist_audit", "sae_grad_layout", "sae_param_checksum", "sae_set_topk_options", "sae_get_topk_state", "sae_set_topk_state",
    "sae_latent_buffer", "sae_topk_indices", "sae_decode", "sae_multi_topk_buffers",
    "sae_step", "sae_eval", "sae_eval_into", "sae_set_eval_precision", "sae_read_metrics", "sae_latent_colmax", "sae_debug_read", "sae_profile", "sae_profile_period", "sae_kernel_times",
    "sae_kernel_name", "sae_dominant_kernel",
]


def _check(rc: int) -> None:
    if rc != 0:
        raise EngineError(f"libfreud_sae error {rc}: {load().sae_last_error().decode()}")


def _np_f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _shaped(name: str, a, shape) -> np.ndarray:
    """fp32 contiguous copy of `a` with exactly `shape` elements (the C side copies prod(shape) floats from the pointer:
    a short buffer would be a host heap over-read)."""
    arr = _np_f32(a)
    if arr.size != int(np.prod(shape)):
        raise EngineError(f"{name}: got {arr.shape} ({arr.size} elements), the engine expects {tuple(shape)}")
    return arr.reshape(shape)


class SaeEngine:
    """One engine context = one SAE on one GPU.  Thin, typed wrapper over the C ABI."""

    def __init__(self, variant: str, d_model: int, n_dict: int, max_rows: int, *, optimizer: str = "radam",
                 recon_alpha: float = 1.0, k: int = 0, auxk_alpha: float = 0.0, clip_thresh: float = 1.0,
                 weight_decay: float = 0.0, betas=(0.9, 0.999), eps: Optional[float] = None, device_id: int = 0,
                 force_generic: bool = False, debug_flags: int = 0, force_gemm128: bool = False,
                 topk_dense_backward: bool = False, multi_topk: bool = False, precision: str = "bf16"):
        if variant not in VARIANT:
            raise AssertionError(f"Invalid autoencoder variant: {variant}, must be 'l1' or 'topk'")
        if optimizer not in OPTIMIZER:
            raise ValueError(f"Invalid optimizer: {optimizer}, must be 'radam' or 'adam'")
        self._lib = load()
        if eps is None:  # train_sae.py:374-379: RAdam(eps=1e-5), Adam default 1e-8
            eps = 1e-5 if optimizer == "radam" else 1e-8
        cfg = SaeConfig()
        cfg.variant, cfg.d_model, cfg.n_dict, cfg.k = VARIANT[variant], d_model, n_dict, k
        cfg.optimizer, cfg.device_id, cfg.max_rows = OPTIMIZER[optimizer], device_id, max_rows
        cfg.recon_alpha, cfg.auxk_alpha, cfg.clip_thresh = recon_alpha, auxk_alpha, clip_thresh
        cfg.weight_decay, cfg.beta1, cfg.beta2, cfg.eps = weight_decay, betas[0], betas[1], eps
        cfg.debug_flags = debug_flags                 # timing experiments only (results become wrong)
        cfg.force_generic = 1 if force_generic else 0   # 1 = generic three-GEMM backward even where a fused kernel exists
        cfg.force_gemm128 = 1 if force_gemm128 else 0   # 1 = 128x128 GEMM tiles even where the 256x256 kernel applies
        cfg.topk_dense_backward = int(topk_dense_backward)   # 0 = CSC sparse backward; 1 = dense GEMM + mask; 2 = sparse dpre + dense dW GEMMs
        cfg.multi_topk = 1 if multi_topk else 0       # TopKAutoEncoderConfig.multi_topk
        if precision not in PRECISION:
            raise ValueError(f"Invalid precision: {precision}, must be one of {sorted(PRECISION)}")
        cfg.precision = PRECISION[precision]
        self.variant, self.d, self.n, self.max_rows, self.device_id = variant, d_model, n_dict, max_rows, device_id
        self.k, self.multi_topk, self.precision = k, bool(multi_topk), precision
        self._ctx = C.c_void_p()
        _check(self._lib.sae_create(C.byref(cfg), C.byref(self._ctx)))

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.sae_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters ---------------------------------------------------------------------------
    def param_shapes(self) -> Dict[str, tuple]:
        """Reference state_dict keys and shapes, in the engine's parameter order."""
        if self.variant == "l1":   # l1autoencoder.py:56-60
            return {"decoder.weight": (self.d, self.n), "encoder_bias": (self.n,)}
        return {"encoder.weight": (self.n, self.d), "encoder.bias": (self.n,),
                "W_dec": (self.n, self.d), "b_dec": (self.d,)}   # topkautoencoder.py:62-70

    def set_params(self, params: Dict[str, np.ndarray]) -> None:
        arrs = [_shaped(k, params[k], shape) for k, shape in self.param_shapes().items()]
        ptrs = [a.ctypes.data_as(C.c_void_p) for a in arrs] + [None] * (4 - len(arrs))
        _check(self._lib.sae_set_params(self._ctx, *ptrs, 0))

    def get_params(self) -> Dict[str, np.ndarray]:
        out = {k: np.empty(shape, dtype=np.float32) for k, shape in self.param_shapes().items()}
        ptrs = [a.ctypes.data_as(C.c_void_p) for a in out.values()] + [None] * (4 - len(out))
        _check(self._lib.sae_get_params(self._ctx, *ptrs, 0))
        return out

    def set_opt_state(self, step: int, exp_avg: Dict[str, np.ndarray], exp_avg_sq: Dict[str, np.ndarray]) -> None:
        shapes = self.param_shapes()
        a = [_shaped(f"exp_avg[{k}]", exp_avg[k], s) for k, s in shapes.items()]
        b = [_shaped(f"exp_avg_sq[{k}]", exp_avg_sq[k], s) for k, s in shapes.items()]
        pa = (C.c_void_p * 4)(*([x.ctypes.data for x in a] + [None] * (4 - len(a))))
        pb = (C.c_void_p * 4)(*([x.ctypes.data for x in b] + [None] * (4 - len(b))))
        _check(self._lib.sae_set_opt_state(self._ctx, int(step), pa, pb, 0))

    def get_opt_state(self):
        shapes = self.param_shapes()
        a = {k: np.empty(s, dtype=np.float32) for k, s in shapes.items()}
        b = {k: np.empty(s, dtype=np.float32) for k, s in shapes.items()}
        pa = (C.c_void_p * 4)(*([x.ctypes.data for x in a.values()] + [None] * (4 - len(a))))
        pb = (C.c_void_p * 4)(*([x.ctypes.data for x in b.values()] + [None] * (4 - len(b))))
        step = C.c_int64(0)
        _check(self._lib.sae_get_opt_state(self._ctx, C.byref(step), pa, pb, 0))
        return int(step.value), a, b

    def set_topk_options(self, dead_feature_threshold: float, rows_per_file: int) -> None:
        """TopK only: autoencoder_config["dead_feature_threshold"] (train_sae.py:438) and T of the [B][T][d] batch."""
        self._dead_threshold, self._rows_per_file = float(dead_feature_threshold), int(rows_per_file)
        _check(self._lib.sae_set_topk_options(self._ctx, float(dead_feature_threshold), int(rows_per_file)))

    def get_topk_state(self) -> np.ndarray:
        """TopK only: num_frames_since_fired[n_dict] (int64) - train_sae.py:412-415 keeps it in the live process only."""
        out = np.zeros(self.n, dtype=np.int64)
        _check(self._lib.sae_get_topk_state(self._ctx, out.ctypes.data_as(C.POINTER(C.c_int64)), self.n))
        return out

    def set_topk_state(self, num_frames_since_fired) -> None:
        a = np.ascontiguousarray(np.asarray(num_frames_since_fired, dtype=np.int64).reshape(-1))
        _check(self._lib.sae_set_topk_state(self._ctx, a.ctypes.data_as(C.POINTER(C.c_int64)), a.size))

    def set_dead_feature_threshold(self, v: float) -> None:
        self.set_topk_options(v, getattr(self, "_rows_per_file", 0))

    # -- the hot path -------------------------------------------------------------------------
    @staticmethod
    def _x_args(x):
        """x: a torch CUDA tensor [M, d] (or [B, T, d]) fp32/fp16/bf16, contiguous."""
        import torch
        if not x.is_cuda:
            raise EngineError("activations must live in HBM (torch CUDA tensor); there is no CPU path")
        if not x.is_contiguous():
            x = x.contiguous()
        name = {torch.float32: "float32", torch.float16: "float16", torch.bfloat16: "bfloat16"}.get(x.dtype)
        if name is None:
            raise EngineError(f"unsupported activation dtype {x.dtype}")
        rows = x.numel() // x.shape[-1]
        return x, x.data_ptr(), rows, DTYPE[name]

    def _note_shape(self, x) -> None:
        """TopK: x.mean(0) is over the files of a [B, T, d] batch -> tell the engine T when it changes."""
        if self.variant == "topk" and x.dim() == 3 and getattr(self, "_rows_per_file", None) != x.shape[1]:
            self.set_topk_options(getattr(self, "_dead_threshold", 1e300), x.shape[1])

    @staticmethod
    def _stream(stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        return C.c_void_p(s.cuda_stream)

    def forward_backward(self, x, stream=None) -> None:
        self._note_shape(x)
        x, ptr, rows, dt = self._x_args(x)
        _check(self._lib.sae_forward_backward(self._ctx, C.c_void_p(ptr), rows, dt, self._stream(stream)))
        err, self._cb_error = getattr(self, "_cb_error", None), None
        if err is not None:
            raise err

    # -- data parallel (include/freud_sae.h, "data-parallel exactness") ------------------------------------
    def batch_stats(self, x, stream=None) -> None:
        """This rank's statistics of batch x (what the losses normalise by) into the statistics buffer."""
        self._note_shape(x)
        x, ptr, rows, dt = self._x_args(x)
        _check(self._lib.sae_batch_stats(self._ctx, C.c_void_p(ptr), rows, dt, self._stream(stream)))

    def stats_tensor(self):
        """The statistics of the last batch_stats() as a float64 torch CUDA tensor aliasing the engine's buffer:
        all-reduce (sum) it over the ranks before forward_backward()."""
        import torch
        p, n = C.c_void_p(), C.c_int64()
        _check(self._lib.sae_stats_buffer(self._ctx, C.byref(p), C.byref(n)))

        class _Alias:
            __cuda_array_interface__ = {"shape": (int(n.value),), "typestr": "<f8", "data": (int(p.value), False), "version": 2}

        return torch.as_tensor(_Alias(), device=f"cuda:{self.device_id}")

    def set_dp_world(self, world: int) -> None:
        _check(self._lib.sae_set_dp_world(self._ctx, int(world)))

    @staticmethod
    def dist_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(load().sae_dist_unique_id(buf, 128))
        return buf.raw

    def dist_init(self, unique_id: bytes, rank: int, world: int) -> None:
        """Give the context its own RCCL communicator: forward_backward() / step() then run the data-parallel protocol
        (statistics and gradient all-reduces on a communication stream) inside the engine."""
        buf = C.create_string_buffer(bytes(unique_id), len(unique_id))
        _check(self._lib.sae_dist_init(self._ctx, buf, len(unique_id), int(rank), int(world)))

    def dist_world(self) -> int:
        return int(self._lib.sae_dist_world(self._ctx))

    def dist_set_payload(self, dtype: str) -> None:
        """"float32" (default, exact) or "bfloat16" (fused d=384 path: half the all-reduce bytes)."""
        _check(self._lib.sae_dist_set_payload(self._ctx, DTYPE[dtype]))

    def p2p_export(self) -> bytes:
        """This rank's blob for the peer exchange (hipIpc handles of the gradient / statistics / flag buffers)."""
        nbytes = int(self._lib.sae_p2p_blob_bytes())
        buf = C.create_string_buffer(nbytes)
        _check(self._lib.sae_p2p_export(self._ctx, buf, nbytes))
        return buf.raw

    def p2p_init(self, blobs: Sequence[bytes], rank: int, world: int) -> None:
        """Map the peers from the blobs of ALL ranks (rank order) and run the self-test exchange.  Collective.  Afterwards
        forward_backward() / step() run the data-parallel protocol with the engine's own exchange kernels."""
        assert len(blobs) == world and all(len(b) == len(blobs[0]) for b in blobs)
        joined = b"".join(blobs)
        buf = C.create_string_buffer(joined, len(joined))
        _check(self._lib.sae_p2p_init(self._ctx, buf, len(blobs[0]), int(rank), int(world)))

    def p2p_leave(self) -> None:
        """Undo p2p_init (a peer's self-test failed although this rank's passed: the ranks fall back together)."""
        _check(self._lib.sae_p2p_leave(self._ctx))

    def dist_set_overlap(self, nranges: int) -> None:
        """Fused d=384 backward in `nranges` column-tile ranges, each exchanged under the next one's backward."""
        _check(self._lib.sae_dist_set_overlap(self._ctx, int(nranges)))

    def dist_check(self) -> None:
        """Synchronise and raise if an in-engine exchange failed (a peer never arrived)."""
        _check(self._lib.sae_dist_check(self._ctx))

    def dist_poll(self) -> None:
        """The same report without synchronising (host-mapped failure word): free after every step."""
        _check(self._lib.sae_dist_poll(self._ctx))

    def dist_audit(self, snapshot) -> None:
        """snapshot: a float32 CUDA tensor shaped like grad_tensor() (kept alive by the caller) -- every gradient exchange
        first copies the segments it is about to sum into it; None switches the snapshots off (freud_amd/dp.py: audit)."""
        if snapshot is None:
            _check(self._lib.sae_dist_audit(self._ctx, None))
            return
        g = self.grad_tensor()
        assert snapshot.is_cuda and snapshot.dtype == g.dtype and snapshot.numel() == g.numel() and snapshot.is_contiguous()
        _check(self._lib.sae_dist_audit(self._ctx, C.c_void_p(snapshot.data_ptr())))

    def grad_layout(self) -> tuple:
        """(parameter-gradient floats, loss scalars, did_fire flags) of grad_tensor(), in that order."""
        out = (C.c_int64 * 3)()
        _check(self._lib.sae_grad_layout(self._ctx, out))
        return tuple(int(v) for v in out)

    def param_checksum(self) -> tuple:
        """(parameters, first moments, second moments, step): order-independent 64-bit checksums of the fp32 bits.  Replicas
        of a data-parallel run must agree on all four (train() compares them over the host channel).  Synchronises."""
        out = (C.c_uint64 * 4)()
        _check(self._lib.sae_param_checksum(self._ctx, out))
        return tuple(int(v) for v in out)

    def optimizer_step(self, lr: float, grad_scale: float = 1.0, stream=None) -> None:
        _check(self._lib.sae_optimizer_step(self._ctx, float(lr), float(grad_scale), self._stream(stream)))

    def step(self, x, lr: float, stream=None) -> None:
        self._note_shape(x)
        x, ptr, rows, dt = self._x_args(x)
        _check(self._lib.sae_step(self._ctx, C.c_void_p(ptr), rows, dt, float(lr), self._stream(stream)))

    def eval(self, x, stream=None) -> None:
        self._note_shape(x)
        x, ptr, rows, dt = self._x_args(x)
        _check(self._lib.sae_eval(self._ctx, C.c_void_p(ptr), rows, dt, self._stream(stream)))

    def set_eval_precision(self, precision: str) -> None:
        """"bf16" (default): eval() / eval_into() run the training kernels' arithmetic (CPU autocast's).  "fp32": fp32 end to end --
        what the reference's validate() computes on device='cpu' (train_sae.py:162-166) and selects bestval.pth by."""
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"Invalid evaluation precision: {precision}, must be 'bf16' or 'fp32'")
        _check(self._lib.sae_set_eval_precision(self._ctx, 3 if precision == "fp32" else PRECISION["bf16"]))
        self.eval_precision = precision

    def eval_into(self, x, metrics_row, colmax_row=None, stream=None) -> None:
        """eval() of one file with its 8 loss scalars (and per-feature latent maxima) left in the given fp32 CUDA rows:
        no host synchronisation (validate() reads all rows once at the end)."""
        self._note_shape(x)
        x, ptr, rows, dt = self._x_args(x)
        assert metrics_row.is_cuda and metrics_row.is_contiguous() and metrics_row.numel() >= NUM_METRICS
        cm = None
        if colmax_row is not None:
            assert colmax_row.is_cuda and colmax_row.is_contiguous() and colmax_row.numel() >= self.n
            cm = C.c_void_p(colmax_row.data_ptr())
        _check(self._lib.sae_eval_into(self._ctx, C.c_void_p(ptr), rows, dt, C.c_void_p(metrics_row.data_ptr()), cm,
                                       self._stream(stream)))

    # -- inference ---------------------------------------------------------
    def latent_buffer(self):
        """(device pointer, row stride in elements) of the bf16 latent of the last forward."""
        ptr, ld = C.c_void_p(), C.c_int64()
        _check(self._lib.sae_latent_buffer(self._ctx, C.byref(ptr), C.byref(ld)))
        return ptr.value, ld.value

    def topk_indices_tensor(self, rows: int, device):
        """int32 [rows][k] torch view of the top-k indices of the last forward (TopK contexts)."""
        import torch
        ptr, k = C.c_void_p(), C.c_int()
        _check(self._lib.sae_topk_indices(self._ctx, C.byref(ptr), C.byref(k)))

        class _Alias:
            __cuda_array_interface__ = {"shape": (rows, k.value), "typestr": "<i4", "data": (ptr.value, False), "version": 2}

        return torch.as_tensor(_Alias(), device=device)

    def multi_topk_buffers(self, rows: int, device):
        """TopK with multi_topk: (bf16 [rows][n_dict] dense 4k activations, int32 [rows][4k] indices) of the last forward."""
        import torch
        dptr, ld, iptr, k4 = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int()
        _check(self._lib.sae_multi_topk_buffers(self._ctx, C.byref(dptr), C.byref(ld), C.byref(iptr), C.byref(k4)))

        class _D:
            __cuda_array_interface__ = {"shape": (rows, ld.value), "typestr": "<i2", "data": (dptr.value, False), "version": 2}

        class _I:
            __cuda_array_interface__ = {"shape": (rows, k4.value), "typestr": "<i4", "data": (iptr.value, False), "version": 2}

        dense = torch.as_tensor(_D(), device=device).view(torch.bfloat16)[:, : self.n]
        return dense, torch.as_tensor(_I(), device=device)

    def decode(self, latent, out, stream=None) -> None:
        """out[rows][d_model] (fp32 CUDA tensor) = decode(latent[rows][>= n_dict]) (fp32 or bf16 CUDA tensor, row-major)."""
        import torch
        assert latent.is_cuda and out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
        assert latent.dim() == 2 and latent.stride(1) == 1
        dt = {torch.float32: DTYPE["float32"], torch.bfloat16: DTYPE["bfloat16"]}[latent.dtype]
        _check(self._lib.sae_decode(self._ctx, C.c_void_p(latent.data_ptr()), dt, int(latent.stride(0)), int(latent.shape[0]),
                                    C.c_void_p(out.data_ptr()), self._stream(stream)))

    def metrics(self, stream=None) -> np.ndarray:
        out = np.zeros(NUM_METRICS, dtype=np.float32)
        _check(self._lib.sae_read_metrics(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), self._stream(stream)))
        return out

    def latent_colmax(self, stream=None) -> np.ndarray:
        """max over rows of |latent| per dictionary feature for the last forward (validate())."""
        out = np.empty(self.n, dtype=np.float32)
        _check(self._lib.sae_latent_colmax(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), self.n, self._stream(stream)))
        return out

    def grad_buffer(self):
        """(device pointer, number of floats) of the flat gradient + metrics buffer (for all-reduce)."""
        p, n = C.c_void_p(), C.c_int64()
        _check(self._lib.sae_grad_buffer(self._ctx, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def set_grad_ready_callback(self, fn) -> None:
        """fn(offset, count) is called from inside forward_backward() each time the range [offset, offset + count) of the
        gradient buffer is final in stream order (data-parallel overlap: start that range's all-reduce).  None removes
        it.  Exceptions raised by fn are re-raised by forward_backward()."""
        self._cb_error = None
        if fn is None:
            self._cb = None
            _check(self._lib.sae_set_grad_ready_callback(self._ctx, C.cast(None, GRAD_READY_FN), None))
            return

        def tramp(_user, offset, count, _stream):
            try:
                fn(int(offset), int(count))
            except BaseException as e:      # never unwind through the C frames
                self._cb_error = e

        self._cb = GRAD_READY_FN(tramp)     # keep the thunk alive
        _check(self._lib.sae_set_grad_ready_callback(self._ctx, self._cb, None))

    def grad_tensor(self):
        """The gradient buffer as a torch CUDA tensor aliasing the engine's HBM (no copy)."""
        import torch
        ptr, n = self.grad_buffer()

        class _Alias:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}

        return torch.as_tensor(_Alias(), device=f"cuda:{self.device_id}")

    # -- inspection -----------------------------------------------------------------------------
    def debug_read(self, which: int, count: int) -> np.ndarray:
        out = np.empty(count, dtype=np.float32)
        _check(self._lib.sae_debug_read(self._ctx, which, out.ctypes.data_as(C.POINTER(C.c_float)), count))
        return out

    def profile(self, level: int, period: Optional[int] = None) -> None:
        """level 1 brackets the dominant kernel with HIP events on every `period`-th step (default 8)."""
        if period is not None:
            _check(self._lib.sae_profile_period(self._ctx, int(period)))
        _check(self._lib.sae_profile(self._ctx, level))

    def kernel_times(self) -> Dict[str, tuple]:
        n = 32
        ms = (C.c_float * n)()
        cnt = (C.c_int32 * n)()
        _check(self._lib.sae_kernel_times(self._ctx, ms, cnt, n))
        out = {}
        for i in range(n):
            name = self._lib.sae_kernel_name(i)
            if name is None:
                break
            out[name.decode()] = (float(ms[i]), int(cnt[i]))
        return out

    def dominant_kernel(self) -> str:
        return self._lib.sae_kernel_name(self._lib.sae_dominant_kernel(self._ctx)).decode()

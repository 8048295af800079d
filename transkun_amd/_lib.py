"""Bindings to the HIP library.

The compute calls of the Python mirror go through torch ops: `ops()` loads libsemicrf_torch.so, a LibTorch stable-ABI
shim (csrc/torch_ops.cpp) that registers the C ABI of include/semicrf_hip.h as `torch.ops.semicrf.*` (dispatch key CUDA =
HIP tensors): torch's dispatcher picks the tensors' device and current stream.  `load()` is the raw ctypes binding
to libsemicrf_hip.so itself -- workspace sizes, the implementation switch, the debug status word, and the ABI tests.

torch is imported first on purpose: the library needs libamdhip64.so.7 and must bind to the
HIP runtime torch has already loaded (one runtime per process), not to a second copy.
There is NO fallback: a missing library raises, and a GPU tensor is only ever computed by the HIP kernels.  CPU tensors are
dispatched (by torch's dispatcher, on the tensors' device) to the product's own host kernels in the same shim
(csrc/cpu_ops.cpp) -- the reference class runs wherever its tensors live (crfMinimalExample, BASELINE config #1).
"""
from __future__ import annotations

import collections
import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEMICRF_LIB") or os.path.join(_HERE, "libsemicrf_hip.so")   # SEMICRF_LIB: development variants

OP_LOGZ_FWD, OP_LOGZ_BWD, OP_VITERBI, OP_EVAL_PATH, OP_INTERVAL_SCORE = range(5)
LEN_MODES = {"linear": 0, "sqrt": 1, "none": 2}

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_sz = ctypes.c_size_t

_SIGS = {
    "semicrf_abi_version": (ctypes.c_int, []),
    "semicrf_last_error": (ctypes.c_char_p, []),
    "semicrf_workspace_bytes": (_sz, [_i, _i, _i]),
    "semicrf_workspace_register": (_i, [_vp, _sz]),
    "semicrf_workspace_unregister": (_i, [_vp]),
    "semicrf_set_impl": (None, [_i]),
    "semicrf_get_impl": (ctypes.c_int, []),
    "semicrf_debug_device_status": (ctypes.c_int, []),
    "semicrf_async_error": (ctypes.c_int, []),
    "semicrf_logz_bwd_f": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "semicrf_logprob_bwd_f": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "semicrf_debug_wg_ticket": (_i, [_i, _i, _i]),
    "semicrf_debug_score_variant": (None, [_i]),
    "semicrf_logz_fwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "semicrf_logz_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "semicrf_viterbi": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i64, _vp, _vp, _sz, _vp]),
    "semicrf_eval_path": (_i, [_vp, _vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "semicrf_eval_path_bwd": (_i, [_vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "semicrf_logprob_fwd": (_i, [_vp, _vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "semicrf_logprob_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "interval_score_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, ctypes.c_float, _i, _i, _vp, _vp, _vp]),
    "interval_score_fwd_p": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, ctypes.c_float, _i, _i, _i, _i, _vp, _vp, _vp]),
    "interval_score_bwd_ws_p": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _i, _i, _vp, _vp, _vp, _i64, _i64, _i64, _vp, ctypes.c_size_t, _vp]),
    "interval_score_bwd_fused_ws_p": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _i, _i, _vp, _vp, _vp, _i64, _i64, _i64, _vp, ctypes.c_size_t, _vp]),
    "interval_score_path_bwd_p": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _i, _i, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "interval_score_fwd_pc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, ctypes.c_float, _i, _i, _i, _i, _vp, _vp, _vp]),
    "interval_score_bwd_ws_pc": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, ctypes.c_size_t, _vp]),
    "interval_score_bwd_fused_ws_pc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, ctypes.c_size_t, _vp]),
    "interval_score_path_bwd_pc": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "semicrf_beta": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "scorer_proj_nn": (_i, [_vp, _i64, _i64, _i, _vp, _i64, _i, _vp, _i64, _vp, _vp, _vp, _i, _i, _vp]),
    "scorer_proj_nn3": (_i, [_vp, _i64, _i64, _i, _vp, _i64, _i, _vp, _i64, _vp, _vp, _vp, _i, _i, _vp, ctypes.c_size_t, _vp]),
    "scorer_proj_nn3_workspace_bytes": (ctypes.c_size_t, [_i, _i]),
    "scorer_proj_tn_workspace_bytes": (_sz, [_i64, _i, _i]),
    "scorer_merge_weights_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "scorer_stage_linear": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "scorer_merge_weights_bwd_workspace_bytes": (_sz, [_i]),
    "scorer_merge_weights_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "scorer_proj_tn": (_i, [_vp, _i64, _i64, _i, _i, _i, _vp, _i64, _i, _vp, _i64, _vp, _vp, _sz, _vp]),
    "interval_score_bwd_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "interval_score_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "interval_score_bwd_fused_ws": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _vp, _vp, _vp, _i64, _i64, _i64, _vp, ctypes.c_size_t, _vp]),
    "interval_score_path_bwd": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "interval_features_gather": (_i, [_vp, _i, _i, _i, _i64, _vp, _i64, _vp, _i, _vp, _vp, _vp, _vp]),
    "interval_features_gather_bwd": (_i, [_vp, _vp, _i, _i, _i, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "interval_score_bwd_ws": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, ctypes.c_float, _i, _vp, _vp, _vp, _i64, _i64, _i64, _vp, ctypes.c_size_t, _vp]),
    "interval_score_bwd_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "segment_onset_filter": (_i, [_vp, _vp, _i, _i, _vp, _i64, _vp, _vp, _vp]),
    "segment_events": (_i, [_vp, _i64, _vp, _i, _i, _vp, _vp, _i, ctypes.c_double, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
}

EXPORTED = tuple(_SIGS)
_lib = None


class SemiCRFLibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SemiCRFLibraryError(
                f"{LIB_PATH} not found: build it with `python -m transkun_amd._build` "
                "(hipcc --offload-arch=gfx950).  transkun_amd never falls back to another implementation.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


_dbg = None
DEBUG_LIB_PATH = os.path.join(_HERE, "libsemicrf_hip_debug.so")


def load_debug():
    """The DEBUG build of the HIP library (same C ABI; transkun_amd/_build.py: build_debug): reference kernels that the release
    library no longer carries, for the parity tests and tools/ -- through ctypes only, never behind the torch ops."""
    global _dbg
    if _dbg is None:
        if not os.path.exists(DEBUG_LIB_PATH):
            raise SemiCRFLibraryError(f"{DEBUG_LIB_PATH} not found: build it with `python -m transkun_amd._build`")
        lib = ctypes.CDLL(DEBUG_LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _dbg = lib
    return _dbg


_ops = None
TORCH_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libsemicrf_torch.so")     # next to the HIP library it links


def ops():
    """torch.ops.semicrf (the stable-ABI shim, loaded once).  Raises if it has not been built."""
    global _ops
    if _ops is None:
        load()                                  # libsemicrf_hip.so first: the shim links against it
        if not os.path.exists(TORCH_LIB_PATH):
            raise SemiCRFLibraryError(f"{TORCH_LIB_PATH} not found: build it with `python -m transkun_amd._build`")
        torch.ops.load_library(TORCH_LIB_PATH)
        _ops = torch.ops.semicrf
    return _ops


_marshal = None


def marshal():
    """The CPython helper module that packs / unpacks interval lists (csrc/pymarshal.c, built by _build.build())."""
    global _marshal
    if _marshal is None:
        try:
            from . import _semicrf_marshal as m
        except ImportError as ex:
            raise SemiCRFLibraryError("transkun_amd/_semicrf_marshal.so not found: build it with "
                                      "`python -m transkun_amd._build`") from ex
        _marshal = m
    return _marshal


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().semicrf_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(t: torch.Tensor, name: str) -> None:
    """For the entry points that exist on the GPU only (interval scorer kernels, attribute gather, segment loop)."""
    if not t.is_cuda:
        raise RuntimeError(f"transkun_amd: `{name}` is on {t.device}; this entry point only runs on an AMD GPU through its HIP kernels")


def require_device(t: torch.Tensor, name: str) -> None:
    """The CRF entry points take GPU tensors (HIP kernels) or CPU tensors (the shim's host kernels); nothing else."""
    if not (t.is_cuda or t.device.type == "cpu"):
        raise RuntimeError(f"transkun_amd: `{name}` is on {t.device}; supported: an AMD GPU (HIP kernels) or the CPU (host kernels)")


_WS_BYTES = {}


def workspace(op: int, T: int, B: int, device) -> torch.Tensor:
    if torch.device(device).type == "cpu":
        return torch.empty(0, dtype=torch.uint8)          # the host kernels allocate their own scratch
    key = (op, T, B)
    n = _WS_BYTES.get(key)
    if n is None:
        n = _WS_BYTES[key] = max(int(load().semicrf_workspace_bytes(op, T, B)), 256)
    return torch.empty(n, dtype=torch.uint8, device=device)


_LEASES: Optional[collections.OrderedDict] = None
_LEASE_MAX = 16


def leased_workspace(op: int, T: int, B: int, device, kind: str = "") -> torch.Tensor:
    """A sweep workspace that is registered with the library (semicrf_workspace_register): one per (device, stream, op,
    T, B), kept alive here, filled once -- every launch leaves it clean for the next one on the same stream.
    SEMICRF_NO_LEASE=1 falls back to a fresh buffer per call (filled by every launch)."""
    global _LEASES
    device = torch.device(device)
    if device.type == "cpu" or os.environ.get("SEMICRF_NO_LEASE") or os.environ.get("SEMICRF_DEBUG_KEEP_WS"):
        return workspace(op, T, B, device)
    if _LEASES is None:
        _LEASES = collections.OrderedDict()
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream(device).cuda_stream, op, T, B, kind)
    ws = _LEASES.get(key)
    if ws is not None:
        _LEASES.move_to_end(key)
        return ws
    ws = workspace(op, T, B, device)
    with torch.cuda.device(device):
        rc = load().semicrf_workspace_register(ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()))
    if rc != 0:
        check(rc, "semicrf_workspace_register")
    _LEASES[key] = ws
    while len(_LEASES) > _LEASE_MAX:
        _, old = _LEASES.popitem(last=False)
        load().semicrf_workspace_unregister(ctypes.c_void_p(old.data_ptr()))
    return ws


def set_impl(impl: int) -> None:
    load().semicrf_set_impl(int(impl))


def device_status() -> int:
    """Synchronising debug hook: sticky device status word (0 = OK), cleared on read."""
    return int(load().semicrf_debug_device_status())


def async_error() -> int:
    """The asynchronous error word (semicrf_async_error): nonzero when a sweep enqueued earlier gave up on a bounded wait; reading
    clears it; never synchronises."""
    return int(load().semicrf_async_error())


def get_impl() -> int:
    return int(load().semicrf_get_impl())

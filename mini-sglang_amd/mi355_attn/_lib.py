"""ctypes binding of libmi355attn.so (declarations: include/mi355attn.h).

There is no CPU fallback: a missing library, a missing symbol or a non-zero return code raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# M355_LIB_PATH: another build of the library - tools/ and the cross-check tests point it at lib/libmi355attn_dev.so (the dev
# build with the A/B reference kernels, M355_* switches and timing probes); "dev" is short for that path
_DEFAULT = os.path.join(os.path.dirname(_HERE), "lib", "libmi355attn.so")
DEV_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libmi355attn_dev.so")
LIB_PATH = {"": _DEFAULT, "dev": DEV_LIB_PATH}.get(os.environ.get("M355_LIB_PATH", ""), os.environ.get("M355_LIB_PATH"))

DTYPE_BF16, DTYPE_F16, DTYPE_F32 = 0, 1, 2

NO_SLOT = -(1 << 63)                 # M355_NO_SLOT (include/mi355attn.h): out_cache_loc of a token without a slot
PLAN_ERROR_WORD = 6                  # M355_PLAN_ERROR_WORD

MAX_RANKS = 8                        # M355_MAX_RANKS (include/mi355tp.h)


class OprojRank(ctypes.Structure):   # M355OprojRank (include/mi355tp.h)
    _fields_ = [("x", c_void_p), ("x_row_stride", c_int64), ("w", c_void_p), ("w_row_stride", c_int64),
                ("out", c_void_p), ("out_row_stride", c_int64), ("recv", c_void_p * MAX_RANKS),
                ("flags", c_void_p * MAX_RANKS), ("rank", c_int32), ("reserved", c_int32)]


# name -> (restype, argtypes); must list every symbol include/mi355attn.h and include/mi355tp.h declare
SIGNATURES = {
    "m355_version": (c_int32, []),
    "m355_strerror": (c_char_p, [c_int32]),
    "m355_kv_store": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32,
                                c_int32, c_int32, c_int64, c_int32, c_void_p]),
    "m355_out_cache_loc": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "m355_prefill_prepare": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "m355_decode_prepare": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                      c_void_p, c_void_p, c_void_p]),
    "m355_page_ring_init": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "m355_page_ring_pop": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "m355_page_ring_push": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    "m355_decode_prepare_ring": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                           c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "m355_decode_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int64]),
    "m355_decode_attn": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int32,
                                   c_void_p, c_int64, c_int32, c_void_p]),
    "m355_decode_plan_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int64, c_int64]),
    "m355_decode_plan": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "m355_decode_planned_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int64, c_int64]),
    "m355_decode_attn_planned": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32,
                                           c_int32, c_int32, c_int32, c_float, c_int64, c_void_p, c_int64, c_int32,
                                           c_void_p]),
    "m355_decode_store_attn_planned": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                                 c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_float,
                                                 c_int64, c_void_p, c_int64, c_int32, c_void_p]),
    "m355_decode_store_attn_planned_static": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                                        c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                                        c_int32, c_float, c_int64, c_int32, c_void_p]),
    "m355_prefill_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "m355_prefill_attn": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_int64, c_int32,
                                    c_void_p]),
    "m355_comm_alloc": (c_int32, [c_int64, POINTER(c_void_p)]),
    "m355_comm_free": (c_int32, [c_void_p]),
    "m355_comm_zero": (c_int32, [c_void_p, c_int64]),
    "m355_comm_export": (c_int32, [c_void_p, c_void_p]),
    "m355_comm_open": (c_int32, [c_void_p, POINTER(c_void_p)]),
    "m355_comm_close": (c_int32, [c_void_p]),
    "m355_oproj_recv_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "m355_oproj_flag_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "m355_oproj_allreduce": (c_int32, [POINTER(OprojRank), c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_uint32,
                                       c_void_p, c_void_p, c_int32, c_void_p]),
    "m355_rope_kv_store": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                     c_int64, c_int32, c_void_p]),
}

_lib = None


class MI355AttnError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the library once; raise if it or any declared symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MI355AttnError(
            f"{LIB_PATH} not found: build it with `make -C mini-sglang_amd/csrc` (or __graft_entry__.build()); "
            "this backend has no fallback path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MI355AttnError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().m355_strerror(code).decode()
        raise MI355AttnError(f"{what} failed: {msg} (code {code})")


def dtype_code(torch_dtype) -> int:
    import torch
    try:
        return {torch.bfloat16: DTYPE_BF16, torch.float16: DTYPE_F16, torch.float32: DTYPE_F32}[torch_dtype]
    except KeyError:
        raise MI355AttnError(f"unsupported dtype {torch_dtype}") from None


def require_cuda(t, name: str):
    if not t.is_cuda:
        raise MI355AttnError(f"{name} must live on the GPU (got {t.device}); libmi355attn has no CPU path")
    return t


_raw_stream = None


def current_stream() -> int:
    """hipStream_t of torch's current stream on the current device (the stream every entry point launches on).
    torch.cuda.current_stream() costs ~8 us of Python per call; the raw getter is a plain C call."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream

"""ctypes binding of liblvattn.so, the C-ABI library declared in include/lv_attention.h and
include/lv_comm.h.

The library is the only compute path: if it is missing, or a call fails, a RuntimeError is raised.
There is no CPU or PyTorch fallback here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblvattn.so")

ABI_VERSION = 9

# name -> (restype, argtypes); must list every symbol include/*.h declares.
SIGNATURES = {
    "lv_abi_version": (c_int, []),
    "lv_error_string": (c_char_p, [c_int]),
    "lv_last_error": (c_char_p, []),
    "lv_store_kvcache": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                 c_int64, c_int, c_int, c_int64, c_void_p]),
    "lv_qk_norm_rope_store": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                      c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "lv_add_rms_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                c_int64, c_int, c_float, c_void_p]),
    "lv_silu_and_mul": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "lv_prefill_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                c_int, c_int, c_int, c_float, c_int64, c_void_p, c_size_t, c_int,
                                c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "lv_prefill_order_lanes": (c_int, [c_int, c_int]),
    "lv_prefill_tile_rows": (c_int, [c_int, c_int, c_int]),
    "lv_prefill_order_columns": (c_int, [c_int]),
    "lv_comm_all_reduce_add_rms_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "lv_prefill_num_splits": (c_int, [c_int, c_int, c_int, c_int]),
    "lv_prefill_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "lv_decode_num_splits": (c_int, [c_int, c_int, c_int, c_int]),
    "lv_decode_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "lv_decode_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                               c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                               c_void_p, c_size_t, c_int, c_void_p, c_int, c_void_p]),
    "lv_decode_attn_static": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                      c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                      c_int, c_void_p, c_int, c_void_p]),
    "lv_decode_attn_planned": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                       c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                       c_void_p, c_size_t, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "lv_prefill_max_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "lv_comm_create": (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    "lv_comm_get_handle": (c_int, [c_void_p, c_void_p]),
    "lv_comm_connect": (c_int, [c_void_p, c_char_p]),
    "lv_comm_all_reduce": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "lv_comm_failed": (c_int, [c_void_p]),
    "lv_comm_set_one_shot_inflow": (c_int, [c_void_p, c_size_t]),
    "lv_comm_one_shot_inflow": (c_size_t, [c_void_p]),
    "lv_comm_destroy": (c_int, [c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load liblvattn.so once. torch must be imported first on a GPU process so that the HIP
    runtime torch already mapped (same soname) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C light_vllm_amd/csrc` (hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.lv_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"liblvattn.so ABI version {got} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        lib = load()
        msg = lib.lv_last_error().decode("utf-8", "replace")
        kind = lib.lv_error_string(code).decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({code}: {kind}): {msg}")

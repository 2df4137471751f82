"""ctypes binding of libnvx_hip.so (the C ABI declared in include/nvx.h).

The library is the product: if it is missing or fails to load, importing anything that computes
raises -- there is no CPU or PyTorch fallback on this path.
"""
from __future__ import annotations

import ctypes
import os

import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# NVX_LIB_PATH: another build of the same library (A/B timing of two versions of a kernel in one job)
LIB_PATH = os.environ.get("NVX_LIB_PATH") or os.path.join(_HERE, "libnvx_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nvx.h")


def header_abi_version() -> int:
    """NVX_ABI_VERSION as include/nvx.h defines it: the version this binding's SIGNATURES table is written for."""
    with open(HEADER_PATH) as f:
        m = re.search(r"^#define\s+NVX_ABI_VERSION\s+(\d+)", f.read(), re.M)
    if m is None:
        raise NvxLibraryError(f"{HEADER_PATH} does not define NVX_ABI_VERSION")
    return int(m.group(1))


_c_i64, _c_i32, _c_f32, _c_vp, _c_u64 = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/nvx.h one to one
SIGNATURES = {
    "nvx_abi_version": (_c_i32, []),
    "nvx_last_error": (ctypes.c_char_p, []),
    "nvx_store_kv": (_c_i32, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp]),
    "nvx_rmsnorm": (_c_i32, [_c_vp, _c_i64, _c_vp, _c_vp, _c_i32, _c_vp, _c_vp, _c_i64, _c_i64, _c_f32, _c_vp]),
    "nvx_rope": (_c_i32, [_c_vp, _c_vp, _c_i64, _c_vp, _c_i64, _c_vp, _c_i64, _c_vp, _c_vp, _c_i64, _c_i32, _c_i32, _c_vp]),
    "nvx_qknorm_rope_store": (_c_i32, [_c_vp, _c_i64, _c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_f32, _c_vp, _c_vp, _c_vp,
                                       _c_vp, _c_i64, _c_i64, _c_i32, _c_i32, _c_vp]),
    "nvx_silu_mul": (_c_i32, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    "nvx_decode_plan_ws_ints": (_c_i64, [_c_i64, _c_i32]),
    "nvx_decode_partial_ws_floats": (_c_i64, [_c_i64, _c_i32, _c_i32]),
    "nvx_decode_plan": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i32, _c_vp, _c_i64, _c_vp]),
    "nvx_paged_decode_qkv": (_c_i32, [_c_vp, _c_i64, _c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_f32, _c_vp, _c_vp, _c_i64, _c_vp, _c_vp,
                                      _c_vp, _c_vp, _c_i64, _c_i32, _c_i32, _c_f32, _c_vp]),
    "nvx_paged_decode": (_c_i32, [_c_vp, _c_i64, _c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64,
                                  _c_i32, _c_i32, _c_f32, _c_vp]),
    "nvx_paged_decode_noalloc_max_batch": (_c_i64, []),
    "nvx_paged_decode_noalloc": (_c_i32, [_c_vp, _c_i64, _c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64,
                                          _c_i32, _c_i32, _c_f32, _c_vp]),
    "nvx_prefill_plan_ws_ints": (_c_i64, [_c_i64]),
    "nvx_prefill_plan": (_c_i32, [_c_vp, _c_i64, _c_vp, _c_i64, _c_vp]),
    "nvx_paged_prefill": (_c_i32, [_c_vp, _c_i64, _c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64,
                                   _c_i32, _c_i32, _c_f32, _c_vp]),
    "nvx_linear_decode_splits": (_c_i32, [_c_i64, _c_i64]),
    "nvx_linear_decode": (_c_i32, [_c_vp, _c_i64, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i32, _c_i32, _c_vp]),
    "nvx_rmsnorm_splitk": (_c_i32, [_c_vp, _c_i32, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_f32, _c_vp]),
    "nvx_sample_ws_bytes": (_c_i64, [_c_i64]),
    "nvx_sample": (_c_i32, [_c_vp, _c_i32, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_u64, _c_u64, _c_vp, _c_vp, _c_i64, _c_vp]),
    "nvx_sampler_probs": (_c_i32, [_c_vp, _c_i32, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp]),
    "nvx_sample_shard": (_c_i32, [_c_vp, _c_i32, _c_i64, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_u64, _c_u64, _c_vp, _c_vp, _c_i64, _c_vp]),
    "nvx_sample_pick": (_c_i32, [_c_vp, _c_i64, _c_i64, _c_vp, _c_vp]),
    "nvx_block_table_write_rows": (_c_i32, [_c_vp, _c_vp, _c_vp, _c_i64, _c_vp, _c_i64, _c_i64, _c_vp]),
    "nvx_decode_step_setup": (_c_i32, [_c_vp, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "nvx_peer_exchange_bytes": (_c_i64, [_c_i32, _c_i64, _c_i64]),
    "nvx_peer_create": (_c_i32, [_c_i32, _c_i32, _c_i64, _c_i64, _c_i64, ctypes.POINTER(_c_vp)]),
    "nvx_peer_export": (_c_i32, [_c_vp, ctypes.c_char_p]),
    "nvx_peer_import": (_c_i32, [_c_vp, _c_i32, ctypes.c_char_p]),
    "nvx_peer_status": (_c_i32, [_c_vp]),
    "nvx_peer_destroy": (_c_i32, [_c_vp]),
    "nvx_lmhead_max_rows": (_c_i64, []),
    "nvx_lmhead_rows_per_pass": (_c_i64, []),
    "nvx_lmhead_packed_elems": (_c_i64, [_c_i64, _c_i64]),
    "nvx_lmhead_ws_bytes": (_c_i64, [_c_i64]),
    "nvx_lmhead_pack": (_c_i32, [_c_vp, _c_i64, _c_i64, _c_vp, _c_vp]),
    "nvx_lmhead_sample": (_c_i32, [_c_vp, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_u64, _c_u64, _c_vp, _c_i64, _c_vp,
                                   _c_vp, _c_vp, _c_i64, _c_vp]),
    "nvx_linear_stream": (_c_i32, [_c_vp, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp]),
    "nvx_splitk_sum": (_c_i32, [_c_vp, _c_i32, _c_i64, _c_i64, _c_i32, _c_vp, _c_vp]),
    "nvx_linear_block_max_rows": (_c_i64, []),
    "nvx_linear_block_slices": (_c_i32, [_c_i64, _c_i64, _c_i32]),
    "nvx_linear_block": (_c_i32, [_c_vp, _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_i32, _c_i32, _c_vp, _c_vp]),
    "nvx_peer_allreduce_norm": (_c_i32, [_c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_f32, _c_vp]),
}

_lib = None


class NvxLibraryError(RuntimeError):
    pass


# entry points that also exist as `<name>_f16` (IEEE half build of the same source, same signature)
F16_ENTRY_POINTS = ("nvx_rmsnorm", "nvx_rope", "nvx_qknorm_rope_store", "nvx_silu_mul", "nvx_linear_decode", "nvx_rmsnorm_splitk",
                    "nvx_paged_decode", "nvx_paged_decode_qkv", "nvx_paged_prefill", "nvx_peer_allreduce_norm", "nvx_lmhead_sample", "nvx_linear_stream", "nvx_splitk_sum",
                    "nvx_linear_block")
for _name in F16_ENTRY_POINTS:
    SIGNATURES[_name + "_f16"] = SIGNATURES[_name]


def load():
    """Load the HIP library once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NvxLibraryError(
            f"{LIB_PATH} not found: build it with `make -C nano_vllm_x_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NvxLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    want, got = header_abi_version(), lib.nvx_abi_version()
    if got != want:
        raise NvxLibraryError(f"{LIB_PATH} reports ABI version {got}, include/nvx.h declares {want}: rebuild it "
                              "(`make -C nano_vllm_x_amd/csrc`)")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().nvx_last_error().decode(errors="replace")
        raise RuntimeError(f"{what or 'nvx'} failed ({rc}): {msg}")

"""The C-ABI library loads on a machine without a GPU and exports every symbol include/nvx.h declares
(no compute call is made here)."""
from __future__ import annotations

import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nvx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nvx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_binds():
    from nano_vllm_x_amd import _lib

    names = declared_symbols()
    assert len(names) >= 20
    assert set(_lib.SIGNATURES) | {"nvx_abi_version", "nvx_last_error"} >= set(names), set(names) - set(_lib.SIGNATURES)
    assert set(_lib.SIGNATURES) <= set(names), set(_lib.SIGNATURES) - set(names)


def test_binding_passes_as_many_arguments_as_the_header_declares():
    """A prototype that gains a parameter must gain it in the ctypes table too: a missing trailing argument
    is not an error at the call, the callee just reads a stale register."""
    from nano_vllm_x_amd import _lib

    text = open(os.path.join(ROOT, "include", "nvx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    seen = 0
    for name, args in re.findall(r"\b(nvx_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        n = 0 if args.strip() in ("", "void") else args.count(",") + 1
        if name in _lib.SIGNATURES:
            assert len(_lib.SIGNATURES[name][1]) == n, f"{name}: header declares {n} parameters, binding passes {len(_lib.SIGNATURES[name][1])}"
            seen += 1
    assert seen == len(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from nano_vllm_x_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build first: make -C nano_vllm_x_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/nvx.h but not exported"
    lib.nvx_abi_version.restype = ctypes.c_int
    assert lib.nvx_abi_version() == _lib.header_abi_version(), "library built from another include/nvx.h: rebuild"
    # pure host helpers answer without a device
    lib.nvx_decode_plan_ws_ints.restype = ctypes.c_int64
    lib.nvx_decode_plan_ws_ints.argtypes = [ctypes.c_int64, ctypes.c_int]
    assert lib.nvx_decode_plan_ws_ints(256, 8) > 0
    lib.nvx_linear_decode_splits.argtypes = [ctypes.c_int64, ctypes.c_int64]
    assert lib.nvx_linear_decode_splits(1024, 3072) == 4 and lib.nvx_linear_decode_splits(4096, 1024) == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nano_vllm_x_amd import _lib

    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "absent.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    try:
        _lib.load()
    except _lib.NvxLibraryError as e:
        assert "absent.so" in str(e) or "build" in str(e).lower()
    else:
        raise AssertionError("loading a missing library must raise")


def test_build_entry_point_runs_end_to_end():
    """`python __graft_entry__.py` is the documented build command: it must compile (a no-op when the objects
    are current), load the library and agree with it on the ABI version."""
    import __graft_entry__ as g

    g.build()


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    from nano_vllm_x_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "header_abi_version", lambda: 10 ** 6)
    try:
        _lib.load()
    except _lib.NvxLibraryError as e:
        assert "ABI version" in str(e)
    else:
        raise AssertionError("a version mismatch must raise")


def test_every_kernel_entry_point_has_a_custom_op():
    """`torch.ops.nvx.<name>` exists for every launching entry point of the C ABI (size queries and diagnostics
    aside): the operator surface the Python host calls the kernels through."""
    import torch

    import nano_vllm_x_amd.torch_ops  # noqa: F401

    op_of = {"nvx_store_kv": "store_kvcache", "nvx_rmsnorm": "rmsnorm", "nvx_rope": "rope", "nvx_silu_mul": "silu_mul",
             "nvx_qknorm_rope_store": "qknorm_rope_store", "nvx_linear_decode": "linear_decode", "nvx_rmsnorm_splitk": "rmsnorm_splitk",
             "nvx_decode_plan": "decode_plan", "nvx_paged_decode": "paged_decode", "nvx_paged_decode_qkv": "paged_decode_qkv",
             "nvx_paged_decode_noalloc": "paged_decode_noalloc", "nvx_prefill_plan": "prefill_plan", "nvx_paged_prefill": "paged_prefill",
             "nvx_sample": "sample", "nvx_sampler_probs": "sampler_probs", "nvx_block_table_write_rows": "block_table_write_rows",
             "nvx_decode_step_setup": "decode_step_setup", "nvx_peer_allreduce_norm": "peer_allreduce_norm",
             "nvx_sample_shard": "sample_shard", "nvx_sample_pick": "sample_pick", "nvx_lmhead_pack": "lmhead_pack",
             "nvx_lmhead_sample": "lmhead_sample", "nvx_linear_stream": "linear_stream", "nvx_splitk_sum": "splitk_sum",
             "nvx_linear_block": "linear_block"}
    helpers = {"nvx_abi_version", "nvx_last_error", "nvx_linear_decode_splits", "nvx_decode_plan_ws_ints", "nvx_decode_partial_ws_floats",
               "nvx_prefill_plan_ws_ints", "nvx_sample_ws_bytes", "nvx_paged_decode_noalloc_max_batch",
               "nvx_lmhead_max_rows", "nvx_lmhead_rows_per_pass", "nvx_lmhead_packed_elems", "nvx_lmhead_ws_bytes",
               "nvx_linear_block_max_rows", "nvx_linear_block_slices",
               # start-up calls of the direct-peer exchange (allocation, handle exchange, status, teardown): host side
               "nvx_peer_exchange_bytes", "nvx_peer_create", "nvx_peer_export", "nvx_peer_import", "nvx_peer_status", "nvx_peer_destroy"}
    # the float16 entry points have no operator of their own: the operator of the bfloat16 one picks them by tensor dtype
    from nano_vllm_x_amd._lib import F16_ENTRY_POINTS
    assert set(F16_ENTRY_POINTS) <= set(op_of)
    half = {name + "_f16" for name in F16_ENTRY_POINTS}
    assert set(declared_symbols()) == set(op_of) | helpers | half, set(declared_symbols()) ^ (set(op_of) | helpers | half)
    for sym, op in op_of.items():
        assert hasattr(torch.ops.nvx, op), f"{sym} has no torch.ops.nvx.{op}"


def test_the_only_users_of_m0_in_the_device_code_are_the_lds_dma_copies(tmp_path):
    """The LDS-DMA copies of the attention kernels set m0 (the LDS destination) from inline assembly, and the toolchain
    refuses m0 in a clobber list.  That is sound only while the compiler itself keeps nothing in m0 in those kernels:
    every instruction of the library's gfx950 code that names m0 must be a move issued for such a copy, one per copy."""
    import shutil
    import subprocess

    from nano_vllm_x_amd import _lib

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        import pytest
        pytest.skip("no llvm-objdump in this image")
    so = tmp_path / "libnvx_hip.so"
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    code_objects = [p for p in tmp_path.iterdir() if "gfx950" in p.name]
    assert code_objects, "no gfx950 code object found in the library"
    moves = copies = 0
    for co in code_objects:
        text = subprocess.run([objdump, "-d", str(co)], check=True, capture_output=True, text=True).stdout
        for line in text.splitlines():
            ins = line.split("//")[0].strip()
            if not ins or ins.endswith(":"):
                continue
            if re.search(r"\bglobal_load_lds_", ins):
                copies += 1
            if re.search(r"\bm0\b", ins):
                assert re.match(r"s_mov_b32 m0, s\d+$", ins), f"m0 is used by something other than an LDS-DMA set-up: {ins}"
                moves += 1
    assert copies > 0 and moves == copies, f"{moves} moves to m0 for {copies} LDS-DMA copies"

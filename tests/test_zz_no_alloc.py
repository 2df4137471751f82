"""nvx_paged_decode_noalloc (the entry point that allocates nothing: plan and partials in an array built into the
library) against nvx_decode_plan + nvx_paged_decode with caller-provided workspaces, through the C ABI.
Bit-exact, on two calls in a row, the second with a smaller input (the array keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(lib, seed, lens, hq, hkv, n_slots=20000):
    gen = torch.Generator().manual_seed(seed)
    b = len(lens)
    lens_np = np.asarray(lens, dtype=np.int32)
    total = int(lens_np.sum())
    idx = torch.randint(0, n_slots, (total,), generator=gen).to(torch.int32).to(DEV)
    kv_len = torch.from_numpy(lens_np).to(DEV)
    kv_start = torch.from_numpy(np.concatenate(([0], np.cumsum(lens_np[:-1]))).astype(np.int64)).to(DEV)
    kc = torch.randn(n_slots, hkv, 128, generator=gen).to(torch.bfloat16).to(DEV)
    vc = torch.randn(n_slots, hkv, 128, generator=gen).to(torch.bfloat16).to(DEV)
    q = torch.randn(b, hq * 128, generator=gen).to(torch.bfloat16).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    # existing entry points, caller-provided workspaces
    plan = torch.zeros(int(lib.nvx_decode_plan_ws_ints(b, hkv)), dtype=torch.int32, device=DEV)
    part = torch.empty(int(lib.nvx_decode_partial_ws_floats(b, hq, hkv)), dtype=torch.float32, device=DEV)
    want = torch.full((b, hq * 128), 7.0, dtype=torch.bfloat16, device=DEV)
    rc = lib.nvx_decode_plan(kv_len.data_ptr(), kv_start.data_ptr(), None, b, hkv, plan.data_ptr(), plan.numel(), st)
    assert rc == 0, lib.nvx_last_error()
    rc = lib.nvx_paged_decode(q.data_ptr(), q.stride(0), kc.data_ptr(), vc.data_ptr(), n_slots, idx.data_ptr(), plan.data_ptr(),
                              part.data_ptr(), want.data_ptr(), b, hq, hkv, 128 ** -0.5, st)
    assert rc == 0, lib.nvx_last_error()
    # the entry point that allocates nothing
    got = torch.full((b, hq * 128), -7.0, dtype=torch.bfloat16, device=DEV)
    rc = lib.nvx_paged_decode_noalloc(q.data_ptr(), q.stride(0), kc.data_ptr(), vc.data_ptr(), n_slots, idx.data_ptr(),
                                      kv_len.data_ptr(), kv_start.data_ptr(), got.data_ptr(), b, hq, hkv, 128 ** -0.5, st)
    assert rc == 0, lib.nvx_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(want.float()).all()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (seed, float((got.float() - want.float()).abs().max()))
    return int(plan[0])     # number of chunks the plan made


def test_noalloc_decode_equals_the_workspace_entry_points_on_two_calls_in_a_row():
    from nano_vllm_x_amd import _lib

    lib = _lib.load()
    assert lib.nvx_paged_decode_noalloc_max_batch() == 256
    rng = np.random.default_rng(0)
    big = rng.integers(1, 3000, size=200).tolist() + [1, 2, 0, 4097]          # split sequences, an empty row, tiny rows
    n_big = _case(lib, 1, big, 16, 8)
    assert n_big > len(big), "the first call was meant to split sequences (partials in the built-in array)"
    small = [900, 5, 1500, 1, 0, 33]                                           # smaller batch, fewer and other chunks
    n_small = _case(lib, 2, small, 16, 8)
    assert n_small < n_big
    _case(lib, 3, [4000, 17], 32, 8)           # other head counts over the same array
    _case(lib, 4, rng.integers(1, 700, size=256).tolist(), 8, 1)


def test_noalloc_decode_rejects_what_the_array_cannot_hold():
    from nano_vllm_x_amd import _lib

    lib = _lib.load()
    z = torch.zeros(16, dtype=torch.int64, device=DEV)
    args = (z.data_ptr(), 64 * 128, z.data_ptr(), z.data_ptr(), 16, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    assert lib.nvx_paged_decode_noalloc(*args, 257, 16, 8, ctypes.c_float(0.1), st) == -4          # batch beyond the maximum
    assert b"at most 256" in lib.nvx_last_error()
    assert lib.nvx_paged_decode_noalloc(*args, 256, 64, 8, ctypes.c_float(0.1), st) == -4          # partials beyond the array
    assert b"partial floats" in lib.nvx_last_error()
    torch.cuda.synchronize()


_CAPTURE_CODE = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from nano_vllm_x_amd import _lib
DEV = "cuda:0"
lib = _lib.load()
gen = torch.Generator().manual_seed(11)
hq, hkv, n_slots = 16, 8, 20000
lens = np.asarray([900, 5, 1500, 1, 0, 33, 2600, 129], dtype=np.int32)      # split sequences, an empty row
b, total = len(lens), int(lens.sum())
idx = torch.randint(0, n_slots, (total,), generator=gen).to(torch.int32).to(DEV)
kv_len = torch.from_numpy(lens).to(DEV)
kv_start = torch.from_numpy(np.concatenate(([0], np.cumsum(lens[:-1]))).astype(np.int64)).to(DEV)
kc = torch.randn(n_slots, hkv, 128, generator=gen).to(torch.bfloat16).to(DEV)
vc = torch.randn(n_slots, hkv, 128, generator=gen).to(torch.bfloat16).to(DEV)
q = torch.randn(b, hq * 128, generator=gen).to(torch.bfloat16).to(DEV)
plan = torch.zeros(int(lib.nvx_decode_plan_ws_ints(b, hkv)), dtype=torch.int32, device=DEV)
part = torch.empty(int(lib.nvx_decode_partial_ws_floats(b, hq, hkv)), dtype=torch.float32, device=DEV)
want = torch.empty(b, hq * 128, dtype=torch.bfloat16, device=DEV)
got = torch.full((b, hq * 128), -7.0, dtype=torch.bfloat16, device=DEV)

def workspace_path():
    st = torch.cuda.current_stream().cuda_stream
    assert lib.nvx_decode_plan(kv_len.data_ptr(), kv_start.data_ptr(), None, b, hkv, plan.data_ptr(), plan.numel(), st) == 0
    assert lib.nvx_paged_decode(q.data_ptr(), q.stride(0), kc.data_ptr(), vc.data_ptr(), n_slots, idx.data_ptr(), plan.data_ptr(),
                                part.data_ptr(), want.data_ptr(), b, hq, hkv, 128 ** -0.5, st) == 0, lib.nvx_last_error()

workspace_path()
torch.cuda.synchronize()
# the FIRST call of the entry point in this process is made while the stream is being captured
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.nvx_paged_decode_noalloc(q.data_ptr(), q.stride(0), kc.data_ptr(), vc.data_ptr(), n_slots, idx.data_ptr(),
                                      kv_len.data_ptr(), kv_start.data_ptr(), got.data_ptr(), b, hq, hkv, 128 ** -0.5, st)
    assert rc == 0, lib.nvx_last_error()
torch.cuda.synchronize()
assert bool((got.float() == -7.0).all()), "a captured call must not run"
for rep in range(3):
    q.copy_(torch.randn(b, hq * 128, generator=gen).to(torch.bfloat16))     # same buffers, new contents
    if rep == 2:
        kv_len[2] = 700                                                       # and another plan: the graph re-plans on replay
    workspace_path()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(want.float()).all()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (rep, float((got.float() - want.float()).abs().max()))
print("capture-ok")
"""


def test_noalloc_decode_first_called_under_graph_capture_replays_bit_exactly():
    """The callers this entry point exists for include code under hipGraph capture.  In a fresh process its very first
    call (the one that looks up the device address of the built-in array) is made inside a capture; replays, on new
    contents of the same buffers and on a changed length, must equal the workspace entry points bit for bit."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CAPTURE_CODE % root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "capture-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]

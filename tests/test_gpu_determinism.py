"""Run-to-run determinism of the attention kernels at full size, through the C ABI.

Both kernels move data behind the compiler's back (copies straight into LDS issued from inline
assembly, streaming loads consumed tiles later).  A missing wait in such code does not give wrong
answers every time: it gives answers that depend on memory timing.  A bit-for-bit comparison of repeated
launches on the same inputs is the cheapest detector of that class of fault (one was found this way: slot
ids prefetched into registers by loads issued from inline assembly, which the register allocator is free
to copy before the data has arrived).  Integer outputs and bf16 outputs alike must be identical.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from nano_vllm_x_amd import _lib, ops as _ops

    _lib.load()
    return _ops


def _lens(n, lo, hi, seed):
    return [int(x) for x in np.random.default_rng(seed).integers(lo, hi, size=n)]


def test_prefill_repeated_launches_are_identical(ops):
    hq, hkv = 16, 8
    lens = _lens(256, 100, 1025, 0)                     # the bench workload's shape: far more work units than CUs
    t = sum(lens)
    gen = torch.Generator(device=DEV).manual_seed(3)
    q = torch.randn(t, hq * 128, generator=gen, device=DEV).to(torch.bfloat16)
    kc = torch.randn(t + 64, hkv, 128, generator=gen, device=DEV).to(torch.bfloat16)
    vc = torch.randn(t + 64, hkv, 128, generator=gen, device=DEV).to(torch.bfloat16)
    indptr = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).to(DEV)
    idx = torch.randperm(t + 64, generator=gen, device=DEV)[:t].to(torch.int32)   # scattered rows: no two tiles arrive alike
    plan = ops.prefill_plan_ws(len(lens), DEV)
    ops.prefill_plan(indptr, plan)
    first = ops.paged_prefill(q, kc, vc, indptr, indptr, idx, plan, hq, hkv, 128 ** -0.5).clone()
    for _ in range(6):
        again = ops.paged_prefill(q, kc, vc, indptr, indptr, idx, plan, hq, hkv, 128 ** -0.5)
        torch.cuda.synchronize()
        assert torch.equal(first, again), "prefill attention differs between two launches on the same inputs"


@pytest.mark.parametrize("batch", [29, 123])
def test_decode_repeated_launches_are_identical(ops, batch):
    hq, hkv = 16, 8
    lens = np.asarray(_lens(batch, 100, 1025, 1), dtype=np.int32)
    total = int(lens.sum())
    gen = torch.Generator(device=DEV).manual_seed(4)
    slots = total + 64
    kc = torch.randn(slots, hkv, 128, generator=gen, device=DEV).to(torch.bfloat16)
    vc = torch.randn(slots, hkv, 128, generator=gen, device=DEV).to(torch.bfloat16)
    q = torch.randn(batch, hq * 128, generator=gen, device=DEV).to(torch.bfloat16)
    idx = torch.randperm(slots, generator=gen, device=DEV)[:total].to(torch.int32)
    kv_len = torch.from_numpy(lens).to(DEV)
    kv_start = torch.from_numpy(np.concatenate(([0], np.cumsum(lens[:-1]))).astype(np.int64)).to(DEV)
    ws = ops.DecodeWorkspace(batch, hq, hkv, DEV)
    ops.decode_plan(kv_len, kv_start, ws)
    first = ops.paged_decode(q, kc, vc, idx, batch, ws, 128 ** -0.5).clone()
    for _ in range(6):
        again = ops.paged_decode(q, kc, vc, idx, batch, ws, 128 ** -0.5)
        torch.cuda.synchronize()
        assert torch.equal(first, again), "decode attention differs between two launches on the same inputs"

"""The LM head that draws as it computes (nvx_lmhead_pack / nvx_lmhead_sample) against the two launches it replaces.

Reference path: ParallelLMHead.forward + Sampler.forward (nanovllmx/layers/vocab_parallel_embedding.py:147-151,
layers/sampler.py:39-55).  Three things are pinned:
  * the packed weight is a pure permutation of the rows (bit for bit against the same permutation written in torch);
  * the logits the kernel computes (written out on request) are the fp32 products rounded to the 16-bit format: within
    one unit in the last place of an fp32 matmul of the same operands (summation order is all that can differ);
  * the token it draws is, bit for bit, the token nvx_sample draws from THOSE logits with the same seed and step (same
    noise per (row, token), same tie rule), for sampled and greedy rows, whole rows and tensor-parallel shards.
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


def _pack_reference(w):
    n, k = w.shape
    pad = (-n) % 16
    if pad:
        w = torch.cat([w, torch.zeros(pad, k, dtype=w.dtype, device=w.device)])
    t = w.shape[0] // 16
    # [tile][row r][k-step][k-block q][8] -> [tile][k-step][q][r][8]
    return w.view(t, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


@pytest.mark.parametrize("n,k", [(64, 1024), (1000, 1024), (151936, 1024), (48, 256)])
def test_pack_is_the_documented_permutation(ops, n, k):
    gen = torch.Generator().manual_seed(n)
    w = torch.randn(n, k, generator=gen).to(torch.bfloat16).to(DEV)
    got = ops.lmhead_pack(w)
    assert torch.equal(got.view(torch.int16), _pack_reference(w).view(torch.int16))


def _case(ops, b, n_rows, dtype, seed, temps=None, vocab=None):
    gen = torch.Generator().manual_seed(seed)
    k = 1024
    vocab = vocab or n_rows
    w = (torch.randn(n_rows, k, generator=gen) * k ** -0.5).to(dtype).to(DEV)
    x = torch.randn(b, k, generator=gen).to(dtype).to(DEV)
    t = torch.full((b,), 0.6) if temps is None else torch.tensor(temps, dtype=torch.float32)
    t = t.to(DEV)
    packed = ops.lmhead_pack(w)
    padded = (n_rows + 15) // 16 * 16
    logits = torch.full((b, padded), float("nan"), dtype=dtype, device=DEV)
    ids = ops.lmhead_sample(packed, n_rows, x, t, 0, vocab, 7, 3, logits_out=logits)
    torch.cuda.synchronize()
    # (a) the logits: fp32 products, one rounding
    ref = x.float() @ w.float().t()
    got = logits[:, :n_rows].float()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert bool(((got - ref).abs() <= ref.abs() * ulp + 2e-6).all()), float((got - ref).abs().max())
    assert bool((logits[:, n_rows:].float() == 0).all())       # padding rows of the packed weight are zeros
    # (b) the draw: nvx_sample on those very logits
    lg = logits[:, :vocab].contiguous()
    if dtype == torch.float16:
        lg = lg.float()
    full = torch.full((b,), vocab, dtype=torch.int64, device=DEV)
    want = ops.sample(lg, t, full, torch.ones(b, device=DEV), torch.zeros(b, device=DEV), 7, 3)
    assert torch.equal(ids, want), (ids.tolist(), want.tolist())
    greedy = (t <= 0).nonzero().flatten().tolist()
    for r in greedy:
        assert int(ids[r]) == int(lg[r].float().argmax())
    # (c) keys: the same tokens through sample_pick
    keys = ops.lmhead_sample(packed, n_rows, x, t, 0, vocab, 7, 3, keys=True)
    assert torch.equal(ops.sample_pick(keys.view(1, b)), want)
    return w, x, t, want


@pytest.mark.parametrize("b", [1, 5, 16, 17, 29, 48, 64, 65, 123, 200])
def test_fused_lm_head_draws_the_token_of_the_unfused_path(ops, b):
    _case(ops, b, 4096, torch.bfloat16, 100 + b)


def test_full_vocabulary_ragged_rows_greedy_rows_and_padding_tokens(ops):
    _case(ops, 29, 151936, torch.bfloat16, 1)
    # rows not a multiple of 16 (the last tile is padded with zero rows that must never be drawn), some rows greedy
    _case(ops, 7, 1000, torch.bfloat16, 2, temps=[0.6, 0.0, 1.3, -1.0, 0.6, 2.0, 0.05])
    # a vocabulary that ends before the weight does (padding of a vocabulary split over ranks)
    _case(ops, 9, 1024, torch.bfloat16, 3, vocab=1001)
    _case(ops, 33, 2048, torch.float16, 4)
    # 65..128 rows: two row groups in one launch on paired workgroups (full vocabulary: one workgroup per compute unit), and
    # the same height where the grid cannot be paired (19 workgroups): one pass after the other
    _case(ops, 100, 151936, torch.bfloat16, 5)
    _case(ops, 70, 1200, torch.bfloat16, 6, temps=[0.6] * 35 + [0.0] * 35)
    _case(ops, 128, 4096, torch.float16, 7)


def test_shards_of_the_vocabulary_pick_the_token_of_the_whole_row(ops):
    gen = torch.Generator().manual_seed(9)
    b, k, v = 21, 1024, 6016
    w = (torch.randn(v, k, generator=gen) * k ** -0.5).to(torch.bfloat16).to(DEV)
    x = torch.randn(b, k, generator=gen).to(torch.bfloat16).to(DEV)
    t = torch.full((b,), 0.8, device=DEV)
    whole = ops.lmhead_sample(ops.lmhead_pack(w), v, x, t, 0, v, 11, 5)
    for shards in (2, 4):
        per = v // shards
        keys = torch.stack([ops.lmhead_sample(ops.lmhead_pack(w[s * per:(s + 1) * per].contiguous()), per, x, t, s * per, v, 11, 5, keys=True)
                            for s in range(shards)])
        assert torch.equal(ops.sample_pick(keys), whole)


def test_limits_are_refused_on_the_host(ops):
    w = torch.zeros(64, 1024, dtype=torch.bfloat16, device=DEV)
    packed = ops.lmhead_pack(w)
    t = torch.ones(1025, device=DEV)
    with pytest.raises(RuntimeError):
        ops.lmhead_sample(packed, 64, torch.zeros(1025, 1024, dtype=torch.bfloat16, device=DEV), t, 0, 64, 1, 1)
    with pytest.raises(RuntimeError):      # another hidden size than the kernel is built for
        ops.lmhead_sample(ops.lmhead_pack(torch.zeros(64, 512, dtype=torch.bfloat16, device=DEV)), 64,
                          torch.zeros(4, 512, dtype=torch.bfloat16, device=DEV), t[:4], 0, 64, 1, 1)
    assert ops.lmhead_max_rows() == 1024


@pytest.mark.parametrize("m,n,k", [(1, 4096, 4096), (29, 6144, 4096), (64, 1024, 12288), (17, 1000, 1024), (33, 24576, 4096)])
def test_streamed_projection_and_its_consumers(ops, m, n, k):
    """nvx_linear_stream: per 1024-wide slice of k, fp32 products of the same 16-bit operands; summed by nvx_splitk_sum
    (plain and SiluAndMul) and by nvx_rmsnorm_splitk exactly as the K-split skinny projection's partials are."""
    from oracle.layers_oracle import rmsnorm_oracle, silu_mul_oracle

    gen = torch.Generator().manual_seed(m + n)
    dt = torch.bfloat16
    w = (torch.randn(n, k, generator=gen) * k ** -0.5).to(dt).to(DEV)
    x = torch.randn(m, k, generator=gen).to(dt).to(DEV)
    part = ops.linear_stream(x, ops.lmhead_pack(w), n)
    assert part.shape == (k // 1024, m, n)
    for s in range(k // 1024):
        ref = x[:, s * 1024:(s + 1) * 1024].float() @ w[:, s * 1024:(s + 1) * 1024].float().t()
        assert bool(((part[s] - ref).abs() <= ref.abs() * 1e-5 + 2e-5).all()), (s, float((part[s] - ref).abs().max()))
    full = x.float() @ w.float().t()
    y = ops.splitk_sum(part, dt)
    assert y.dtype == dt and bool(((y.float() - full).abs() <= full.abs() * 2.0 ** -8 + 1e-5).all())
    if n % 8 == 0:
        act = ops.splitk_sum(part, dt, silu_mul=True)
        want = silu_mul_oracle(y.cpu())       # the oracle on the very sum the kernel rounded
        ulp = (act.cpu().view(torch.int16).int() - want.view(torch.int16).int()).abs().max().item()
        assert act.shape == (m, n // 2) and ulp <= 2, ulp
    nw = (1 + 0.1 * torch.randn(n, generator=gen)).to(dt).to(DEV)
    res = torch.randn(m, n, generator=gen).to(dt).to(DEV)
    if n % 4 == 0 and n <= 8192:
        yn, ro = ops.rmsnorm_splitk(part, nw, 1e-6, residual=res)
        yr, rr = rmsnorm_oracle(y.cpu(), nw.cpu(), 1e-6, res.cpu())
        assert torch.equal(ro.cpu().view(torch.int16), rr.view(torch.int16))
        assert (yn.cpu().view(torch.int16).int() - yr.view(torch.int16).int()).abs().max().item() <= 1


@pytest.mark.parametrize("n,k", [(4096, 4096), (2048, 512), (4096, 12288), (48, 1024), (4112, 1536)])
def test_blocked_projection_matches_fp32_matmul_and_its_fused_epilogues(ops, n, k):
    """nvx_linear_block (the weight streamed from the tiled copy, the rows through LDS in chunks of k) against a torch
    fp32 matmul of the same 16-bit operands: mode 0 within the output rounding (2^-8 |ref| + 1e-3), the k-slices'
    partials of mode 2 sum to the product within fp32 summation error and feed nvx_rmsnorm_splitk as the partials of the
    skinny kernel do, and the SiluAndMul epilogue (mode 1) is bit-equal to nvx_silu_mul of mode 0's rounded output (same
    accumulation order, same roundings).  Ragged rows (17, 33, 65, 100, 129, 200), ragged tiles (n = 48, 4112), both formats."""
    gen = torch.Generator().manual_seed(n + k)
    for dtype in (torch.bfloat16, torch.float16):
        w = (torch.randn(n, k, generator=gen) * k ** -0.5).to(dtype).to(DEV)
        packed = ops.lmhead_pack(w)
        slices = ops.linear_block_slices(n, k, 2)
        assert slices >= 1 and (k // 512) % slices == 0
        for m in (1, 17, 32, 33, 65, 100, 128, 129, 192, 200, 256):
            x = torch.randn(m, k, generator=gen).to(dtype).to(DEV)
            want = x.float() @ w.float().t()
            y = ops.linear_block(x, packed, n, mode=0)
            ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
            assert ((y.float() - want).abs() <= want.abs() * ulp + 1e-3).all(), (m, (y.float() - want).abs().max())
            parts = ops.linear_block(x, packed, n, mode=2, k_slices=slices)
            assert parts.shape == (slices, m, n)
            torch.testing.assert_close(parts.sum(0), want, rtol=1e-4, atol=2e-4)
            if n % 32 == 0:
                assert torch.equal(ops.linear_block(x, packed, n, mode=1), ops.silu_mul(y))
            if n % 8 == 0 and dtype == torch.bfloat16:
                nw = torch.randn(n, generator=gen).to(dtype).to(DEV)
                res = torch.randn(m, n, generator=gen).to(dtype).to(DEV)
                summed = parts[0].clone()             # the slices added in order, as the norm kernel adds them
                for sl in range(1, slices):
                    summed += parts[sl]
                y_ref, r_ref = ops.rmsnorm(summed.to(dtype), nw, 1e-6, residual=res)
                y2, r2 = ops.rmsnorm_splitk(parts, nw, 1e-6, residual=res)
                assert torch.equal(r2, r_ref)
                assert (y2.view(torch.int16).int() - y_ref.view(torch.int16).int()).abs().max() <= 2
    with pytest.raises(RuntimeError, match="256 rows"):
        ops.linear_block(torch.zeros(257, k, dtype=packed.dtype, device=DEV), packed, n)
    assert ops.linear_block_slices(4096, 4096 + 128, 2) == 0      # k not a whole number of ring turns

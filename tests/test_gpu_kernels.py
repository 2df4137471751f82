"""GPU parity: every HIP kernel, called through the C ABI, against the oracle and the golden vectors.

Tolerances (stated per test):
  * integer/byte work (KV scatter, block-table kernels): bit-exact;
  * RoPE: bit-exact against the reference's eager output (every rounding is reproduced);
  * RMSNorm, SiluAndMul, fused qk-norm+RoPE: <= 1 bf16 ulp (fp32 reduction order / expf differ);
  * decode attention (fp32 probabilities): |out - fp32 oracle| <= 1e-3 * max(1, |oracle|) + the rounding of
    the bf16 output (half an ulp = 2^-9 relative) -- the fp32-accumulated value is what is held to 1e-3;
  * prefill attention: the probabilities enter the PV MFMA as bf16 (as in every tensor-core flash
    kernel, the reference's backend included), which adds at most 2^-9 * max|V| to the bound above;
  * sampler probabilities: rtol 2e-5 on kept tokens, masks identical.
"""
import math

import numpy as np
import pytest
import torch

from helpers import bf16_ulp_diff, from_bits, golden_json, golden_npz, to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from nano_vllm_x_amd import _lib, ops as _ops

    _lib.load()
    return _ops


# The 16-bit format the attention cases below are built in.  tests/test_gpu_float16.py re-runs the edge cases with
# float16 (the *_f16 entry points); the rounding terms of the tolerances follow the format (ULP: one unit in the last
# place relative to the value, 2^-8 for bfloat16's 8 significant bits, 2^-11 for float16's 11).
DTYPE = torch.bfloat16


def ULP() -> float:
    return 2.0 ** -8 if DTYPE == torch.bfloat16 else 2.0 ** -11


def attn_close(got: torch.Tensor, want_f32: torch.Tensor, p_bf16_vmax: float = 0.0):
    got = got.float().cpu()
    want = want_f32.float().cpu()
    tol = 1e-3 * torch.clamp(want.abs(), min=1.0) + want.abs() * ULP() + p_bf16_vmax * ULP() / 2 + 1e-6
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} outside tolerance; max err {(got - want).abs().max():.3e}"


def decode_p_rounding(hq: int, hkv: int, vc: torch.Tensor) -> float:
    """max|V| when the decode kernel for this head ratio rounds the probabilities to bf16 before the value product (the
    matrix-core variant -- used from 8 query heads per kv head, and from 4 with one or two kv heads -- as the prefill
    kernel and FlashInfer's tensor-core decode do); 0 for the variant that keeps them in fp32."""
    g = hq // hkv
    return float(vc.float().abs().max()) if g >= 8 or (g >= 4 and hkv <= 2) else 0.0


# the two decode kernels: rows in registers + vector arithmetic (G <= 4), rows through LDS + MFMA (G >= 8)
DECODE_KERNEL_SHAPES = [(16, 8), (16, 2)]


# -----------------------------------------------------------------------------------------------
def test_store_kv_golden(ops):
    g = golden_npz("g4_store_kv.npz")
    hq, hkv, d, t, slots, layers, start_layer, layer = g["meta"].tolist()
    qkv = from_bits(g["qkv"]).to(DEV)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    kc, vc = from_bits(g["k_before"]).to(DEV), from_bits(g["v_before"]).to(DEV)
    li = layer - start_layer
    ops.store_kvcache(k.view(-1, hkv, d), v.view(-1, hkv, d), kc[li], vc[li], torch.from_numpy(g["idx"]).to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(to_bits(kc), g["k_after"]) and np.array_equal(to_bits(vc), g["v_after"])


def test_store_kv_large_and_skip(ops):
    from oracle.layers_oracle import store_kv_oracle

    gen = torch.Generator().manual_seed(1)
    t, hkv, d, slots = 3000, 8, 128, 5000
    qkv = torch.randn(t, (16 + 2 * hkv) * d, generator=gen).to(torch.bfloat16)
    perm = torch.randperm(slots, generator=gen)[:t]
    perm[::7] = -1
    k = qkv[:, 16 * d:(16 + hkv) * d]
    v = qkv[:, (16 + hkv) * d:]
    kc0 = torch.randn(slots, hkv, d, generator=gen).to(torch.bfloat16)
    vc0 = torch.randn(slots, hkv, d, generator=gen).to(torch.bfloat16)
    kc_ref, vc_ref = to_bits(kc0).copy(), to_bits(vc0).copy()
    store_kv_oracle(to_bits(k.contiguous()).reshape(t, hkv, d), to_bits(v.contiguous()).reshape(t, hkv, d), kc_ref, vc_ref, perm.numpy())
    qd = qkv.to(DEV)
    kd, vd = kc0.to(DEV), vc0.to(DEV)
    ops.store_kvcache(qd[:, 16 * d:(16 + hkv) * d].view(t, hkv, d), qd[:, (16 + hkv) * d:].view(t, hkv, d), kd, vd, perm.to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(to_bits(kd), kc_ref) and np.array_equal(to_bits(vd), vc_ref)
    # empty input is a no-op
    ops.store_kvcache(qd[:0, :hkv * d].view(0, hkv, d), qd[:0, :hkv * d].view(0, hkv, d), kd, vd, perm[:0].to(DEV))


def test_rmsnorm_golden(ops):
    g = golden_npz("g5_layers.npz")
    for tag in ("h1024", "h4096", "hd128"):
        w, x, r = (from_bits(g[f"rms_{tag}_{n}"]).to(DEV) for n in ("w", "x", "res"))
        y = ops.rmsnorm(x, w, 1e-6)
        y2, r2 = ops.rmsnorm(x, w, 1e-6, r)
        torch.cuda.synchronize()
        assert bf16_ulp_diff(y, from_bits(g[f"rms_{tag}_y"])) <= 1
        assert bf16_ulp_diff(y2, from_bits(g[f"rms_{tag}_y_res"])) <= 1
        assert np.array_equal(to_bits(r2), g[f"rms_{tag}_res_out"])  # one fp32 add, one rounding: exact
        frac = (to_bits(y) == g[f"rms_{tag}_y"]).mean()
        assert frac > 0.98, frac


@pytest.mark.parametrize("hidden", [128, 1024, 2560, 5120, 8192])
def test_rmsnorm_vs_oracle_sizes(ops, hidden):
    from oracle.layers_oracle import rmsnorm_oracle

    gen = torch.Generator().manual_seed(hidden)
    x = (torch.randn(37, hidden, generator=gen) * 2).to(torch.bfloat16)
    r = torch.randn(37, hidden, generator=gen).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(hidden, generator=gen)).to(torch.bfloat16)
    y_ref, r_ref = rmsnorm_oracle(x, w, 1e-6, r)
    y, ro = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6, r.to(DEV))
    assert bf16_ulp_diff(y, y_ref) <= 1 and np.array_equal(to_bits(ro), to_bits(r_ref))
    # fp32 weight and a row-strided input view
    big = torch.zeros(37, hidden + 64, dtype=torch.bfloat16)
    big[:, :hidden] = x
    y3 = ops.rmsnorm(big.to(DEV)[:, :hidden], w.float().to(DEV), 1e-6)
    assert bf16_ulp_diff(y3, rmsnorm_oracle(x, w.float(), 1e-6)) <= 1
    # few wide rows take a workgroup per row, many rows a wave per row: the second kernel on the same widths
    if hidden > 1024:
        many = 1100
        xm = (torch.randn(many, hidden, generator=gen) * 2).to(torch.bfloat16)
        rm = torch.randn(many, hidden, generator=gen).to(torch.bfloat16)
        ym_ref, rm_ref = rmsnorm_oracle(xm, w, 1e-6, rm)
        ym, rom = ops.rmsnorm(xm.to(DEV), w.to(DEV), 1e-6, rm.to(DEV))
        assert bf16_ulp_diff(ym, ym_ref) <= 1 and np.array_equal(to_bits(rom), to_bits(rm_ref))
        # ... and both kernels give the same rows where both apply
        few, _ = ops.rmsnorm(xm[:37].to(DEV), w.to(DEV), 1e-6, rm[:37].to(DEV))
        assert bf16_ulp_diff(few, ym[:37]) <= 1


def test_rope_golden_bit_exact(ops):
    from oracle.layers_oracle import rope_table_oracle

    g = golden_npz("g5_layers.npz")
    table = rope_table_oracle(128, 40960, 1000000).to(DEV)
    pos = torch.from_numpy(g["rope_pos"]).to(DEV)
    q, k = from_bits(g["rope_q"]).to(DEV), from_bits(g["rope_k"]).to(DEV)
    qo, ko = ops.rope(pos, q, k, table)
    torch.cuda.synchronize()
    assert np.array_equal(to_bits(qo), g["rope_q_out"]) and np.array_equal(to_bits(ko), g["rope_k_out"])


def test_silu_mul_golden(ops):
    g = golden_npz("g5_layers.npz")
    y = ops.silu_mul(from_bits(g["silu_x"]).to(DEV))
    assert bf16_ulp_diff(y, from_bits(g["silu_y"])) <= 1
    assert (to_bits(y) == g["silu_y"]).mean() > 0.98


def test_fused_qknorm_rope_store_vs_unfused_oracle(ops):
    from oracle.layers_oracle import rmsnorm_oracle, rope_rows_oracle, rope_table_oracle

    gen = torch.Generator().manual_seed(3)
    t, hq, hkv, d, slots = 19, 16, 8, 128, 64
    qkv = (torch.randn(t, (hq + 2 * hkv) * d, generator=gen) * 1.5).to(torch.bfloat16)
    qw = (1 + 0.2 * torch.randn(d, generator=gen)).to(torch.bfloat16)
    kw = (1 + 0.2 * torch.randn(d, generator=gen)).to(torch.bfloat16)
    pos = torch.randint(0, 4096, (t,), generator=gen)
    slot = torch.randperm(slots, generator=gen)[:t]
    slot[3] = -1
    table = rope_table_oracle(128, 4096, 1000000)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    qn = rmsnorm_oracle(q.reshape(t, hq, d), qw, 1e-6).reshape(t, hq * d)
    kn = rmsnorm_oracle(k.reshape(t, hkv, d), kw, 1e-6).reshape(t, hkv * d)
    q_ref = rope_rows_oracle(qn, table[pos], d)
    k_ref = rope_rows_oracle(kn, table[pos], d)
    kc = torch.zeros(slots, hkv, d, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros(slots, hkv, d, dtype=torch.bfloat16, device=DEV)
    q_out = ops.qknorm_rope_store(qkv.to(DEV), pos.to(DEV), table.to(DEV), qw.to(DEV), kw.to(DEV), 1e-6, slot.to(DEV), kc, vc, hq, hkv)
    torch.cuda.synchronize()
    # a 1-ulp difference after the norm passes through the rotation, whose differences can cancel: the
    # bound is absolute, 2 bf16 ulps of the operands' magnitude (2^-7 * max|x| of the tensor), not of the result
    def close(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
    assert close(q_out, q_ref)
    live = slot >= 0
    assert close(kc.cpu()[slot[live]].reshape(-1, hkv * d), k_ref[live])
    assert np.array_equal(to_bits(vc.cpu()[slot[live]].reshape(-1, hkv * d)), to_bits(v[live].contiguous()))
    untouched = torch.ones(slots, dtype=torch.bool)
    untouched[slot[live]] = False
    assert not kc.cpu()[untouched].any() and not vc.cpu()[untouched].any()
    assert (to_bits(q_out) == to_bits(q_ref)).mean() > 0.97


# -----------------------------------------------------------------------------------------------
def _random_cache(gen, slots, hkv, d=128):
    kc = torch.randn(slots, hkv, d, generator=gen).to(DTYPE)
    vc = torch.randn(slots, hkv, d, generator=gen).to(DTYPE)
    return kc, vc


def _decode_case(ops, seed, lens, hq, hkv, slots=None, strided_table=False):
    from oracle.attention_oracle import paged_attention_oracle

    gen = torch.Generator().manual_seed(seed)
    b = len(lens)
    total = sum(lens)
    slots = slots or max(64, 2 * total)
    kc, vc = _random_cache(gen, slots, hkv)
    perm = torch.randperm(slots, generator=gen)[:total].to(torch.int32)
    q = torch.randn(b, hq, 128, generator=gen).to(DTYPE)
    indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    ref = paged_attention_oracle(q.float(), kc, vc, np.arange(b + 1), indptr, perm.numpy(), 128 ** -0.5)
    ws = ops.DecodeWorkspace(b, hq, hkv, DEV)
    if strided_table:   # rows of a 2-D table instead of a packed CSR list
        cap = max(lens) + 5
        table = torch.full((b, cap), -7, dtype=torch.int32)
        for i in range(b):
            table[i, :lens[i]] = perm[indptr[i]:indptr[i + 1]]
        kv_indices = table.flatten().to(DEV)
        kv_start = (torch.arange(b, dtype=torch.int64) * cap).to(DEV)
    else:
        kv_indices = perm.to(DEV)
        kv_start = torch.from_numpy(indptr[:-1].astype(np.int64)).to(DEV)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=DEV)
    ops.decode_plan(kv_len, kv_start, ws)
    out = ops.paged_decode(q.to(DEV).view(b, hq * 128), kc.to(DEV), vc.to(DEV), kv_indices, b, ws, 128 ** -0.5)
    torch.cuda.synchronize()
    attn_close(out.view(b, hq, 128), ref, p_bf16_vmax=decode_p_rounding(hq, hkv, vc))
    return ws


# (16, 8) Qwen3-0.6B; (32, 8) Qwen3-8B on one rank; (4, 1) Qwen3-8B over 8 ranks; (16, 2) Qwen3-32B over 4 ranks; (64, 8)
# Qwen3-32B on one rank; the rest: Qwen3-0.6B over 2, 4, 8 ranks and other ratios
HEAD_SHAPES = [(16, 8), (32, 8), (16, 2), (64, 8), (4, 1), (8, 2), (8, 1), (2, 2), (8, 4), (4, 2), (2, 1)]


@pytest.mark.parametrize("hq,hkv", HEAD_SHAPES)
def test_decode_attention_small(ops, hq, hkv):
    _decode_case(ops, 10 + hq, [1, 2, 31, 32, 33, 127, 128, 129, 300], hq, hkv)


@pytest.mark.parametrize("hq,hkv", DECODE_KERNEL_SHAPES)
def test_decode_attention_split_sequences(ops, hq, hkv):
    # one long sequence alone: the plan must cut it into many chunks and the merge kernel must run
    ws = _decode_case(ops, 21, [4000], hq, hkv)
    assert int(ws.plan[0]) > 1
    _decode_case(ops, 22, [3000, 5, 700, 1], hq, hkv, strided_table=True)


@pytest.mark.parametrize("hq,hkv", DECODE_KERNEL_SHAPES)
def test_decode_attention_bench_like_batch(ops, hq, hkv):
    rng = np.random.default_rng(5)
    lens = rng.integers(107, 1100, size=123).tolist()
    _decode_case(ops, 23, lens, hq, hkv)


@pytest.mark.parametrize("hq,hkv", DECODE_KERNEL_SHAPES)
def test_decode_zero_length_row_gives_zeros(ops, hq, hkv):
    gen = torch.Generator().manual_seed(2)
    kc, vc = _random_cache(gen, 64, hkv)
    q = torch.randn(2, hq * 128, generator=gen).to(DTYPE).to(DEV)
    ws = ops.DecodeWorkspace(2, hq, hkv, DEV)
    kv_len = torch.tensor([0, 3], dtype=torch.int32, device=DEV)
    kv_start = torch.tensor([0, 2], dtype=torch.int64, device=DEV)
    ops.decode_plan(kv_len, kv_start, ws)
    out = ops.paged_decode(q, kc.to(DEV), vc.to(DEV), torch.arange(8, dtype=torch.int32, device=DEV), 2, ws, 0.1)
    assert not out[0].any() and out[1].any()


def test_matrix_core_decode_kernel_at_every_head_ratio():
    """NVX_DEC_RING=1 sends every head ratio to the LDS-ring + MFMA decode kernel (by default only ratios >= 8 go
    there); the switch is read once per process, so the check runs in a process of its own: plain and raw-projection
    entry points against the oracle for ratios 1, 2, 4, 8, with split sequences, ragged lengths and a 2-D table."""
    import os
    import subprocess
    import sys

    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_kernels as t
from nano_vllm_x_amd import _lib, ops
_lib.load()
for hq, hkv in [(2, 2), (4, 2), (16, 4), (16, 2), (64, 8), (4, 1)]:
    t.decode_p_rounding = lambda hq, hkv, vc: float(vc.float().abs().max())     # every ratio rounds P to bf16 here
    t._decode_case(ops, 50 + hq, [1, 2, 15, 16, 17, 31, 33, 128, 129, 300, 1000], hq, hkv)
    t._decode_case(ops, 60 + hq, [3000, 5, 700, 1], hq, hkv, strided_table=True)
    t.test_decode_attention_from_raw_qkv_matches_the_two_kernel_path(ops, hq, hkv)
print("ring-ok")
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, NVX_DEC_RING="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ring-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# -----------------------------------------------------------------------------------------------
def _prefill_case(ops, seed, q_lens, kv_lens, hq, hkv, check_oracle=True):
    from oracle.attention_oracle import paged_attention_oracle

    gen = torch.Generator().manual_seed(seed)
    b = len(q_lens)
    t, total = sum(q_lens), sum(kv_lens)
    slots = max(64, 2 * total)
    kc, vc = _random_cache(gen, slots, hkv)
    perm = torch.randperm(slots, generator=gen)[:total].to(torch.int32)
    q = torch.randn(t, hq, 128, generator=gen).to(DTYPE)
    qo = np.concatenate(([0], np.cumsum(q_lens))).astype(np.int32)
    kp = np.concatenate(([0], np.cumsum(kv_lens))).astype(np.int32)
    qo_d, kp_d = torch.from_numpy(qo).to(DEV), torch.from_numpy(kp).to(DEV)
    plan = ops.prefill_plan_ws(b, DEV)
    ops.prefill_plan(qo_d, plan)
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    out = ops.paged_prefill(q.to(DEV).view(t, hq * 128), kcd, vcd, qo_d, kp_d, perm.to(DEV), plan, hq, hkv, 128 ** -0.5)
    torch.cuda.synchronize()
    if check_oracle:
        ref = paged_attention_oracle(q.float(), kc, vc, qo, kp, perm.numpy(), 128 ** -0.5)
        attn_close(out.view(t, hq, 128), ref, p_bf16_vmax=float(vc.float().abs().max()))
    # independent on-device check: every query row as its own decode row over its visible keys
    row_start, row_len = [], []
    for i in range(b):
        off = kv_lens[i] - q_lens[i]
        for j in range(q_lens[i]):
            row_start.append(int(kp[i]))
            row_len.append(j + off + 1)
    ws = ops.DecodeWorkspace(t, hq, hkv, DEV)
    kv_len = torch.tensor(row_len, dtype=torch.int32, device=DEV)
    ops.decode_plan(kv_len, torch.tensor(row_start, dtype=torch.int64, device=DEV), ws)
    out2 = ops.paged_decode(q.to(DEV).view(t, hq * 128), kcd, vcd, perm.to(DEV), t, ws, 128 ** -0.5)
    torch.cuda.synchronize()
    # two 16-bit outputs of the same fp32 quantity: 1e-3 + one rounding each + the 16-bit probabilities of the MFMA path
    d = (out.float() - out2.float()).abs()
    tol = 2e-3 * torch.clamp(out2.float().abs(), min=1.0) + out2.float().abs() * 2 * ULP() + float(vc.float().abs().max()) * ULP() / 2
    assert not (d > tol).any(), f"prefill vs per-row decode: max diff {d.max():.3e}"


@pytest.mark.parametrize("hq,hkv", HEAD_SHAPES)
def test_prefill_attention_small(ops, hq, hkv):
    q_lens = [1, 5, 31, 32, 33, 64, 65, 130, 200]
    _prefill_case(ops, 30 + hq, q_lens, q_lens, hq, hkv)


def test_prefill_attention_with_cached_prefix(ops):
    # queries are the LAST q_len positions of kv_len keys (attention.py:67-81 with cached_kv_len > 0)
    _prefill_case(ops, 41, [3, 40, 1, 100], [50, 40, 300, 164], 16, 8)


def test_prefill_attention_bench_like(ops):
    rng = np.random.default_rng(7)
    lens = rng.integers(100, 1025, size=24).tolist()
    # against the fp32 oracle as well as the per-row decode kernel: 24 sequences of the bench's prompt lengths
    _prefill_case(ops, 42, lens, lens, 16, 8)


def test_prefill_many_short_sequences(ops):
    """More sequences than threads in a workgroup (the owner lookup walks the plan table in several
    passes), empty sequences in between, queries shorter than their key lists."""
    rng = np.random.default_rng(7)
    q_lens = [int(x) for x in rng.integers(0, 40, size=700)]
    q_lens[0], q_lens[-1] = 0, 0
    kv_lens = [q + int(x) if q else 0 for q, x in zip(q_lens, rng.integers(0, 30, size=700))]
    _prefill_case(ops, 44, q_lens, kv_lens, 16, 8)
    _prefill_case(ops, 45, q_lens[:300], kv_lens[:300], 8, 2)


def test_prefill_matches_oracle_long_rows(ops):
    _prefill_case(ops, 43, [1024, 517], [1024, 517], 16, 8)


# -----------------------------------------------------------------------------------------------
def test_sampler_probs_golden(ops):
    g = golden_npz("g6_sampler.npz")
    logits = from_bits(g["logits"]).to(DEV)
    args = [torch.from_numpy(g[n]).to(DEV) for n in ("temps", "top_k", "top_p", "min_p")]
    probs = ops.sampler_probs(logits, args[0], args[1], args[2], args[3]).cpu().numpy()
    assert np.array_equal(probs > 0, g["keep_after_minp"])
    np.testing.assert_allclose(probs, g["probs"], rtol=2e-5, atol=1e-9)
    # fp32 logits input path
    probs32 = ops.sampler_probs(logits.float(), args[0], args[1], args[2], args[3]).cpu().numpy()
    np.testing.assert_allclose(probs32, g["probs"], rtol=2e-5, atol=1e-9)


def test_sampler_probs_full_vocab_vs_oracle(ops):
    from oracle.layers_oracle import sampler_masks_oracle

    gen = torch.Generator().manual_seed(9)
    v = 151936
    logits = (torch.randn(5, v, generator=gen) * 3).to(torch.bfloat16)
    temps = np.array([0.6, 1.0, 0.9, 0.7, 1.2], dtype=np.float32)
    top_k = np.array([v, 40, v, 1000, 3], dtype=np.int64)
    top_p = np.array([1.0, 1.0, 0.8, 0.95, 1.0], dtype=np.float32)
    min_p = np.array([0.0, 0.0, 0.0, 0.02, 0.0], dtype=np.float32)
    _, keep, probs_ref = sampler_masks_oracle(logits.float().numpy(), temps, top_k, top_p, min_p)
    probs = ops.sampler_probs(logits.to(DEV), *(torch.from_numpy(a).to(DEV) for a in (temps, top_k, top_p, min_p))).cpu().numpy()
    got_keep = probs > 0
    # the top-p cut compares a 150k-term fp32 running sum with 1 - p: tokens within 1e-4 of the cut may differ
    diff = got_keep != keep
    assert diff.sum(axis=1).max() <= 64, diff.sum(axis=1)
    same = ~diff.any(axis=1)
    assert same[[0, 1, 4]].all()     # rows without top-p must match exactly
    np.testing.assert_allclose(probs[same], probs_ref[same], rtol=5e-5, atol=1e-9)


def test_sampler_draws_follow_the_distribution(ops):
    gen = torch.Generator().manual_seed(11)
    v, n = 64, 20000
    row = (torch.randn(v, generator=gen) * 1.5).to(torch.bfloat16)
    logits = row.repeat(n, 1).to(DEV)
    ones = torch.ones(n, device=DEV)
    t = ones * 0.8
    ids = ops.sample(logits, t, torch.full((n,), v, dtype=torch.int64, device=DEV), ones, ones * 0, seed=123, offset=7)
    ids2 = ops.sample(logits, t, torch.full((n,), v, dtype=torch.int64, device=DEV), ones, ones * 0, seed=123, offset=7)
    assert torch.equal(ids, ids2)                      # reproducible for a given (seed, offset)
    ids3 = ops.sample(logits, t, torch.full((n,), v, dtype=torch.int64, device=DEV), ones, ones * 0, seed=123, offset=8)
    assert not torch.equal(ids, ids3)
    p = torch.softmax(row.float() / 0.8, dim=-1).numpy()
    counts = np.bincount(ids.cpu().numpy(), minlength=v).astype(np.float64)
    chi2 = ((counts - n * p) ** 2 / (n * p)).sum()
    assert chi2 < 130, chi2          # 63 degrees of freedom: P(chi2 > 130) < 1e-6
    # filtered rows: draws stay inside the kept set, top_k = 1 is the argmax
    ids_k = ops.sample(logits[:4000], t[:4000], torch.full((4000,), 5, dtype=torch.int64, device=DEV), ones[:4000], ones[:4000] * 0, 1, 0)
    top5 = set(torch.topk(row.float(), 5).indices.tolist())
    assert set(ids_k.cpu().tolist()) <= top5 and len(set(ids_k.cpu().tolist())) > 1
    ids_1 = ops.sample(logits[:16], t[:16], torch.ones(16, dtype=torch.int64, device=DEV), ones[:16], ones[:16] * 0, 1, 0)
    assert (ids_1 == int(torch.argmax(row.float()))).all()


# -----------------------------------------------------------------------------------------------
def test_block_table_kernels(ops):
    rows, cap, max_bs = 6, 16, 8
    table = torch.full((rows, cap), -1, dtype=torch.int32, device=DEV)
    kv = torch.tensor([10, 11, 12, 20, 21, 30], dtype=torch.int32, device=DEV)
    indptr = torch.tensor([0, 3, 5, 6], dtype=torch.int32, device=DEV)
    ops.block_table_write_rows(kv, indptr, torch.tensor([4, 0, 2], dtype=torch.int32, device=DEV), table)
    want = torch.full((rows, cap), -1, dtype=torch.int32)
    want[4, :3] = torch.tensor([10, 11, 12]); want[0, :2] = torch.tensor([20, 21]); want[2, :1] = 30
    assert torch.equal(table.cpu(), want)
    step = torch.zeros(4 + 4 * max_bs, dtype=torch.int32)
    step[0] = 3
    step[4:7] = torch.tensor([4, 0, 2])                       # row
    step[4 + max_bs:7 + max_bs] = torch.tensor([3, 2, 1])     # pos
    step[4 + 2 * max_bs:7 + 2 * max_bs] = torch.tensor([13, 22, 31])  # slot
    step[4 + 3 * max_bs:7 + 3 * max_bs] = torch.tensor([7, 8, 9])     # token
    kv_start = torch.zeros(max_bs, dtype=torch.int64, device=DEV)
    kv_len = torch.zeros(max_bs, dtype=torch.int32, device=DEV)
    out_slots, positions, input_ids = (torch.zeros(max_bs, dtype=torch.int64, device=DEV) for _ in range(3))
    pad_start = 5 * cap
    ops.decode_step_setup(step.to(DEV), max_bs, 4, table, pad_start, kv_start, kv_len, out_slots, positions, input_ids)
    want[4, 3] = 13; want[0, 2] = 22; want[2, 1] = 31
    assert torch.equal(table.cpu(), want)
    assert kv_start[:4].tolist() == [4 * cap, 0, 2 * cap, pad_start]
    assert kv_len[:4].tolist() == [4, 3, 2, 1]
    assert out_slots[:4].tolist() == [13, 22, 31, -1]          # padded row: scatter skipped (attention.py:263-265)
    assert positions[:4].tolist() == [3, 2, 1, 0] and input_ids[:4].tolist() == [7, 8, 9, 0]
    # a negative token -1-j stands for the id the step before sampled for its row j, still on the device
    step[4 + 3 * max_bs:7 + 3 * max_bs] = torch.tensor([-1, 8, -3])
    prev = torch.arange(100, 100 + max_bs, dtype=torch.int64, device=DEV)
    ops.decode_step_setup(step.to(DEV), max_bs, 4, table, pad_start, kv_start, kv_len, out_slots, positions, input_ids, prev_ids=prev)
    torch.cuda.synchronize()
    assert input_ids[:4].tolist() == [100, 8, 102, 0]


def test_bad_arguments_raise(ops):
    x = torch.zeros(4, 100, dtype=torch.bfloat16, device=DEV)   # hidden not a multiple of 8
    with pytest.raises(RuntimeError):
        ops.rmsnorm(x, torch.ones(100, dtype=torch.bfloat16, device=DEV), 1e-6)
    with pytest.raises(RuntimeError):
        ops.silu_mul(torch.zeros(4, 10, dtype=torch.bfloat16))   # CPU tensor: no fallback


# -----------------------------------------------------------------------------------------------
def test_torch_ops_namespace_calls_the_same_kernels(ops):
    import nano_vllm_x_amd.torch_ops  # noqa: F401  (registers torch.ops.nvx.*)

    gen = torch.Generator().manual_seed(4)
    x = torch.randn(7, 512, generator=gen).to(torch.bfloat16).to(DEV)
    assert torch.equal(torch.ops.nvx.silu_mul(x), ops.silu_mul(x))
    w = torch.randn(512, generator=gen).to(torch.bfloat16).to(DEV)
    assert torch.equal(torch.ops.nvx.rmsnorm(x, w, 1e-6), ops.rmsnorm(x, w, 1e-6))
    y0, r0 = ops.rmsnorm(x, w, 1e-6, residual=x.clone())
    y1, r1 = torch.ops.nvx.rmsnorm_residual(x, x.clone(), w, 1e-6)
    assert torch.equal(y0, y1) and torch.equal(r0, r1)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.nvx.silu_mul(x.cpu())   # no CPU kernel is registered


@pytest.mark.parametrize("v,f32", [(151936, False), (151936, True), (1000, False), (50257, False)])
def test_sharded_sampling_draws_the_token_of_the_gathered_row(ops, v, f32):
    """Tensor-parallel ranks draw over their own shards of the (64-padded) vocabulary and exchange one 64-bit key per
    row; the largest key must be the token nvx_sample draws from the whole row -- same noise per token, same tie
    rule -- for sampled rows at several temperatures and for a greedy row, with 2, 4 and 8 shards."""
    import nano_vllm_x_amd.torch_ops  # noqa: F401  (registers torch.ops.nvx.*)

    gen = torch.Generator().manual_seed(31)
    b = 37
    padded = (v + 63) // 64 * 64
    full = torch.zeros(b, padded)
    full[:, :v] = torch.randn(b, v, generator=gen) * 2
    full[5, :v] = 0.0                                   # a row of ties: the lowest token id among the best scores wins
    logits = (full if f32 else full.to(torch.bfloat16)).to(DEV)
    t = (0.3 + torch.rand(b, generator=gen)).to(DEV)
    t[3] = 0.0                                          # greedy
    k = torch.full((b,), v, dtype=torch.int64, device=DEV)
    p = torch.ones(b, device=DEV)
    mp = torch.zeros(b, device=DEV)
    for seed, offset in ((77, 5), (3, 1234567)):
        want = ops.sample(logits[:, :v], t, k, p, mp, seed, offset).cpu()
        for tp in (2, 4, 8):
            per = padded // tp
            keys = torch.stack([ops.sample_shard(logits[:, r * per:(r + 1) * per], t, r * per, v, seed, offset) for r in range(tp)])
            got = ops.sample_pick(keys).cpu()
            assert torch.equal(got, want), f"{tp} shards: {(got != want).sum().item()} of {b} rows differ"
            assert int(got.max()) < v
        via_op = torch.ops.nvx.sample_pick(torch.stack([torch.ops.nvx.sample_shard(logits[:, r * (padded // 2):(r + 1) * (padded // 2)], t,
                                                                                    r * (padded // 2), v, seed, offset) for r in range(2)]))
        assert torch.equal(via_op.cpu(), want)


def test_sampler_sliced_rows_draw_the_same_tokens(ops):
    """With a workspace, unfiltered rows are cut into slices over several workgroups; the draw must be the
    token the one-workgroup path draws, and filtered rows in the same batch must be unaffected."""
    from nano_vllm_x_amd import _lib

    gen = torch.Generator().manual_seed(12)
    v, b = 151936, 7
    logits = (torch.randn(b, v, generator=gen) * 2).to(torch.bfloat16).to(DEV)
    t = torch.full((b,), 0.6, device=DEV)
    k = torch.full((b,), v, dtype=torch.int64, device=DEV)
    p = torch.ones(b, device=DEV)
    mp = torch.zeros(b, device=DEV)
    k[2], p[4] = 50, 0.9          # two filtered rows among unfiltered ones
    lib = _lib.load()

    def run(ws):
        out = torch.full((b,), -1, dtype=torch.int64, device=DEV)
        rc = lib.nvx_sample(logits.data_ptr(), 0, logits.stride(0), t.data_ptr(), k.data_ptr(), p.data_ptr(), mp.data_ptr(), b, v,
                            77, 5, out.data_ptr(), ws.data_ptr() if ws is not None else None, ws.numel() * 8 if ws is not None else 0,
                            torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.nvx_last_error()
        torch.cuda.synchronize()
        return out.cpu()

    ws = torch.empty(int(lib.nvx_sample_ws_bytes(b)) // 8, dtype=torch.int64, device=DEV)
    whole, sliced = run(None), run(ws)
    assert torch.equal(whole, sliced)
    assert ((whole >= 0) & (whole < v)).all()
    # the top-k row drew one of its 50 best tokens
    top50 = torch.topk(logits[2].float(), 50).indices.cpu()
    assert whole[2] in top50
    # a workspace that is too small is refused, not overrun
    small = torch.empty(4, dtype=torch.int64, device=DEV)
    rc = lib.nvx_sample(logits.data_ptr(), 0, logits.stride(0), t.data_ptr(), k.data_ptr(), p.data_ptr(), mp.data_ptr(), b, v, 77, 5,
                        torch.empty(b, dtype=torch.int64, device=DEV).data_ptr(), small.data_ptr(), small.numel() * 8,
                        torch.cuda.current_stream().cuda_stream)
    assert rc == -4


# -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [1, 5, 16, 29, 64, 123, 256])
@pytest.mark.parametrize("n,k", [(4096, 1024), (1024, 2048), (1024, 3072), (48, 512), (2048, 1024), (512, 1024), (1024, 1536),
                                 (1024, 256)])
def test_linear_decode_matches_fp32_matmul(ops, m, n, k):
    """fp32-accumulated MFMA product against torch fp32 matmul of the same bf16 operands: the only
    difference is summation order, so |err| <= 2^-8 |ref| (output rounding) + 1e-3."""
    gen = torch.Generator().manual_seed(m * 7 + n + k)
    x = torch.randn(m, k, generator=gen).to(torch.bfloat16).to(DEV)
    w = (torch.randn(n, k, generator=gen) * k ** -0.5).to(torch.bfloat16).to(DEV)
    want = x.float() @ w.float().t()
    if k in (256, 512, 768, 1024, 1536):    # one workgroup can hold the whole K slice: direct bf16 output
        got = ops.linear_decode(x, w)
        tol = want.abs() * 2.0 ** -8 + 1e-3
        assert ((got.float() - want).abs() <= tol).all(), (got.float() - want).abs().max()
    # split-K partials + the RMSNorm that sums them == RMSNorm of the unsplit projection
    s = ops.linear_decode_splits(n, k)
    assert s >= 1 or k == 256
    if s:
        parts = ops.linear_decode(x, w, mode=2, k_splits=s)
        assert parts.shape == (s, m, n)
        torch.testing.assert_close(parts.sum(0), want, rtol=1e-4, atol=1e-4)
        if n % 8 == 0:
            nw = torch.randn(n, generator=gen).to(torch.bfloat16).to(DEV)
            res = torch.randn(m, n, generator=gen).to(torch.bfloat16).to(DEV)
            y_ref, r_ref = ops.rmsnorm(parts.sum(0).to(torch.bfloat16), nw, 1e-6, residual=res)
            y, r = ops.rmsnorm_splitk(parts, nw, 1e-6, residual=res)
            assert bf16_ulp_diff(r, r_ref) <= 1 and bf16_ulp_diff(y, y_ref) <= 2
            assert bf16_ulp_diff(ops.rmsnorm_splitk(parts, nw, 1e-6), ops.rmsnorm(parts.sum(0).to(torch.bfloat16), nw, 1e-6)) <= 2


@pytest.mark.parametrize("m", [37, 49, 64, 123, 200, 256])
def test_linear_decode_strided_input_and_fused_silu(ops, m):
    gen = torch.Generator().manual_seed(8 + m)
    k, inter = 1024, 3072
    wide = torch.randn(m, k + 64, generator=gen).to(torch.bfloat16).to(DEV)
    x = wide[:, :k]                                     # rows 1088 elements apart
    w = (torch.randn(2 * inter, k, generator=gen) * k ** -0.5).to(torch.bfloat16).to(DEV)
    gu = ops.linear_decode(x, w)                        # plain projection ...
    want = ops.silu_mul(gu)                             # ... then the stand-alone activation
    got = ops.linear_decode(x, w, mode=1)
    assert got.shape == (m, inter)
    assert torch.equal(got, want)                       # same accumulation, same roundings: same bits


def test_linear_decode_rejects_bad_shapes(ops):
    x = torch.zeros(300, 64, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(64, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="256 rows"):
        ops.linear_decode(x, w)
    with pytest.raises(RuntimeError, match="512, 768 or 1024"):
        ops.linear_decode(x[:4], w)
    assert ops.linear_decode_splits(64, 96) == 0


# -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (16, 2), (2, 1), (8, 2), (8, 1)])
def test_decode_attention_from_raw_qkv_matches_the_two_kernel_path(ops, hq, hkv):
    """nvx_paged_decode_qkv == nvx_qknorm_rope_store followed by nvx_paged_decode: the pool rows it
    writes are bit-identical, the attention output agrees to the decode tolerance (the new token joins
    the running softmax at another point of the sum)."""
    gen = torch.Generator().manual_seed(31 + hq)
    lens = [1, 2, 17, 128, 129, 300, 1500]          # includes the new token
    b, total = len(lens), sum(lens)
    slots = 2 * total + 64
    kc, vc = _random_cache(gen, slots, hkv)
    perm = torch.randperm(slots - 1, generator=gen)[:total].to(torch.int32) + 1
    indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    new_slots = torch.tensor([int(perm[indptr[i + 1] - 1]) for i in range(b)], dtype=torch.int64)
    new_slots[3] = -1                                # a padding-style row: nothing stored, pool row read instead
    qkv = torch.randn(b, (hq + 2 * hkv) * 128, generator=gen).to(torch.bfloat16).to(DEV)
    pos = torch.tensor([n - 1 for n in lens], dtype=torch.int64, device=DEV)
    from nano_vllm_x_amd.layers.rotary_embedding import build_cos_sin_cache
    table = build_cos_sin_cache(128, 4096, 1e6).to(DEV)
    qw = (1 + 0.1 * torch.randn(128, generator=gen)).to(torch.bfloat16).to(DEV)
    kw = (1 + 0.1 * torch.randn(128, generator=gen)).to(torch.bfloat16).to(DEV)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=DEV)
    kv_start = torch.from_numpy(indptr[:-1]).to(DEV)
    idx = perm.to(DEV)
    ns = new_slots.to(DEV)

    kc1, vc1 = kc.clone().to(DEV), vc.clone().to(DEV)
    ws1 = ops.DecodeWorkspace(b, hq, hkv, DEV)
    q = ops.qknorm_rope_store(qkv, pos, table, qw, kw, 1e-6, ns, kc1, vc1, hq, hkv)
    ops.decode_plan(kv_len, kv_start, ws1)
    want = ops.paged_decode(q, kc1, vc1, idx, b, ws1, 128 ** -0.5)

    kc2, vc2 = kc.clone().to(DEV), vc.clone().to(DEV)
    ws2 = ops.DecodeWorkspace(b, hq, hkv, DEV)
    ops.decode_plan(kv_len, kv_start, ws2, out_slots=ns)
    got = ops.paged_decode_qkv(qkv, pos, table, qw, kw, 1e-6, kc2, vc2, idx, b, ws2, 128 ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    # (the matrix-core variant rounds P to bf16 against the running maximum of the moment, which the two paths reach in a
    # different order: the new row is reduced last on the fused path, in its place in the slot list on the other)
    attn_close(got.view(b, hq, 128), want.view(b, hq, 128).float(), p_bf16_vmax=decode_p_rounding(hq, hkv, vc))


# -----------------------------------------------------------------------------------------------
# Full-size runs, checked through properties that need no CPU oracle at that size.
def _full_size_decode(ops, perm_seed, lens, hq=16, hkv=8, slots=1_100_000, scale_v=1.0):
    """Logical K/V content is a function of (sequence, position) only; `perm_seed` picks which physical
    slots hold it."""
    b, total = len(lens), int(sum(lens))
    gen = torch.Generator(device=DEV).manual_seed(99)
    k_log = torch.randn(total, hkv, 128, generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    v_log = (torch.randn(total, hkv, 128, generator=gen, device=DEV, dtype=torch.float32) * scale_v).to(torch.bfloat16)
    q = torch.randn(b, hq * 128, generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    pg = torch.Generator().manual_seed(perm_seed)
    perm = torch.randperm(slots, generator=pg)[:total].to(torch.int32).to(DEV)
    kc = torch.zeros(slots, hkv, 128, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    kc[perm.long()] = k_log
    vc[perm.long()] = v_log
    indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    ws = ops.DecodeWorkspace(b, hq, hkv, DEV)
    ops.decode_plan(torch.tensor(lens, dtype=torch.int32, device=DEV), torch.from_numpy(indptr[:-1]).to(DEV), ws)
    out = ops.paged_decode(q, kc, vc, perm, b, ws, 128 ** -0.5)
    torch.cuda.synchronize()
    return out.view(b, hq, 128), v_log, indptr


def test_decode_full_size_is_invariant_to_slot_placement_and_linear_in_v(ops):
    """256 sequences x 4096 keys (the largest decode batch the engine's configuration allows: 1 M KV
    rows per layer): the result may not depend on WHERE the rows live, doubling V must exactly double it
    (power-of-two scaling commutes with every rounding), and every output lies within the range of the
    values it averages."""
    lens = [4096] * 256
    a, v_log, indptr = _full_size_decode(ops, 1, lens)
    b2, _, _ = _full_size_decode(ops, 2, lens)
    assert torch.equal(a, b2), "output depends on the physical slot assignment"
    c, _, _ = _full_size_decode(ops, 1, lens, scale_v=2.0)
    assert torch.equal(c.float(), a.float() * 2.0)
    v = v_log.float().view(256, 4096, 8, 128)
    lo = v.amin(dim=1).repeat_interleave(2, dim=1)       # (seq, q head, dim) bounds from the kv head's values
    hi = v.amax(dim=1).repeat_interleave(2, dim=1)
    af = a.float()
    assert ((af >= lo - 2 ** -6 * lo.abs() - 1e-3) & (af <= hi + 2 ** -6 * hi.abs() + 1e-3)).all()


@pytest.mark.parametrize("hq,hkv", DECODE_KERNEL_SHAPES)
def test_decode_ragged_batch_with_empty_and_single_rows(ops, hq, hkv):
    """Ragged lengths including 0 and 1, and a one-row batch, against the oracle."""
    _decode_case(ops, 41, [1], hq, hkv)
    _decode_case(ops, 42, [3, 1, 257, 1, 64, 1000, 2, 15, 16, 17], hq, hkv)
    gen = torch.Generator().manual_seed(3)
    kc, vc = _random_cache(gen, 64, hkv)
    ws = ops.DecodeWorkspace(3, hq, hkv, DEV)
    q = torch.randn(3, hq * 128, generator=gen).to(DTYPE).to(DEV)
    ops.decode_plan(torch.tensor([0, 0, 0], dtype=torch.int32, device=DEV), torch.zeros(3, dtype=torch.int64, device=DEV), ws)
    out = ops.paged_decode(q, kc.to(DEV), vc.to(DEV), torch.zeros(4, dtype=torch.int32, device=DEV), 3, ws, 0.1)
    assert not out.any(), "sequences with no keys must produce zeros"


def test_prefill_full_size_causality(ops):
    """The bench's 142 827 prompt tokens in one launch: changing the LAST key/value row of every sequence
    may change only the last query's output (causality), bit for bit elsewhere."""
    import bench

    prompts, _ = bench.workload(256)
    lens = [len(p) for p in prompts]
    t_tok = sum(lens)
    hq, hkv = 16, 8
    gen = torch.Generator(device=DEV).manual_seed(5)
    kc = torch.randn(t_tok, hkv, 128, generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    vc = torch.randn(t_tok, hkv, 128, generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn(t_tok, hq * 128, generator=gen, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    indptr_np = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    indptr = torch.from_numpy(indptr_np).to(DEV)
    idx = torch.arange(t_tok, dtype=torch.int32, device=DEV)
    plan = ops.prefill_plan_ws(len(lens), DEV)
    ops.prefill_plan(indptr, plan)
    base = ops.paged_prefill(q, kc, vc, indptr, indptr, idx, plan, hq, hkv, 128 ** -0.5)
    last = torch.from_numpy(indptr_np[1:] - 1).long().to(DEV)
    kc2, vc2 = kc.clone(), vc.clone()
    kc2[last] = -kc2[last]
    vc2[last] = vc2[last] * 3
    changed = ops.paged_prefill(q, kc2, vc2, indptr, indptr, idx, plan, hq, hkv, 128 ** -0.5)
    torch.cuda.synchronize()
    keep = torch.ones(t_tok, dtype=torch.bool, device=DEV)
    keep[last] = False
    assert torch.equal(base[keep], changed[keep]), "a query saw a key that comes after it"
    assert not torch.equal(base[last], changed[last])
    assert torch.isfinite(base.float()).all()


def test_sampler_edge_cases(ops):
    gen = torch.Generator().manual_seed(77)
    v = 151936
    b = 256
    logits = (torch.randn(b, v, generator=gen) * 2).to(torch.bfloat16).to(DEV)
    ones = torch.ones(b, device=DEV)
    full_k = torch.full((b,), v, dtype=torch.int64, device=DEV)
    # top_k = 1 keeps the maximum and every token tied with it (sampler.py:72-75 masks `sorted < k-th value`),
    # so the draw is a token whose logit equals the row maximum
    top1 = ops.sample(logits, ones * 0.7, torch.ones(b, dtype=torch.int64, device=DEV), ones, ones * 0, 1, 2)
    lf = logits.float()
    assert (lf.gather(1, top1.view(-1, 1)).squeeze(1) == lf.max(-1).values).all()
    # a tiny temperature concentrates the draw on the maximum (unfiltered, sliced path; 256 rows)
    cold = ops.sample(logits, ones * 1e-3, full_k, ones, ones * 0, 3, 4)
    assert (lf.gather(1, cold.view(-1, 1)).squeeze(1) == lf.max(-1).values).all()
    # min_p = 1 keeps only the maximum; top_p -> 0 keeps only the maximum
    only_max = ops.sample(logits[:8], ones[:8], full_k[:8], ones[:8], ones[:8], 5, 6)
    assert (lf[:8].gather(1, only_max.view(-1, 1)).squeeze(1) == lf[:8].max(-1).values).all()
    tiny_p = ops.sample(logits[:8], ones[:8], full_k[:8], ones[:8] * 1e-6, ones[:8] * 0, 7, 8)
    assert (lf[:8].gather(1, tiny_p.view(-1, 1)).squeeze(1) == lf[:8].max(-1).values).all()
    # a vocabulary that is not a multiple of 4 (Philox blocks serve 4 tokens) and all-equal logits: uniform draw
    flat = torch.zeros(4000, 1001, dtype=torch.bfloat16, device=DEV)
    n = flat.shape[0]
    draws = ops.sample(flat, torch.ones(n, device=DEV), torch.full((n,), 1001, dtype=torch.int64, device=DEV),
                       torch.ones(n, device=DEV), torch.zeros(n, device=DEV), 9, 10)
    assert int(draws.min()) >= 0 and int(draws.max()) <= 1000
    assert draws.unique().numel() > 900            # 4000 uniform draws over 1001 tokens hit almost all of them
    # same seed and offset: same draws; another offset: other draws
    again = ops.sample(flat, torch.ones(n, device=DEV), torch.full((n,), 1001, dtype=torch.int64, device=DEV),
                       torch.ones(n, device=DEV), torch.zeros(n, device=DEV), 9, 10)
    other = ops.sample(flat, torch.ones(n, device=DEV), torch.full((n,), 1001, dtype=torch.int64, device=DEV),
                       torch.ones(n, device=DEV), torch.zeros(n, device=DEV), 9, 11)
    assert torch.equal(draws, again) and not torch.equal(draws, other)


@pytest.mark.parametrize("hq,hkv", DECODE_KERNEL_SHAPES)
def test_decode_masks_invalid_slot_ids_and_survives_extreme_scores(ops, hq, hkv):
    from oracle.attention_oracle import paged_attention_oracle

    gen = torch.Generator().manual_seed(6)
    slots = 256
    kc, vc = _random_cache(gen, slots, hkv)
    kc = kc * 40            # q.k * scale reaches the hundreds: exp2 of the raw score would overflow
    q = (torch.randn(2, hq, 128, generator=gen) * 4).to(DTYPE)
    valid = [torch.randperm(slots, generator=gen)[:70].to(torch.int32), torch.randperm(slots, generator=gen)[:9].to(torch.int32)]
    # the listed slots of sequence 0 are interleaved with ids outside [0, slots): those are masked keys
    listed0 = torch.stack([valid[0], torch.full((70,), -1, dtype=torch.int32)], 1).flatten()
    listed0[5] = slots + 3
    listed0[7] = 2 ** 31 - 1
    kept0 = torch.tensor([s for s in listed0.tolist() if 0 <= s < slots], dtype=torch.int32)
    flat = torch.cat([listed0, valid[1]]).to(DEV)
    lens = [listed0.numel(), 9]
    ws = ops.DecodeWorkspace(2, hq, hkv, DEV)
    ops.decode_plan(torch.tensor(lens, dtype=torch.int32, device=DEV), torch.tensor([0, lens[0]], dtype=torch.int64, device=DEV), ws)
    out = ops.paged_decode(q.view(2, -1).to(DEV), kc.to(DEV), vc.to(DEV), flat, 2, ws, 128 ** -0.5).view(2, hq, 128)
    ref_idx = torch.cat([kept0, valid[1]]).numpy()
    ref = paged_attention_oracle(q.float(), kc, vc, np.arange(3), np.array([0, kept0.numel(), kept0.numel() + 9]), ref_idx, 128 ** -0.5)
    assert torch.isfinite(out.float()).all()
    attn_close(out, ref, p_bf16_vmax=decode_p_rounding(hq, hkv, vc))


def test_norm_and_rope_edge_rows(ops):
    # all-zero rows normalise to zero (rsqrt(eps) is finite); huge rows do not overflow the fp32 sum of squares
    w = torch.ones(1024, dtype=torch.bfloat16, device=DEV)
    x = torch.zeros(3, 1024, dtype=torch.bfloat16, device=DEV)
    x[1] = 3.0e4
    x[2, 0] = float("nan")
    y = ops.rmsnorm(x, w, 1e-6)
    assert not y[0].any() and torch.allclose(y[1].float(), torch.ones(1024, device=DEV), atol=2 ** -7)
    assert torch.isnan(y[2].float()).all()          # a NaN stays visible
    # positions outside the cos/sin table are clamped, not read out of bounds
    from nano_vllm_x_amd.layers.rotary_embedding import build_cos_sin_cache
    table = build_cos_sin_cache(128, 64, 1e6).to(DEV)
    gen = torch.Generator().manual_seed(2)
    q = torch.randn(3, 2 * 128, generator=gen).to(torch.bfloat16).to(DEV)
    k = torch.randn(3, 128, generator=gen).to(torch.bfloat16).to(DEV)
    pos = torch.tensor([63, 64, 10 ** 9], dtype=torch.int64, device=DEV)
    q1, k1 = ops.rope(pos, q, k, table)
    q2, k2 = ops.rope(torch.tensor([63, 63, 63], dtype=torch.int64, device=DEV), q, k, table)
    assert torch.equal(q1, q2) and torch.equal(k1, k2)


def test_sampler_resident_and_streaming_kernels_agree(ops):
    """Filtered bf16 rows that are 16-byte aligned stay on the compute unit as packed keys; any other row is
    re-read from memory per pass by the generic kernel.  Same thresholds, same tie handling, same draws: the
    masked distributions and the sampled tokens must be identical, bit for bit, whichever kernel serves a row."""
    gen = torch.Generator().manual_seed(21)
    v, b = 151936, 12
    base = (torch.randn(b, v, generator=gen) * 2.5).to(torch.bfloat16)
    base[3] = (torch.randn(v, generator=gen) * 0.05).to(torch.bfloat16)        # flat row: many ties at every level
    base[7, :5000] = base[7, 5000:10000]                                         # duplicated scores
    aligned = base.to(DEV)
    wide = torch.zeros(b, v + 3, dtype=torch.bfloat16, device=DEV)               # row stride 151939: rows not 16-byte aligned
    wide[:, :v] = aligned
    shifted = wide[:, :v]
    assert shifted.stride(0) % 8 != 0
    t = torch.tensor([0.6, 1.0, 0.3, 1.0, 0.8, 1.3, 0.7, 1.0, 0.5, 2.0, 0.9, 1.0], device=DEV)
    k = torch.tensor([v, 50, v, v, 1, 1000, v, 20, v, v, 3, v], dtype=torch.int64, device=DEV)
    p = torch.tensor([0.9, 1.0, 0.5, 0.95, 1.0, 0.8, 0.9, 0.7, 0.999, 0.1, 1.0, 1.0], device=DEV)
    mp = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.01, 0.0, 0.0, 0.2, 0.0, 0.0, 0.05], device=DEV)
    pa = ops.sampler_probs(aligned, t, k, p, mp)
    pb = ops.sampler_probs(shifted, t, k, p, mp)
    assert torch.equal(pa, pb)
    kept = (pa > 0).sum(-1)
    assert int(kept[4]) == 1 and int(kept[10]) <= 3 + 8 and int(kept[1]) >= 50     # top-k keeps ties of the k-th score
    assert torch.allclose(pa.sum(-1), torch.ones(b, device=DEV), atol=1e-4)
    for off in range(6):
        ia = ops.sample(aligned, t, k, p, mp, seed=11, offset=off)
        ib = ops.sample(shifted, t, k, p, mp, seed=11, offset=off)
        assert torch.equal(ia, ib), off
        assert bool((pa[torch.arange(b), ia] > 0).all()), "a token outside the kept set was drawn"
    # against the CPU restatement of the reference's masks
    from oracle.layers_oracle import sampler_masks_oracle
    _, keep2, probs = sampler_masks_oracle(base.float().numpy(), t.cpu().numpy(), k.cpu().numpy(), p.cpu().numpy(), mp.cpu().numpy())
    got_keep = (pa > 0).cpu().numpy()
    # rows without top-p must match exactly; a top-p cut compares a 150k-term fp32 running sum with 1 - p in the
    # reference, so tokens within 1e-4 of the cut may differ (more on the rows built to sit on ties)
    for r in range(b):
        n_diff = int((got_keep[r] != keep2[r]).sum())
        if float(p[r]) >= 1.0:
            assert n_diff == 0, (r, n_diff)
        else:
            assert n_diff <= (64 if r not in (3, 7) else max(64, int(0.002 * keep2[r].sum()))), (r, n_diff)
    for r in (1, 4, 10, 11):
        np.testing.assert_allclose(pa[r].cpu().numpy(), probs[r], rtol=5e-5, atol=1e-9)


def test_sampler_temperature_zero_is_greedy(ops):
    """Extension: a temperature <= 0 (the reference divides by it unguarded) selects the first maximum, whatever
    the filters say, in the draw and in the distribution."""
    gen = torch.Generator().manual_seed(5)
    v, b = 4096, 6
    logits = (torch.randn(b, v, generator=gen) * 2).to(torch.bfloat16).to(DEV)
    logits[2, 100] = logits[2, 7] = logits[2].max() + 1         # a tie: the lower id wins
    t = torch.tensor([0.0, 0.0, 0.0, -1.0, 0.7, 0.0], device=DEV)
    k = torch.tensor([v, 5, v, v, v, 1], dtype=torch.int64, device=DEV)
    p = torch.tensor([1.0, 0.5, 1.0, 0.9, 1.0, 1.0], device=DEV)
    mp = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.3], device=DEV)
    ids = ops.sample(logits, t, k, p, mp, seed=1, offset=1)
    am = logits.float().argmax(-1)
    for r in (0, 1, 2, 3, 5):
        assert int(ids[r]) == int(am[r]), r
    assert int(ids[2]) == 7
    probs = ops.sampler_probs(logits, t, k, p, mp)
    for r in (0, 1, 2, 3, 5):
        assert float(probs[r, am[r]]) == 1.0 and float(probs[r].sum()) == 1.0
    f32 = ops.sample(logits.float(), t, k, p, mp, seed=1, offset=1)
    assert torch.equal(f32[[0, 1, 2, 3, 5]], ids[[0, 1, 2, 3, 5]])


def test_attention_path_custom_ops_match_the_direct_calls(ops):
    """torch.ops.nvx.{decode_plan, paged_decode, paged_decode_noalloc, prefill_plan, paged_prefill, linear_decode,
    rmsnorm_splitk}: operator objects over the same C-ABI launches (workspaces passed as tensors)."""
    import nano_vllm_x_amd.torch_ops  # noqa: F401

    gen = torch.Generator().manual_seed(31)
    hq, hkv, slots = 16, 8, 4096
    lens = np.asarray([700, 33, 1, 1500, 64], dtype=np.int32)
    b = len(lens)
    idx = torch.randperm(slots, generator=gen)[:int(lens.sum())].to(torch.int32).to(DEV)
    kv_len = torch.from_numpy(lens).to(DEV)
    kv_start = torch.from_numpy(np.concatenate(([0], np.cumsum(lens[:-1]))).astype(np.int64)).to(DEV)
    kc = torch.randn(slots, hkv, 128, generator=gen).to(torch.bfloat16).to(DEV)
    vc = torch.randn(slots, hkv, 128, generator=gen).to(torch.bfloat16).to(DEV)
    q = torch.randn(b, hq * 128, generator=gen).to(torch.bfloat16).to(DEV)
    ws = ops.DecodeWorkspace(b, hq, hkv, DEV)
    ops.decode_plan(kv_len, kv_start, ws)
    want = ops.paged_decode(q, kc, vc, idx, b, ws, 128 ** -0.5)
    ws2 = ops.DecodeWorkspace(b, hq, hkv, DEV)
    torch.ops.nvx.decode_plan(kv_len, kv_start, None, hkv, ws2.plan)
    got = torch.ops.nvx.paged_decode(q, kc, vc, idx, ws2.plan, ws2.partial, hq, hkv, 128 ** -0.5)
    assert torch.equal(got, want)
    assert torch.equal(torch.ops.nvx.paged_decode_noalloc(q, kc, vc, idx, kv_len, kv_start, hq, hkv, 128 ** -0.5), want)
    assert ops.DecodeWorkspace.from_tensors(ws2.plan, ws2.partial, hq, hkv).max_batch == b
    # prefill
    q_lens = [40, 130, 7]
    t = sum(q_lens)
    qo = torch.tensor(np.concatenate(([0], np.cumsum(q_lens))), dtype=torch.int32, device=DEV)
    pidx = torch.randperm(slots, generator=gen)[:t].to(torch.int32).to(DEV)
    qp = torch.randn(t, hq * 128, generator=gen).to(torch.bfloat16).to(DEV)
    plan = ops.prefill_plan_ws(len(q_lens), DEV)
    ops.prefill_plan(qo, plan)
    want_p = ops.paged_prefill(qp, kc, vc, qo, qo, pidx, plan, hq, hkv, 128 ** -0.5)
    plan2 = torch.zeros_like(plan)
    torch.ops.nvx.prefill_plan(qo, plan2)
    assert torch.equal(torch.ops.nvx.paged_prefill(qp, kc, vc, qo, qo, pidx, plan2, hq, hkv, 128 ** -0.5), want_p)
    # a K-split projection and the norm that sums it
    x = torch.randn(9, 2048, generator=gen).to(torch.bfloat16).to(DEV)
    w = (torch.randn(1024, 2048, generator=gen) * 0.02).to(torch.bfloat16).to(DEV)
    g = torch.ones(1024, dtype=torch.bfloat16, device=DEV)
    s = ops.linear_decode_splits(1024, 2048)
    parts = torch.ops.nvx.linear_decode(x, w, 2, s)
    assert torch.equal(parts, ops.linear_decode(x, w, mode=2, k_splits=s))
    y0, r0 = ops.rmsnorm_splitk(parts, g, 1e-6, residual=x[:, :1024].contiguous())
    y1, r1 = torch.ops.nvx.rmsnorm_splitk_residual(parts, x[:, :1024].contiguous(), g, 1e-6)
    assert torch.equal(y0, y1) and torch.equal(r0, r1)

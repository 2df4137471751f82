"""The float16 build of the library (entry points `*_f16`, compiled from the same sources with -DNVX_F16) against the
oracle, which restates the reference's eager ops in whatever dtype it is handed (the reference resolves the dtype from
the checkpoint and defaults to float16, nanovllmx/core/model_runner.py:14-23).  The reference's own fixtures are
bfloat16, so what pins the float16 path is the oracle alone; tolerances are the bfloat16 ones with float16's unit in
the last place (2^-11 relative instead of 2^-8)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F16 = torch.float16


@pytest.fixture(scope="module")
def ops():
    from nano_vllm_x_amd import _lib, ops as _ops

    _lib.load()
    return _ops


def f16_ulp_diff(a: torch.Tensor, b: torch.Tensor) -> int:
    """Largest distance, in float16 units in the last place, between two float16 tensors (sign-magnitude bit order)."""
    def key(t):
        i = t.detach().cpu().contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return int((key(a) - key(b)).abs().max())


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy()


def attn_close(got, want_f32, p_rounding_vmax=0.0):
    """|err| <= 1e-3 max(1, |ref|) + one float16 ulp of the output (+ 2^-12 max|V| where the probabilities are rounded to
    float16 for the matrix-core product)."""
    got, want = got.float().cpu(), want_f32.float().cpu()
    tol = 1e-3 * torch.clamp(want.abs(), min=1.0) + want.abs() * 2.0 ** -11 + p_rounding_vmax * 2.0 ** -12 + 1e-6
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} outside tolerance; max err {(got - want).abs().max():.3e}"


@pytest.mark.parametrize("hidden", [128, 1024, 4096, 5120])
def test_rmsnorm_f16_vs_oracle(ops, hidden):
    from oracle.layers_oracle import rmsnorm_oracle

    gen = torch.Generator().manual_seed(hidden)
    x = (torch.randn(37, hidden, generator=gen) * 2).to(F16)
    r = torch.randn(37, hidden, generator=gen).to(F16)
    w = (1 + 0.1 * torch.randn(hidden, generator=gen)).to(F16)
    y_ref, r_ref = rmsnorm_oracle(x, w, 1e-6, r)
    y, ro = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6, r.to(DEV))
    assert y.dtype == F16 and ro.dtype == F16
    assert f16_ulp_diff(y, y_ref) <= 1 and np.array_equal(bits(ro), bits(r_ref))
    y1 = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6)
    assert f16_ulp_diff(y1, rmsnorm_oracle(x, w, 1e-6)) <= 1


def test_rope_f16_bit_exact_vs_oracle(ops):
    from oracle.layers_oracle import rope_rows_oracle, rope_table_oracle

    gen = torch.Generator().manual_seed(7)
    t, hq, hkv = 23, 16, 8
    table = rope_table_oracle(128, 4096, 1000000)
    pos = torch.randint(0, 4096, (t,), generator=gen)
    q = torch.randn(t, hq * 128, generator=gen).to(F16)
    k = torch.randn(t, hkv * 128, generator=gen).to(F16)
    qo, ko = ops.rope(pos.to(DEV), q.to(DEV), k.to(DEV), table.to(DEV))
    assert np.array_equal(bits(qo), bits(rope_rows_oracle(q, table[pos], 128)))
    assert np.array_equal(bits(ko), bits(rope_rows_oracle(k, table[pos], 128)))


def test_silu_mul_f16_vs_oracle(ops):
    from oracle.layers_oracle import silu_mul_oracle

    gen = torch.Generator().manual_seed(9)
    x = (torch.randn(29, 2 * 3072, generator=gen) * 3).to(F16)
    y = ops.silu_mul(x.to(DEV))
    ref = silu_mul_oracle(x)
    # silu is rounded to float16 before the product (as eager arithmetic does); where the fast exponential lands it on the
    # other side of a rounding boundary, that one ulp passes through the product and its own rounding
    assert y.dtype == F16 and f16_ulp_diff(y, ref) <= 2
    assert (bits(y) == bits(ref)).mean() > 0.98


def test_fused_qknorm_rope_store_f16_vs_unfused_oracle(ops):
    from oracle.layers_oracle import rmsnorm_oracle, rope_rows_oracle, rope_table_oracle

    gen = torch.Generator().manual_seed(3)
    t, hq, hkv, d, slots = 19, 16, 8, 128, 64
    qkv = (torch.randn(t, (hq + 2 * hkv) * d, generator=gen) * 1.5).to(F16)
    qw = (1 + 0.2 * torch.randn(d, generator=gen)).to(F16)
    kw = (1 + 0.2 * torch.randn(d, generator=gen)).to(F16)
    pos = torch.randint(0, 4096, (t,), generator=gen)
    slot = torch.randperm(slots, generator=gen)[:t]
    slot[3] = -1
    table = rope_table_oracle(128, 4096, 1000000)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    qn = rmsnorm_oracle(q.reshape(t, hq, d), qw, 1e-6).reshape(t, hq * d)
    kn = rmsnorm_oracle(k.reshape(t, hkv, d), kw, 1e-6).reshape(t, hkv * d)
    q_ref, k_ref = rope_rows_oracle(qn, table[pos], d), rope_rows_oracle(kn, table[pos], d)
    kc = torch.zeros(slots, hkv, d, dtype=F16, device=DEV)
    vc = torch.zeros(slots, hkv, d, dtype=F16, device=DEV)
    q_out = ops.qknorm_rope_store(qkv.to(DEV), pos.to(DEV), table.to(DEV), qw.to(DEV), kw.to(DEV), 1e-6, slot.to(DEV), kc, vc, hq, hkv)
    torch.cuda.synchronize()

    def close(a, b):   # two float16 ulps of the operands' magnitude (a 1-ulp difference after the norm passes through the rotation)
        a, b = a.float().cpu(), b.float().cpu()
        return float((a - b).abs().max()) <= 2.0 ** -10 * float(b.abs().max())
    assert close(q_out, q_ref)
    live = slot >= 0
    assert close(kc.cpu()[slot[live]].reshape(-1, hkv * d), k_ref[live])
    assert np.array_equal(bits(vc.cpu()[slot[live]].reshape(-1, hkv * d)), bits(v[live].contiguous()))


def _cache(gen, slots, hkv):
    return torch.randn(slots, hkv, 128, generator=gen).to(F16), torch.randn(slots, hkv, 128, generator=gen).to(F16)


def _uses_matrix_cores(hq, hkv):
    g = hq // hkv
    return g >= 8 or (g >= 4 and hkv <= 2)


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (16, 2), (4, 1), (2, 2)])
def test_decode_attention_f16_vs_oracle(ops, hq, hkv):
    from oracle.attention_oracle import paged_attention_oracle

    gen = torch.Generator().manual_seed(100 + hq)
    lens = [1, 2, 15, 16, 17, 33, 128, 129, 300, 2000]
    b, total = len(lens), sum(lens)
    slots = 2 * total
    kc, vc = _cache(gen, slots, hkv)
    perm = torch.randperm(slots, generator=gen)[:total].to(torch.int32)
    q = torch.randn(b, hq, 128, generator=gen).to(F16)
    indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    ref = paged_attention_oracle(q.float(), kc, vc, np.arange(b + 1), indptr, perm.numpy(), 128 ** -0.5)
    ws = ops.DecodeWorkspace(b, hq, hkv, DEV)
    ops.decode_plan(torch.tensor(lens, dtype=torch.int32, device=DEV), torch.from_numpy(indptr[:-1].astype(np.int64)).to(DEV), ws)
    out = ops.paged_decode(q.to(DEV).view(b, hq * 128), kc.to(DEV), vc.to(DEV), perm.to(DEV), b, ws, 128 ** -0.5)
    assert out.dtype == F16
    attn_close(out.view(b, hq, 128), ref, float(vc.float().abs().max()) if _uses_matrix_cores(hq, hkv) else 0.0)


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8), (16, 2)])
def test_prefill_attention_f16_vs_oracle(ops, hq, hkv):
    from oracle.attention_oracle import paged_attention_oracle

    gen = torch.Generator().manual_seed(200 + hq)
    q_lens, kv_lens = [1, 5, 64, 65, 200, 33], [1, 5, 64, 65, 200, 133]   # the last one has 100 cached keys
    t, total = sum(q_lens), sum(kv_lens)
    slots = 2 * total
    kc, vc = _cache(gen, slots, hkv)
    perm = torch.randperm(slots, generator=gen)[:total].to(torch.int32)
    q = torch.randn(t, hq, 128, generator=gen).to(F16)
    qo = np.concatenate(([0], np.cumsum(q_lens))).astype(np.int32)
    kp = np.concatenate(([0], np.cumsum(kv_lens))).astype(np.int32)
    ref = paged_attention_oracle(q.float(), kc, vc, qo, kp, perm.numpy(), 128 ** -0.5)
    plan = ops.prefill_plan_ws(len(q_lens), DEV)
    qo_d, kp_d = torch.from_numpy(qo).to(DEV), torch.from_numpy(kp).to(DEV)
    ops.prefill_plan(qo_d, plan)
    out = ops.paged_prefill(q.to(DEV).view(t, hq * 128), kc.to(DEV), vc.to(DEV), qo_d, kp_d, perm.to(DEV), plan, hq, hkv, 128 ** -0.5)
    assert out.dtype == F16
    attn_close(out.view(t, hq, 128), ref, float(vc.float().abs().max()))


@pytest.mark.parametrize("hq,hkv", [(16, 8), (16, 2)])
def test_decode_from_raw_qkv_f16_matches_the_two_kernel_path(ops, hq, hkv):
    from oracle.layers_oracle import rope_table_oracle

    gen = torch.Generator().manual_seed(300 + hq)
    lens = [1, 2, 17, 129, 700]
    b, total = len(lens), sum(lens)
    slots = 2 * total + 8
    kc, vc = _cache(gen, slots, hkv)
    perm = torch.randperm(slots, generator=gen)[:total].to(torch.int32)
    indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    new_slots = perm[torch.from_numpy(indptr[1:] - 1)].to(torch.int64)      # the last listed slot takes the new row
    qkv = (torch.randn(b, (hq + 2 * hkv) * 128, generator=gen) * 1.5).to(F16).to(DEV)
    pos = torch.tensor([n - 1 for n in lens], dtype=torch.int64, device=DEV)
    table = rope_table_oracle(128, 4096, 1000000).to(DEV)
    qw = (1 + 0.2 * torch.randn(128, generator=gen)).to(F16).to(DEV)
    kw = (1 + 0.2 * torch.randn(128, generator=gen)).to(F16).to(DEV)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=DEV)
    kv_start = torch.from_numpy(indptr[:-1]).to(DEV)
    idx, ns = perm.to(DEV), new_slots.to(DEV)
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
    attn_close(got.view(b, hq, 128), want.view(b, hq, 128).float(), float(vc.float().abs().max()) if _uses_matrix_cores(hq, hkv) else 0.0)


@pytest.fixture
def half_cases(monkeypatch):
    """The attention cases of test_gpu_kernels.py, built in float16 (tolerances follow: see ULP() there)."""
    import test_gpu_kernels as t

    monkeypatch.setattr(t, "DTYPE", F16)
    return t


@pytest.mark.parametrize("hq,hkv", [(16, 8), (16, 2)])   # the register kernel and the LDS-ring + MFMA kernel
def test_decode_edge_cases_in_float16(ops, half_cases, hq, hkv):
    """Split sequences (merge kernel), a 2-D table, zero-length and one-key rows, slot ids out of range, scores in the
    hundreds: the cases of the bfloat16 suite, through nvx_paged_decode_f16."""
    half_cases.test_decode_attention_split_sequences(ops, hq, hkv)
    half_cases.test_decode_zero_length_row_gives_zeros(ops, hq, hkv)
    half_cases.test_decode_ragged_batch_with_empty_and_single_rows(ops, hq, hkv)
    half_cases.test_decode_masks_invalid_slot_ids_and_survives_extreme_scores(ops, hq, hkv)


def test_prefill_edge_cases_in_float16(ops, half_cases):
    """Cached prefixes, 700 short and empty sequences (multi-pass owner lookup), rows of a thousand keys: through
    nvx_paged_prefill_f16, against the oracle and against per-row decode."""
    half_cases.test_prefill_attention_with_cached_prefix(ops)
    half_cases.test_prefill_many_short_sequences(ops)
    half_cases.test_prefill_matches_oracle_long_rows(ops)


@pytest.mark.parametrize("m", [1, 29, 64, 200])
def test_decode_projections_f16_vs_fp32_reference(ops, m):
    """nvx_linear_decode_f16 (modes 0, 1, 2) and nvx_rmsnorm_splitk_f16 against fp32 arithmetic on the same float16 values."""
    import torch.nn.functional as F

    gen = torch.Generator().manual_seed(400 + m)
    x = torch.randn(m, 1024, generator=gen).to(F16)
    w = (torch.randn(2048, 1024, generator=gen) * 1024 ** -0.5).to(F16)
    ref = F.linear(x.float(), w.float())
    y0 = ops.linear_decode(x.to(DEV), w.to(DEV), mode=0)
    # fp32-accumulated products of the same float16 operands: what differs is the summation order (an absolute term, seen
    # on results near zero) and the output rounding
    assert y0.dtype == F16 and bool(((y0.float().cpu() - ref).abs() <= ref.abs() * 2.0 ** -11 + 1e-4).all()), (y0.float().cpu() - ref).abs().max()
    gate, up = ref[:, :1024].to(F16).float(), ref[:, 1024:].to(F16).float()
    act_ref = ((gate / (1 + torch.exp(-gate))).to(F16).float() * up).to(F16)
    y1 = ops.linear_decode(x.to(DEV), w.to(DEV), mode=1)
    assert float((y1.float().cpu() - act_ref.float()).abs().max()) <= 2.0 ** -9 * float(act_ref.float().abs().max())
    # K-split projection + the norm that sums its partials
    x2 = torch.randn(m, 2048, generator=gen).to(F16)
    w2 = (torch.randn(1024, 2048, generator=gen) * 2048 ** -0.5).to(F16)
    res = torch.randn(m, 1024, generator=gen).to(F16)
    nw = (1 + 0.1 * torch.randn(1024, generator=gen)).to(F16)
    splits = ops.linear_decode_splits(1024, 2048)
    part = ops.linear_decode(x2.to(DEV), w2.to(DEV), mode=2, k_splits=splits)
    y, ro = ops.rmsnorm_splitk(part, nw.to(DEV), 1e-6, residual=res.to(DEV))
    from oracle.layers_oracle import rmsnorm_oracle
    proj = F.linear(x2.float(), w2.float()).to(F16)
    y_ref, r_ref = rmsnorm_oracle(proj, nw, 1e-6, res)
    assert y.dtype == F16
    # the projection's last bit may differ (summation order), which the residual add and the norm pass on
    assert float((ro.float().cpu() - r_ref.float()).abs().max()) <= 2.0 ** -10 * float(r_ref.float().abs().max())
    assert float((y.float().cpu() - y_ref.float()).abs().max()) <= 2.0 ** -9 * float(y_ref.float().abs().max())


def test_tiny_model_in_float16_against_the_fp32_oracle():
    """End to end in float16: the HIP runner on a float16 checkpoint (prefill + two decode steps of the tiny fixture model)
    against the fp32 CPU oracle on the same float16 weights.  The bound is asserted per logit, relative to the largest
    logit of the step: float16 keeps 11 significant bits where bfloat16 keeps 8, and the error shrinks accordingly."""
    import dataclasses

    from helpers import golden_json
    from model_checks import check_logits
    from nano_vllm_x_amd.core.model_runner import ModelRunner
    from nano_vllm_x_amd.qwen3 import Qwen3Config
    from oracle.qwen3_oracle import cpu_engine_greedy_oracle
    from test_gpu_model import _run_fixture
    from tiny_model import tiny_state_dict

    meta = golden_json("g7_tiny_qwen3.json")
    sd16 = {k: v.to(F16) for k, v in tiny_state_dict(meta["config"]).items()}
    cfg = dataclasses.replace(Qwen3Config.from_dict(meta["config"]), torch_dtype="float16")
    for eager in (True, False):
        r = ModelRunner(model="tiny", max_bs=4, enforce_eager=eager, context_len=64, config=cfg)
        r.load_model(state_dict_iter=sd16.items())
        assert r.dtype == F16 and r.checkpoint_dtype == F16
        r.initialize_kv_cache(64)
        got = _run_fixture(r, meta)
        assert r.kv_cache.k_cache.dtype == F16
        if eager:
            first = got
        else:
            assert r.stats["graph_steps"] == 2
            for step in range(3):
                assert np.array_equal(got[step], first[step]), step
    # the same steps on the CPU in fp32 with the same (float16-valued) weights, teacher-forced with the fixture's tokens
    n = len(meta["prompts"])
    want = cpu_engine_greedy_oracle(meta["config"], {k: v.float() for k, v in sd16.items()}, [list(p) for p in meta["prompts"]], [3] * n, 64, 4, 64,
                                    forced={i: [meta["feed"][0][i], meta["feed"][1][i], 0] for i in range(n)})
    # ... and with the oracle itself computing in float16 (the reference's eager arithmetic in its default dtype; that run
    # is 1.7e-3 of the largest logit away from the fp32 one)
    want16 = cpu_engine_greedy_oracle(meta["config"], sd16, [list(p) for p in meta["prompts"]], [3] * n, 64, 4, 64, dtype=F16,
                                      forced={i: [meta["feed"][0][i], meta["feed"][1][i], 0] for i in range(n)})
    for step in range(3):
        ref = np.stack([want["logits"][i][step].float().numpy() for i in range(n)])
        check_logits("tiny_f16_hip_vs_fp32_cpu", first[step], ref, f"step {step}")
        ref16 = np.stack([want16["logits"][i][step].float().numpy() for i in range(n)])
        check_logits("tiny_f16_hip_vs_f16_cpu", first[step], ref16, f"step {step}")

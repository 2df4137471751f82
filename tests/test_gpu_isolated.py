"""The hand-written path ALONE, end to end, in float16 -- and what "1e-3 per-token logits" can mean for it.

Every model-level comparison elsewhere in the suite mixes two things: the hand-written kernels (RMSNorm, q/k-norm +
rotary + KV scatter, paged prefill/decode attention, SiluAndMul) and the dense projections, which are library GEMMs or
the skinny projection kernels and sum in another order than a CPU matmul does.  Here the projections are taken out of
the comparison: on BOTH sides every projection multiplies in fp32 and rounds once to float16 (NVX_FP32_LINEAR=1 on the
device, `linear_fp32=True` in the oracle), and the LM head gives fp32 logits from the float16 hidden states.  What is
left to differ is the hand-written path.  Three runs of the same teacher-forced history (prefill of six ragged
prompts + nine decode steps replayed from the captured hipGraph):

    hip    the device path
    ref16  oracle.qwen3_oracle in float16: the reference's eager arithmetic in its default dtype
           (nanovllmx/core/model_runner.py:14-23), every op's result rounded to float16, around the call-site semantics
           of its attention (nanovllmx/layers/attention.py:354-373)
    ref32  the same oracle in fp32 on the same (float16-valued) weights: exact arithmetic, for this purpose

Measured on MI355X (tools/isolated_probe.py), three layers, 16/8 heads, 61 440 logits of magnitude <= 4.8:
ref16 vs ref32 6.4e-3 max / 1.43e-3 rms; hip vs ref32 6.2e-3 / 1.44e-3; hip vs ref16 5.5e-3 / 1.18e-3; the op-by-op device
path (no fusion) 5.6e-3 / 1.18e-3.  So a max-norm bound of 1e-3 * max(1, |ref|) per logit is BELOW what the float16
format itself does to this model: the reference's own eager float16 arithmetic is 5.9 x that bound away from exact
arithmetic, and any two float16 implementations that round at different points (a fused q/k-norm + rotary keeps fp32
between the two, an attention kernel rounds its output once) differ by about as much.  No single op exceeds it -- each
is within one unit in the last place or 1e-3 of its own oracle in test_gpu_float16.py / test_gpu_kernels.py; the excess
is those last-place differences carried through three layers of norms and a 512-term LM-head sum.

What IS asserted, because it is what the tolerance is for (the device path must not be a worse float16 implementation
than the reference's): hip is no further from exact arithmetic than ref16 is (rms with 10 % slack, max with 50 %), hip and ref16
are no further apart than either is from exact arithmetic (same slack), the rms difference per logit is within
1.5e-3, every arg-max agrees unless ref32's own top two are closer than the error, and the fraction of the 1e-3 bound
that was reached is recorded (gpurun_out/model_tolerances_observed.json) so a change in it is seen.
"""
import dataclasses

import numpy as np
import pytest
import torch

from model_checks import record

pytestmark = pytest.mark.gpu
F16 = torch.float16
TOL = 1e-3


@pytest.mark.parametrize("hq,hkv", [(16, 8), (32, 8)])
def test_hand_written_path_alone_is_as_close_to_exact_arithmetic_as_eager_float16(hq, hkv, monkeypatch):
    from nano_vllm_x_amd import qwen3
    from nano_vllm_x_amd.core.engine import EngineCore
    from nano_vllm_x_amd.core.model_runner import ModelRunner
    from oracle.qwen3_oracle import cpu_engine_greedy_oracle
    from test_gpu_configs import _forced_run

    monkeypatch.setenv("NVX_FP32_LINEAR", "1")
    cfg = qwen3.Qwen3Config(vocab_size=1024, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                            num_attention_heads=hq, num_key_value_heads=hkv, head_dim=128, max_position_embeddings=512,
                            tie_word_embeddings=False, torch_dtype="float16")
    sd16 = {k: v.to(F16) for k, v in qwen3.random_checkpoint(cfg, 5)}
    rng = np.random.default_rng(hq)
    lens = [3, 17, 40, 64, 65, 150]
    prompts = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in lens]
    n_new, slots, max_bs, ctx = 10, 1024, 8, 256

    want = cpu_engine_greedy_oracle(dataclasses.asdict(cfg), sd16, prompts, [n_new] * len(prompts), slots, max_bs, ctx, dtype=F16,
                                    linear_fp32=True)
    assert want["steps"] == n_new
    forced_cpu = {i: want["streams"][i] for i in range(len(prompts))}
    exact = cpu_engine_greedy_oracle(dataclasses.asdict(cfg), {k: v.float() for k, v in sd16.items()}, prompts, [n_new] * len(prompts),
                                     slots, max_bs, ctx, forced=forced_cpu)

    runner = ModelRunner(model="isolated", max_bs=max_bs, enforce_eager=False, context_len=ctx, config=cfg)
    runner.load_model(state_dict_iter=sd16.items())
    assert qwen3.FP32_LINEAR and runner.dtype == F16
    runner.initialize_kv_cache(slots)
    core = EngineCore(kv_cache_size=slots, max_bs=max_bs, context_len=ctx)
    reqs = [(str(i), p, n_new) for i, p in enumerate(prompts)]
    forced = {str(i): want["streams"][i] for i in range(len(prompts))}
    with torch.inference_mode():
        _toks, logits, _ = _forced_run(core, runner, reqs, forced)
    assert runner.stats["graph_steps"] == n_new - 1 and runner.stats["eager_decode_steps"] == 0, runner.stats

    def stack(r):
        return np.stack([torch.stack(r["logits"][i]).double().numpy() for i in range(len(prompts))])

    hip = np.stack([np.stack(logits[str(i)]).astype(np.float64) for i in range(len(prompts))])
    ref16, ref32 = stack(want), stack(exact)
    assert hip.shape == ref16.shape == ref32.shape == (len(prompts), n_new, cfg.vocab_size)

    def gap(a, b):
        d = np.abs(a - b)
        return float(d.max()), float(np.sqrt((d ** 2).mean())), float((d / (TOL * np.maximum(1.0, np.abs(b)))).max())

    fmt_max, fmt_rms, fmt_frac = gap(ref16, ref32)        # what float16 itself costs: the reference's arithmetic vs exact
    hip_max, hip_rms, hip_frac = gap(hip, ref32)
    d_max, d_rms, d_frac = gap(hip, ref16)
    tag = f"isolated_f16[{hq}/{hkv}]"
    record(tag + "_hip_vs_ref16_max_abs", d_max)
    record(tag + "_hip_vs_ref16_fraction_of_1e-3_bound", d_frac)
    record(tag + "_hip_vs_fp32_max_abs", hip_max)
    record(tag + "_ref16_vs_fp32_max_abs", fmt_max)
    print(f"isolated {hq}/{hkv}: hip vs ref16 max {d_max:.3e} rms {d_rms:.3e} ({d_frac:.2f} x the 1e-3 bound); "
          f"hip vs fp32 max {hip_max:.3e} rms {hip_rms:.3e}; ref16 vs fp32 max {fmt_max:.3e} rms {fmt_rms:.3e} ({fmt_frac:.2f} x)")
    # the rms over 61 440 logits is a stable statistic (10 % slack); the maximum is one sample of a 4.5-sigma tail (50 %)
    slack, slack_max = 1.10, 1.50
    assert hip_max <= slack_max * fmt_max and hip_rms <= slack * fmt_rms, "the device path is further from exact arithmetic than eager float16 is"
    assert d_max <= slack_max * max(fmt_max, hip_max) and d_rms <= slack * max(fmt_rms, hip_rms), \
        "the device path and the float16 oracle are further apart than either is from exact arithmetic"
    assert d_rms <= 1.5e-3
    # arg-max: the same token unless exact arithmetic itself rates the two candidates closer than the float16 error
    flat_h, flat_x = hip.reshape(-1, cfg.vocab_size), ref32.reshape(-1, cfg.vocab_size)
    for r in np.nonzero(flat_h.argmax(-1) != ref16.reshape(-1, cfg.vocab_size).argmax(-1))[0]:
        top2 = np.sort(flat_x[r])[-2:]
        assert top2[1] - top2[0] <= 2 * fmt_max, f"row {r}: arg-max differs although the top two logits are {top2[1] - top2[0]:.3e} apart"
    monkeypatch.delenv("NVX_FP32_LINEAR")
    qwen3.FP32_LINEAR = False

"""The five configurations the headline metric is quoted on, each exercised on the GPU.

  1. Qwen3-0.6B, one rank, 4 requests of 128 prompt + 128 new tokens: the CPU engine (oracle/) and the HIP
     engine (LLM.generate) on the same weights, arg-max sampling, compared step by step.
  2. Qwen3-0.6B, one rank, the 256-request bench workload: bench.py itself; its batches and kernels are covered at
     full size in test_gpu_kernels.py (`*_bench_like*`, `*_full_size_*`) and test_gpu_determinism.py.
  3. Qwen3-8B on one rank with prefix reuse: 32 query / 8 KV heads; a workload sharing a 512-token prefix.
  4. Qwen3-8B over 8 ranks: what ONE rank computes (4 query heads / 1 KV head, hidden 4096, 1536 MLP columns).
  5. Qwen3-32B over 4 ranks x 2 stages: one rank's shapes (16 / 2 heads, hidden 5120, 6400 MLP columns), and a
     4096-token context through chunked prefill.
What the ranks exchange (all-reduce, gather, stage hand-off) is covered in test_gpu_tp.py / test_tp_gloo.py.
"""
import asyncio
import random

import numpy as np
import pytest
import torch

from model_checks import bf16_ulp, bound, check_logits, record

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------
def _forced_run(core, runner, requests, forced=None):
    """Drive scheduler + runner by hand: arg-max tokens (or `forced[seq_id][step]`), logits kept per request."""
    from nano_vllm_x_amd.core.common import ForwardMode, SamplingParams

    for sid, prompt, n in requests:
        core.add_sequence(sid, list(prompt), SamplingParams(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=n))
    toks = {sid: [] for sid, _, _ in requests}
    logits = {sid: [] for sid, _, _ in requests}
    rows = 0
    steps = 0
    while core.has_unfinished_sequences():
        batch = core.scheduler.schedule()
        assert batch is not None
        if batch.forward_mode == ForwardMode.PREFILL:
            rows += sum(len(s.token_ids) - s.cached_kv_len for s in batch.seqs)
        lg = runner.compute_logits(batch).float()
        pick = lg.argmax(dim=-1).tolist()
        lg = lg.cpu().numpy()
        for j, s in enumerate(batch.seqs):
            if s.partial_of is not None:
                continue
            if forced is not None:
                pick[j] = forced[s.seq_id][len(toks[s.seq_id])]
            toks[s.seq_id].append(pick[j])
            logits[s.seq_id].append(lg[j])
        core.update_from_output(batch, pick)
        steps += 1
        assert steps < 5000
    return toks, logits, rows


# ---------------------------------------------------------------------------------------------------
def test_config1_four_requests_128_in_128_out_cpu_engine_vs_hip_engine():
    """Both engines hold the same bf16-representable weights (qwen3.random_checkpoint, seed 0).  The CPU engine
    computes in fp32, so its logits are the reference point; the HIP engine computes in bf16 with fp32
    accumulation.  (a) LLM.generate with top_k = 1 runs freely: every stream must equal the CPU engine's up to
    its first differing token, and that token must be one the CPU logits rate within the logit error bound of
    their maximum (after it the two histories differ and cannot be compared).  (b) The HIP engine is then
    fed the CPU engine's tokens, so that all 4 x 128 steps are compared on identical histories: logits within
    the bound, same arg-max unless the CPU logits' top two are closer than the bound allows to tell apart."""
    from nano_vllm_x_amd.core.common import SamplingParams
    from nano_vllm_x_amd.llm import LLM
    from nano_vllm_x_amd.qwen3 import PRESETS, Qwen3Config, random_checkpoint
    from oracle.qwen3_oracle import cpu_engine_greedy_oracle

    cfg = Qwen3Config(**PRESETS["Qwen3-0.6B"])
    rnd = random.Random(1)
    prompts = [[rnd.randint(0, 10000) for _ in range(128)] for _ in range(4)]
    n_new = 128
    sd = {k: v.to(torch.bfloat16).float() for k, v in random_checkpoint(cfg, 0)}
    ref = cpu_engine_greedy_oracle(cfg.__dict__, sd, prompts, [n_new] * 4, kv_slots=4 * 257 + 8, max_bs=4, context_len=4096)
    del sd
    assert ref["steps"] == n_new and ref["generated"] == 4 * n_new

    llm = LLM("Qwen3-0.6B", max_bs=4, context_len=4096, kv_cache_size=4096, seed=0)

    async def free_run():
        async def one(i):
            sp = SamplingParams(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=n_new)
            return [o.token_id async for o in llm.generate(list(prompts[i]), sp, sequence_id=f"c{i}")]
        return await asyncio.gather(*(one(i) for i in range(4)))

    got = asyncio.run(free_run())
    name = "config1_cpu_fp32_engine_vs_hip_engine"
    lim = bound(name)
    same_to_the_end = 0
    for i in range(4):
        want = ref["streams"][i]
        assert len(got[i]) == n_new
        t = next((k for k in range(n_new) if got[i][k] != want[k]), None)
        if t is None:
            same_to_the_end += 1
            continue
        row = ref["logits"][i][t].numpy()
        gap = float(row.max() - row[got[i][t]])
        assert gap <= 2 * lim * float(np.abs(row).max()), \
            f"request {i} leaves the CPU engine's stream at token {t} for an id the CPU logits rate {gap:.4f} below their maximum"
    stats = llm.engine.model_runner.stats
    assert stats["graph_steps"] > 0 and stats["eager_decode_steps"] == 0

    # (b) identical histories
    eng = llm.engine
    eng.reset()
    reqs = [(f"f{i}", prompts[i], n_new) for i in range(4)]
    forced = {f"f{i}": ref["streams"][i] for i in range(4)}
    with torch.inference_mode():
        _toks, logits, _ = _forced_run(eng, eng.model_runner, reqs, forced)
    worst = 0.0
    for i in range(4):
        a = np.stack(logits[f"f{i}"])
        b = torch.stack(ref["logits"][i]).numpy()
        # ties: the fp32 reference separates logits that bf16 arithmetic cannot; allow what the bound allows
        scale = float(np.abs(b).max())
        worst = max(worst, check_logits(name, a, b, f"request {i}", tie_ulps=2 * lim * scale / bf16_ulp(scale)))
    record("config1_free_running_streams_equal_to_the_cpu_engines_of_4", same_to_the_end)
    print(f"config 1: {same_to_the_end}/4 free-running streams identical to the CPU engine's over {n_new} tokens; "
          f"forced: worst max|dlogit|/max|logit| = {worst:.3e}")
    llm.shutdown()


# ---------------------------------------------------------------------------------------------------
RANK_SHAPES = {
    # name: (hidden, per-rank MLP columns, per-rank query heads, per-rank KV heads)
    "config3_qwen3_8b_tp1": (4096, 12288, 32, 8),
    "config4_qwen3_8b_tp8": (4096, 12288 // 8, 32 // 8, 1),
    "config5_qwen3_32b_tp4_stage": (5120, 25600 // 4, 64 // 4, 8 // 4),
}


@pytest.mark.parametrize("name", list(RANK_SHAPES))
def test_one_rank_of_the_larger_configurations_hip_path_vs_library_path(name):
    """Prefill + three decode steps of a two-layer model with exactly the shapes one rank of the configuration
    holds, HIP path (fused q/k-norm + rotary + scatter, paged attention, skinny projections where they apply)
    against the op-by-op path with library GEMMs."""
    from nano_vllm_x_amd import qwen3
    from test_gpu_model import _fused_vs_library

    hidden, inter, hq, hkv = RANK_SHAPES[name]
    cfg = qwen3.Qwen3Config(vocab_size=2048, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=2,
                            num_attention_heads=hq, num_key_value_heads=hkv, head_dim=128, max_position_embeddings=256,
                            tie_word_embeddings=False)
    _fused_vs_library(cfg, name)


def test_config3_shared_512_token_prefix_with_prefix_reuse_at_8b_head_geometry():
    """32 query / 8 KV heads (Qwen3-8B on one rank), six requests behind a 512-token system prompt that a first
    request has left in the prefix tree.  With prefix_reuse the prefill computes only the suffixes (queries
    right-aligned to 512 cached keys); fed the same tokens, its logits must match the flag-off run's at every
    step, and exactly 6 x 512 prompt rows must not have been computed."""
    from nano_vllm_x_amd.core.engine import EngineCore
    from nano_vllm_x_amd.core.model_runner import ModelRunner
    from nano_vllm_x_amd.qwen3 import Qwen3Config

    cfg = Qwen3Config(vocab_size=2048, hidden_size=1024, intermediate_size=2048, num_hidden_layers=2, num_attention_heads=32,
                      num_key_value_heads=8, head_dim=128, max_position_embeddings=2048, tie_word_embeddings=False)
    rng = np.random.default_rng(3)
    system = rng.integers(0, 2048, size=512).tolist()
    warm = [("warm", system + [5], 2)]
    reqs = [(f"r{i}", system + rng.integers(0, 2048, size=int(rng.integers(8, 64))).tolist(), int(rng.integers(3, 9))) for i in range(6)]
    out = {}
    for reuse in (False, True):
        runner = ModelRunner(model="cfg3", max_bs=8, enforce_eager=False, context_len=1024, config=cfg, seed=2)
        runner.load_model()
        runner.initialize_kv_cache(8192)
        core = EngineCore(kv_cache_size=8192, max_bs=8, context_len=1024, prefix_reuse=reuse)
        core.scheduler.on_release = runner.release_sequence
        with torch.inference_mode():
            wt, _, _ = _forced_run(core, runner, warm, out[False][3] if reuse else None)
            toks, logits, rows = _forced_run(core, runner, reqs, out[False][0] if reuse else None)
        out[reuse] = (toks, logits, rows, wt, core.scheduler.kv_manager.prefix_hit_tokens, runner.stats["graph_steps"])
    assert out[False][4] == 0 and out[True][4] == 6 * 512
    assert out[True][2] == out[False][2] - 6 * 512
    assert out[True][5] > 0
    for sid, _, _ in reqs:
        check_logits("config3_prefix_reuse_vs_recompute", np.stack(out[True][1][sid]), np.stack(out[False][1][sid]), sid)


def test_config5_4096_token_context_in_chunks_at_32b_rank_geometry():
    """16 query / 2 KV heads (one rank of Qwen3-32B over 4), prompts of 1024-3072 tokens up to a 4096-token
    context, prefill in chunks of 1024 tokens (a later chunk attends to up to 3072 cached keys) and hipGraph
    decode over block-table rows of that length; against the same model prefilled in one step."""
    from nano_vllm_x_amd.core.engine import EngineCore
    from nano_vllm_x_amd.core.model_runner import ModelRunner
    from nano_vllm_x_amd.qwen3 import Qwen3Config

    cfg = Qwen3Config(vocab_size=2048, hidden_size=1024, intermediate_size=2048, num_hidden_layers=2, num_attention_heads=16,
                      num_key_value_heads=2, head_dim=128, max_position_embeddings=4096, tie_word_embeddings=False)
    rng = np.random.default_rng(8)
    reqs = [(f"l{i}", rng.integers(0, 2048, size=int(n)).tolist(), 6) for i, n in enumerate((3072, 1024, 2500, 1777))]
    out = {}
    for chunked in (False, True):
        runner = ModelRunner(model="cfg5", max_bs=4, enforce_eager=False, context_len=4096, config=cfg, seed=4)
        runner.load_model()
        runner.initialize_kv_cache(16384)
        core = EngineCore(kv_cache_size=16384, max_bs=4, context_len=4096, max_prefill_tokens=1024 if chunked else None,
                          chunked_prefill=chunked)
        core.scheduler.on_release = runner.release_sequence
        with torch.inference_mode():
            toks, logits, rows = _forced_run(core, runner, reqs, out[False][0] if chunked else None)
        out[chunked] = (toks, logits, rows, dict(runner.stats))
    assert out[True][2] == out[False][2] == sum(len(p) for _, p, _ in reqs)
    assert out[False][3]["prefill_steps"] == 1 and out[True][3]["prefill_steps"] == -(-sum(len(p) for _, p, _ in reqs) // 1024)
    assert out[True][3]["graph_steps"] == 5 and out[True][3]["eager_decode_steps"] == 0
    for sid, _, _ in reqs:
        check_logits("config5_chunked_vs_one_step_prefill", np.stack(out[True][1][sid]), np.stack(out[False][1][sid]), sid)


# ---------------------------------------------------------------------------------------------------
def test_a_batch_replayed_from_a_larger_graph_gets_the_same_bits():
    """ModelRunner.prune_slow_graph_sizes drops a captured size that is slower than a larger one; its batches then run
    padded to the larger size.  Rows of a batch do not see the padding rows: with every size but the largest dropped
    (margin 0), the same teacher-forced history gives bit-identical logits at every step."""
    from nano_vllm_x_amd import qwen3
    from nano_vllm_x_amd.core.engine import EngineCore
    from nano_vllm_x_amd.core.model_runner import ModelRunner

    cfg = qwen3.Qwen3Config(vocab_size=1024, hidden_size=1024, intermediate_size=1536, num_hidden_layers=2, num_attention_heads=16,
                            num_key_value_heads=8, head_dim=128, max_position_embeddings=512, tie_word_embeddings=False)
    rng = np.random.default_rng(3)
    prompts = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in (5, 40, 17, 64, 9)]
    n_new, slots, max_bs, ctx = 6, 2048, 48, 256
    runner = ModelRunner(model="graph-sizes", max_bs=max_bs, enforce_eager=False, context_len=ctx, config=cfg, seed=1)
    runner.load_model(state_dict_iter=qwen3.random_checkpoint(cfg, 9))
    runner.initialize_kv_cache(slots)
    assert runner.graph.captured_bs[-1] == max_bs and len(runner.graph.captured_bs) + len(runner.graph_sizes_dropped) == 7
    assert set(runner.graph_step_us_by_size) == {1, 2, 4, 8, 16, 32, 48}

    def run(forced=None):
        core = EngineCore(kv_cache_size=slots, max_bs=max_bs, context_len=ctx)
        reqs = [(str(i), p, n_new) for i, p in enumerate(prompts)]
        with torch.inference_mode():
            toks, logits, _ = _forced_run(core, runner, reqs, forced)
        for s in list(core.scheduler.running):
            runner.release_sequence(s)
        return toks, logits

    toks, want = run()
    before = runner.stats["graph_steps"]
    assert before == n_new - 1
    runner.prune_slow_graph_sizes(margin=0.0)
    assert runner.graph.captured_bs == [max_bs] and runner.graph.match_bs(5) == max_bs
    _toks, got = run(forced=toks)
    assert runner.stats["graph_steps"] == 2 * before and runner.stats["eager_decode_steps"] == 0
    for sid in want:
        for a, b in zip(want[sid], got[sid]):
            assert np.array_equal(a, b)

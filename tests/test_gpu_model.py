"""GPU end-to-end: the Qwen3 module graph over the HIP ops against the reference's tiny-model logits,
eager vs hipGraph decode, and the engine loop."""
import numpy as np
import pytest
import torch

from helpers import from_bits, golden_json, golden_npz
from model_checks import check_close, check_logits

pytestmark = pytest.mark.gpu


def _tiny_runner(enforce_eager, max_bs=4, kv_slots=64, context_len=64):
    from nano_vllm_x_amd.core.model_runner import ModelRunner
    from nano_vllm_x_amd.qwen3 import Qwen3Config
    from tiny_model import tiny_state_dict

    meta = golden_json("g7_tiny_qwen3.json")
    cfg = Qwen3Config.from_dict(meta["config"])
    r = ModelRunner(model="tiny", max_bs=max_bs, enforce_eager=enforce_eager, context_len=context_len, config=cfg)
    r.load_model(state_dict_iter=tiny_state_dict(meta["config"]).items())
    r.initialize_kv_cache(kv_slots)
    return r, meta


def _run_fixture(runner, meta):
    from nano_vllm_x_amd.core.common import ForwardBatch, ForwardMode, Sequence

    seqs = [Sequence(seq_id=f"t{i}", token_ids=list(p), num_tokens=len(p), prompt_len=len(p), kv_indices=list(s))
            for i, (p, s) in enumerate(zip(meta["prompts"], meta["slots"]))]
    outs = []
    for step in range(3):
        mode = ForwardMode.PREFILL if step == 0 else ForwardMode.DECODE
        logits = runner.compute_logits(ForwardBatch(forward_mode=mode, num_seqs=3, seqs=seqs))
        outs.append(logits.float().cpu().numpy())
        if step < 2:
            for i, s in enumerate(seqs):
                s.cached_kv_len = len(s.kv_indices)
                s.token_ids.append(meta["feed"][step][i])
                s.kv_indices.append(meta["next_slot"][i] + step)
    return outs


def test_tiny_qwen3_logits_match_reference_eager_and_graph():
    """Reference logits come from bf16 GEMMs on the CPU; here hipBLASLt accumulates in another order, so
    agreement is to bf16 working precision: the bound is twice the error observed on MI355X (model_checks)."""
    g = golden_npz("g7_tiny_qwen3.npz")
    r_eager, meta = _tiny_runner(enforce_eager=True)
    got_eager = _run_fixture(r_eager, meta)
    for step in range(3):
        check_logits("g7_reference_vs_hip_eager", got_eager[step], g[f"logits{step}"], f"step {step}")
    # K rows written by the fused norm+rotary+scatter kernel vs the reference's cache
    check_close("g7_k_cache_vs_reference", r_eager.kv_cache.k_cache.float().cpu().numpy(), from_bits(g["k_cache"]).float().numpy())
    r_graph, meta = _tiny_runner(enforce_eager=False)
    got_graph = _run_fixture(r_graph, meta)
    assert r_graph.stats["graph_steps"] == 2
    for step in range(3):   # same kernels, same order: identical bits
        assert np.array_equal(got_graph[step], got_eager[step]), step
    # the op-by-op (unfused) module graph gives the same logits as the fused kernel
    r_eager.model.fused = False
    r_eager.kv_cache.k_cache.zero_(); r_eager.kv_cache.v_cache.zero_()
    got_unfused = _run_fixture(r_eager, meta)
    for step in range(3):
        check_logits("g7_unfused_vs_fused", got_unfused[step], got_eager[step], f"step {step}")
        check_logits("g7_reference_vs_hip_unfused", got_unfused[step], g[f"logits{step}"], f"step {step}")


def test_float32_checkpoint_is_served_in_bfloat16():
    """The kernels are built for the two 16-bit formats (a float16 checkpoint runs in float16: test_gpu_float16.py).  A
    checkpoint whose config says float32 loads and runs with its tensors converted to bfloat16 (same exponent range):
    the logits are those of the same values handed over as a bfloat16 checkpoint, bit for bit."""
    import dataclasses

    from nano_vllm_x_amd.core.model_runner import ModelRunner
    from nano_vllm_x_amd.qwen3 import Qwen3Config
    from tiny_model import tiny_state_dict

    meta = golden_json("g7_tiny_qwen3.json")
    as_f32 = {k: v.float() for k, v in tiny_state_dict(meta["config"]).items()}
    base = Qwen3Config.from_dict(meta["config"])

    def run(cfg, sd):
        r = ModelRunner(model="tiny", max_bs=4, enforce_eager=True, context_len=64, config=cfg)
        r.load_model(state_dict_iter=sd.items())
        r.initialize_kv_cache(64)
        return r, _run_fixture(r, meta)

    r32, got = run(dataclasses.replace(base, torch_dtype="float32"), as_f32)
    assert r32.checkpoint_dtype == torch.float32 and r32.dtype == torch.bfloat16
    _, want = run(base, {k: v.to(torch.bfloat16) for k, v in as_f32.items()})   # the same values, converted beforehand
    for step in range(3):
        assert np.array_equal(got[step], want[step]), step
        check_logits("g7_reference_vs_hip_eager", got[step], golden_npz("g7_tiny_qwen3.npz")[f"logits{step}"], f"float32 checkpoint, step {step}")


def test_engine_greedy_streams_identical_eager_vs_graph():
    from nano_vllm_x_amd.core.common import SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    streams = []
    for eager in (True, False):
        runner, meta = _tiny_runner(enforce_eager=eager, max_bs=4, kv_slots=400, context_len=64)
        core = EngineCore(kv_cache_size=400, max_bs=4, context_len=64)
        core.scheduler.on_release = runner.release_sequence
        rng = np.random.default_rng(0)
        for i in range(7):
            prompt = rng.integers(0, 256, size=int(rng.integers(3, 30))).tolist()
            core.add_sequence(f"s{i}", prompt, SamplingParams(temperature=1.0, top_k=1, ignore_eos=True,
                                                              max_new_tokens=int(rng.integers(2, 12))))
        got = {}
        steps = 0
        while core.has_unfinished_sequences():
            batch = core.scheduler.schedule()
            toks = runner.execute_model(batch)
            for o in core.update_from_output(batch, toks):
                got.setdefault(o.seq_id, []).append(o.new_token_id)
            steps += 1
            assert steps < 200
        streams.append(got)
        if not eager:
            assert runner.stats["graph_steps"] > 0
    assert streams[0] == streams[1]


def test_llm_generate_api():
    import asyncio

    from nano_vllm_x_amd.core.common import SamplingParams
    from nano_vllm_x_amd.llm import LLM
    from nano_vllm_x_amd.qwen3 import PRESETS

    PRESETS["tiny-test"] = dict(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                                tie_word_embeddings=False)

    async def main():
        llm = LLM("tiny-test", max_bs=4, context_len=64, kv_cache_size=512)
        async def one(prompt, n):
            outs = [o async for o in llm.generate(prompt, SamplingParams(temperature=0.7, ignore_eos=True, max_new_tokens=n))]
            return outs
        res = await asyncio.gather(one([1, 2, 3, 4], 5), one("hi there", 3), one([9] * 20, 7))
        llm.shutdown()
        return res

    res = asyncio.run(main())
    assert [len(r) for r in res] == [5, 3, 7]
    assert all(r[-1].is_finished and r[-1].finish_reason.name == "LENGTH" for r in res)
    assert res[0][-1].num_prompt_tokens == 4 and res[0][-1].num_generated_tokens == 5


@pytest.mark.parametrize("extra", [{}, {"prefix_reuse": True}, {"max_prefill_tokens": 32, "chunked_prefill": True}],
                         ids=["plain", "prefix_reuse", "chunked_prefill"])
def test_run_ahead_generates_the_same_streams(extra):
    """run_ahead queues decode step N+1 before step N's ids reach the host; its input ids are read on the
    device from the sampler's output buffer.  Same batches, same rows, same (seed, step) draws: the token
    streams must be bit-identical to the ordinary loop's, with sampling at temperature 0.8."""
    import asyncio

    from nano_vllm_x_amd.core.common import SamplingParams
    from nano_vllm_x_amd.llm import LLM
    from nano_vllm_x_amd.qwen3 import PRESETS

    PRESETS["tiny-ahead"] = dict(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                 num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                                 tie_word_embeddings=False)
    rng = np.random.default_rng(12)
    reqs = [(rng.integers(0, 256, size=int(rng.integers(3, 40))).tolist(), int(rng.integers(2, 30))) for _ in range(11)]

    async def main(ahead):
        llm = LLM("tiny-ahead", max_bs=8, context_len=128, kv_cache_size=1024, seed=3, run_ahead=ahead, **extra)

        async def one(i, prompt, n):
            sp = SamplingParams(temperature=0.8, ignore_eos=(i != 4), eos_token_id=255, max_new_tokens=n)   # one request may stop on EOS
            return [o.token_id async for o in llm.generate(list(prompt), sp, sequence_id=f"r{i}")]

        res = await asyncio.gather(*(one(i, p, n) for i, (p, n) in enumerate(reqs)))
        stats = (llm.engine.ahead_steps, dict(llm.engine.model_runner.stats))
        llm.shutdown()
        return res, stats

    plain, (n0, st0) = asyncio.run(main(False))
    ahead, (n1, st1) = asyncio.run(main(True))
    assert n0 == 0 and n1 > 0
    assert st0["decode_steps"] == st1["decode_steps"] and st1["eager_decode_steps"] == 0
    assert plain == ahead
    assert [len(r) for r in plain if r] and all(len(r) <= n for r, (_, n) in zip(plain, reqs))


def test_prefix_reuse_generates_the_same_tokens_with_fewer_prefill_rows():
    """Requests sharing a cached system prompt: with Scheduler(prefix_reuse=True) the shared rows are
    read from the pool instead of recomputed.  Greedy streams must equal those of the reference-faithful
    path (flag off); the prefill must have computed fewer rows."""
    from nano_vllm_x_amd.core.common import ForwardMode, SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    rng = np.random.default_rng(1)
    system = rng.integers(0, 256, size=24).tolist()
    requests = [("warm", system + [5], 3)] + [(f"r{i}", system + rng.integers(0, 256, size=int(rng.integers(2, 9))).tolist(),
                                               int(rng.integers(3, 8))) for i in range(6)]
    results = {}
    for reuse in (False, True):
        runner, _ = _tiny_runner(enforce_eager=False, max_bs=4, kv_slots=600, context_len=64)
        core = EngineCore(kv_cache_size=600, max_bs=4, context_len=64, prefix_reuse=reuse)
        core.scheduler.on_release = runner.release_sequence
        got, prefill_rows = {}, 0

        def run_all():
            nonlocal prefill_rows
            while core.has_unfinished_sequences():
                batch = core.scheduler.schedule()
                if batch.forward_mode == ForwardMode.PREFILL:
                    prefill_rows += sum(len(s.token_ids) - s.cached_kv_len for s in batch.seqs)
                toks = runner.execute_model(batch)
                for o in core.update_from_output(batch, toks):
                    got.setdefault(o.seq_id, []).append(o.new_token_id)

        sid, prompt, n = requests[0]
        core.add_sequence(sid, list(prompt), SamplingParams(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=n))
        run_all()                       # the warm-up request finishes and its rows enter the tree
        for sid, prompt, n in requests[1:]:
            core.add_sequence(sid, list(prompt), SamplingParams(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=n))
        run_all()
        results[reuse] = (got, prefill_rows, core.scheduler.kv_manager.prefix_hit_tokens)
    assert results[True][0] == results[False][0]
    assert results[False][2] == 0 and results[True][2] == 6 * 24
    assert results[True][1] == results[False][1] - 6 * 24


def test_chunked_prefill_generates_the_same_tokens():
    """Scheduler(chunked_prefill=True, max_prefill_tokens=...): prompts longer than the budget are computed
    over several prefill steps (the later chunks attend to the rows the earlier ones wrote).  The streams of
    arg-max tokens must equal those of the one-step prefill (arg-max of the logits, not the sampler with
    top_k = 1: that keeps ties, which bf16 logits of a tiny random model have, and breaks them by a draw that
    depends on the row's place in the batch)."""
    from nano_vllm_x_amd.core.common import ForwardMode, SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    rng = np.random.default_rng(2)
    requests = [(f"c{i}", rng.integers(0, 256, size=int(n)).tolist(), int(rng.integers(3, 7))) for i, n in enumerate((50, 7, 33, 61, 12))]
    results = {}
    for chunked in (False, True):
        runner, _ = _tiny_runner(enforce_eager=False, max_bs=8, kv_slots=600, context_len=64)
        core = EngineCore(kv_cache_size=600, max_bs=8, context_len=64, max_prefill_tokens=24 if chunked else None,
                          chunked_prefill=chunked)
        core.scheduler.on_release = runner.release_sequence
        for sid, prompt, n in requests:
            core.add_sequence(sid, list(prompt), SamplingParams(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=n))
        got, prefill_steps = {}, 0
        while core.has_unfinished_sequences():
            batch = core.scheduler.schedule()
            prefill_steps += batch.forward_mode == ForwardMode.PREFILL
            toks = runner.compute_logits(batch).float().argmax(dim=-1).tolist()
            for o in core.update_from_output(batch, toks):
                got.setdefault(o.seq_id, []).append(o.new_token_id)
        results[chunked] = (got, prefill_steps)
    assert results[True][0] == results[False][0]
    assert results[False][1] == 1 and results[True][1] == 7      # 163 prompt tokens in steps of at most 24
    assert all(len(results[True][0][sid]) == n for sid, _, n in requests)


def test_decode_projections_fused_path_matches_library_gemm_path():
    """A mid-size Qwen3 (hidden 512) whose decode projections qualify for ops.linear_decode (+ fused
    SiluAndMul, + K-split partials summed inside RMSNorm) against the same model with every projection a
    library GEMM (module flag fused=False runs the op-by-op graph).  Same bf16 operands, fp32
    accumulation in both, different summation order: logits agree to 2^-6 * max|logit|."""
    from nano_vllm_x_amd import qwen3

    cfg = qwen3.Qwen3Config(vocab_size=1024, hidden_size=512, intermediate_size=1536, num_hidden_layers=3,
                            num_attention_heads=4, num_key_value_heads=2, head_dim=128, max_position_embeddings=256,
                            tie_word_embeddings=False)
    calls = _fused_vs_library(cfg, "hidden512")
    assert calls == 3 * 3 * 4, calls        # 3 decode steps x 3 layers x (qkv, o, gate_up, down)


@pytest.mark.parametrize("tp", [2, 4, 8])
def test_shapes_of_one_tensor_parallel_rank_of_qwen3_0_6b(tp):
    """The kernels see per-rank shapes under tensor parallelism: Qwen3-0.6B over 2, 4 or 8 ranks leaves a
    rank 8/4, 4/2 or 2/1 query/KV heads and 1536, 768 or 384 MLP columns at hidden size 1024.  A one-rank
    model of exactly that geometry (two layers, small vocabulary) runs the HIP path against the
    library-GEMM op-by-op path; what the ranks exchange is covered by the two-rank tests."""
    from nano_vllm_x_amd import qwen3

    cfg = qwen3.Qwen3Config(vocab_size=2048, hidden_size=1024, intermediate_size=3072 // tp, num_hidden_layers=2,
                            num_attention_heads=16 // tp, num_key_value_heads=8 // tp, head_dim=128,
                            max_position_embeddings=256, tie_word_embeddings=False)
    assert _fused_vs_library(cfg, f"0.6B/tp{tp}") > 0


def _fused_vs_library(cfg, tag: str) -> int:
    from nano_vllm_x_amd import qwen3
    from nano_vllm_x_amd.core.common import ForwardBatch, ForwardMode, Sequence
    from nano_vllm_x_amd.core.model_runner import ModelRunner

    rng = np.random.default_rng(4)
    prompts = [rng.integers(0, cfg.vocab_size, size=int(n)).tolist() for n in (5, 17, 33, 9, 21)]
    outs = {}
    calls = {"n": 0}
    real = qwen3.ops.linear_decode

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    for fused in (True, False):
        r = ModelRunner(model="mid", max_bs=8, enforce_eager=True, context_len=128, config=cfg, seed=3)
        r.load_model()
        r.model.fused = fused
        r.initialize_kv_cache(512)
        slot = 0
        seqs = []
        for i, p in enumerate(prompts):
            seqs.append(Sequence(seq_id=f"m{i}", token_ids=list(p), num_tokens=len(p), prompt_len=len(p),
                                 kv_indices=list(range(slot, slot + len(p)))))
            slot += len(p) + 8
        got = []
        qwen3.ops.linear_decode = counting if fused else real
        try:
            for step in range(4):
                mode = ForwardMode.PREFILL if step == 0 else ForwardMode.DECODE
                logits = r.compute_logits(ForwardBatch(forward_mode=mode, num_seqs=len(seqs), seqs=seqs)).float().cpu().numpy()
                got.append(logits)
                for i, s in enumerate(seqs):
                    s.cached_kv_len = len(s.kv_indices)
                    s.token_ids.append(int(rng.integers(0, cfg.vocab_size)) if fused else outs["fed"][step][i])
                    s.kv_indices.append(s.kv_indices[-1] + 1)
                if fused:
                    outs.setdefault("fed", []).append([s.token_ids[-1] for s in seqs])
        finally:
            qwen3.ops.linear_decode = real
        outs[fused] = got
    for step in range(4):
        check_logits(f"fused_vs_library[{tag}]", outs[True][step], outs[False][step], f"step {step}")
    return calls["n"]


def test_small_pool_forces_preemption_and_eviction_under_graph_decode():
    """A KV pool far smaller than the workload: sequences are preempted and re-prefilled, cached prefixes
    are evicted, block-table rows are released and reassigned -- all under hipGraph decode.  Every request
    must still finish with exactly the requested number of tokens, and nothing may leak."""
    from nano_vllm_x_amd.core.common import ForwardMode, SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    pool = 220
    runner, _ = _tiny_runner(enforce_eager=False, max_bs=4, kv_slots=pool, context_len=64)
    core = EngineCore(kv_cache_size=pool, max_bs=4, context_len=64)
    core.scheduler.on_release = runner.release_sequence
    rng = np.random.default_rng(7)
    want = {}
    for i in range(14):
        n_prompt, n_new = int(rng.integers(8, 40)), int(rng.integers(5, 20))
        want[f"p{i}"] = n_new
        core.add_sequence(f"p{i}", rng.integers(0, 256, size=n_prompt).tolist(),
                          SamplingParams(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=n_new))
    got = {k: 0 for k in want}
    prefills = steps = 0
    while core.has_unfinished_sequences():
        batch = core.scheduler.schedule()
        assert batch is not None
        prefills += batch.forward_mode == ForwardMode.PREFILL
        toks = runner.execute_model(batch)
        for o in core.update_from_output(batch, toks):
            got[o.seq_id] += 1
            if o.is_finished:
                assert o.num_generated_tokens == want[o.seq_id]
        steps += 1
        assert steps < 2000
    assert prefills > 4, "the pool was meant to be too small for one admission wave"
    assert not runner.block_table.row_of, "block-table rows leaked"
    mgr = core.scheduler.kv_manager
    in_tree = []
    todo = [mgr.radix_tree.root]
    while todo:
        n = todo.pop()
        in_tree.extend(n.value)
        todo.extend(n.children.values())
    # reference accounting (leaks duplicate slots of shared prefixes by design: none here, prompts are random)
    assert mgr.kv_cache_allocator.num_free() + len(set(in_tree)) == pool
    assert runner.stats["graph_steps"] > 0

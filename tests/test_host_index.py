"""Product host logic (allocator, radix tree, KV manager, scheduler, engine core) vs the reference's golden traces."""
import pytest

from helpers import bench_workload, golden_json
from trace_drivers import (ProductEngineAdapter, ProductRadixAdapter, dump_node_tree, replay_bench_trace,
                           replay_radix_program, replay_small_trace)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_radix_program(i):
    replay_radix_program(ProductRadixAdapter, golden_json("g1_radix.json")["traces"][i])


def test_corner_cases():
    from nano_vllm_x_amd.core.kv_cache import KVCacheAllocator, RadixTree

    c = golden_json("g1_radix.json")["corner"]
    a = KVCacheAllocator(8)
    assert [a.alloc(3), a.alloc(2)] == c["alloc_order"]
    a.free([1, 0])
    assert [a.alloc(2), a.alloc(4), a.alloc(1)] == c["after_free"]
    t = RadixTree(KVCacheAllocator(8))
    with pytest.raises(IndexError):
        t.match_prefix([])
    with pytest.raises(IndexError):
        t.insert([], [])
    t.insert([5, 6, 7], [0, 1])
    t.insert([5, 6, 8, 9], [])
    assert dump_node_tree(t.root) == c["ragged_tree"]
    assert [t.match_prefix([5, 6, 8])[0], t.match_prefix([5, 6, 7, 1])[0]] == c["ragged_match"]
    assert dump_node_tree(t.root) == c["ragged_tree_after_match"]


def test_scheduler_bench_trace():
    g = golden_json("g2_sched_bench.json")
    prompts, max_tokens = bench_workload()
    replay_bench_trace(ProductEngineAdapter, g, prompts, max_tokens)


def test_scheduler_small_trace():
    replay_small_trace(ProductEngineAdapter, golden_json("g2_sched_small.json"))


def test_kv_exhaustion_preempts_instead_of_spinning():
    """Where the reference loops forever, the victim is preempted and re-prefilled later."""
    from nano_vllm_x_amd.core.common import ForwardMode, SamplingParams, SequenceStatus
    from nano_vllm_x_amd.core.engine import EngineCore

    e = EngineCore(kv_cache_size=20, max_bs=4, context_len=64)
    for name in ("a", "b"):
        e.add_sequence(name, list(range(9)), SamplingParams(ignore_eos=True, max_new_tokens=6))
    b = e.scheduler.schedule()
    assert b.forward_mode == ForwardMode.PREFILL and b.num_seqs == 2
    e.update_from_output(b, [1, 1])
    b = e.scheduler.schedule()           # 18 used, both need one more: fits exactly
    assert b.forward_mode == ForwardMode.DECODE and b.num_seqs == 2
    e.update_from_output(b, [1, 1])
    b = e.scheduler.schedule()           # pool exhausted: "b" is preempted to make room for "a"
    assert [s.seq_id for s in b.seqs] == ["a"]
    sb = e.scheduler.get_sequence("b")
    assert sb.status == SequenceStatus.WAITING and sb.kv_indices == [] and e.scheduler.waiting[0] is sb
    e.update_from_output(b, [1])
    steps = 0
    while e.has_unfinished_sequences() and steps < 100:
        b = e.scheduler.schedule()
        if b is None:
            break
        e.update_from_output(b, [1] * b.num_seqs)
        steps += 1
    assert not e.has_unfinished_sequences()


def test_single_sequence_too_large_returns_none():
    from nano_vllm_x_amd.core.common import SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    e = EngineCore(kv_cache_size=8, max_bs=4, context_len=64)
    e.add_sequence("a", [1, 2, 3, 4, 5, 6, 7, 8], SamplingParams(ignore_eos=True, max_new_tokens=4))
    b = e.scheduler.schedule()
    e.update_from_output(b, [9])
    assert e.scheduler.schedule() is None     # no slot, nobody to preempt: the step yields nothing
    assert [s.seq_id for s in e.scheduler.running] == ["a"]


def test_prefill_token_budget_is_opt_in_and_never_starves():
    """Scheduler(max_prefill_tokens=...) caps the tokens of one prefill step; None (default) is the
    reference's unbounded admission (scheduler.py:67-80)."""
    from nano_vllm_x_amd.core.common import ForwardMode, SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    lens = [30, 50, 10, 200, 5, 5]

    def prefill_batches(budget):
        core = EngineCore(kv_cache_size=2000, max_bs=8, max_prefill_tokens=budget)
        for i, n in enumerate(lens):
            core.add_sequence(f"s{i}", list(range(n)), SamplingParams(ignore_eos=True, max_new_tokens=2))
        out = []
        while core.has_unfinished_sequences():
            b = core.scheduler.schedule()
            if b.forward_mode == ForwardMode.PREFILL:
                out.append([len(s.token_ids) for s in b.seqs])
            core.update_from_output(b, [1] * b.num_seqs)
        return out

    assert prefill_batches(None) == [lens]
    capped = prefill_batches(100)
    assert capped == [[30, 50, 10], [200], [5, 5]]          # FCFS order kept; the 200-token prompt runs alone
    assert all(sum(b) <= 100 or len(b) == 1 for b in capped)


def test_context_length_stop_rule_is_opt_in():
    """EngineCore(stop_at_context_len=True) ends a sequence with LENGTH at context_len tokens; off (default)
    only the reference's rules apply (engine.py:143-154 has no such rule)."""
    from nano_vllm_x_amd.core.common import FinishReason, SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    def run(flag):
        core = EngineCore(kv_cache_size=400, max_bs=4, context_len=16, stop_at_context_len=flag)
        core.add_sequence("a", list(range(10)), SamplingParams(ignore_eos=True, max_new_tokens=20))
        core.add_sequence("b", list(range(40)), SamplingParams(ignore_eos=True, max_new_tokens=3))   # prompt cut to its last 16 tokens
        ends = {}
        while core.has_unfinished_sequences():
            b = core.scheduler.schedule()
            for o in core.update_from_output(b, [1] * b.num_seqs):
                if o.is_finished:
                    ends[o.seq_id] = (o.num_prompt_tokens + o.num_generated_tokens, o.finish_reason)
        return ends

    assert run(False) == {"a": (30, FinishReason.LENGTH), "b": (19, FinishReason.LENGTH)}
    assert run(True) == {"a": (16, FinishReason.LENGTH), "b": (17, FinishReason.LENGTH)}   # b: one token, then at the limit


def test_chunked_prefill_computes_every_token_once_and_keeps_order():
    """Scheduler(chunked_prefill=True): a prompt larger than the step budget is computed over several
    prefill steps; every prompt token is computed exactly once, a cut prompt yields no token until its last
    chunk, decode steps wait for prompts under way, and the final streams have the lengths asked for."""
    from nano_vllm_x_amd.core.common import ForwardMode, SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    lens = [30, 250, 10, 120]
    core = EngineCore(kv_cache_size=2000, max_bs=8, max_prefill_tokens=100, chunked_prefill=True)
    for i, n in enumerate(lens):
        # disjoint token ranges: sequences that share a prefix leak their duplicate slots at finish (a quirk of
        # the reference that is reproduced), which would blur the slot count at the end
        core.add_sequence(f"s{i}", list(range(1000 * i, 1000 * i + n)), SamplingParams(ignore_eos=True, max_new_tokens=3))
    computed = {f"s{i}": [] for i in range(len(lens))}
    steps, outs = [], {f"s{i}": 0 for i in range(len(lens))}
    while core.has_unfinished_sequences():
        b = core.scheduler.schedule()
        assert b is not None
        if b.forward_mode == ForwardMode.PREFILL:
            sizes = [len(s.token_ids) - s.cached_kv_len for s in b.seqs]
            assert sum(sizes) <= 100
            for s, n in zip(b.seqs, sizes):
                assert len(s.kv_indices) == len(s.token_ids)          # what the model is shown is self-consistent
                computed[s.seq_id].append((s.cached_kv_len, s.cached_kv_len + n))
            steps.append(sizes)
        else:
            assert not core.scheduler.prefilling                       # no decode while a prompt is under way
        for o in core.update_from_output(b, [100000 + len(steps)] * b.num_seqs):
            outs[o.seq_id] += 1
    assert steps[:4] == [[30, 70], [100], [80, 10, 10], [100]] and steps[4] == [10]
    for i, n in enumerate(lens):
        spans = computed[f"s{i}"]
        assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert outs[f"s{i}"] == 3
    assert len(core.scheduler.kv_manager.kv_cache_allocator.free_slots) + _slots_in_tree(core.scheduler.kv_manager) == 2000


def _slots_in_tree(mgr):
    total, stack = 0, [mgr.radix_tree.root]
    while stack:
        node = stack.pop()
        total += len(node.value)
        stack.extend(node.children.values())
    return total


def test_chunked_prefill_abort_midway_returns_every_slot():
    from nano_vllm_x_amd.core.common import SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    core = EngineCore(kv_cache_size=500, max_bs=4, max_prefill_tokens=64, chunked_prefill=True)
    core.add_sequence("a", list(range(200)), SamplingParams(ignore_eos=True, max_new_tokens=2))
    b = core.scheduler.schedule()
    core.update_from_output(b, [1])
    assert core.scheduler.prefilling and core.scheduler.sequences["a"].cached_kv_len == 64
    core.abort_sequence("a")
    assert not core.has_unfinished_sequences()
    assert len(core.scheduler.kv_manager.kv_cache_allocator.free_slots) == 500

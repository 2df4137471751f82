"""Oracle vs golden traces frozen from the reference: allocator, radix tree, manager, scheduler."""
import pytest

from helpers import bench_workload, golden_json
from trace_drivers import (OracleEngineAdapter, OracleRadixAdapter, replay_bench_trace,
                           replay_radix_program, replay_small_trace)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_radix_program(i):
    replay_radix_program(OracleRadixAdapter, golden_json("g1_radix.json")["traces"][i])


def test_corner_cases():
    from oracle.kv_index_oracle import RadixOracle, SlotStackOracle

    c = golden_json("g1_radix.json")["corner"]
    a = SlotStackOracle(8)
    assert [a.alloc(3), a.alloc(2)] == c["alloc_order"]
    a.free([1, 0])
    assert [a.alloc(2), a.alloc(4), a.alloc(1)] == c["after_free"]
    t = RadixOracle(SlotStackOracle(8))
    assert c["match_empty"] == "IndexError" and c["insert_empty"] == "IndexError"
    with pytest.raises(IndexError):
        t.match_prefix([])
    with pytest.raises(IndexError):
        t.insert([], [])
    t.insert([5, 6, 7], [0, 1])
    t.insert([5, 6, 8, 9], [])
    assert t.dump() == c["ragged_tree"]
    assert [t.match_prefix([5, 6, 8])[0], t.match_prefix([5, 6, 7, 1])[0]] == c["ragged_match"]
    assert t.dump() == c["ragged_tree_after_match"]


def test_scheduler_bench_trace():
    g = golden_json("g2_sched_bench.json")
    prompts, max_tokens = bench_workload()
    assert sum(map(len, prompts)) == g["sum_prompt"] and sum(max_tokens) == g["sum_max_tokens"]
    replay_bench_trace(OracleEngineAdapter, g, prompts, max_tokens)


def test_scheduler_small_trace():
    replay_small_trace(OracleEngineAdapter, golden_json("g2_sched_small.json"))


def test_livelock_is_reported_not_spun():
    from oracle.sched_oracle import EngineOracle, SamplingOracle, SchedulerLivelock

    e = EngineOracle(kv_cache_size=8, max_bs=4, context_len=64)
    e.add("a", [1, 2, 3, 4, 5, 6, 7, 8], SamplingOracle(ignore_eos=True, max_new_tokens=4))
    b = e.sched.schedule()
    e.absorb(b, [9])
    with pytest.raises(SchedulerLivelock):
        e.sched.schedule()

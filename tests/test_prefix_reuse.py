"""Admission-time prefix reuse (Scheduler(prefix_reuse=True)): host-side slot accounting.

The reference never consults its radix tree at admission (the only match_prefix call is inside
cache_sequence, nanovllmx/core/kv_cache.py:343, reached only from finish_sequence,
core/scheduler.py:189), so there is no reference behaviour to pin here; these tests state the
invariants of the extension: shared slots are never freed or double-owned, nothing leaks, pinned
paths survive eviction, and with the flag off nothing changes (tests/test_host_index.py replays the
reference traces).
"""
from __future__ import annotations

import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nano_vllm_x_amd.core.common import ForwardMode, SamplingParams
from nano_vllm_x_amd.core.engine import EngineCore


def tree_slots(tree):
    out, todo = [], [tree.root]
    while todo:
        n = todo.pop()
        out.extend(n.value)
        assert len(n.value) == len(n.key) or n is tree.root, "reuse mode keeps one slot per keyed token"
        todo.extend(n.children.values())
    return out


def check_accounting(core, size):
    mgr = core.scheduler.kv_manager
    in_tree = tree_slots(mgr.radix_tree)
    assert len(in_tree) == len(set(in_tree)), "a slot is owned by two tree nodes"
    free = mgr.kv_cache_allocator.free_slots
    assert len(free) == len(set(free))
    owned = []
    for seq in list(core.scheduler.running) + list(core.scheduler.waiting):
        shared = mgr.adopted.get(seq.seq_id, (0, None))[0]
        assert seq.kv_indices[:shared] == [s for s in seq.kv_indices[:shared] if s in set(in_tree)]
        owned.extend(seq.kv_indices[shared:])
    everything = in_tree + free + owned
    assert len(everything) == len(set(everything)), "a slot has two owners"
    assert len(everything) == size, f"{size - len(everything)} slots leaked"


def drain(core, token=7, on_batch=None):
    steps = []
    while core.has_unfinished_sequences():
        b = core.scheduler.schedule()
        assert b is not None
        if on_batch:
            on_batch(b)
        steps.append((b.forward_mode, [(s.seq_id, s.cached_kv_len, len(s.kv_indices)) for s in b.seqs]))
        core.update_from_output(b, [token] * b.num_seqs)
    return steps


def sp(n):
    return SamplingParams(temperature=0.0, ignore_eos=True, max_tokens=n)


def test_second_request_adopts_the_cached_prefix():
    size = 400
    core = EngineCore(kv_cache_size=size, max_bs=4, prefix_reuse=True)
    system = list(range(100, 150))
    core.add_sequence("a", system + [1, 2, 3], sp(60))
    first = drain(core)
    assert first[0][1] == [("a", 0, 53)]
    check_accounting(core, size)
    a_slots = core.scheduler.kv_manager.radix_tree.match_prefix(system)[0]
    assert len(a_slots) == 50

    core.add_sequence("b", system + [9, 8], sp(58))
    seen = {}

    def grab(b):
        if b.forward_mode == ForwardMode.PREFILL:
            s = b.seqs[0]
            seen["cached"], seen["slots"] = s.cached_kv_len, list(s.kv_indices)

    drain(core, on_batch=grab)
    assert seen["cached"] == 50 and seen["slots"][:50] == a_slots       # rows reused, not recomputed
    assert not set(seen["slots"][50:]) & set(a_slots)
    mgr = core.scheduler.kv_manager
    assert (mgr.prefix_hit_tokens, mgr.prefix_query_tokens) == (50, 53 + 52)
    assert not mgr.adopted
    check_accounting(core, size)


def test_identical_prompt_keeps_one_token_to_compute():
    size = 300
    core = EngineCore(kv_cache_size=size, max_bs=4, prefix_reuse=True)
    prompt = list(range(40))
    core.add_sequence("a", list(prompt), sp(45))
    drain(core)
    core.add_sequence("b", list(prompt), sp(45))
    steps = drain(core)
    mode, rows = steps[0]
    assert mode == ForwardMode.PREFILL and rows == [("b", 39, 40)]
    check_accounting(core, size)


def test_simultaneous_duplicates_are_freed_not_leaked():
    # the reference leaks the duplicate slots of a finished sequence whose prefix is already cached
    size = 300
    core = EngineCore(kv_cache_size=size, max_bs=4, prefix_reuse=True)
    prompt = list(range(30))
    core.add_sequence("a", list(prompt), sp(33))
    core.add_sequence("b", list(prompt), sp(36))      # admitted in the same prefill: nothing to adopt yet
    drain(core)
    check_accounting(core, size)
    assert core.scheduler.kv_manager.prefix_hit_tokens == 0


def test_pinned_prefix_survives_eviction_and_preemption_unpins():
    size = 128
    core = EngineCore(kv_cache_size=size, max_bs=4, prefix_reuse=True)
    system = list(range(200, 240))
    core.add_sequence("warm", system + [1], sp(44))
    drain(core)
    core.add_sequence("b", system + [2, 3], sp(120))
    batch = core.scheduler.schedule()
    assert batch.seqs[0].cached_kv_len == 40
    shared = list(batch.seqs[0].kv_indices[:40])
    core.update_from_output(batch, [7])
    # a big newcomer forces eviction: only unpinned leaves may go
    core.add_sequence("c", list(range(1000, 1060)), sp(70))
    while core.scheduler.get_sequence("c") is not None:
        b = core.scheduler.schedule()
        core.update_from_output(b, [7] * b.num_seqs)
        check_accounting(core, size)
    seq_b = core.scheduler.get_sequence("b")
    assert seq_b is not None and seq_b.kv_indices[:40] == shared
    assert core.scheduler.kv_manager.radix_tree.match_prefix(system)[0] == shared
    core.scheduler.preempt(seq_b)
    assert not seq_b.kv_indices and not core.scheduler.kv_manager.adopted
    check_accounting(core, size)
    drain(core)
    check_accounting(core, size)


def test_randomised_workload_never_leaks_or_double_owns():
    rnd = random.Random(3)
    size = 900
    core = EngineCore(kv_cache_size=size, max_bs=6, prefix_reuse=True)
    prefixes = [[rnd.randint(0, 50) for _ in range(rnd.randint(5, 40))] for _ in range(4)]
    pending = []
    for i in range(60):
        p = list(rnd.choice(prefixes)) + [rnd.randint(0, 50) for _ in range(rnd.randint(1, 30))]
        pending.append((f"s{i}", p, len(p) + rnd.randint(1, 25)))
    step = 0
    while pending or core.has_unfinished_sequences():
        for _ in range(rnd.randint(0, 3)):
            if pending:
                sid, p, m = pending.pop()
                core.add_sequence(sid, p, sp(m))
        if step % 17 == 5 and core.scheduler.running:
            core.scheduler.preempt(rnd.choice(list(core.scheduler.running)))
        if step % 23 == 7 and core.scheduler.sequences:
            core.abort_sequence(rnd.choice(list(core.scheduler.sequences)))
        b = core.scheduler.schedule()
        if b is not None:
            for s in b.seqs:
                assert s.cached_kv_len < len(s.token_ids) and len(s.kv_indices) == len(s.token_ids)
            core.update_from_output(b, [rnd.randint(0, 50) for _ in range(b.num_seqs)])
        check_accounting(core, size)
        step += 1
    mgr = core.scheduler.kv_manager
    assert mgr.prefix_hit_tokens > 0 and not mgr.adopted


def test_flag_off_is_the_reference_path():
    core = EngineCore(kv_cache_size=300, max_bs=4)
    system = list(range(100, 150))
    core.add_sequence("a", system + [1], sp(55))
    drain(core)
    core.add_sequence("b", system + [2], sp(55))
    steps = drain(core)
    assert steps[0][1] == [("b", 0, 51)]            # nothing adopted
    assert core.scheduler.kv_manager.prefix_query_tokens == 0

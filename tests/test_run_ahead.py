"""Engine.step_overlapped with run_ahead: decode step N+1 is queued before step N's ids are known.

The device side is a stand-in here (the GPU test in test_gpu_model.py runs the real one); what is checked
is the host logic: the same tokens reach the same sequences, the scheduler hands out the same slots in the
same order, steps are really queued ahead, and no placeholder survives."""
from __future__ import annotations

import os
import sys
from collections import deque

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nano_vllm_x_amd.core.common import ForwardMode, SamplingParams   # noqa: E402
from nano_vllm_x_amd.core.engine import Engine, EngineCore   # noqa: E402


class _Table:
    row_capacity = 4096


class _Runner:
    """Samples id = hash of (sequence, its length) -- what a model would do is irrelevant; what matters is
    that a step queued ahead is given exactly the batch a step launched after the ids would have been."""

    def __init__(self):
        self.block_table = _Table()
        self.log = []          # (mode, [(seq_id, tokens known to the host, slots)], ids_on_device)
        self.pending = 0

    def graph_step_possible(self, batch):
        return batch.forward_mode == ForwardMode.DECODE

    def launch_step(self, batch, ids_on_device=None):
        self.pending += 1
        assert self.pending <= 2
        ids = [(sum(map(ord, s.seq_id)) * 31 + len(s.token_ids) * 7) % 1000 for s in batch.seqs]
        if ids_on_device is not None:
            assert all(s.token_ids[-1] == Engine.PLACEHOLDER for s in batch.seqs)
            assert [self.last_batch[j] for j in ids_on_device] == [s.seq_id for s in batch.seqs]     # rows of the step before
        else:
            assert all(t >= 0 for s in batch.seqs for t in s.token_ids)
        self.last_batch = [s.seq_id for s in batch.seqs]
        self.log.append((batch.forward_mode, [(s.seq_id, len(s.token_ids), tuple(s.kv_indices)) for s in batch.seqs],
                         ids_on_device is not None))
        return (batch, ids, 0.0)

    def finish_step(self, handle):
        self.pending -= 1
        return handle[1]

    def release_sequence(self, seq):
        pass


def _engine(run_ahead: bool, kv: int = 4096) -> Engine:
    eng = Engine.__new__(Engine)
    EngineCore.__init__(eng, kv_cache_size=kv, max_bs=4, context_len=256)
    eng.model_runner = _Runner()
    eng._inflight = eng._ahead = eng._placeholders = None
    eng._late_aborts, eng._pp_aborts, eng.ahead_steps = [], [], 0
    eng.ahead_retired = 0
    eng.run_ahead, eng.pp_size, eng._pp_window = run_ahead, 1, deque(maxlen=1)
    eng.pp_stats = {"ticks": 0, "stage_busy_ticks": 0}
    return eng


def _drive(eng, requests, abort=None, late=None):
    for sid, n_prompt, params in requests:
        eng.add_sequence(sid, list(range(1, n_prompt + 1)), params)
    got = {}
    calls = 0
    while eng.has_unfinished_sequences() or eng.step_in_flight:
        for o in eng.step_overlapped():
            got.setdefault(o.seq_id, []).append((o.new_token_id, o.is_finished, o.finish_reason, o.num_generated_tokens))
        calls += 1
        if abort and calls == abort[1]:
            eng.abort_sequence(abort[0])
        if late and calls == late[0]:
            eng.add_sequence(late[1][0], list(late[1][1]), late[1][2])
        assert calls < 2000
    return got


REQS = [("a", 5, SamplingParams(ignore_eos=True, max_new_tokens=9)),
        ("b", 9, SamplingParams(ignore_eos=True, max_tokens=20)),
        ("c", 3, SamplingParams(ignore_eos=True, max_new_tokens=30)),
        ("d", 7, SamplingParams(ignore_eos=False, eos_token_id=999999, max_new_tokens=4)),      # may stop on its token
        ("e", 4, SamplingParams(ignore_eos=True, max_new_tokens=25)),                           # fifth of max_bs 4: admitted later
        ]


@pytest.mark.parametrize("case", ["plain", "abort", "late_arrival", "tight_pool"])
def test_run_ahead_changes_nothing_but_the_launch_order(case):
    kw = {}
    if case == "abort":
        kw["abort"] = ("c", 12)
    if case == "late_arrival":
        kw["late"] = (15, ("f", [5, 6, 7], SamplingParams(ignore_eos=True, max_new_tokens=6)))
    kv = 80 if case == "tight_pool" else 4096        # 80 slots: eviction and preemption paths are reached
    runs = []
    for ahead in (False, True):
        eng = _engine(ahead, kv)
        got = _drive(eng, REQS, **kw)
        runs.append((got, eng.model_runner.log, eng.ahead_steps, eng.scheduler.kv_manager.kv_cache_allocator.num_free()))
        if ahead and case == "plain":
            assert eng.ahead_retired > 0        # steps after which a sequence ends by length are queued ahead too
        for s in eng.scheduler.sequences.values():
            assert Engine.PLACEHOLDER not in s.token_ids
    (got0, log0, ahead0, free0), (got1, log1, ahead1, free1) = runs
    assert ahead0 == 0 and ahead1 > 0
    if case == "late_arrival":
        # a request that arrives while a step is queued ahead is admitted one step later than without:
        # every sequence still gets the same tokens, the interleaving (and with it the slot numbers) shifts
        assert got0 == got1
        return
    assert free0 == free1
    if case != "abort":      # an abort lands one step later when a step is queued ahead: outputs may differ for that sequence
        assert got0 == got1
    else:
        assert {k: v for k, v in got0.items() if k != "c"} == {k: v for k, v in got1.items() if k != "c"}
    # every step ran on the same sequences with the same slots; only who supplied the newest id differs
    strip = lambda log: [(m, rows) for m, rows, _ in log]   # noqa: E731
    if case != "abort":
        assert strip(log0) == strip(log1)
    assert any(dev for _, _, dev in log1) and not any(dev for _, _, dev in log0)

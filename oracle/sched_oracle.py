"""Oracle: continuous-batching scheduler and finish rules (integer, bit-exact).

TEST INFRASTRUCTURE.  Restates nanovllmx/core/scheduler.py:13-193 and the per-step
bookkeeping of nanovllmx/core/engine.py:50-69,119-154.  Pinned by
tests/golden/g2_sched_bench.json and g2_sched_small.json (produced by the reference).

One deliberate difference: where the reference would spin forever (KV exhaustion during a
decode step with no other running sequence, scheduler.py:84-104 -- the sequence is put back
at the head of `running` and immediately popped again by the enclosing loop) this oracle
raises `SchedulerLivelock` instead of hanging the test run.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field

from .kv_index_oracle import KVManagerOracle

WAITING, RUNNING, FINISHED = "WAITING", "RUNNING", "FINISHED"
PREFILL, DECODE = "PREFILL", "DECODE"


class SchedulerLivelock(RuntimeError):
    pass


@dataclass
class SamplingOracle:  # common.py:4-16
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = -1
    min_p: float = 0.0
    max_tokens: int | None = None
    max_new_tokens: int | None = None
    ignore_eos: bool = False
    eos_token_id: int = -1


@dataclass(eq=False)
class SeqOracle:  # common.py:44-54
    seq_id: str
    token_ids: list
    prompt_len: int
    sampling: SamplingOracle
    status: str = WAITING
    num_tokens: int = 0
    kv_indices: list = field(default_factory=list)
    cached_kv_len: int = 0


@dataclass
class BatchOracle:  # common.py:61-65
    mode: str
    seqs: list

    @property
    def num_seqs(self):
        return len(self.seqs)


class SchedulerOracle:
    def __init__(self, kv_cache_size: int, max_bs: int = 32):
        self.max_bs = max_bs
        self.by_id: dict[str, SeqOracle] = {}
        self.waiting: deque[SeqOracle] = deque()
        self.running: deque[SeqOracle] = deque()
        self.scheduled: set[str] = set()
        self.kv = KVManagerOracle(kv_cache_size)

    # scheduler.py:52-58
    def add(self, seq: SeqOracle):
        seq.status = WAITING
        self.waiting.append(seq)
        self.by_id[seq.seq_id] = seq

    # scheduler.py:119-128
    def _grow(self, seq: SeqOracle):
        need = len(seq.token_ids) - len(seq.kv_indices)
        if need > 0:
            seq.kv_indices.extend(self.kv.alloc_slots(need))

    # scheduler.py:60-117
    def schedule(self):
        picked = []
        while self.waiting and len(picked) < self.max_bs:
            s = self.waiting.popleft()
            s.status = RUNNING
            self._grow(s)  # RuntimeError escapes, as in the reference
            picked.append(s)
            self.running.append(s)
            self.scheduled.add(s.seq_id)
        if picked:
            return BatchOracle(PREFILL, picked)

        seen = []
        while self.running and len(picked) < self.max_bs:
            s = self.running.popleft()
            seen.append(s)
            if s.seq_id in self.scheduled:
                continue
            ok = False
            while not ok:
                try:
                    self._grow(s)
                    ok = True
                except RuntimeError:
                    if self.running:
                        victim = self.running.pop()
                        self.preempt(victim)  # no-op: victim is no longer in `running`
                    else:
                        raise SchedulerLivelock(
                            "reference would spin: KV exhausted with nothing left to preempt")
            picked.append(s)
            self.scheduled.add(s.seq_id)
        self.running.extendleft(reversed(seen))
        if picked:
            return BatchOracle(DECODE, picked)
        return None

    # scheduler.py:140-153
    def preempt(self, seq: SeqOracle):
        if seq.status == FINISHED:
            return
        if not any(x is seq for x in self.running):
            return
        self.running.remove(seq)
        self.scheduled.discard(seq.seq_id)
        if seq.kv_indices:  # scheduler.py:130-138
            self.kv.free_slots(seq.kv_indices)
            seq.kv_indices.clear()
            seq.cached_kv_len = 0
        seq.status = WAITING
        self.waiting.appendleft(seq)

    # scheduler.py:161-171
    def update(self, seq: SeqOracle, token: int):
        if seq.status != RUNNING:
            return
        self.scheduled.remove(seq.seq_id)
        seq.cached_kv_len = len(seq.kv_indices)
        seq.token_ids.append(token)
        seq.num_tokens = len(seq.token_ids)

    # scheduler.py:173-190
    def finish(self, seq: SeqOracle):
        if seq.status == FINISHED:
            return
        if seq.status == WAITING:
            self.waiting.remove(seq)
        if seq.status == RUNNING:
            self.running.remove(seq)
            self.scheduled.discard(seq.seq_id)
        seq.status = FINISHED
        self.kv.cache_sequence(seq, finished=True)
        self.by_id.pop(seq.seq_id, None)

    def busy(self) -> bool:
        return bool(self.waiting or self.running)


class EngineOracle:
    """Scheduler + the step bookkeeping of engine.py (no model: tokens are supplied)."""

    def __init__(self, kv_cache_size: int, max_bs: int, context_len: int):
        self.context_len = context_len
        self.sched = SchedulerOracle(kv_cache_size, max_bs)

    # engine.py:50-69
    def add(self, seq_id: str, prompt, sampling: SamplingOracle):
        prompt = list(prompt)
        if len(prompt) > self.context_len:
            prompt = prompt[-self.context_len:]
        self.sched.add(SeqOracle(seq_id=seq_id, token_ids=prompt, prompt_len=len(prompt),
                                 sampling=sampling, num_tokens=len(prompt)))

    # engine.py:72-79
    def abort(self, seq_id: str):
        s = self.sched.by_id.get(seq_id)
        if s is not None:
            self.sched.finish(s)

    # engine.py:143-154
    @staticmethod
    def done(seq: SeqOracle):
        sp = seq.sampling
        if seq.token_ids[-1] == sp.eos_token_id and not sp.ignore_eos:
            return True, "STOP"
        if sp.max_tokens and seq.num_tokens >= sp.max_tokens:
            return True, "LENGTH"
        if sp.max_new_tokens and seq.num_tokens >= seq.prompt_len + sp.max_new_tokens:
            return True, "LENGTH"
        return False, None

    # engine.py:119-141
    def absorb(self, batch: BatchOracle, tokens):
        out = []
        for s, t in zip(batch.seqs, tokens):
            self.sched.update(s, t)
            fin, why = self.done(s)
            if fin:
                self.sched.finish(s)
            out.append((s.seq_id, t, fin, why, s.prompt_len, s.num_tokens - s.prompt_len))
        return out

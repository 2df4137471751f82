"""Continuous-batching scheduler: prefill-first, first-come-first-served, one KV slot per token.

API of nanovllmx/core/scheduler.py:13-193.  Decisions (which sequences run, in what order, which
slots they get) are bit-exact with the reference on every trace the reference can complete; they
are pinned by tests/golden/g2_sched_*.json.

Three departures, all where the reference cannot complete:
  * KV exhaustion during a decode step.  The reference pops its victim off `running` and then calls
    preempt(), whose "is it running?" test now fails, so nothing is freed, the victim is lost, and
    once `running` is empty the step loops forever (scheduler.py:84-104,146).  Here the victim is
    really preempted (slots freed, back to the head of `waiting`), which is what that code is
    written to do; if nothing is left to preempt the sequence stays queued and the step returns.
  * KV exhaustion at admission.  The reference lets the allocator's RuntimeError escape from
    schedule() (scheduler.py:71); here the prompt waits at the head of the queue while running
    sequences finish, and the error is raised only when nothing is running.
  * Queue membership is by object identity and O(1) per test, instead of field-wise dataclass
    comparison of whole token lists (scheduler.py:146,148,184).

One addition, off by default: `prefix_reuse=True` consults the radix tree when a sequence is admitted
(the reference builds the tree but never reads it at admission, so none of its requests reuses a
cached prefix).  The matched prefix's slots become the head of the sequence's block table, the prefill
computes only the rest, and the slot accounting at finish/preemption distinguishes shared from owned
slots.  With the flag off every decision is the reference's.
"""
from __future__ import annotations

from collections import deque
from typing import Callable, Deque, Optional

from .common import ForwardBatch, ForwardMode, Sequence, SequenceStatus
from .kv_cache import KVCacheManager


class Scheduler:
    def __init__(self, kv_cache_size: int, max_bs: int = 32, prefix_reuse: bool = False,
                 max_prefill_tokens: Optional[int] = None):
        self.kv_cache_size = kv_cache_size
        self.max_bs = max_bs
        self.prefix_reuse = prefix_reuse
        # Optional cap on the tokens one prefill step computes (the reference has none: a burst of long
        # prompts becomes one arbitrarily large batch, scheduler.py:67-80).  None keeps the reference's
        # behaviour; with a cap, admission stops before the step would exceed it (one sequence is always
        # admitted, however long).
        self.max_prefill_tokens = max_prefill_tokens
        self.sequences: dict = {}
        self.waiting: Deque[Sequence] = deque()
        self.running: Deque[Sequence] = deque()
        self.scheduled: set = set()            # ids picked for the step in flight
        self.kv_manager = KVCacheManager(size=kv_cache_size)
        self.on_release: Optional[Callable[[Sequence], None]] = None   # told when a sequence gives up its slots

    # -- intake -----------------------------------------------------------------------------------
    def add_sequence(self, seq: Sequence):
        seq.status = SequenceStatus.WAITING
        self.waiting.append(seq)
        self.sequences[seq.seq_id] = seq

    def get_sequence(self, sequence_id: str) -> Optional[Sequence]:
        return self.sequences.get(sequence_id)

    def has_unfinished_sequences(self) -> bool:
        return bool(self.waiting or self.running)

    # -- slots ------------------------------------------------------------------------------------
    def alloc_kv_slots(self, seq: Sequence):
        missing = len(seq.token_ids) - len(seq.kv_indices)
        if missing > 0:
            seq.kv_indices.extend(self.kv_manager.alloc_slots(missing))

    def free_kv_slots(self, seq: Sequence):
        if self.prefix_reuse:
            self.kv_manager.release_sequence(seq)
            return
        if seq.kv_indices:
            self.kv_manager.free_slots(seq.kv_indices)
            seq.kv_indices.clear()
            seq.cached_kv_len = 0

    # -- one step ---------------------------------------------------------------------------------
    def schedule(self) -> Optional[ForwardBatch]:
        picked = self._admit_waiting()
        if picked:
            return ForwardBatch(forward_mode=ForwardMode.PREFILL, num_seqs=len(picked), seqs=picked)
        picked = self._extend_running()
        if picked:
            return ForwardBatch(forward_mode=ForwardMode.DECODE, num_seqs=len(picked), seqs=picked)
        return None

    def _admit_waiting(self):
        """Up to max_bs queued sequences, each with slots for ALL its tokens; no token budget
        (scheduler.py:67-80).  When the pool cannot hold the next prompt the sequence goes back to the
        head of the queue and admission stops for this step (the reference lets the RuntimeError
        escape and the engine dies); it is re-raised only if nothing is running that could ever
        release memory."""
        picked = []
        budget = self.max_prefill_tokens
        while self.waiting and len(picked) < self.max_bs:
            if budget is not None and picked:
                head = self.waiting[0]
                if sum(len(s.token_ids) - s.cached_kv_len for s in picked) + len(head.token_ids) > budget:
                    break
            seq = self.waiting.popleft()
            try:
                if self.prefix_reuse and not seq.kv_indices:
                    self.kv_manager.adopt_cached_prefix(seq)
                self.alloc_kv_slots(seq)
            except RuntimeError:
                if self.prefix_reuse:
                    self.kv_manager.release_sequence(seq)
                self.waiting.appendleft(seq)
                if not picked and not self.running:
                    raise
                break
            seq.status = SequenceStatus.RUNNING
            picked.append(seq)
            self.running.append(seq)
            self.scheduled.add(seq.seq_id)
        return picked

    def _extend_running(self):
        """One more slot for every running sequence not already in flight (scheduler.py:82-115)."""
        running = self.running
        # Steady state (every running sequence fits in one batch, none is in flight, the pool has a slot
        # for each): one allocator call for the whole batch.  The slots come off the stack in the same
        # order as one call per sequence would take them, so the assignment is the reference's.
        if running and len(running) <= self.max_bs and not self.scheduled:
            need = [s for s in running if len(s.token_ids) > len(s.kv_indices)]
            if all(len(s.token_ids) - len(s.kv_indices) == 1 for s in need):
                got = self.kv_manager.kv_cache_allocator.alloc(len(need))
                if got is not None:
                    for s, slot in zip(need, got):
                        s.kv_indices.append(slot)
                    self.scheduled.update(s.seq_id for s in running)
                    return list(running)
        picked, passed = [], []
        while running and len(picked) < self.max_bs:
            seq = running.popleft()
            passed.append(seq)
            if seq.seq_id in self.scheduled:
                continue
            if self._grow_or_preempt(seq):
                picked.append(seq)
                self.scheduled.add(seq.seq_id)
            else:
                break
        running.extendleft(reversed(passed))
        return picked

    def _grow_or_preempt(self, seq: Sequence) -> bool:
        while True:
            try:
                self.alloc_kv_slots(seq)
                return True
            except RuntimeError:
                if not self.running:
                    return False       # nothing left to preempt: stays at the head of `running`
                self._preempt_detached(self.running.pop())

    # -- preemption -------------------------------------------------------------------------------
    def _preempt_detached(self, seq: Sequence):
        """Preempt a sequence that has already been taken off `running`."""
        self.scheduled.discard(seq.seq_id)
        self.free_kv_slots(seq)
        seq.status = SequenceStatus.WAITING
        self.waiting.appendleft(seq)
        if self.on_release:
            self.on_release(seq)

    def preempt(self, seq: Sequence):
        """Send a running sequence back to the head of the waiting queue; its KV rows are dropped and
        will be recomputed by a new prefill over prompt + generated tokens (scheduler.py:140-153)."""
        if seq.status == SequenceStatus.FINISHED or not any(s is seq for s in self.running):
            return
        self.running.remove(seq)
        self._preempt_detached(seq)

    # -- results ----------------------------------------------------------------------------------
    def update_sequence(self, seq: Sequence, new_token_id: int):
        if seq.status != SequenceStatus.RUNNING:
            return
        self.scheduled.remove(seq.seq_id)
        seq.cached_kv_len = len(seq.kv_indices)
        seq.token_ids.append(new_token_id)
        seq.num_tokens = len(seq.token_ids)

    def finish_sequence(self, seq: Sequence):
        """Retire a sequence; its tokens and slots go to the prefix tree (scheduler.py:173-190)."""
        if seq.status == SequenceStatus.FINISHED:
            return
        if seq.status == SequenceStatus.WAITING:
            self.waiting.remove(seq)
        elif seq.status == SequenceStatus.RUNNING:
            self.running.remove(seq)
            self.scheduled.discard(seq.seq_id)
        seq.status = SequenceStatus.FINISHED
        if self.prefix_reuse:
            self.kv_manager.cache_sequence_shared(seq)
        else:
            self.kv_manager.cache_sequence(seq)
        self.sequences.pop(seq.seq_id, None)
        if self.on_release:
            self.on_release(seq)

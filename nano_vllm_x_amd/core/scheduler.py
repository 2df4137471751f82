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

A second addition, also off by default: `chunked_prefill=True` (with `max_prefill_tokens`) lets a prompt
that does not fit the step's token budget be computed over several prefill steps.  The sequence holds all
its slots from admission (as every admitted sequence does here); a chunk step presents a truncated copy of
it to the model -- tokens and slots up to the chunk's end, `cached_kv_len` at the chunk's start -- so the
prefill path that serves cached prefixes computes exactly that range, and the token the model samples for
a truncated copy is dropped.  The last chunk is an ordinary prefill of the real sequence.  Chunks do not
mix with decode rows (the policy stays prefill-first).
"""
from __future__ import annotations

from collections import deque
from typing import Callable, Deque, Optional

from .common import ForwardBatch, ForwardMode, Sequence, SequenceStatus
from .kv_cache import KVCacheManager


class Scheduler:
    def __init__(self, kv_cache_size: int, max_bs: int = 32, prefix_reuse: bool = False,
                 max_prefill_tokens: Optional[int] = None, chunked_prefill: bool = False):
        self.kv_cache_size = kv_cache_size
        self.max_bs = max_bs
        self.prefix_reuse = prefix_reuse
        # Optional cap on the tokens one prefill step computes (the reference has none: a burst of long
        # prompts becomes one arbitrarily large batch, scheduler.py:67-80).  None keeps the reference's
        # behaviour; with a cap, admission stops before the step would exceed it (one sequence is always
        # admitted, however long).
        self.max_prefill_tokens = max_prefill_tokens
        if chunked_prefill and not max_prefill_tokens:
            raise ValueError("chunked_prefill needs max_prefill_tokens (the size of a chunk step)")
        self.chunked_prefill = chunked_prefill
        self.prefilling: dict = {}             # seq_id -> sequence whose prompt is only partly computed (admission order)
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
        if self.chunked_prefill:
            picked = self._schedule_chunks()
            if picked:
                return ForwardBatch(forward_mode=ForwardMode.PREFILL, num_seqs=len(picked), seqs=picked)
            if self.prefilling:
                return None                    # every partly computed prompt is in flight
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

    # -- chunked prefill -----------------------------------------------------------------------------
    @staticmethod
    def _chunk_view(seq: Sequence, end: int) -> Sequence:
        """What the model is shown for a chunk that stops before the prompt does: the sequence cut at `end`."""
        view = Sequence(seq_id=seq.seq_id, status=SequenceStatus.RUNNING, num_tokens=end, prompt_len=seq.prompt_len,
                        token_ids=seq.token_ids[:end], sampling_params=seq.sampling_params, kv_indices=seq.kv_indices[:end],
                        cached_kv_len=seq.cached_kv_len)
        view.partial_of = seq
        return view

    def _schedule_chunks(self):
        """One prefill step of at most max_prefill_tokens computed tokens: first the next chunks of prompts
        already under way, then new admissions, the last of which may itself be cut."""
        budget = self.max_prefill_tokens
        picked = []
        for seq in list(self.prefilling.values()):
            if budget <= 0 or len(picked) >= self.max_bs:
                break
            if seq.seq_id in self.scheduled:
                continue
            left = len(seq.token_ids) - seq.cached_kv_len
            take = min(left, budget)
            budget -= take
            self.scheduled.add(seq.seq_id)
            picked.append(seq if take == left else self._chunk_view(seq, seq.cached_kv_len + take))
        while self.waiting and len(picked) < self.max_bs and budget > 0:
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
            self.running.append(seq)
            self.scheduled.add(seq.seq_id)
            left = len(seq.token_ids) - seq.cached_kv_len
            take = min(left, budget)
            budget -= take
            if take == left:
                picked.append(seq)
            else:
                self.prefilling[seq.seq_id] = seq
                picked.append(self._chunk_view(seq, seq.cached_kv_len + take))
        return picked

    def finish_chunk(self, view: Sequence):
        """A chunk step has run: the rows up to the view's end are written; the sampled token is dropped."""
        seq = view.partial_of
        self.scheduled.discard(seq.seq_id)
        if seq.status == SequenceStatus.RUNNING and seq.seq_id in self.prefilling:
            seq.cached_kv_len = len(view.token_ids)

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
                # the victim is the youngest running sequence that is not in flight (with one batch in flight
                # none of `running` is here: the picked ones were taken off it; with pipeline stages an
                # earlier batch's sequences are still being computed and keep their slots)
                victim = next((s for s in reversed(self.running) if s.seq_id not in self.scheduled), None)
                if victim is None:
                    return False       # nothing left to preempt: stays at the head of `running`
                self.running.remove(victim)
                self._preempt_detached(victim)

    # -- preemption -------------------------------------------------------------------------------
    def _preempt_detached(self, seq: Sequence):
        """Preempt a sequence that has already been taken off `running`."""
        self.scheduled.discard(seq.seq_id)
        self.prefilling.pop(seq.seq_id, None)
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
        if self.prefilling:
            self.prefilling.pop(seq.seq_id, None)   # its last chunk has run
        seq.cached_kv_len = len(seq.kv_indices)
        seq.token_ids.append(new_token_id)
        seq.num_tokens = len(seq.token_ids)

    def detach_finishing(self, seq: Sequence):
        """First half of finish_sequence for a running sequence whose last token is known to be its last but
        not yet known by value: it leaves `running` now (the next step is scheduled without it) and
        finish_sequence(seq, detached=True) does the rest once the token is in place."""
        self.running.remove(seq)
        self.scheduled.discard(seq.seq_id)

    def finish_sequence(self, seq: Sequence, detached: bool = False):
        """Retire a sequence; its tokens and slots go to the prefix tree (scheduler.py:173-190)."""
        if seq.status == SequenceStatus.FINISHED:
            return
        if detached:
            pass                      # already off the queues (detach_finishing)
        elif seq.status == SequenceStatus.WAITING:
            self.waiting.remove(seq)
        elif seq.status == SequenceStatus.RUNNING:
            self.running.remove(seq)
            self.scheduled.discard(seq.seq_id)
        seq.status = SequenceStatus.FINISHED
        if self.prefilling.pop(seq.seq_id, None) is not None:
            # aborted with its prompt only partly computed: the rows past cached_kv_len hold nothing, so the
            # sequence gives its slots back instead of entering the prefix tree
            if self.prefix_reuse:
                self.kv_manager.release_sequence(seq)
            else:
                self.kv_manager.free_slots(seq.kv_indices)
                seq.kv_indices.clear()
                seq.cached_kv_len = 0
        elif self.prefix_reuse:
            self.kv_manager.cache_sequence_shared(seq)
        else:
            self.kv_manager.cache_sequence(seq)
        self.sequences.pop(seq.seq_id, None)
        if self.on_release:
            self.on_release(seq)

"""Step loop: schedule -> run the model -> fold the sampled tokens back into the sequences.

`EngineCore` is the model-free half (admission, finish rules; nanovllmx/core/engine.py:50-79,
119-154); `Engine` adds the in-process model runner and `step()` (engine.py:81-100).  The reference
spreads this over an engine process and one worker process per GPU joined by pickled queues
(engine_client.py, executor.py, worker_client.py); here every rank runs the same deterministic loop
in its own process (SPMD), so a step needs no host-side message at all -- only the collectives inside
the model and one broadcast of the sampled ids.
"""
from __future__ import annotations

from collections import deque
from dataclasses import replace
from typing import Optional

from .common import EngineOutput, FinishReason, ForwardBatch, SamplingParams, Sequence, SequenceStatus
from .scheduler import Scheduler


class EngineCore:
    def __init__(self, kv_cache_size: int, max_bs: int, context_len: int = 2048, prefix_reuse: bool = False,
                 max_prefill_tokens: Optional[int] = None, stop_at_context_len: bool = False, chunked_prefill: bool = False):
        self.context_len = context_len
        self.prefix_reuse = prefix_reuse
        self.max_prefill_tokens = max_prefill_tokens
        # The reference has no stop rule tied to the context length: a request without max_tokens and with
        # ignore_eos runs until the pool is exhausted.  Opt-in: finish with LENGTH once a sequence is
        # context_len tokens long (default off = the reference's rules only).
        self.stop_at_context_len = stop_at_context_len
        self.chunked_prefill = chunked_prefill
        self.scheduler = Scheduler(kv_cache_size=kv_cache_size, max_bs=max_bs, prefix_reuse=prefix_reuse,
                                   max_prefill_tokens=max_prefill_tokens, chunked_prefill=chunked_prefill)

    def add_sequence(self, sequence_id: str, prompt_token_ids: list, sampling_params: SamplingParams):
        if len(prompt_token_ids) > self.context_len:   # keep the tail (engine.py:59-60)
            prompt_token_ids = prompt_token_ids[-self.context_len:]
        n = len(prompt_token_ids)
        # the engine keeps its own copy of the parameters: the runner caches their device form while a batch holds the
        # same objects, so a caller that edits its object after submission must not reach a running sequence
        sampling_params = replace(sampling_params)
        self.scheduler.add_sequence(Sequence(seq_id=sequence_id, token_ids=prompt_token_ids, num_tokens=n,
                                             prompt_len=n, sampling_params=sampling_params))

    def abort_sequence(self, sequence_id: str):
        seq = self.scheduler.get_sequence(sequence_id)
        if seq is not None:
            self.scheduler.finish_sequence(seq)

    def has_unfinished_sequences(self) -> bool:
        return self.scheduler.has_unfinished_sequences()

    def _is_sequence_finished(self, seq: Sequence):
        """Stop rules in the reference's order (engine.py:143-154).  NB `max_tokens` caps the TOTAL
        length, prompt included.  The optional context-length rule comes last."""
        sp = seq.sampling_params
        if not sp.ignore_eos and seq.token_ids[-1] == sp.eos_token_id:
            return True, FinishReason.STOP
        if sp.max_tokens and seq.num_tokens >= sp.max_tokens:
            return True, FinishReason.LENGTH
        if sp.max_new_tokens and seq.num_tokens >= seq.prompt_len + sp.max_new_tokens:
            return True, FinishReason.LENGTH
        if self.stop_at_context_len and seq.num_tokens >= self.context_len:
            return True, FinishReason.LENGTH
        return False, None

    def _length_limit(self, seq: Sequence) -> int:
        """Smallest total length at which one of the LENGTH rules of _is_sequence_finished fires (0 = none).
        A sequence's sampling parameters never change, so this is computed once per sequence."""
        sp = seq.sampling_params
        limits = []
        if sp.max_tokens:
            limits.append(sp.max_tokens)
        if sp.max_new_tokens:
            limits.append(seq.prompt_len + sp.max_new_tokens)
        if self.stop_at_context_len:
            limits.append(self.context_len)
        return min(limits) if limits else 0

    def fold_output(self, batch: ForwardBatch, output_ids: list) -> list:
        """Apply one step's sampled ids to the sequences (Scheduler.update_sequence, the stop rules of
        _is_sequence_finished and Scheduler.finish_sequence, engine.py:119-141) and return one
        (seq_id, token, finished, reason, prompt tokens, generated tokens) tuple per sequence.  This is
        the host work that stands between two device steps, so the per-sequence part is written out
        in line: same decisions in the same order as the three calls it replaces."""
        outs = []
        sched = self.scheduler
        scheduled, prefilling, finish = sched.scheduled, sched.prefilling, sched.finish_sequence
        running = SequenceStatus.RUNNING
        for seq, tok in zip(batch.seqs, output_ids):
            if seq.partial_of is not None:   # a chunk that stops before the prompt does: no token yet
                sched.finish_chunk(seq)
                continue
            toks = seq.token_ids
            if seq.status is running:
                scheduled.remove(seq.seq_id)
                if prefilling:
                    prefilling.pop(seq.seq_id, None)   # its last chunk has run
                seq.cached_kv_len = len(seq.kv_indices)
                toks.append(tok)
                seq.num_tokens = len(toks)
            n = seq.num_tokens
            limit = seq.length_limit
            if limit is None:
                limit = seq.length_limit = self._length_limit(seq)
            sp = seq.sampling_params
            if not sp.ignore_eos and toks[-1] == sp.eos_token_id:
                why = FinishReason.STOP
            elif limit and n >= limit:
                why = FinishReason.LENGTH
            else:
                outs.append((seq.seq_id, tok, False, None, seq.prompt_len, n - seq.prompt_len))
                continue
            finish(seq)
            outs.append((seq.seq_id, tok, True, why, seq.prompt_len, n - seq.prompt_len))
        return outs

    def update_from_output(self, batch: ForwardBatch, output_ids: list) -> list:
        return [EngineOutput(*t) for t in self.fold_output(batch, output_ids)]


class Engine(EngineCore):
    """Engine with the reference's nine constructor arguments (engine.py:13-24, llm.py:17-28)."""

    def __init__(self, model: str, gpu_memory_utilization: float, max_bs: int, tp_size: int = 1, pp_size: int = 1,
                 nccl_port: int = 29500, device_ids: Optional[list] = None, enforce_eager: bool = False,
                 context_len: int = 2048, kv_cache_size: Optional[int] = None, seed: int = 0, prefix_reuse: bool = False,
                 max_prefill_tokens: Optional[int] = None, stop_at_context_len: bool = False, chunked_prefill: bool = False,
                 run_ahead: bool = False):
        from .model_runner import ModelRunner

        # pp_size > 1: every rank runs this same engine and scheduler.  Time advances in ticks: at tick t
        # the scheduler picks batch t (among the sequences not in flight), stage r runs batch t - r, and the
        # ids of batch t - (pp_size - 1), which the last stage has just finished, reach every rank -- up to
        # pp_size batches in flight, as engine.py:102-117 keeps them
        self.model_runner = ModelRunner(model=model, max_bs=max_bs, tp_size=tp_size, nccl_port=nccl_port,
                                        device_ids=device_ids, enforce_eager=enforce_eager, context_len=context_len,
                                        seed=seed, pp_size=pp_size)
        self.model_runner.load_model()
        if kv_cache_size is None:
            kv_cache_size = self.model_runner.profile_kv_cache_size(gpu_memory_utilization)
        self.kv_cache_size = kv_cache_size
        self.model_runner.initialize_kv_cache(kv_cache_size)
        super().__init__(kv_cache_size=kv_cache_size, max_bs=max_bs, context_len=context_len, prefix_reuse=prefix_reuse,
                         max_prefill_tokens=max_prefill_tokens, stop_at_context_len=stop_at_context_len,
                         chunked_prefill=chunked_prefill)
        self.scheduler.on_release = self.model_runner.release_sequence
        self._inflight = None
        # run_ahead: step_overlapped() queues decode step N+1 behind step N before N's ids have reached the
        # host, whenever N's outcome cannot change what N+1 is (see _launch_ahead)
        self.run_ahead = run_ahead and pp_size == 1
        self._ahead = None            # handle of the step queued behind the one in flight
        self._placeholders = None     # in-flight step already folded with placeholder tokens: their positions
        self._late_aborts: list = []
        self.ahead_steps = 0         # steps queued ahead
        self.ahead_retired = 0       # sequences retired by a step that had another queued behind it
        self.pp_size = pp_size
        self._pp_window: deque = deque(maxlen=max(pp_size, 1))   # the batches of the last pp_size ticks, newest last
        self._pp_aborts: list = []
        self.pp_stats = {"ticks": 0, "stage_busy_ticks": 0}

    def step(self) -> list:
        if self._inflight is None and not self.begin_step():
            return []
        return self.end_step()

    # The two halves of a step, for a driver that wants the host free while the device works
    # (LLM._run_engine hands out the previous step's tokens between them).
    def begin_step(self) -> bool:
        """Schedule a batch and queue it on the device.  False if there is nothing to run."""
        assert self._inflight is None
        if self.pp_size > 1:
            return self._begin_tick()
        batch = self.scheduler.schedule()
        if not batch:
            return False
        self._inflight = self.model_runner.launch_step(batch)
        return True

    def _begin_tick(self, admit: bool = True) -> bool:
        """`admit=False` only moves the batches already in flight on by one stage (draining)."""
        from .. import distributed as tpd

        win = self._pp_window
        win.append((self.scheduler.schedule() or None) if admit else None)      # a tick without a new batch is a bubble
        if all(b is None for b in win):
            win.clear()
            return False
        r = tpd.pp_rank()
        mine = win[-1 - r] if len(win) > r else None
        out = win[0] if len(win) == self.pp_size else None
        self.pp_stats["ticks"] += 1
        self.pp_stats["stage_busy_ticks"] += mine is not None
        self._inflight = self.model_runner.launch_tick(mine, out)
        return True

    def end_step(self) -> list:
        """Wait for the step in flight and fold its tokens into the sequences."""
        return [EngineOutput(*t) for t in self._finish_inflight()]

    def step_overlapped(self) -> list:
        """step() for a driver that keeps the device busy: wait for the step in flight (or run one), fold
        its tokens in, queue the NEXT step, and only then build the output objects -- whatever the host
        does between the arrival of one step's ids and the launch of the next is time the device idles.
        With run_ahead the next step is queued even before the wait, where that is possible."""
        if self._inflight is None and not self.begin_step():
            return []
        if self.run_ahead and self._ahead is None:
            self._launch_ahead()
        raw = self._finish_inflight()
        if self._inflight is None and self.has_unfinished_sequences():
            self.begin_step()
        return [EngineOutput(*t) for t in raw]

    PLACEHOLDER = -1     # stands in a sequence's token list for an id that is still on the device

    def _launch_ahead(self) -> bool:
        """Queue decode step N+1 while step N (self._inflight) is running, if N+1 is fully determined
        without N's ids: N is a graph decode step over ALL running sequences, nothing waits to be
        admitted, no sequence of N can stop on its token (ignore_eos), and the pool has a free slot for
        every sequence that goes on (no eviction, no preemption).  Sequences that reach a length limit
        with N's token leave the queues now and are retired when the token is in place (the retirement
        only feeds the prefix tree, which nothing reads or evicts from in between; with prefix_reuse it
        could also free slots, so such steps are not queued ahead then).  N+1 has the remaining
        sequences in the same order, one more slot each -- the scheduler is run as usual and must pick
        exactly that -- and its input ids are N's sampled ids, which the first node of the decode graph
        reads on the device.  The sequences carry a placeholder for N's token until finish_step(N)."""
        handle = self._inflight
        batch = handle[0] if handle is not None else None
        sched, runner = self.scheduler, self.model_runner
        if batch is None or self._placeholders is not None or self._late_aborts or not runner.graph_step_possible(batch):
            return False
        n = batch.num_seqs
        if sched.waiting or sched.prefilling or len(sched.running) != n or len(sched.scheduled) != n:
            return False
        cap = runner.block_table.row_capacity
        running = SequenceStatus.RUNNING
        rows_on, leaving = [], set()
        for j, seq in enumerate(batch.seqs):
            limit = seq.length_limit
            if limit is None:
                limit = seq.length_limit = self._length_limit(seq)
            if seq.status is not running or not seq.sampling_params.ignore_eos:
                return False
            if limit and seq.num_tokens + 1 >= limit:
                leaving.add(j)
            elif len(seq.kv_indices) + 1 > cap:
                return False
            else:
                rows_on.append(j)
        if not rows_on or (leaving and self.prefix_reuse) or sched.kv_manager.kv_cache_allocator.num_free() < len(rows_on):
            return False
        # fold step N with placeholders: what Scheduler.update_sequence does, minus the token's value
        scheduled = sched.scheduled
        spots = []
        for j, seq in enumerate(batch.seqs):
            scheduled.remove(seq.seq_id)
            seq.cached_kv_len = len(seq.kv_indices)
            spots.append(len(seq.token_ids))
            seq.token_ids.append(self.PLACEHOLDER)
            seq.num_tokens = len(seq.token_ids)
            if j in leaving:
                sched.detach_finishing(seq)
        nxt = sched.schedule()
        assert nxt is not None and nxt.forward_mode == batch.forward_mode and len(nxt.seqs) == len(rows_on) \
            and all(a is batch.seqs[j] for a, j in zip(nxt.seqs, rows_on)), "the step queued ahead must continue the same sequences"
        self._placeholders = (spots, leaving)
        self._ahead = runner.launch_step(nxt, ids_on_device=rows_on)
        self.ahead_steps += 1
        return True

    def _finish_inflight(self) -> list:
        handle, self._inflight = self._inflight, None
        if handle is None:
            return []
        ids = self.model_runner.finish_step(handle)
        if self._placeholders is not None:
            # folded already, with placeholders: put the ids in and retire the sequences that were known to end here
            (spots, leaving), self._placeholders = self._placeholders, None
            raw = []
            for j, (seq, tok, at) in enumerate(zip(handle[0].seqs, ids, spots)):
                seq.token_ids[at] = tok
                if j in leaving:
                    self.scheduler.finish_sequence(seq, detached=True)
                    self.ahead_retired += 1
                    raw.append((seq.seq_id, tok, True, FinishReason.LENGTH, seq.prompt_len, at + 1 - seq.prompt_len))
                else:
                    raw.append((seq.seq_id, tok, False, None, seq.prompt_len, at + 1 - seq.prompt_len))
            self._inflight, self._ahead = self._ahead, None
        else:
            raw = self.fold_output(handle[0], ids) if handle[0] is not None else []
        if self._pp_aborts or self._late_aborts:
            waiting, self._pp_aborts, self._late_aborts = self._pp_aborts + self._late_aborts, [], []
            for seq_id in waiting:
                self.abort_sequence(seq_id)
        return raw

    def abort_sequence(self, sequence_id: str):
        """With batches in flight across stages an abort waits until the sequence's step has come back:
        a later stage still has to run the batch exactly as the earlier stages saw it."""
        if self.pp_size > 1 and sequence_id in self.scheduler.scheduled:
            self._pp_aborts.append(sequence_id)
            return
        if self._placeholders is not None:    # its token list holds a placeholder until the step in flight is back
            self._late_aborts.append(sequence_id)
            return
        super().abort_sequence(sequence_id)

    @property
    def step_in_flight(self) -> bool:
        return self._inflight is not None or self._ahead is not None or any(b is not None for b in self._pp_window)

    def reset(self):
        """Forget every sequence and every cached prefix (fresh scheduler, same pool and graphs)."""
        for handle in (self._inflight, self._ahead):
            if handle is not None:
                self.model_runner.finish_step(handle)
        self._inflight = self._ahead = self._placeholders = None
        self._late_aborts = []
        while self.pp_size > 1 and self._begin_tick(admit=False):   # drain: every stage must see every batch in flight
            self.end_step()
        for seq_id in list(self.model_runner.block_table.row_of):
            self.model_runner.block_table.release(seq_id)
        self.model_runner._sp_key = None
        self.scheduler = Scheduler(kv_cache_size=self.kv_cache_size, max_bs=self.scheduler.max_bs, prefix_reuse=self.prefix_reuse,
                                   max_prefill_tokens=self.max_prefill_tokens, chunked_prefill=self.chunked_prefill)
        self.scheduler.on_release = self.model_runner.release_sequence

    def wait_until_ready(self):
        return True

    def shutdown(self):
        self.model_runner.shutdown()

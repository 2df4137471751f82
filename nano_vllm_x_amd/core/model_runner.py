"""Per-rank model runner: weights, KV pool, per-step input preparation, eager and graph execution.

Interface of nanovllmx/core/model_runner.py (`load_model` :79-131, `initialize_kv_cache` :133-157,
`profile_kv_cache_size` :159-173, `prepare_input` :175-189, `prepare_sampling_params` :191-219,
`prepare_last_hidden_states` :221-227, `execute_model` :229-274, `capture_graph` :332-417).

What differs, by design, on MI355X:
  * a decode step is ONE hipGraph launch that starts with the block-table update and the split plan
    (both device side), so the host uploads 4 integers per sequence and nothing else;
  * list -> tensor conversion is numpy on whole batches into one staging buffer;
  * the sampled ids are broadcast from rank 0, so every rank's scheduler stays in lockstep without
    any host-side message passing.
"""
from __future__ import annotations

import gc
import os
import sys
import time
from typing import Optional

import numpy as np
import torch

from .. import distributed as tpd
from .. import ops
from ..layers.attention import AttentionBackend, AttentionMetadata, BlockTable, PagedDecodeWrapper, attention_kv_cache
from ..layers.sampler import Sampler
from ..qwen3 import Qwen3Config, Qwen3ForCausalLM, load_config
from .common import ForwardBatch, ForwardMode
from .hip_graph import HIPGraph, decode_batch_ladder
from .kv_cache import KVCachePool

# Largest step whose LM head + draw go through ops.lmhead_sample: it holds 64 rows per compute unit, takes 65..128 rows as two
# row groups on paired workgroups in one launch (a tile fetched from HBM by one is found in L2 by the other), and more rows as
# passes that follow one another, where the library projection + sampler is the faster pair (Qwen3-0.6B, us: 60 vs 86 at 29
# rows, 74 vs 97 at 64, 88 vs 109 at 96, 105 vs 119 at 123, 215 vs 155 at 192; tools/kernel_bench.py lmhead)
LMHEAD_FUSED_MAX_ROWS = int(os.environ.get("NVX_LMHEAD_FUSED_MAX_ROWS", "128"))

_DTYPES = {"half": torch.float16, "float16": torch.float16, "float": torch.float32, "float32": torch.float32,
           "bfloat16": torch.bfloat16}


def get_model_config_per_gpu(cfg: Qwen3Config, tp_size: int):
    """Per-rank head counts (model_runner.py:43-44): query heads split, kv heads split or replicated."""
    return cfg.num_attention_heads // tp_size, max(1, cfg.num_key_value_heads // tp_size), cfg.head_dim


class ModelRunner:
    def __init__(self, model: str, max_bs: int, tp_size: int = 1, nccl_port: int = 29500,
                 device_ids: Optional[list] = None, enforce_eager: bool = False, context_len: int = 2048,
                 seed: int = 0, config: Optional[Qwen3Config] = None, pp_size: int = 1):
        if not torch.cuda.is_available():
            raise RuntimeError("ModelRunner needs an MI355X: the HIP path has no CPU fallback")
        self.model_path, self.max_bs, self.context_len = model, max_bs, context_len
        # a pipeline stage waits for its neighbour inside a step, which a captured graph cannot do: stages
        # launch their kernels one by one
        self.enforce_eager, self.seed = enforce_eager or pp_size > 1, seed
        self.rank = tpd.init_tensor_parallel(tp_size, port=nccl_port, device_ids=device_ids, pp_size=pp_size)
        self.tp_size, self.pp_size = tp_size, pp_size
        if tp_size * pp_size == 1:
            torch.cuda.set_device(device_ids[0] if device_ids else 0)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._config_override = config
        self.graph: Optional[HIPGraph] = None
        self.block_table: Optional[BlockTable] = None
        self._sp_key, self._sp_dev, self._sp_unfiltered = None, None, False
        self._stage_output, self._defer_stage_send = None, False
        self._last_done = 0.0
        self.stats = {"prefill_steps": 0, "decode_steps": 0, "graph_steps": 0, "eager_decode_steps": 0, "prefill_s": 0.0, "decode_s": 0.0}
        self.graph_sizes_dropped: list = []               # decode batch sizes whose graph was slower than a larger one's
        self.graph_step_us_by_size: dict = {}
        self.graph_capture_error: Optional[str] = None    # set when a tensor-parallel decode graph could not be captured
        # NVX_TP_SAMPLE=gather: tensor-parallel ranks always gather the logits and sample whole rows, as the reference does
        self.shard_sampling = os.environ.get("NVX_TP_SAMPLE", "shard") != "gather"

    # -- set-up -----------------------------------------------------------------------------------
    def load_model(self, state_dict_iter=None):
        self.hf_config = self._config_override or load_config(self.model_path)
        cfg = self.hf_config
        # The kernels are built for bfloat16 and for float16 (the reference resolves the dtype from the checkpoint and
        # defaults to float16, model_runner.py:14-23,38-39); a float32 checkpoint is served in bfloat16 (same exponent
        # range), its tensors converted as they are loaded.
        self.checkpoint_dtype = _DTYPES[cfg.torch_dtype]
        self.dtype = self.checkpoint_dtype if self.checkpoint_dtype in (torch.bfloat16, torch.float16) else torch.bfloat16
        if self.checkpoint_dtype != self.dtype and self.rank == 0:
            print(f"[nano_vllm_x_amd] checkpoint dtype {cfg.torch_dtype}: weights are converted to bfloat16", file=sys.stderr, flush=True)
        self.num_heads, self.num_kv_heads, self.head_dim = get_model_config_per_gpu(cfg, self.tp_size)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(self.dtype)
        try:
            with torch.device(self.device):
                self.model = Qwen3ForCausalLM(cfg)
        finally:
            torch.set_default_dtype(prev)
        self.model.finalize()
        self.model.eval()
        self.start_layer, self.end_layer = self.model.model.start_layer, self.model.model.end_layer
        self.num_layers = self.end_layer - self.start_layer     # of this pipeline stage
        if state_dict_iter is not None:
            self.model.load_weights(state_dict_iter)
        else:
            files = _safetensors_files(self.model_path)
            if files:
                self.model.load_weights(_iter_safetensors(files))
            else:
                self.model.init_random(self.seed)
        self.model.pack_decode_weights()
        self.sampler = Sampler(seed=self.seed)
        # Decode-sized steps whose rows have no sampling filter take the LM head and the draw in ONE pass over the weight
        # (ops.lmhead_sample): the tiled copy of this rank's LM-head rows it streams is made here, once.  NVX_LMHEAD_FUSED=0
        # keeps the two-launch path (library projection + sampler) everywhere.
        self._lmhead_packed = None
        if (os.environ.get("NVX_LMHEAD_FUSED", "1") != "0" and tpd.pp_is_last() and cfg.hidden_size == 1024
                and self.dtype in (torch.bfloat16, torch.float16)):
            w = self.model.lm_head.weight
            self._lmhead_packed = ops.lmhead_pack(w.detach())
            self._lmhead_rows = w.shape[0]
            self._lmhead_ws = ops.lmhead_workspace(max(self.max_bs, 1), self.device)
        # tensor-parallel ranks: all-reduces of decode-sized tensors go peer to peer (set up and proven here, before
        # the decode graph is captured; every rank falls back to the library all-reduce together if it cannot be)
        tpd.init_peer_allreduce(self.max_bs, cfg.hidden_size, self.dtype, self.device)

    def profile_kv_cache_size(self, gpu_memory_utilization: float = 0.9) -> int:
        per_token = self.num_layers * self.num_kv_heads * self.head_dim * 2 * self.dtype.itemsize
        gc.collect()
        torch.cuda.empty_cache()
        free, _total = torch.cuda.mem_get_info(self.device)
        # ranks that share a device (rehearsals of several ranks on a one-GPU box) share its free memory too
        n = int(free * gpu_memory_utilization / tpd.ranks_on_my_device(self.device)) // per_token
        return tpd.tp_all_gather_min_int(n, self.device)

    def initialize_kv_cache(self, kv_cache_size: int):
        self.kv_cache_size = kv_cache_size
        self.kv_cache = KVCachePool(dtype=self.dtype, device=self.device, num_tokens=kv_cache_size,
                                    num_layers=self.num_layers, num_heads=self.num_kv_heads, head_dim=self.head_dim,
                                    start_layer=self.start_layer, end_layer=self.end_layer)
        self.attn_backend = AttentionBackend(kv_cache=self.kv_cache, num_qo_heads=self.num_heads,
                                             num_kv_heads=self.num_kv_heads, head_dim=self.head_dim, dtype=self.dtype,
                                             device=self.device)
        n_rows = max(1024, 4 * self.max_bs) + 1
        self.block_table = BlockTable(self.max_bs, n_rows, 2 * self.context_len, self.device)
        self.decode_ws = ops.DecodeWorkspace(self.max_bs, self.num_heads, self.num_kv_heads, self.device)
        self._sample_out = torch.zeros(self.max_bs, dtype=torch.int64, device=self.device)
        # two pinned landing buffers with an event each: a step queued behind another must not overwrite the
        # ids the host has not read yet, and the host waits for ITS step's ids, not for the whole stream
        self._sample_hosts = [torch.zeros(self.max_bs, dtype=torch.int64, pin_memory=True) for _ in range(2)]
        self._sample_events = [torch.cuda.Event() for _ in range(2)]
        self._sample_turn = 0
        if not self.enforce_eager:
            self.graph = HIPGraph()
            try:
                self.capture_graph()
            except RuntimeError as e:
                if self.tp_size == 1 or os.environ.get("NVX_REQUIRE_GRAPH") == "1":
                    raise
                self.graph_capture_error = str(e).splitlines()[0]
                # a collective that cannot be captured on this RCCL build: every rank hits the same error at
                # the same point, so all fall back together to eager decode steps (same kernels, launched
                # one by one)
                print(f"[nano_vllm_x_amd] rank {self.rank}: decode graph capture failed ({e}); running decode eagerly",
                      file=sys.stderr, flush=True)
                self.graph = None
                torch.cuda.synchronize()

    # -- per-step host preparation -------------------------------------------------------------------
    def prepare_input(self, batch: ForwardBatch):
        ids = [t for s in batch.seqs for t in s.token_ids[s.cached_kv_len:]]
        pos = np.concatenate([np.arange(s.cached_kv_len, len(s.token_ids), dtype=np.int64) for s in batch.seqs])
        ids_t = torch.from_numpy(np.asarray(ids, dtype=np.int64)).to(self.device, non_blocking=True)
        return ids_t, torch.from_numpy(pos).to(self.device, non_blocking=True)

    def prepare_sampling_params(self, batch: ForwardBatch):
        """One (4, B) fp32 upload: temperature, min_p, top_p, top_k (-1 -> vocabulary size,
        model_runner.py:202-206).  Returns the four per-row device tensors.  A sequence's parameters never
        change, so the upload is skipped while the batch holds the same parameter objects, in the same order,
        as the step before (sequence ids do not identify them: a caller may reuse an id for a new request)."""
        key = [s.sampling_params for s in batch.seqs]    # the objects themselves: an id string can be reused
        old = self._sp_key
        if old is not None and len(old) == len(key) and all(a is b for a, b in zip(old, key)):
            return self._sp_dev
        v = self.hf_config.vocab_size
        n = batch.num_seqs
        host = np.empty((4, n), dtype=np.float32)
        for i, s in enumerate(batch.seqs):
            sp = s.sampling_params
            host[0, i], host[1, i], host[2, i] = sp.temperature, sp.min_p, sp.top_p
            host[3, i] = v if sp.top_k == -1 else min(sp.top_k, v)
        # no row with a filter (greedy rows, temperature <= 0, ignore theirs): tensor-parallel ranks can then draw from
        # their own shards of the vocabulary and exchange 8 bytes per row instead of the logits
        self._sp_unfiltered = bool(np.all((host[0] <= 0) | ((host[3] >= v) & (host[2] >= 1.0) & (host[1] <= 0.0))))
        dev = torch.from_numpy(host).to(self.device, non_blocking=True)
        self._sp_key, self._sp_dev = key, (dev[0], dev[1], dev[2], dev[3].to(torch.int64))
        return self._sp_dev

    @staticmethod
    def prepare_last_hidden_states(batch: ForwardBatch, hidden_states: torch.Tensor) -> torch.Tensor:
        last = np.cumsum([len(s.token_ids) - s.cached_kv_len for s in batch.seqs]) - 1
        if len(last) == hidden_states.shape[0]:
            return hidden_states   # decode: every row is a last row
        return hidden_states[torch.from_numpy(last).to(hidden_states.device)]

    # -- execution -----------------------------------------------------------------------------------
    @torch.inference_mode()
    def graph_step_possible(self, batch: ForwardBatch) -> bool:
        return batch.forward_mode == ForwardMode.DECODE and self.graph is not None and self.graph.is_captured \
            and batch.num_seqs <= self.graph.max_bs and self.block_table.fits(batch.seqs)

    @torch.inference_mode()
    def compute_logits(self, batch: ForwardBatch, ids_on_device=None, shard: bool = False):
        """Forward pass of one scheduled batch -> logits (num_seqs, vocab) of each sequence's last token
        (None on a pipeline stage other than the last: its output went to the next stage).
        `ids_on_device`: the input ids are ids the step before sampled (see launch_step).
        `shard`: return (this rank's columns of the logits, id of their first token) instead of gathering the ranks'."""
        hidden = self.compute_hidden(batch, ids_on_device)
        if hidden is None:
            return None
        return self.model.compute_logits_shard(hidden) if shard else self.model.compute_logits(hidden)

    @torch.inference_mode()
    def compute_hidden(self, batch: ForwardBatch, ids_on_device=None):
        """Forward pass of one scheduled batch -> the normed hidden state of each sequence's last token (num_seqs, hidden);
        None on a pipeline stage other than the last."""
        decode = batch.forward_mode == ForwardMode.DECODE
        tpd.check_peer_status()      # results leave through here too (tests, tools): never on top of a failed exchange
        if self.graph_step_possible(batch):
            hidden = self._execute_model_graph(batch, ids_on_device)
            self.stats["graph_steps"] += 1
        elif ids_on_device is not None:
            raise RuntimeError("a step that takes its input ids from the device must be a graph decode step")
        else:
            hidden = self._execute_model_eager(batch)
            if decode:
                self.stats["eager_decode_steps"] += 1
        self.stats["decode_steps" if decode else "prefill_steps"] += 1
        if hidden is None:
            return None
        return self.prepare_last_hidden_states(batch, hidden)

    @torch.inference_mode()
    def execute_model(self, batch: ForwardBatch) -> list:
        return self.finish_step(self.launch_step(batch))

    @torch.inference_mode()
    def launch_step(self, batch: ForwardBatch, ids_on_device=None):
        """Queue one step on the device (forward, sampling, copy of the sampled ids to pinned memory)
        and return without waiting for it; finish_step() collects the ids.  Between the two the host is
        free (the engine hands the previous step's tokens to their consumers there).

        `ids_on_device` (a list: for every sequence of this batch, its row in the step launched before it):
        the sequences' newest token is whatever that step samples -- not known to the host yet.  The first
        node of the decode graph then takes each row's input id from the sampler's output buffer, so the
        step can be queued while the one before it is still running."""
        if batch.num_seqs == 0:
            return None
        t0 = time.perf_counter()
        temperatures, min_ps, top_ps, top_ks = self.prepare_sampling_params(batch)
        n = batch.num_seqs
        if (self._lmhead_packed is not None and self.pp_size == 1 and self._sp_unfiltered and n <= min(self.max_bs, LMHEAD_FUSED_MAX_ROWS)
                and (self.tp_size == 1 or self.shard_sampling)):
            # LM head and draw in one pass over this rank's rows of the weight; tensor-parallel ranks exchange one 64-bit key
            # per row, as in the sharded path below
            hidden = self.compute_hidden(batch, ids_on_device)
            hidden = hidden if hidden.stride(-1) == 1 else hidden.contiguous()
            self.sampler.calls += 1
            start, vocab = self.model.lm_head.start, self.hf_config.vocab_size
            if self.tp_size > 1:
                keys = ops.lmhead_sample(self._lmhead_packed, self._lmhead_rows, hidden, temperatures, start, vocab, self.sampler.seed,
                                         self.sampler.calls, ws=self._lmhead_ws, keys=True)
                ids = ops.sample_pick(tpd.tp_all_gather_rows(keys), out=self._sample_out[:n])
            else:
                ids = ops.lmhead_sample(self._lmhead_packed, self._lmhead_rows, hidden, temperatures, start, vocab, self.sampler.seed,
                                        self.sampler.calls, ws=self._lmhead_ws, out=self._sample_out[:n])
            self.stats["fused_lm_head_steps"] = self.stats.get("fused_lm_head_steps", 0) + 1
            return (batch, self._ids_to_host(ids, n), t0)
        if self.tp_size > 1 and self.pp_size == 1 and self._sp_unfiltered and n <= self.max_bs and self.shard_sampling:
            # every rank draws over its own shard of the vocabulary; the ranks exchange one 64-bit key per row and all
            # pick the same token: no gather of the logits (vocab_parallel_embedding.py:147-151), no broadcast of the ids
            local, first = self.compute_logits(batch, ids_on_device, shard=True)
            keys = self.sampler.forward_shard(local, temperatures, first, self.hf_config.vocab_size)
            ids = ops.sample_pick(tpd.tp_all_gather_rows(keys), out=self._sample_out[:n])
            self.stats["sharded_sampling_steps"] = self.stats.get("sharded_sampling_steps", 0) + 1
            return (batch, self._ids_to_host(ids, n), t0)
        logits = self.compute_logits(batch, ids_on_device)
        if logits is None:      # not the last pipeline stage: the ids arrive with the broadcast
            ids = self._sample_out[:n] if n <= self.max_bs else torch.empty(n, dtype=torch.int64, device=self.device)
        else:
            ids = self.sampler(logits, temperatures, min_ps, top_ps, top_ks, out=self._sample_out[:n] if n <= self.max_bs else None)
        tpd.broadcast_sampled_ids(ids)
        return (batch, self._ids_to_host(ids, n), t0)

    def _ids_to_host(self, ids: torch.Tensor, n: int):
        """Start the copy of a step's ids into the next pinned buffer; returns what finish_step needs."""
        if n > self.max_bs:
            return ids
        self._sample_turn ^= 1
        k = self._sample_turn
        self._sample_hosts[k][:n].copy_(ids, non_blocking=True)
        self._sample_events[k].record()
        return k

    @torch.inference_mode()
    def launch_tick(self, mine: Optional[ForwardBatch], out: Optional[ForwardBatch]):
        """One tick of the pipeline with several batches in flight: this stage runs `mine` (the batch
        that reached it, if any), then every rank takes part in the broadcast of the ids of `out` (the
        batch the LAST stage ran this tick, if any), and only then is this stage's output sent on -- the
        next stage posts its receive at the start of its next tick, after the same broadcast, so the
        order of operations is the same chain on every rank and cannot form a cycle.  On the last
        stage `mine is out`."""
        t0 = time.perf_counter()
        logits = None
        if mine is not None and mine.num_seqs:
            self._defer_stage_send = True
            try:
                logits = self.compute_logits(mine)
            finally:
                self._defer_stage_send = False
        ids = None
        if out is not None and out.num_seqs:
            n = out.num_seqs
            if tpd.pp_is_last():
                assert mine is out and logits is not None
                temperatures, min_ps, top_ps, top_ks = self.prepare_sampling_params(out)
                ids = self.sampler(logits, temperatures, min_ps, top_ps, top_ks, out=self._sample_out[:n] if n <= self.max_bs else None)
            else:
                ids = self._sample_out[:n] if n <= self.max_bs else torch.empty(n, dtype=torch.int64, device=self.device)
            tpd.broadcast_sampled_ids(ids)
            ids = self._ids_to_host(ids, n)
        else:
            out = None
        self.send_stage_output()
        return (out, ids, t0)

    def finish_step(self, handle) -> list:
        if handle is None or handle[0] is None:
            return []
        batch, ids, t0 = handle
        n = batch.num_seqs
        if isinstance(ids, int):     # index of the pinned buffer the ids land in
            self._sample_events[ids].synchronize()
            out = self._sample_hosts[ids][:n].tolist()
        else:
            out = ids.tolist()
        key = "decode_s" if batch.forward_mode == ForwardMode.DECODE else "prefill_s"
        try:
            tpd.check_peer_status()      # a direct-peer exchange that gave up waiting must not pass as a result
        except RuntimeError as e:
            raise RuntimeError(f"{e} [rank {self.rank}, {key[:-2]} step of {n} sequences, steps so far {self.stats}]") from None
        # launch to completion; a step queued behind another is charged from the moment that one completed
        now = time.perf_counter()
        self.stats[key] = self.stats.get(key, 0.0) + (now - max(t0, self._last_done))
        self._last_done = now
        return out

    def _execute_model_eager(self, batch: ForwardBatch) -> torch.Tensor:
        input_ids, positions = self.prepare_input(batch)
        md = self.attn_backend.build_metadata(batch)
        with attention_kv_cache(self.model, md):
            hidden = self.model(input_ids, positions) if self.pp_size == 1 else self._run_stage(input_ids, positions)
        # keep the device block table in step with what the scheduler decided: rows are (re)written from the
        # step's own slot lists, sequence by sequence, so a decode step that had to run outside the graph
        # (no free row, a list longer than a row) does not keep its batch off the graph path afterwards
        if self.graph is not None:
            t = self.attn_backend.last_table
            self.block_table.write_rows(batch.seqs, t["paged_kv_indices"], t["paged_kv_indptr"])
        return hidden

    def _run_stage(self, input_ids: torch.Tensor, positions: torch.Tensor) -> Optional[torch.Tensor]:
        """This pipeline stage's share of a forward pass (worker.py:84-104): take (hidden, residual) from
        the previous stage, run the local layers, hand the pair on -- one message per direction, the two
        tensors stacked.  Returns the normed hidden states on the last stage, None elsewhere."""
        hidden = residual = None
        if not tpd.pp_is_first():
            pair = tpd.pp_recv_prev((2, input_ids.shape[0], self.hf_config.hidden_size), self.dtype, self.device)
            hidden, residual = pair[0], pair[1]
        hidden, residual = self.model.model.forward_stage(input_ids, positions, hidden, residual, self.model.fused)
        if tpd.pp_is_last():
            return hidden
        self._stage_output = torch.stack((hidden, residual))
        if not self._defer_stage_send:
            self.send_stage_output()
        return None

    def send_stage_output(self):
        """Hand the pair computed by _run_stage to the next stage (a no-op if there is none waiting)."""
        out, self._stage_output = self._stage_output, None
        if out is not None:
            tpd.pp_send_next(out)

    def _execute_model_graph(self, batch: ForwardBatch, ids_on_device=None) -> torch.Tensor:
        bs = batch.num_seqs
        padded = self.graph.match_bs(bs)
        self.block_table.stage_decode_step(batch.seqs, ids_on_device)
        self.graph.replay(padded)
        return self.graph.get_output_buffer("hidden_states")[:bs]

    def _decode_forward(self, bs: int) -> torch.Tensor:
        """The body of a captured decode step for a padded batch of `bs` rows."""
        bt = self.block_table
        torch.ops.nvx.decode_step_setup(bt.step_dev, bt.max_bs, bs, bt.table, bt.pad_start, bt.kv_start, bt.kv_len, bt.out_slots,
                                        bt.positions, bt.input_ids, self._sample_out)
        torch.ops.nvx.decode_plan(bt.kv_len[:bs], bt.kv_start[:bs], bt.out_slots[:bs], self.num_kv_heads, self.decode_ws.plan)
        wrapper = self._graph_wrappers[bs]
        md = AttentionMetadata(forward_mode=ForwardMode.DECODE, kv_cache=self.kv_cache, output_kv_indices=bt.out_slots[:bs],
                               wrapper=wrapper)
        with attention_kv_cache(self.model, md):
            return self.model(bt.input_ids[:bs], bt.positions[:bs])

    @torch.inference_mode()
    def capture_graph(self):
        """Capture one decode graph per ladder size, largest first so the pool is sized once
        (model_runner.py:393).  Static inputs: the step buffer of the block table; static output:
        `hidden_states` (max_bs, hidden)."""
        assert self.graph is not None
        bt = self.block_table
        sizes = decode_batch_ladder(self.max_bs)
        if self.tp_size > 1:
            # a batch whose all-reduce goes through a library that cannot be captured (gloo, in rehearsals of several ranks on
            # one device) keeps eager steps; the sizes below it are captured as usual
            sizes = [bs for bs in sizes if tpd.decode_graph_capturable(bs)]
            if not sizes:
                raise RuntimeError("the all-reduce of this group cannot be captured at any decode batch size")
        hidden_out = torch.zeros((self.max_bs, self.hf_config.hidden_size), dtype=self.dtype, device=self.device)
        self.graph.set_input_buffer("step", bt.step_dev)
        self.graph.set_input_buffer("input_ids", bt.input_ids)
        self.graph.set_input_buffer("positions", bt.positions)
        self.graph.set_output_buffer("hidden_states", hidden_out)
        self._graph_wrappers = {}
        bt.step_dev.zero_()   # batch = 0: every row is a padding row during warm-up and capture
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for bs in reversed(sizes):
                w = PagedDecodeWrapper(None, "NHD", use_cuda_graph=True)
                w.bind_table(bt.table.view(-1), bt.kv_start, bt.kv_len, bs, self.decode_ws, qkv_fused=True)
                self._graph_wrappers[bs] = w
                self._decode_forward(bs)   # warm-up outside capture (library handles, allocator pool)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for bs in reversed(sizes):
            with self.graph.capture(bs):
                hidden_out[:bs] = self._decode_forward(bs)
        torch.cuda.synchronize()
        if os.environ.get("NVX_GRAPH_PRUNE", "1") != "0":
            self.prune_slow_graph_sizes()

    def prune_slow_graph_sizes(self, reps: int = 3, margin: float = 1.04):
        """A padded batch costs what its projections cost at that many rows, and a BLAS library's choice of kernel is not
        monotonic in the rows (Qwen3-8B's down projection on this stack: 90 us at 192 rows, 52 at 208).  Every captured
        size is replayed on padding rows and timed; a size that is slower than a LARGER one (by more than `margin`) is
        dropped, so its batches replay the larger graph.  All ranks of a group take the slowest rank's times and so
        drop the same sizes."""
        sizes = list(self.graph.captured_bs)
        if len(sizes) < 2:
            return
        self.block_table.step_dev.zero_()     # batch = 0: every row is a padding row, as during capture
        times = {}
        for bs in sizes:
            self.graph.replay(bs)             # (first replay of a graph uploads it)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                self.graph.replay(bs)
            b.record()
            b.synchronize()
            us = int(a.elapsed_time(b) * 1e3 / reps)
            times[bs] = -tpd.all_ranks_min_int(-us, self.device)      # (the slowest rank's; one rank: its own)
        dropped = []
        for i, bs in enumerate(sizes[:-1]):
            best_larger = min(times[b2] for b2 in sizes[i + 1:])
            if times[bs] > margin * best_larger:
                dropped.append(bs)
        for bs in dropped:
            self.graph.drop(bs)
        self.graph_sizes_dropped, self.graph_step_us_by_size = dropped, times
        torch.cuda.synchronize()

    # -- bookkeeping ---------------------------------------------------------------------------------
    def release_sequence(self, seq):
        if self.block_table is not None:
            self.block_table.release(seq.seq_id)

    def shutdown(self):
        if self.graph is not None:
            self.graph.clear()
        tpd.shutdown()


def _safetensors_files(path: str):
    import glob
    import os

    return sorted(glob.glob(os.path.join(path, "*.safetensors"))) if os.path.isdir(path) else []


def _iter_safetensors(files):
    """(name, tensor) pairs of a checkpoint directory (nanovllmx/model_loader.py:22-48)."""
    from safetensors import safe_open

    for f in files:
        with safe_open(f, framework="pt", device="cpu") as sf:
            for name in sf.keys():
                yield name, sf.get_tensor(name)

"""Paged attention op, its metadata builder and the two kernel wrappers.

Interface of nanovllmx/layers/attention.py: `AttentionMetadata` (:17-22), `AttentionBackend`
(:24-295), `attention_kv_cache` (:297-306), `Attention` (:309-373).  The two wrapper classes stand
where the third-party paged-attention wrappers stood in the reference (constructed at :49-59,
planned at :106-132, run at :360-369) and accept the same `plan(**kwargs)` spellings, the injected
`_sm_scale` attribute and `run(q, (k_cache, v_cache))`; behind them are the HIP kernels.

Block-table building is host integer work and is bit-exact with the reference (pinned by
tests/golden/g3_metadata.json); it is done with numpy on whole batches -- one host-to-device copy
per array instead of one per sequence (attention.py:92-101,134-141).

Beyond the reference's surface, `AttentionBackend` keeps a device-resident block table
(`BlockTable`) so that a decode step uploads three integers per sequence instead of every slot id
of every sequence (attention.py:252-261 re-sends them all, every step).
"""
from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
from torch import nn

from .. import ops
from .. import torch_ops  # noqa: F401  (registers torch.ops.nvx.*, which the wrappers below launch through)
from ..core.common import ForwardBatch, ForwardMode
from ..core.kv_cache import KVCachePool


# ---------------------------------------------------------------------------------------------
# host-side block table (integer, bit-exact)
# ---------------------------------------------------------------------------------------------
def block_table_from_batch(seqs) -> dict:
    """CSR block table of a batch at page size 1 (attention.py:66-101,134-141), as numpy arrays:
    qo_indptr / paged_kv_indptr (int32, B+1), paged_kv_indices (int32), paged_kv_last_page_len
    (int32, all ones), output_kv_indices (int64: slots of the tokens computed in this step)."""
    n = len(seqs)
    q_lens = np.fromiter((len(s.token_ids) - s.cached_kv_len for s in seqs), dtype=np.int64, count=n)
    kv_lens = np.fromiter((len(s.kv_indices) for s in seqs), dtype=np.int64, count=n)
    qo_indptr = np.zeros(n + 1, dtype=np.int32)
    kv_indptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(q_lens, out=qo_indptr[1:])
    np.cumsum(kv_lens, out=kv_indptr[1:])
    flat = np.empty(int(kv_indptr[-1]), dtype=np.int32)
    fresh = np.empty(sum(len(s.kv_indices) - s.cached_kv_len for s in seqs), dtype=np.int64)
    a = f = 0
    for s in seqs:
        kv = s.kv_indices
        flat[a:a + len(kv)] = kv
        a += len(kv)
        new = kv[s.cached_kv_len:]
        fresh[f:f + len(new)] = new
        f += len(new)
    return {"qo_indptr": qo_indptr, "paged_kv_indptr": kv_indptr, "paged_kv_indices": flat,
            "paged_kv_last_page_len": np.ones(n, dtype=np.int32), "output_kv_indices": fresh}


# ---------------------------------------------------------------------------------------------
# kernel wrappers
# ---------------------------------------------------------------------------------------------
def _as_nhd(cache: torch.Tensor) -> torch.Tensor:
    """(slots, 1, Hkv, D) page view of attention.py:356-357 -> (slots, Hkv, D)."""
    return cache.squeeze(1) if cache.dim() == 4 else cache


class PagedPrefillWrapper:
    """Causal variable-length attention over paged K/V; MFMA kernel."""

    def __init__(self, float_workspace_buffer=None, kv_layout: str = "NHD", backend: str = "auto", **_):
        if kv_layout != "NHD":
            raise NotImplementedError("only the NHD layout is built")
        self._sm_scale: Optional[float] = None
        self._plan_ws = None
        self._planned = None

    def plan(self, qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len, num_qo_heads, num_kv_heads,
             head_dim_qk, head_dim_vo=None, page_size=1, causal=True, q_data_type=None, kv_data_type=None, **_):
        if page_size != 1 or head_dim_qk != 128 or (head_dim_vo or head_dim_qk) != 128 or not causal:
            raise NotImplementedError("built for page_size 1, head_dim 128, causal")
        b = qo_indptr.numel() - 1
        if self._plan_ws is None or self._plan_ws.numel() < 3 * (b + 2):
            self._plan_ws = ops.prefill_plan_ws(max(b, 256), qo_indptr.device)
        torch.ops.nvx.prefill_plan(qo_indptr, self._plan_ws)
        self._planned = (qo_indptr, paged_kv_indptr, paged_kv_indices, num_qo_heads, num_kv_heads)

    def run(self, q, paged_kv_cache, out=None):
        qo_indptr, kv_indptr, kv_indices, hq, hkv = self._planned
        k_cache, v_cache = (_as_nhd(c) for c in paged_kv_cache)
        t = q.shape[0]
        scale = self._sm_scale if self._sm_scale is not None else 128 ** -0.5
        if out is None:
            o = torch.ops.nvx.paged_prefill(q.reshape(t, hq * 128), k_cache, v_cache, qo_indptr, kv_indptr, kv_indices,
                                            self._plan_ws, hq, hkv, scale)
        else:   # a caller-provided output buffer: the operator form allocates its result
            o = ops.paged_prefill(q.reshape(t, hq * 128), k_cache, v_cache, qo_indptr, kv_indptr, kv_indices, self._plan_ws,
                                  hq, hkv, scale, out=out)
        return o.view(t, hq, 128)


class PagedDecodeWrapper:
    """One query token per sequence over paged K/V; split-KV streaming kernel.

    The three `*_buffer` arguments make the wrapper graph-safe in the reference's sense
    (attention.py:187-196): plan() copies into them instead of rebinding, so a captured run() keeps
    reading the same addresses."""

    def __init__(self, float_workspace_buffer=None, kv_layout: str = "NHD", use_tensor_cores: bool = False,
                 use_cuda_graph: bool = False, backend: str = "auto", paged_kv_indptr_buffer=None,
                 paged_kv_indices_buffer=None, paged_kv_last_page_len_buffer=None, **_):
        if kv_layout != "NHD":
            raise NotImplementedError("only the NHD layout is built")
        self._sm_scale: Optional[float] = None
        self._static = use_cuda_graph
        self._indptr_buf, self._indices_buf = paged_kv_indptr_buffer, paged_kv_indices_buffer
        self._ws: Optional[ops.DecodeWorkspace] = None
        self._kv_start = self._kv_len = self._kv_indices = None
        self._batch = 0
        self.qkv_fused = False
        self._new_slots = None

    def _ensure(self, batch: int, hq: int, hkv: int, device):
        if self._ws is None or self._ws.max_batch < batch or (self._ws.n_q_heads, self._ws.n_kv_heads) != (hq, hkv):
            cap = max(batch, 64)
            self._ws = ops.DecodeWorkspace(cap, hq, hkv, device)
            self._kv_start = torch.zeros(cap, dtype=torch.int64, device=device)
            self._kv_len = torch.zeros(cap, dtype=torch.int32, device=device)

    def expect_new_rows(self, out_slots: Optional[torch.Tensor]):
        """Tell the next plan() which slot each sequence's new K/V row goes to (int64, one per sequence; it must
        be the last entry of the sequence's slot list).  The plan then carries them, and run_qkv() may be used:
        norm + rotary + scatter of the new row inside the attention launch.  Kept apart from plan() because
        plan()'s keyword set is the one the reference passes."""
        self._new_slots = out_slots

    def plan(self, indptr, indices, last_page_len, num_qo_heads, num_kv_heads, head_dim, page_size=1,
             q_data_type=None, kv_data_type=None, **_):
        if page_size != 1 or head_dim != 128:
            raise NotImplementedError("built for page_size 1, head_dim 128")
        b = indptr.numel() - 1
        self._ensure(b, num_qo_heads, num_kv_heads, indptr.device)
        self._batch = b
        self._kv_start[:b].copy_(indptr[:-1])
        torch.sub(indptr[1:], indptr[:-1], out=self._kv_len[:b])
        self._kv_indices = indices if not self._static or self._indices_buf is None else self._indices_buf
        new_slots, self._new_slots = self._new_slots, None
        if new_slots is not None and new_slots.numel() != b:
            new_slots = None
        torch.ops.nvx.decode_plan(self._kv_len[:b], self._kv_start[:b], new_slots, num_kv_heads, self._ws.plan)
        self.qkv_fused = new_slots is not None

    def bind_table(self, kv_indices, kv_start, kv_len, batch: int, ws: ops.DecodeWorkspace, qkv_fused: bool = False):
        """Use caller-owned static buffers (the device-resident block table); the plan kernel is then
        launched by the caller as part of its captured graph.  `qkv_fused` says that the caller's plan
        carries the new rows' slots, so run_qkv may be used."""
        self._kv_indices, self._kv_start, self._kv_len, self._batch, self._ws = kv_indices, kv_start, kv_len, batch, ws
        self.qkv_fused = qkv_fused

    def run_qkv(self, qkv, positions, cos_sin_cache, q_weight, k_weight, eps, paged_kv_cache, out=None):
        """Attention straight from the raw fused projection (see ops.paged_decode_qkv): replaces the
        q/k norm + rotary + scatter launch and run()."""
        k_cache, v_cache = (_as_nhd(c) for c in paged_kv_cache)
        scale = self._sm_scale if self._sm_scale is not None else 128 ** -0.5
        if out is None and qkv.shape[0] == self._batch:
            ws = self._ws
            return torch.ops.nvx.paged_decode_qkv(qkv, positions, cos_sin_cache, q_weight, k_weight, eps, k_cache, v_cache,
                                                  self._kv_indices, ws.plan, ws.partial, ws.n_q_heads, ws.n_kv_heads, scale)
        return ops.paged_decode_qkv(qkv, positions, cos_sin_cache, q_weight, k_weight, eps, k_cache, v_cache, self._kv_indices,
                                    self._batch, self._ws, scale, out=out)

    def run(self, q, paged_kv_cache, out=None):
        k_cache, v_cache = (_as_nhd(c) for c in paged_kv_cache)
        b = self._batch
        ws = self._ws
        hq = ws.n_q_heads
        scale = self._sm_scale if self._sm_scale is not None else 128 ** -0.5
        if out is None:
            o = torch.ops.nvx.paged_decode(q.reshape(b, hq * 128), k_cache, v_cache, self._kv_indices, ws.plan, ws.partial, hq,
                                           ws.n_kv_heads, scale)
        else:
            o = ops.paged_decode(q.reshape(b, hq * 128), k_cache, v_cache, self._kv_indices, b, ws, scale, out=out)
        return o.view(b, hq, 128)


# ---------------------------------------------------------------------------------------------
@dataclass
class AttentionMetadata:
    forward_mode: ForwardMode
    kv_cache: KVCachePool
    output_kv_indices: torch.Tensor
    wrapper: object   # PagedPrefillWrapper | PagedDecodeWrapper


class BlockTable:
    """Device-resident block table: one row of `row_capacity` slot ids per running sequence, plus the
    static per-step buffers a captured decode graph reads.  Row `n_rows - 1` is the padding row."""

    def __init__(self, max_bs: int, n_rows: int, row_capacity: int, device):
        self.max_bs, self.n_rows, self.row_capacity, self.device = max_bs, n_rows, row_capacity, device
        self.table = torch.zeros((n_rows, row_capacity), dtype=torch.int32, device=device)
        self.pad_start = (n_rows - 1) * row_capacity
        self.free_rows = list(range(n_rows - 2, -1, -1))
        self.row_of: dict = {}
        self.kv_start = torch.zeros(max_bs, dtype=torch.int64, device=device)
        self.kv_len = torch.ones(max_bs, dtype=torch.int32, device=device)
        self.out_slots = torch.full((max_bs,), -1, dtype=torch.int64, device=device)
        self.positions = torch.zeros(max_bs, dtype=torch.int64, device=device)
        self.input_ids = torch.zeros(max_bs, dtype=torch.int64, device=device)
        self.step_dev = torch.zeros(4 + 4 * max_bs, dtype=torch.int32, device=device)
        # two pinned staging buffers, used in turn: with one step queued behind another the upload of the
        # first may not have run yet when the host fills in the second
        pin = torch.device(device).type == "cuda"
        self._step_hosts = [torch.zeros(4 + 4 * max_bs, dtype=torch.int32, pin_memory=pin) for _ in range(2)]
        self._step_nps = [h.numpy() for h in self._step_hosts]
        self._step_turn = 0

    def assign(self, seq_id: str) -> int:
        if seq_id in self.row_of:
            return self.row_of[seq_id]
        if not self.free_rows:
            raise RuntimeError("block table has no free row (more live sequences than rows)")
        r = self.free_rows.pop()
        self.row_of[seq_id] = r
        return r

    def release(self, seq_id: str):
        r = self.row_of.pop(seq_id, None)
        if r is not None:
            self.free_rows.append(r)

    def write_rows(self, seqs, kv_indices_dev, kv_indptr_dev):
        """After a step that ran from a CSR table (a prefill, or a decode step outside the graph): every
        sequence of the batch that has, or can get, a row long enough for its slot list gets the list
        written there; any other sequence loses its row, so that fits() is False for it and its next
        decode step runs from a CSR table again instead of reading a row nobody filled.  A chunk view (a
        prompt cut before its end) gets no row: its sequence receives one with its last chunk, whose slot
        list is the whole prompt's."""
        rows = []
        for s in seqs:
            if s.partial_of is not None:
                rows.append(-1)
            elif len(s.kv_indices) <= self.row_capacity and (s.seq_id in self.row_of or self.free_rows):
                rows.append(self.assign(s.seq_id))
            else:
                self.release(s.seq_id)
                rows.append(-1)
        if any(r >= 0 for r in rows):   # the kernel skips negative rows
            ops.block_table_write_rows(kv_indices_dev, kv_indptr_dev,
                                       torch.tensor(rows, dtype=torch.int32).to(self.device, non_blocking=True), self.table)

    def stage_decode_step(self, seqs, ids_on_device=None):
        """Fill a pinned step buffer [batch | row | pos | slot | token] and start its upload.  With
        `ids_on_device` (a list: for every sequence its row in the step before) the token of a row is
        written as -1-j: the id that step sampled for its row j and that has not reached the host yet."""
        n, m = len(seqs), self.max_bs
        self._step_turn ^= 1
        buf = self._step_nps[self._step_turn]
        buf[0] = n
        row_of = self.row_of
        buf[4:4 + n] = [row_of[s.seq_id] for s in seqs]
        buf[4 + m:4 + m + n] = [len(s.kv_indices) - 1 for s in seqs]
        buf[4 + 2 * m:4 + 2 * m + n] = [s.kv_indices[-1] for s in seqs]
        if ids_on_device is not None:
            buf[4 + 3 * m:4 + 3 * m + n] = [-1 - j for j in ids_on_device]
        else:
            buf[4 + 3 * m:4 + 3 * m + n] = [s.token_ids[-1] for s in seqs]
        self.step_dev.copy_(self._step_hosts[self._step_turn], non_blocking=True)

    def fits(self, seqs) -> bool:
        return all(s.seq_id in self.row_of and len(s.kv_indices) <= self.row_capacity for s in seqs)


class AttentionBackend:
    def __init__(self, kv_cache: KVCachePool, num_qo_heads: int, num_kv_heads: int, head_dim: int,
                 dtype: torch.dtype, device: torch.device):
        if head_dim != 128:
            raise NotImplementedError("the HIP attention kernels are built for head_dim 128")
        self.kv_cache = kv_cache
        self.num_qo_heads, self.num_kv_heads, self.head_dim = num_qo_heads, num_kv_heads, head_dim
        self.dtype, self.device = dtype, device
        self.prefill_wrapper = PagedPrefillWrapper(None, "NHD", backend="auto")
        self.decode_wrapper = PagedDecodeWrapper(None, "NHD", use_tensor_cores=False, backend="auto")
        self.decode_wrappers_cuda_graph: dict = {}
        self.last_table: dict = {}

    def _up(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(a).to(self.device, non_blocking=True)

    # -- eager path (attention.py:62-149) -----------------------------------------------------------
    def build_metadata(self, batch: ForwardBatch) -> AttentionMetadata:
        t = block_table_from_batch(batch.seqs)
        dev = {k: self._up(v) for k, v in t.items()}
        self.last_table = dev
        if batch.forward_mode == ForwardMode.PREFILL:
            wrapper = self.prefill_wrapper
            wrapper.plan(qo_indptr=dev["qo_indptr"], paged_kv_indices=dev["paged_kv_indices"],
                         paged_kv_indptr=dev["paged_kv_indptr"], paged_kv_last_page_len=dev["paged_kv_last_page_len"],
                         num_qo_heads=self.num_qo_heads, num_kv_heads=self.num_kv_heads, head_dim_qk=self.head_dim,
                         head_dim_vo=self.head_dim, page_size=1, causal=True, q_data_type=self.dtype, kv_data_type=self.dtype)
        elif batch.forward_mode == ForwardMode.DECODE:
            wrapper = self.decode_wrapper
            # the same kernels as a captured step launches: the attention launch makes and stores the new rows
            wrapper.expect_new_rows(dev["output_kv_indices"])
            wrapper.plan(indptr=dev["paged_kv_indptr"], indices=dev["paged_kv_indices"],
                         last_page_len=dev["paged_kv_last_page_len"], num_qo_heads=self.num_qo_heads,
                         num_kv_heads=self.num_kv_heads, head_dim=self.head_dim, page_size=1, q_data_type=self.dtype,
                         kv_data_type=self.dtype)
        else:
            raise NotImplementedError
        return AttentionMetadata(forward_mode=batch.forward_mode, kv_cache=self.kv_cache,
                                 output_kv_indices=dev["output_kv_indices"], wrapper=wrapper)

    # -- graph path with the reference's buffer protocol (attention.py:151-295) ----------------------
    def prepare_for_cuda_graph_capture(self, graph, max_bs: int, context_len: int):
        dev = self.device
        graph.set_input_buffer("paged_kv_indptr_buffer", torch.arange(0, max_bs + 1, dtype=torch.int32, device=dev) * context_len)
        graph.set_input_buffer("paged_kv_indices_buffer", torch.zeros((max_bs * context_len,), dtype=torch.int32, device=dev))
        graph.set_input_buffer("paged_kv_last_page_len_buffer", torch.ones((max_bs,), dtype=torch.int32, device=dev))
        graph.set_input_buffer("output_kv_indices_buffer", torch.zeros((max_bs,), dtype=torch.int64, device=dev))

    def build_metadata_for_cuda_graph_capture(self, graph, bs: int, context_len: int) -> AttentionMetadata:
        indptr = graph.get_input_buffer("paged_kv_indptr_buffer")[:bs + 1]
        indices = graph.get_input_buffer("paged_kv_indices_buffer")
        last = graph.get_input_buffer("paged_kv_last_page_len_buffer")[:bs]
        out_idx = graph.get_input_buffer("output_kv_indices_buffer")[:bs]
        w = PagedDecodeWrapper(None, "NHD", use_tensor_cores=False, use_cuda_graph=True, backend="auto",
                               paged_kv_indptr_buffer=indptr, paged_kv_indices_buffer=indices,
                               paged_kv_last_page_len_buffer=last)
        w.plan(indptr=indptr, indices=indices[:bs * context_len], last_page_len=last, num_qo_heads=self.num_qo_heads,
               num_kv_heads=self.num_kv_heads, head_dim=self.head_dim, page_size=1, q_data_type=self.dtype,
               kv_data_type=self.dtype)
        self.decode_wrappers_cuda_graph[bs] = w
        return AttentionMetadata(forward_mode=ForwardMode.DECODE, kv_cache=self.kv_cache, output_kv_indices=out_idx, wrapper=w)

    def build_metadata_for_cuda_graph_replay(self, graph, batch: ForwardBatch, padded_bs: int) -> AttentionMetadata:
        bs = batch.num_seqs
        t = block_table_from_batch(batch.seqs)
        kv_len_total = int(t["paged_kv_indptr"][-1])
        indptr_buf = graph.get_input_buffer("paged_kv_indptr_buffer")
        indices_buf = graph.get_input_buffer("paged_kv_indices_buffer")
        last_buf = graph.get_input_buffer("paged_kv_last_page_len_buffer")
        out_buf = graph.get_input_buffer("output_kv_indices_buffer")
        # padding rows: one key each, slot 0, no scatter (attention.py:240-265)
        indptr = np.empty(padded_bs + 1, dtype=np.int32)
        indptr[:bs + 1] = t["paged_kv_indptr"]
        indptr[bs + 1:] = kv_len_total + np.arange(1, padded_bs - bs + 1, dtype=np.int32)
        indptr_buf[:padded_bs + 1].copy_(self._up(indptr))
        indices_buf.fill_(0)
        indices_buf[:kv_len_total].copy_(self._up(t["paged_kv_indices"]))
        out_buf.fill_(-1)
        out_buf[:bs].copy_(self._up(t["output_kv_indices"]))
        w = self.decode_wrappers_cuda_graph[padded_bs]
        w.plan(indptr=indptr_buf[:padded_bs + 1], indices=indices_buf[:kv_len_total], last_page_len=last_buf[:padded_bs],
               num_qo_heads=self.num_qo_heads, num_kv_heads=self.num_kv_heads, head_dim=self.head_dim, page_size=1,
               q_data_type=self.dtype, kv_data_type=self.dtype)
        return AttentionMetadata(forward_mode=ForwardMode.DECODE, kv_cache=self.kv_cache,
                                 output_kv_indices=out_buf[:padded_bs], wrapper=w)


@contextmanager
def attention_kv_cache(model: nn.Module, metadata: AttentionMetadata):
    """Hand `metadata` to every Attention module of `model` for the duration of one forward."""
    touched = [m for m in model.modules() if isinstance(m, Attention)]
    for m in touched:
        m.set_attention_metadata(metadata)
    try:
        yield
    finally:
        for m in touched:
            m.set_attention_metadata(None)


class Attention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, scaling: Optional[float], num_kv_heads: int, layer_id: int):
        super().__init__()
        self.num_heads, self.head_dim, self.num_kv_heads = num_heads, head_dim, num_kv_heads
        self.num_kv_groups = num_heads // num_kv_heads
        self.scaling = head_dim ** -0.5 if scaling is None else scaling
        self.layer_id = layer_id
        self.attention_metadata: Optional[AttentionMetadata] = None

    def set_attention_metadata(self, metadata: Optional[AttentionMetadata]):
        self.attention_metadata = metadata

    def forward(self, q: torch.Tensor, k: Optional[torch.Tensor], v: Optional[torch.Tensor], save_kv_cache: bool = True,
                out: Optional[torch.Tensor] = None):
        """Scatter this step's K/V rows into the paged cache (skipped when k is None: the caller fused
        it), then attend over every slot the block table lists (attention.py:332-373)."""
        md = self.attention_metadata
        assert md is not None
        if not q.is_contiguous():
            q = q.contiguous()
        if k is not None:
            assert v is not None
            if save_kv_cache:
                md.kv_cache.set_kv_cache(self.layer_id, md.output_kv_indices, k.view(-1, self.num_kv_heads, self.head_dim),
                                         v.view(-1, self.num_kv_heads, self.head_dim))
        k_cache, v_cache = md.kv_cache.get_kv_cache(self.layer_id)
        md.wrapper._sm_scale = self.scaling
        if md.forward_mode not in (ForwardMode.PREFILL, ForwardMode.DECODE):
            raise NotImplementedError
        o = md.wrapper.run(q.view(-1, self.num_heads, self.head_dim), (k_cache, v_cache), out=out)
        return o.view(-1, self.num_heads * self.head_dim)

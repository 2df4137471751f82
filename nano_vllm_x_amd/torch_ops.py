"""`torch.ops.nvx.*`: the C-ABI kernels as PyTorch custom ops (device key only, no CPU kernels).

Importing this module registers the ops.  Every op is a thin call into `nano_vllm_x_amd.ops` (argument checks,
output allocation, one launch through the C ABI on the current stream).  The paged-attention wrappers of
`layers/attention.py` -- the objects that stand where the reference's third-party attention wrappers stood
(nanovllmx/layers/attention.py:360-369) -- launch their kernels through these operators, so the attention path of
the engine is "custom ops over a C ABI" end to end; workspaces are ordinary tensors passed in and mutated.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

_LIB = torch.library.Library("nvx", "DEF")
_DEFS = {
    "store_kvcache": "(Tensor k, Tensor v, Tensor(a!) k_cache, Tensor(b!) v_cache, Tensor kv_indices) -> ()",
    "rmsnorm": "(Tensor x, Tensor weight, float eps) -> Tensor",
    "rmsnorm_residual": "(Tensor x, Tensor residual, Tensor weight, float eps) -> (Tensor, Tensor)",
    "rope": "(Tensor positions, Tensor q, Tensor k, Tensor cos_sin) -> (Tensor, Tensor)",
    "silu_mul": "(Tensor x) -> Tensor",
    "sample": "(Tensor logits, Tensor temperatures, Tensor top_ks, Tensor top_ps, Tensor min_ps, int seed, int offset) -> Tensor",
    "sampler_probs": "(Tensor logits, Tensor temperatures, Tensor top_ks, Tensor top_ps, Tensor min_ps) -> Tensor",
    "sample_shard": "(Tensor logits, Tensor temperatures, int vocab_offset, int vocab, int seed, int offset) -> Tensor",
    "sample_pick": "(Tensor winners) -> Tensor",
    "lmhead_pack": "(Tensor weight) -> Tensor",
    "linear_stream": "(Tensor x, Tensor packed, int n) -> Tensor",
    "linear_block": "(Tensor x, Tensor packed, int n, int mode, int k_slices) -> Tensor",
    "splitk_sum": "(Tensor partials, bool half, bool silu_mul) -> Tensor",
    "lmhead_sample": "(Tensor packed, int n_rows, Tensor x, Tensor temperatures, int vocab_offset, int vocab, int seed, int offset, "
                     "bool keys) -> Tensor",
    "qknorm_rope_store": "(Tensor qkv, Tensor positions, Tensor cos_sin, Tensor q_weight, Tensor k_weight, float eps, Tensor slots, "
                         "Tensor(a!) k_cache, Tensor(b!) v_cache, int n_q_heads, int n_kv_heads) -> Tensor",
    # paged attention, decode: plan once per step, then one call per layer
    "decode_plan": "(Tensor kv_len, Tensor kv_start, Tensor? out_slots, int n_kv_heads, Tensor(a!) plan_ws) -> ()",
    "paged_decode": "(Tensor q, Tensor k_cache, Tensor v_cache, Tensor kv_indices, Tensor plan_ws, Tensor(a!) partial_ws, "
                    "int n_q_heads, int n_kv_heads, float sm_scale) -> Tensor",
    "paged_decode_qkv": "(Tensor qkv, Tensor positions, Tensor cos_sin, Tensor q_weight, Tensor k_weight, float eps, "
                        "Tensor(a!) k_cache, Tensor(b!) v_cache, Tensor kv_indices, Tensor plan_ws, Tensor(c!) partial_ws, "
                        "int n_q_heads, int n_kv_heads, float sm_scale) -> Tensor",
    "paged_decode_noalloc": "(Tensor q, Tensor k_cache, Tensor v_cache, Tensor kv_indices, Tensor kv_len, Tensor kv_start, "
                            "int n_q_heads, int n_kv_heads, float sm_scale) -> Tensor",
    # paged attention, prefill
    "prefill_plan": "(Tensor qo_indptr, Tensor(a!) plan_ws) -> ()",
    "paged_prefill": "(Tensor q, Tensor k_cache, Tensor v_cache, Tensor qo_indptr, Tensor kv_indptr, Tensor kv_indices, "
                     "Tensor plan_ws, int n_q_heads, int n_kv_heads, float sm_scale) -> Tensor",
    # projections of a decode step and the norm that sums their K-split partials
    "linear_decode": "(Tensor x, Tensor w, int mode, int k_splits) -> Tensor",
    "rmsnorm_splitk": "(Tensor partials, Tensor weight, float eps) -> Tensor",
    "rmsnorm_splitk_residual": "(Tensor partials, Tensor residual, Tensor weight, float eps) -> (Tensor, Tensor)",
    # direct-peer all-reduce of a tensor-parallel group (`group`: the integer ops.PeerGroup.comm), alone or with the norm
    "peer_allreduce": "(int group, Tensor(a!) x) -> ()",
    "peer_allreduce_norm": "(int group, Tensor x, Tensor? residual, Tensor weight, float eps) -> (Tensor, Tensor)",
    # device-resident block table
    "block_table_write_rows": "(Tensor kv_indices, Tensor kv_indptr, Tensor rows, Tensor(a!) table) -> ()",
    "decode_step_setup": "(Tensor step, int stride, int padded_batch, Tensor(a!) table, int pad_start, Tensor(b!) kv_start, "
                         "Tensor(c!) kv_len, Tensor(d!) out_slots, Tensor(e!) positions, Tensor(f!) input_ids, Tensor? prev_ids) -> ()",
}
for _name, _sig in _DEFS.items():
    _LIB.define(_name + _sig)


def _ws(plan_ws, partial_ws, n_q_heads, n_kv_heads):
    return ops.DecodeWorkspace.from_tensors(plan_ws, partial_ws, n_q_heads, n_kv_heads)


def _store_kvcache(k, v, k_cache, v_cache, kv_indices):
    ops.store_kvcache(k, v, k_cache, v_cache, kv_indices)


def _rmsnorm(x, weight, eps):
    return ops.rmsnorm(x, weight, eps)


def _rmsnorm_residual(x, residual, weight, eps):
    return ops.rmsnorm(x, weight, eps, residual=residual)


def _rope(positions, q, k, cos_sin):
    return ops.rope(positions, q, k, cos_sin)


def _silu_mul(x):
    return ops.silu_mul(x)


def _sample(logits, temperatures, top_ks, top_ps, min_ps, seed, offset):
    return ops.sample(logits, temperatures, top_ks, top_ps, min_ps, seed, offset)


def _sampler_probs(logits, temperatures, top_ks, top_ps, min_ps):
    return ops.sampler_probs(logits, temperatures, top_ks, top_ps, min_ps)


def _sample_shard(logits, temperatures, vocab_offset, vocab, seed, offset):
    return ops.sample_shard(logits, temperatures, vocab_offset, vocab, seed, offset)


def _sample_pick(winners):
    return ops.sample_pick(winners)


def _linear_stream(x, packed, n):
    return ops.linear_stream(x, packed, n)


def _linear_block(x, packed, n, mode, k_slices):
    return ops.linear_block(x, packed, n, mode=mode, k_slices=k_slices)


def _splitk_sum(partials, half, silu_mul):
    return ops.splitk_sum(partials, torch.float16 if half else torch.bfloat16, silu_mul)


def _lmhead_pack(weight):
    return ops.lmhead_pack(weight)


def _lmhead_sample(packed, n_rows, x, temperatures, vocab_offset, vocab, seed, offset, keys):
    return ops.lmhead_sample(packed, n_rows, x, temperatures, vocab_offset, vocab, seed, offset, keys=keys)


def _qknorm_rope_store(qkv, positions, cos_sin, q_weight, k_weight, eps, slots, k_cache, v_cache, n_q_heads, n_kv_heads):
    return ops.qknorm_rope_store(qkv, positions, cos_sin, q_weight, k_weight, eps, slots, k_cache, v_cache, n_q_heads, n_kv_heads)


def _decode_plan(kv_len, kv_start, out_slots: Optional[torch.Tensor], n_kv_heads, plan_ws):
    ops.decode_plan(kv_len, kv_start, ops.DecodeWorkspace.from_tensors(plan_ws, None, 0, n_kv_heads), out_slots=out_slots)


def _paged_decode(q, k_cache, v_cache, kv_indices, plan_ws, partial_ws, n_q_heads, n_kv_heads, sm_scale):
    return ops.paged_decode(q, k_cache, v_cache, kv_indices, q.shape[0], _ws(plan_ws, partial_ws, n_q_heads, n_kv_heads), sm_scale)


def _paged_decode_qkv(qkv, positions, cos_sin, q_weight, k_weight, eps, k_cache, v_cache, kv_indices, plan_ws, partial_ws,
                      n_q_heads, n_kv_heads, sm_scale):
    return ops.paged_decode_qkv(qkv, positions, cos_sin, q_weight, k_weight, eps, k_cache, v_cache, kv_indices, qkv.shape[0],
                                _ws(plan_ws, partial_ws, n_q_heads, n_kv_heads), sm_scale)


def _paged_decode_noalloc(q, k_cache, v_cache, kv_indices, kv_len, kv_start, n_q_heads, n_kv_heads, sm_scale):
    return ops.paged_decode_noalloc(q, k_cache, v_cache, kv_indices, kv_len, kv_start, n_q_heads, n_kv_heads, sm_scale)


def _prefill_plan(qo_indptr, plan_ws):
    ops.prefill_plan(qo_indptr, plan_ws)


def _paged_prefill(q, k_cache, v_cache, qo_indptr, kv_indptr, kv_indices, plan_ws, n_q_heads, n_kv_heads, sm_scale):
    return ops.paged_prefill(q, k_cache, v_cache, qo_indptr, kv_indptr, kv_indices, plan_ws, n_q_heads, n_kv_heads, sm_scale)


def _linear_decode(x, w, mode, k_splits):
    return ops.linear_decode(x, w, mode=mode, k_splits=k_splits)


def _rmsnorm_splitk(partials, weight, eps):
    return ops.rmsnorm_splitk(partials, weight, eps)


def _rmsnorm_splitk_residual(partials, residual, weight, eps):
    return ops.rmsnorm_splitk(partials, weight, eps, residual=residual)


def _peer_allreduce(group, x):
    ops.peer_allreduce_norm(ops.PeerGroup._registry[group], x)


def _peer_allreduce_norm(group, x, residual: Optional[torch.Tensor], weight, eps):
    g = ops.PeerGroup._registry[group]
    if residual is None:
        return ops.peer_allreduce_norm(g, x, weight, eps), x.new_empty(0)
    return ops.peer_allreduce_norm(g, x, weight, eps, residual=residual)


def _block_table_write_rows(kv_indices, kv_indptr, rows, table):
    ops.block_table_write_rows(kv_indices, kv_indptr, rows, table)


def _decode_step_setup(step, stride, padded_batch, table, pad_start, kv_start, kv_len, out_slots, positions, input_ids,
                       prev_ids: Optional[torch.Tensor]):
    ops.decode_step_setup(step, stride, padded_batch, table, pad_start, kv_start, kv_len, out_slots, positions, input_ids,
                          prev_ids=prev_ids)


for _name in _DEFS:
    _LIB.impl(_name, globals()["_" + _name], "CUDA")   # PyTorch-ROCm's device key is spelled CUDA

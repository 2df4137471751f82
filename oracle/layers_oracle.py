"""Oracle: RMSNorm, rotary embedding, SiluAndMul, sampler masks, KV scatter (CPU, torch/numpy).

TEST INFRASTRUCTURE.  Eager-mode semantics of
  nanovllmx/layers/layernorm.py:17-36, rotary_embedding.py:5-102, activation.py:15-18,
  sampler.py:20-106, core/kv_cache.py:12-44.
Pinned by tests/golden/g4_store_kv.npz, g5_layers.npz and g6_sampler.npz (reference run in
eager mode on CPU).
"""
from __future__ import annotations

import numpy as np
import torch


# -- RMSNorm ---------------------------------------------------------------------------
def rmsnorm_oracle(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: torch.Tensor | None = None):
    """layernorm.py:17-36: fp32 math; the sum that is normalised is the UNROUNDED fp32
    x+residual, the residual handed back is its rounding to the input dtype; the weight
    multiply happens in fp32 before the single final rounding."""
    dt = x.dtype
    acc = x.to(torch.float32)
    res_out = None
    if residual is not None:
        acc = acc + residual.to(torch.float32)
        res_out = acc.to(dt)
    ms = (acc * acc).mean(dim=-1, keepdim=True)
    y = (acc * torch.rsqrt(ms + eps) * weight.to(torch.float32)).to(dt)
    return y if residual is None else (y, res_out)


# -- rotary ----------------------------------------------------------------------------
def rope_table_oracle(rotary_dim: int, max_pos: int, base: float) -> torch.Tensor:
    """rotary_embedding.py:52-71: fp32 table (max_pos, rotary_dim) = [cos | sin]."""
    inv = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    ang = torch.arange(max_pos, dtype=torch.float)[:, None] * inv[None, :]
    return torch.cat((ang.cos(), ang.sin()), dim=-1)


def rope_rows_oracle(x: torch.Tensor, cos_sin_rows: torch.Tensor, head_size: int) -> torch.Tensor:
    """rotary_embedding.py:5-24,86-101 for rotary_dim == head_size.

    x: (T, H*head_size) in the model dtype.  cos/sin are rounded to that dtype first; halves
    are paired (j, j+D/2) on input and written INTERLEAVED (2j, 2j+1) on output; each product
    and each sum is rounded to the model dtype (eager arithmetic)."""
    t = x.shape[0]
    half = head_size // 2
    xs = x.reshape(t, -1, head_size)
    c = cos_sin_rows[:, :half].to(x.dtype)[:, None, :]
    s = cos_sin_rows[:, half:].to(x.dtype)[:, None, :]
    a, b = xs[..., :half], xs[..., half:]
    even = a * c - b * s
    odd = b * c + a * s
    out = torch.empty_like(xs)
    out[..., 0::2] = even
    out[..., 1::2] = odd
    return out.reshape(x.shape)


# -- SwiGLU gate -------------------------------------------------------------------------
def silu_mul_oracle(x: torch.Tensor) -> torch.Tensor:
    """activation.py:15-18: silu(x[..., :d]) * x[..., d:], each rounded to the input dtype."""
    d = x.shape[-1] // 2
    return torch.nn.functional.silu(x[..., :d]) * x[..., d:]


# -- KV scatter --------------------------------------------------------------------------
def store_kv_oracle(k: np.ndarray, v: np.ndarray, k_cache: np.ndarray, v_cache: np.ndarray, slots: np.ndarray):
    """kv_cache.py:12-44 on raw 16-bit patterns: row i of k/v goes to slot slots[i]; -1 skips."""
    for i, s in enumerate(slots.tolist()):
        if s == -1:
            continue
        k_cache[s] = k[i]
        v_cache[s] = v[i]


# -- sampler -----------------------------------------------------------------------------
def sampler_masks_oracle(logits: np.ndarray, temps, top_k, top_p, min_p):
    """sampler.py:39-48,58-106 up to (not including) the multinomial draw.

    logits (B, V) float32 (already upcast).  Returns (keep_after_topk_topp, keep_after_minp,
    probs) with probs float32.  Arithmetic follows the reference op by op in float32."""
    x = torch.from_numpy(np.array(logits, dtype=np.float32))
    x = x / torch.tensor(temps, dtype=torch.float32)[:, None]
    b, v = x.shape
    srt, idx = x.sort(dim=-1, descending=False)
    kth = srt.gather(1, (v - torch.tensor(top_k, dtype=torch.long))[:, None])
    srt = srt.masked_fill(srt < kth, float("-inf"))
    cum = srt.softmax(dim=-1).cumsum(dim=-1)
    drop = cum <= (1 - torch.tensor(top_p, dtype=torch.float32))[:, None]
    drop[:, -1] = False
    srt = srt.masked_fill(drop, float("-inf"))
    x = torch.empty_like(srt).scatter_(-1, idx, srt)
    keep1 = torch.isfinite(x).numpy()
    pr = torch.softmax(x, dim=-1)
    floor = torch.tensor(min_p, dtype=torch.float32)[:, None] * pr.max(dim=-1, keepdim=True).values
    x = x.masked_fill(pr < floor, float("-inf"))
    keep2 = torch.isfinite(x).numpy()
    probs = torch.softmax(x, dim=-1, dtype=torch.float32).numpy()
    return keep1, keep2, probs

"""Rotary position embedding backed by the HIP kernel (interface of nanovllmx/layers/rotary_embedding.py:27-102)."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


def build_cos_sin_cache(rotary_dim: int, max_position_embeddings: int, base: float) -> torch.Tensor:
    """fp32 (max_pos, rotary_dim) = [cos(t * f_j) | sin(t * f_j)], f_j = base^(-2j/rotary_dim)
    (rotary_embedding.py:52-71; built on the CPU in fp32 so that every rank holds identical bits)."""
    exponents = torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim
    inv_freq = 1.0 / (base ** exponents)
    t = torch.arange(max_position_embeddings, dtype=torch.float)
    angles = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((angles.cos(), angles.sin()), dim=-1)


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 dtype: torch.dtype | None = None) -> None:
        super().__init__()
        if head_size != 128 or rotary_dim != head_size:
            raise NotImplementedError("the HIP rotary kernel is built for rotary_dim == head_size == 128")
        self.head_size, self.rotary_dim = head_size, rotary_dim
        self.max_position_embeddings, self.base, self.dtype = max_position_embeddings, base, dtype
        cache = build_cos_sin_cache(rotary_dim, max_position_embeddings, base)
        if dtype is not None:   # the kernel reads fp32; a narrower table is widened again, rounding kept
            cache = cache.to(dtype).to(torch.float32)
        self.register_buffer("cos_sin_cache", cache, persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                offsets: torch.Tensor | None = None):
        """Pairs element j with j + D/2 of every head and writes the rotated pair to (2j, 2j+1)
        (rotary_embedding.py:19-24); q.k is unchanged by the common permutation."""
        if offsets is not None:
            positions = positions + offsets
        if self.cos_sin_cache.dtype != torch.float32:   # model.to(bf16) casts buffers too
            self.cos_sin_cache = self.cos_sin_cache.float()
        return ops.rope(positions, query, key, self.cos_sin_cache, self.head_size)

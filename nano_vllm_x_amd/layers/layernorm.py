"""RMSNorm module backed by the HIP kernel (interface of nanovllmx/layers/layernorm.py:6-36)."""
from __future__ import annotations

import torch
from torch import nn

from .. import distributed as tpd
from .. import ops


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        """y = x * rsqrt(mean(x^2) + eps) * weight in fp32, rounded once; with `residual`, x + residual is
        normalised unrounded and returned rounded as the new residual: (y, residual)."""
        if isinstance(x, tpd.Unreduced):
            # a row-parallel projection's output, not yet summed over the tensor-parallel ranks: the direct-peer
            # exchange, the sum and this norm are one launch
            return tpd.peer_allreduce_norm(x, self.weight, self.variance_epsilon, residual)
        if x.dtype == torch.float32 and x.dim() == 3:
            # fp32 partial products (splits, rows, hidden) left by a K-split decode projection: the sum is
            # rounded to bf16 first, which is what the unsplit projection would have handed over
            return ops.rmsnorm_splitk(x, self.weight, self.variance_epsilon, residual)
        return ops.rmsnorm(x, self.weight, self.variance_epsilon, residual)

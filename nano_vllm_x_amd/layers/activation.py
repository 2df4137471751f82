"""SwiGLU gate backed by the HIP kernel (interface of nanovllmx/layers/activation.py:5-18)."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


class SiluAndMul(nn.Module):
    """x -> silu(x[..., :d]) * x[..., d:], d = x.shape[-1] // 2."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.silu_mul(x if x.is_contiguous() else x.contiguous())

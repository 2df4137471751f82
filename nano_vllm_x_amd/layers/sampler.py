"""Token sampler backed by the HIP kernel (interface of nanovllmx/layers/sampler.py:4-55)."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


class Sampler(nn.Module):
    """temperature -> top-k -> top-p -> min-p -> categorical draw, one token per row.

    The draw is a Gumbel-max over Philox4x32-10 noise keyed by (seed, call counter): reproducible,
    identical on every tensor-parallel rank, exact in distribution; it is not bit-comparable with
    torch.multinomial, which the reference uses (sampler.py:53)."""

    def __init__(self, seed: int = 0):
        super().__init__()
        self.seed = seed
        self.calls = 0

    def forward(self, logits: torch.Tensor, temperatures: torch.Tensor | None = None, min_ps: torch.Tensor | None = None,
                top_ps: torch.Tensor | None = None, top_ks: torch.Tensor | None = None, out: torch.Tensor | None = None):
        if temperatures is None and min_ps is None and top_ps is None and top_ks is None:
            return torch.argmax(logits, dim=-1)   # sampler.py:28-36
        b, v = logits.shape
        dev = logits.device
        if temperatures is None:
            temperatures = torch.ones(b, dtype=torch.float32, device=dev)
        if min_ps is None:
            min_ps = torch.zeros(b, dtype=torch.float32, device=dev)
        if top_ps is None:
            top_ps = torch.ones(b, dtype=torch.float32, device=dev)
        if top_ks is None:
            top_ks = torch.full((b,), v, dtype=torch.int64, device=dev)
        if logits.dtype == torch.float16:
            logits = logits.float()   # the kernel reads bfloat16 or fp32 logits (the reference samples in fp32: sampler.py:37-38)
        if logits.stride(-1) != 1:
            logits = logits.contiguous()
        self.calls += 1
        return ops.sample(logits, temperatures.float().flatten(), top_ks.to(torch.int64).flatten(), top_ps.float().flatten(),
                          min_ps.float().flatten(), self.seed, self.calls, out=out)

    def forward_shard(self, logits: torch.Tensor, temperatures: torch.Tensor, vocab_offset: int, vocab: int) -> torch.Tensor:
        """A tensor-parallel rank's part of a step whose rows have no filter: the draw over this rank's shard of the
        vocabulary as one 64-bit key per row (int64 bit patterns); `ops.sample_pick` over the ranks' keys gives the
        token `forward` would draw from the gathered row.  Counts as one call, like `forward`, so the ranks' and the
        gathered path's (seed, counter) pairs stay the same."""
        if logits.dtype == torch.float16:
            logits = logits.float()
        if logits.stride(-1) != 1:
            logits = logits.contiguous()
        self.calls += 1
        return ops.sample_shard(logits, temperatures.float().flatten(), vocab_offset, vocab, self.seed, self.calls)

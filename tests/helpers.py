"""Shared test utilities: golden-file loading, bf16 bit views, the bench.py request stream."""
import hashlib
import json
import os
import random

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def golden_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def from_bits(a: np.ndarray) -> torch.Tensor:
    """uint16 bit pattern -> bf16 tensor."""
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def to_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def digest(obj) -> str:
    return hashlib.sha1(json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def bf16_ulp_diff(a: torch.Tensor, b: torch.Tensor) -> int:
    """Largest distance, in bf16 units in the last place, between two bf16 tensors."""
    def key(t):
        i = t.detach().cpu().contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return int((key(a) - key(b)).abs().max())


def bench_workload(num_seqs=256, max_in=1024, max_out=1024, seed=0):
    """Request stream of the reference's bench.py (bench.py:13-31): same RNG call order."""
    rnd = random.Random()
    rnd.seed(seed)
    prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(100, max_in))] for _ in range(num_seqs)]
    max_tokens = [rnd.randint(100, max_out) for _ in range(num_seqs)]
    return prompts, max_tokens

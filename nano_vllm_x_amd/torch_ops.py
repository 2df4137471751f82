"""`torch.ops.nvx.*`: the C-ABI kernels as PyTorch custom ops (device key only, no CPU kernels).

Importing this module registers the ops.  They call the same wrappers as `nano_vllm_x_amd.ops`; the
engine itself calls those wrappers directly (a dispatcher hop per launch buys nothing inside a
captured graph), so this module is for callers that want operator objects: module graphs written
against `torch.ops`, `torch.library.opcheck`, export.
"""
from __future__ import annotations

import torch

from . import ops

_LIB = torch.library.Library("nvx", "DEF")
_LIB.define("store_kvcache(Tensor k, Tensor v, Tensor(a!) k_cache, Tensor(b!) v_cache, Tensor kv_indices) -> ()")
_LIB.define("rmsnorm(Tensor x, Tensor weight, float eps) -> Tensor")
_LIB.define("rmsnorm_residual(Tensor x, Tensor residual, Tensor weight, float eps) -> (Tensor, Tensor)")
_LIB.define("rope(Tensor positions, Tensor q, Tensor k, Tensor cos_sin) -> (Tensor, Tensor)")
_LIB.define("silu_mul(Tensor x) -> Tensor")
_LIB.define("sample(Tensor logits, Tensor temperatures, Tensor top_ks, Tensor top_ps, Tensor min_ps, int seed, int offset) -> Tensor")


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


for _name, _fn in (("store_kvcache", _store_kvcache), ("rmsnorm", _rmsnorm), ("rmsnorm_residual", _rmsnorm_residual),
                   ("rope", _rope), ("silu_mul", _silu_mul), ("sample", _sample)):
    _LIB.impl(_name, _fn, "CUDA")   # PyTorch-ROCm's device key is spelled CUDA

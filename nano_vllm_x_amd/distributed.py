"""Tensor-parallel process group: one process per GPU, RCCL over xGMI (gloo on CPU for tests).

Covers what the hot path uses of nanovllmx/distributed/{parallel_state,communication_op}.py: the TP
group, the in-place all-reduce behind row-parallel linears and the vocab-parallel embedding
(communication_op.py:41-52,153-156), the last-dim all-gather behind the LM head (:9-39,158-162) and a
broadcast of sampled ids.  The path shards by KV head, so attention itself needs no exchange.
"""
from __future__ import annotations

import datetime
import os

import torch
import torch.distributed as dist

_TP_GROUP = None
_TP_RANK = 0
_TP_SIZE = 1


def init_tensor_parallel(tp_size: int, rank: int | None = None, backend: str | None = None, port: int = 29500,
                         device_ids: list | None = None):
    """Join (or create) the process group.  Under torchrun the RANK/WORLD_SIZE/MASTER_* variables are
    used; a single process with tp_size 1 needs no group at all."""
    global _TP_GROUP, _TP_RANK, _TP_SIZE
    if tp_size <= 1:
        _TP_GROUP, _TP_RANK, _TP_SIZE = None, 0, 1
        return 0
    if backend is None:
        # NVX_TP_BACKEND=gloo lets several ranks share one GPU (rehearsals on a one-GPU box: RCCL refuses
        # two ranks on one device); collectives then stage through the host and cannot be graph-captured
        backend = os.environ.get("NVX_TP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        if rank is None:
            rank = int(os.environ.get("RANK", "0"))
        if "MASTER_ADDR" in os.environ and "WORLD_SIZE" in os.environ:
            init_method = "env://"
        else:
            init_method = f"tcp://127.0.0.1:{port}"
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", rank))
            dev = device_ids[local] if device_ids else local
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, init_method=init_method, world_size=tp_size, rank=rank,
                                timeout=datetime.timedelta(seconds=120), **kw)
    assert dist.get_world_size() == tp_size, "world size must equal tp_size (pipeline stages are not built)"
    _TP_GROUP, _TP_RANK, _TP_SIZE = dist.group.WORLD, dist.get_rank(), tp_size
    return _TP_RANK


def tp_rank() -> int:
    return _TP_RANK


def tp_size() -> int:
    return _TP_SIZE


def _via_host(x: torch.Tensor) -> bool:
    """gloo with device tensors (ranks sharing one GPU in tests and rehearsals): its own device staging
    was seen to hang intermittently with two processes on one GPU, so the tensor is staged through the
    host here, synchronously.  RCCL never takes this path."""
    return x.is_cuda and dist.get_backend(_TP_GROUP) == "gloo"


def tp_all_reduce(x: torch.Tensor) -> torch.Tensor:
    if _TP_SIZE > 1:
        if _via_host(x):
            h = x.float().cpu() if x.dtype == torch.bfloat16 else x.cpu()
            dist.all_reduce(h, group=_TP_GROUP)
            x.copy_(h)
        else:
            dist.all_reduce(x, group=_TP_GROUP)
    return x


def tp_all_gather_last_dim(x: torch.Tensor) -> torch.Tensor:
    """Concatenate the ranks' shards along the last dimension."""
    if _TP_SIZE == 1:
        return x
    x = x.contiguous()
    if _via_host(x):
        h = torch.empty(_TP_SIZE * x.numel(), dtype=x.dtype)
        dist.all_gather_into_tensor(h, x.view(-1).cpu(), group=_TP_GROUP)
        flat = h.to(x.device)
    else:
        flat = torch.empty(_TP_SIZE * x.numel(), dtype=x.dtype, device=x.device)   # rank-major concatenation
        dist.all_gather_into_tensor(flat, x.view(-1), group=_TP_GROUP)
    out = flat.view((_TP_SIZE,) + tuple(x.shape))
    return out.movedim(0, -2).reshape(x.shape[:-1] + (_TP_SIZE * x.shape[-1],))


def tp_broadcast(x: torch.Tensor, src: int = 0) -> torch.Tensor:
    if _TP_SIZE > 1:
        if _via_host(x):
            h = x.cpu()
            dist.broadcast(h, src=src, group=_TP_GROUP)
            x.copy_(h)
        else:
            dist.broadcast(x, src=src, group=_TP_GROUP)
    return x


def tp_all_gather_min_int(value: int, device) -> int:
    """Smallest `value` over the ranks (KV pool sizing, model_runner.py:166-169)."""
    if _TP_SIZE == 1:
        return value
    t = torch.tensor([value], dtype=torch.int64, device="cpu" if dist.get_backend(_TP_GROUP) == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=_TP_GROUP)
    return int(t.item())


def barrier():
    if _TP_SIZE > 1:
        dist.barrier(group=_TP_GROUP)


def shutdown():
    global _TP_GROUP, _TP_RANK, _TP_SIZE
    if dist.is_initialized():
        dist.destroy_process_group()
    _TP_GROUP, _TP_RANK, _TP_SIZE = None, 0, 1

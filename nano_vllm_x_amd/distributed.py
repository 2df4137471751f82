"""Process groups: one process per GPU, RCCL over xGMI (gloo on CPU for tests).

Covers what the hot path uses of nanovllmx/distributed/{parallel_state,communication_op}.py: the TP
group, the in-place all-reduce behind row-parallel linears and the vocab-parallel embedding
(communication_op.py:41-52,153-156), the last-dim all-gather behind the LM head (:9-39,158-162) and a
broadcast of sampled ids.  The path shards by KV head, so attention itself needs no exchange.

Pipeline stages (opt-in, pp_size > 1): world = tp_size * pp_size, rank = pp_rank * tp_size + tp_rank
(worker.py:40-41), a TP group per stage, and point-to-point transfers of the activations between the
ranks of neighbouring stages that hold the same TP rank: each sends 1/tp of the tensor and the
receiving stage rebuilds it with an all-gather (communication_op.py:236-346), so an xGMI link carries
every byte once.  Shapes are known on both sides (every rank runs the same scheduler), so no metadata
message precedes the payload.
"""
from __future__ import annotations

import datetime
import os
import sys

import torch
import torch.distributed as dist

_TP_GROUP = None
_TP_RANK = 0
_TP_SIZE = 1
_PP_RANK = 0
_PP_SIZE = 1
_FORCED = False    # one rank, but issue every collective anyway (NVX_FORCE_COLLECTIVES=1)
_PEER = None       # ops.PeerGroup of this stage's TP group: decode-sized all-reduces go peer to peer, not through RCCL
PEER_STATE = "off"  # why the direct-peer all-reduce is (not) in use: reported by bench.py


class Unreduced:
    """The output of a row-parallel projection that has not been summed over the ranks yet: the RMSNorm that consumes
    it does the sum itself (one kernel: exchange with the peers, add, normalise)."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t


def init_tensor_parallel(tp_size: int, rank: int | None = None, backend: str | None = None, port: int = 29500,
                         device_ids: list | None = None, pp_size: int = 1):
    """Join (or create) the process group.  Under torchrun the RANK/WORLD_SIZE/MASTER_* variables are
    used; a single process with tp_size 1 (and one stage) needs no group at all.  Returns the global rank."""
    global _TP_GROUP, _TP_RANK, _TP_SIZE, _PP_RANK, _PP_SIZE, _FORCED
    world = tp_size * pp_size
    if world <= 1:
        _TP_GROUP, _TP_RANK, _TP_SIZE, _PP_RANK, _PP_SIZE = None, 0, 1, 0, 1
        _FORCED = os.environ.get("NVX_FORCE_COLLECTIVES") == "1"
        if _FORCED:
            # A one-rank RCCL group: the model then runs the code path of a tensor-parallel rank (bf16 row-parallel
            # outputs handed to all-reduce, masked embedding, LM-head gather, id broadcast) with every collective
            # really issued -- also inside the captured decode graph.  The only way to exercise RCCL under hipGraph
            # capture on a box with one GPU.
            if not dist.is_initialized():
                dev = device_ids[0] if device_ids else 0
                torch.cuda.set_device(dev)
                dist.init_process_group(backend=backend or "nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0,
                                        device_id=torch.device("cuda", dev) if (backend or "nccl") == "nccl" else None)
            _TP_GROUP = dist.group.WORLD
        return 0
    _FORCED = False
    if backend is None:
        # NVX_TP_BACKEND=gloo lets several ranks share one GPU (rehearsals on a one-GPU box: RCCL refuses
        # two ranks on one device); collectives then stage through the host and cannot be graph-captured
        backend = os.environ.get("NVX_TP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        if rank is None:
            # nobody told this process which rank it is: without a launcher's variables it would sit in the
            # rendezvous below waiting for peers that were never started
            from .launch import require_launcher
            require_launcher(world, f"tp_size={tp_size}, pp_size={pp_size}")
            rank = int(os.environ["RANK"])
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
        dist.init_process_group(backend=backend, init_method=init_method, world_size=world, rank=rank,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("NVX_DIST_TIMEOUT_S", "600"))), **kw)
    assert dist.get_world_size() == world, f"world size {dist.get_world_size()} must equal tp_size * pp_size = {world}"
    me = dist.get_rank()
    _TP_RANK, _TP_SIZE, _PP_RANK, _PP_SIZE = me % tp_size, tp_size, me // tp_size, pp_size
    if pp_size == 1:
        _TP_GROUP = dist.group.WORLD
    else:   # every rank creates every stage's group, in the same order
        for stage in range(pp_size):
            g = dist.new_group(list(range(stage * tp_size, (stage + 1) * tp_size))) if tp_size > 1 else None
            if stage == _PP_RANK:
                _TP_GROUP = g
    return me


def collectives_active() -> bool:
    """True when row-parallel outputs must go through all-reduce (several ranks, or one rank that issues its
    collectives anyway)."""
    return _TP_SIZE > 1 or _FORCED


def tp_rank() -> int:
    return _TP_RANK


def tp_size() -> int:
    return _TP_SIZE


def pp_rank() -> int:
    return _PP_RANK


def pp_size() -> int:
    return _PP_SIZE


def pp_is_first() -> bool:
    return _PP_RANK == 0


def pp_is_last() -> bool:
    return _PP_RANK == _PP_SIZE - 1


def _via_host(x: torch.Tensor) -> bool:
    """gloo with device tensors (ranks sharing one GPU in tests and rehearsals): its own device staging
    was seen to hang intermittently with two processes on one GPU, so the tensor is staged through the
    host here, synchronously.  RCCL never takes this path."""
    return x.is_cuda and dist.get_backend() == "gloo"


def decode_graph_capturable(rows: int) -> bool:
    """Can the all-reduces of a decode step of `rows` rows be recorded into a graph?  RCCL's can; gloo moves tensors
    through host memory, so under gloo only the sizes the direct-peer exchange takes."""
    if not collectives_active() or not dist.is_initialized() or dist.get_backend(_TP_GROUP) != "gloo":
        return True
    return _PEER is not None and rows <= _PEER.max_rows


def init_peer_allreduce(max_rows: int, hidden: int, dtype: torch.dtype, device) -> bool:
    """Set up the direct-peer all-reduce for this stage's TP group (every rank of the group calls this at the same
    point): allocate the landing buffer, exchange the inter-process handles through the process group, map the
    peers, then PROVE the exchange on known data before trusting it -- 32 all-reduces whose result every rank can
    compute by itself.  All ranks must pass, otherwise all fall back to the library all-reduce together.
    NVX_PEER_AR=0 keeps it off; NVX_PEER_TIMEOUT_MS bounds the waits inside the kernels (default 10 s)."""
    global _PEER, PEER_STATE
    from . import ops, torch_ops  # noqa: F401  (registers torch.ops.nvx.*)
    want = os.environ.get("NVX_PEER_AR", "1" if _TP_SIZE > 1 else "0")
    if want == "0" or not collectives_active() or _PEER is not None or torch.device(device).type != "cuda":
        if _PEER is None:
            PEER_STATE = "off (NVX_PEER_AR=0)" if want == "0" and _TP_SIZE > 1 else "off"
        return _PEER is not None
    tp, rank = max(_TP_SIZE, 1), _TP_RANK
    group, err = None, None
    sharing = ranks_on_my_device(device) if tp > 1 else 1
    if os.environ.get("NVX_PEER_SHARED_DEVICE") == "0":
        sharing = 1            # size the group as on GPUs of their own although the ranks share one (tests of the sizing)
    if sharing > 1:
        # Ranks rehearsed on ONE device: a peer's exchange kernel waits for this rank with a workgroup (four waves of
        # ~80 registers) per row, spread over the CUs.  A kernel of this rank that needs most of a SIMD's registers
        # cannot be placed beside the waiting waves of several peers, and a rank that cannot run is a rank everybody
        # waits for until the time-out.  So the whole-SIMD gate/up projection (used from 112 rows) stays off (the
        # library reads the variable at its first projection, which comes later), and below the exchange is kept to so
        # few rows that all peers together put at most one waiting wave on a SIMD; taller batches take the library
        # all-reduce.  Ranks on GPUs of their own never share a CU.
        os.environ.setdefault("NVX_LIN_GATEUP_WIDE_MIN", "1000000")
    # one routing limit for the whole group (a rank that read another NVX_PEER_MAX_PUSH_BYTES would send a tensor through
    # the library while its peers wait for it in the exchange kernel): the smallest any rank holds
    push = ops.PeerGroup.MAX_PUSH_BYTES
    if tp > 1:
        pushes = [None] * tp
        dist.all_gather_object(pushes, push, group=_TP_GROUP)
        push = min(pushes)
    # the landing buffer is sized for the rows that will ever take this path: the decode batch, capped by the limit above
    rows_cap = min(max(max_rows, 1), 512, ops.PeerGroup.rows_within_push_cap(tp, hidden, push))
    if sharing > 1:
        rows_cap = min(rows_cap, max(8, 128 // (sharing - 1)))
    try:
        if tp > 8:
            raise RuntimeError("more than 8 ranks")
        group = ops.PeerGroup(tp, rank, rows_cap, hidden, int(os.environ.get("NVX_PEER_TIMEOUT_MS", "0")))
        group.max_push_bytes = push
        handle = group.export_handle() if tp > 1 else b""
    except RuntimeError as e:      # every rank still takes part in the exchange below, so nobody waits for a rank that failed
        err, handle = str(e), None
    handles = [handle]
    if tp > 1:
        handles = [None] * tp
        dist.all_gather_object(handles, handle, group=_TP_GROUP)
    if err is None and all(h is not None for h in handles):
        try:
            for p in range(tp):
                if p != rank:
                    group.import_handle(p, handles[p])
            # 32 exchanges at the group's full height on data every rank can derive by itself (small integers: sums are
            # exact in both 16-bit formats), so that both slot parities are reused many times; a few milliseconds in all
            rows = group.max_rows
            base = torch.arange(rows * hidden, device=device).view(rows, hidden)
            for it in range(32):
                pattern = (base + it) % 7
                x = (pattern * (rank + 1)).to(dtype)
                torch.ops.nvx.peer_allreduce(group.comm, x)
                torch.cuda.synchronize(device)
                if group.status() != 0 or not torch.equal(x, (pattern * (tp * (tp + 1) // 2)).to(dtype)):
                    raise RuntimeError("self-test: the exchanged sum is wrong" if group.status() == 0 else "self-test: a wait ran out")
        except RuntimeError as e:
            err = str(e)
    elif err is None:
        err = "a peer could not create its end"
    oks = [err is None]
    if tp > 1:
        oks = [None] * tp
        dist.all_gather_object(oks, err is None, group=_TP_GROUP)
    if all(oks):
        _PEER = group
        PEER_STATE = (f"direct-peer ({tp} ranks, exchanges of up to {group.max_rows} rows x {hidden}, "
                      f"{max(group.exchange_bytes >> 10, 1)} KiB landing buffer per rank; taller ones: library all-reduce)")
        return True
    if group is not None:
        group.close()
    PEER_STATE = f"library all-reduce (direct-peer set-up failed: {err or 'on another rank'})"
    if rank == 0:
        print(f"[nano_vllm_x_amd] {PEER_STATE}", file=sys.stderr, flush=True)
    return False


def peer_allreduce_active() -> bool:
    return _PEER is not None


def check_peer_status():
    """Raise if a wait inside a direct-peer exchange ever ran out: the sums since then are not to be trusted."""
    if _PEER is not None and _PEER.status() != 0:
        raise RuntimeError("direct-peer all-reduce: a rank waited longer than the time-out for a peer's rows "
                           "(a peer died or stalled); results since then are invalid")


def tp_reduce_later(x: torch.Tensor):
    """All-reduce of a row-parallel projection's output -- left to the RMSNorm that follows (Unreduced) when the
    direct-peer exchange can take the tensor, done now otherwise."""
    if _PEER is not None and _PEER.prefer(x):
        return Unreduced(x)
    return tp_all_reduce(x)


def tp_finish_reduce(x):
    """The tensor behind an Unreduced (summed now), or x itself."""
    if isinstance(x, Unreduced):
        t = x.t if x.t.is_contiguous() else x.t.contiguous()
        torch.ops.nvx.peer_allreduce(_PEER.comm, t)
        return t
    return x


def peer_allreduce_norm(x: Unreduced, weight: torch.Tensor, eps: float, residual=None):
    """RMSNorm.forward on an Unreduced input: exchange, sum, (+ residual,) normalise in one launch."""
    y, res = torch.ops.nvx.peer_allreduce_norm(_PEER.comm, x.t, residual, weight, float(eps))
    return y if residual is None else (y, res)


def tp_all_reduce(x: torch.Tensor) -> torch.Tensor:
    if _TP_SIZE > 1 or _FORCED:
        if _PEER is not None and x.is_contiguous() and _PEER.prefer(x):
            torch.ops.nvx.peer_allreduce(_PEER.comm, x)
        elif _via_host(x):
            h = x.float().cpu() if x.dtype == torch.bfloat16 else x.cpu()
            dist.all_reduce(h, group=_TP_GROUP)
            x.copy_(h)
        else:
            dist.all_reduce(x, group=_TP_GROUP)
    return x


def tp_all_gather_last_dim(x: torch.Tensor) -> torch.Tensor:
    """Concatenate the ranks' shards along the last dimension."""
    if _TP_SIZE == 1 and not _FORCED:
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


def tp_all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    """(n,) from every rank of the TP group -> (tp, n), rank by rank."""
    if _TP_SIZE == 1 and not _FORCED:
        return x.view(1, -1)
    x = x.contiguous()
    n = max(_TP_SIZE, 1)
    if _via_host(x):
        h = torch.empty(n * x.numel(), dtype=x.dtype)
        dist.all_gather_into_tensor(h, x.view(-1).cpu(), group=_TP_GROUP)
        return h.to(x.device).view(n, -1)
    out = torch.empty(n * x.numel(), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.view(-1), group=_TP_GROUP)
    return out.view(n, -1)


def tp_broadcast(x: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Broadcast within this stage's TP group; `src` is the TP rank of the sender."""
    if _TP_SIZE > 1:
        g_src = _PP_RANK * _TP_SIZE + src
        if _via_host(x):
            h = x.cpu()
            dist.broadcast(h, src=g_src, group=_TP_GROUP)
            x.copy_(h)
        else:
            dist.broadcast(x, src=g_src, group=_TP_GROUP)
    return x


def broadcast_sampled_ids(x: torch.Tensor) -> torch.Tensor:
    """The sampled ids, from TP rank 0 of the last stage to every rank of the job: each rank's scheduler
    then applies the same tokens and stays in lockstep without host-side messages."""
    if _TP_SIZE * _PP_SIZE > 1 or _FORCED:
        src = (_PP_SIZE - 1) * _TP_SIZE
        if _via_host(x):
            h = x.cpu()
            dist.broadcast(h, src=src)
            x.copy_(h)
        else:
            dist.broadcast(x, src=src)
    return x


def all_ranks_min_int(value: int, device) -> int:
    """Smallest `value` over every rank of the job (KV pool sizing, model_runner.py:166-169: all stages
    and shards must agree on the number of slots because they run the same scheduler)."""
    if _TP_SIZE * _PP_SIZE == 1:
        return value
    t = torch.tensor([value], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


tp_all_gather_min_int = all_ranks_min_int


def ranks_on_my_device(device) -> int:
    """How many ranks of the job run on this rank's GPU: 1 with one process per GPU (the product layout); more only when
    several ranks are rehearsed on one device."""
    if _TP_SIZE * _PP_SIZE == 1 or torch.device(device).type != "cuda":
        return 1
    mine = torch.device(device).index
    mine = torch.cuda.current_device() if mine is None else mine
    all_devices = [None] * dist.get_world_size()
    dist.all_gather_object(all_devices, mine)
    return max(1, sum(1 for d in all_devices if d == mine)) if dist.get_backend() == "gloo" else 1


# -- pipeline stages: activations between neighbouring stages ---------------------------------------
def _slice_for_peer(x: torch.Tensor) -> torch.Tensor:
    """This TP rank's 1/tp of the flattened tensor (communication_op.py:277-279); the whole tensor when
    its size does not divide."""
    flat = x.contiguous().view(-1)
    if _TP_SIZE > 1 and flat.numel() % _TP_SIZE == 0:
        return flat.view(_TP_SIZE, -1)[_TP_RANK]
    return flat


def pp_send_next(x: torch.Tensor):
    """Send `x` (identical on the stage's TP ranks) to the next stage."""
    assert not pp_is_last()
    part = _slice_for_peer(x)
    dst = (_PP_RANK + 1) * _TP_SIZE + _TP_RANK
    dist.send(part.cpu() if _via_host(part) else part, dst=dst)


def pp_recv_prev(shape, dtype, device) -> torch.Tensor:
    """Receive a tensor of a known shape from the previous stage."""
    assert not pp_is_first()
    n = 1
    for d in shape:
        n *= d
    sliced = _TP_SIZE > 1 and n % _TP_SIZE == 0
    src = (_PP_RANK - 1) * _TP_SIZE + _TP_RANK
    host = torch.device(device).type == "cuda" and dist.get_backend() == "gloo"
    part = torch.empty(n // _TP_SIZE if sliced else n, dtype=dtype, device="cpu" if host else device)
    dist.recv(part, src=src)
    if host:
        part = part.to(device)
    if sliced:   # rank-major concatenation restores the flattened tensor
        full = torch.empty(n, dtype=dtype, device=device)
        if dist.get_backend() == "gloo" and full.is_cuda:
            h = torch.empty(n, dtype=dtype)
            dist.all_gather_into_tensor(h, part.cpu(), group=_TP_GROUP)
            full.copy_(h)
        else:
            dist.all_gather_into_tensor(full, part, group=_TP_GROUP)
        part = full
    return part.view(tuple(shape))


def barrier():
    if _TP_SIZE * _PP_SIZE > 1:
        dist.barrier()


def shutdown():
    global _TP_GROUP, _TP_RANK, _TP_SIZE, _PP_RANK, _PP_SIZE, _FORCED, _PEER, PEER_STATE
    if _PEER is not None:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist.is_initialized() and _TP_SIZE * _PP_SIZE > 1:
            dist.barrier()          # no rank unmaps a landing buffer a peer's kernel may still write to
        _PEER.close()
        _PEER, PEER_STATE = None, "off"
    if dist.is_initialized():
        dist.destroy_process_group()
    _TP_GROUP, _TP_RANK, _TP_SIZE, _PP_RANK, _PP_SIZE, _FORCED = None, 0, 1, 0, 1, False

"""Direct-peer all-reduce (+ RMSNorm) of a tensor-parallel group: several ranks sharing the one GPU of the test box.

Each rank is a process of its own with its own landing buffer; the peers map it through inter-process handles
exactly as ranks on different GPUs do (there the writes cross xGMI, here they stay on the device -- the protocol,
the tags, the slot reuse and the graph replay are the same).  gloo only carries the 64-byte handles.

The sum is integer-exact work on 16-bit inputs: contributions are added in fp32 in rank order and rounded once, so
the expected bits are computed here with the same additions in torch and compared bit for bit; the norm that
follows is compared bit for bit with nvx_rmsnorm_splitk on the summed tensor (same thread layout, same order of
additions) and within one unit in the last place with the oracle's RMSNorm.
"""
from __future__ import annotations

import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

HIDDEN = 1024
MAX_ROWS = 64
EPS = 1e-6
# (rows, mode) of consecutive exchanges: 0 = sum in place, 1 = sum + norm, 2 = sum + residual + norm
CALLS = [(5, 0), (5, 2), (64, 2), (1, 1), (33, 2), (64, 0), (7, 2), (7, 2), (7, 2), (2, 1), (64, 2), (64, 2)]


def _inputs(call: int, rank: int, rows: int, hidden: int, dtype):
    g = torch.Generator().manual_seed(1000 * call + rank)
    return torch.randn(rows, hidden, generator=g).to(dtype)


def _expected_sum(call: int, tp: int, rows: int, hidden: int, dtype):
    acc = _inputs(call, 0, rows, hidden, dtype).float()
    for r in range(1, tp):
        acc = acc + _inputs(call, r, rows, hidden, dtype).float()
    return acc.to(dtype)


def _setup(tp, rank, port, hidden, max_rows, timeout_ms=0):
    import torch.distributed as dist

    from nano_vllm_x_amd import ops, torch_ops  # noqa: F401

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=tp, rank=rank)
    group = ops.PeerGroup(tp, rank, max_rows, hidden, timeout_ms)
    handles = [None] * tp
    dist.all_gather_object(handles, group.export_handle())
    for p in range(tp):
        if p != rank:
            group.import_handle(p, handles[p])
    dist.barrier()
    return group, dist


def _peer_worker(rank, tp, port, q, dtype_name, hidden):
    try:
        from nano_vllm_x_amd import ops
        from oracle.layers_oracle import rmsnorm_oracle

        dtype = getattr(torch, dtype_name)
        group, dist = _setup(tp, rank, port, hidden, MAX_ROWS)
        dev = torch.device("cuda", 0)
        gw = torch.Generator().manual_seed(7)
        weight = (1.0 + 0.1 * torch.randn(hidden, generator=gw)).to(dtype).to(dev)
        digest = []
        for call, (rows, mode) in enumerate(CALLS):
            x = _inputs(call, rank, rows, hidden, dtype).to(dev)
            want_sum = _expected_sum(call, tp, rows, hidden, dtype).to(dev)
            res = _inputs(call, 99, rows, hidden, dtype).to(dev) if mode == 2 else None
            if mode == 0:
                got = ops.peer_allreduce_norm(group, x)
                assert torch.equal(got, want_sum), f"call {call}: sum differs"
                digest.append(got.cpu())
                continue
            out = ops.peer_allreduce_norm(group, x, weight, EPS, residual=res)
            ref = ops.rmsnorm_splitk(want_sum.float()[None].contiguous(), weight, EPS, residual=res)
            y, r_out = (out, None) if mode == 1 else out
            y_ref, r_ref = (ref, None) if mode == 1 else ref
            if hidden <= 1024:
                assert torch.equal(y, y_ref), f"call {call}: norm differs from the split-K norm of the same sum"
            else:   # wide rows: the split-K norm sums the squares with 1024 threads, the exchange kernel with 256 -- another order
                d = (y.view(torch.int16).int() - y_ref.view(torch.int16).int()).abs().max().item()
                assert d <= 1, f"call {call}: norm is {d} units in the last place from the split-K norm of the same sum"
            if mode == 2:
                assert torch.equal(r_out, r_ref), f"call {call}: residual differs"
            o = rmsnorm_oracle(want_sum.cpu(), weight.cpu(), EPS, None if res is None else res.cpu())
            o_y = o if mode == 1 else o[0]
            ulp = (y.cpu().view(torch.int16).int() - o_y.view(torch.int16).int()).abs().max().item()
            assert ulp <= 1, f"call {call}: {ulp} ulp from the oracle's RMSNorm"
            digest.append(y.cpu())
        assert group.status() == 0

        # rows that are not contiguous (a column view of a wider tensor, as a fused projection's output may be)
        wide = torch.zeros(9, 2 * hidden, dtype=dtype, device=dev)
        wide[:, hidden:] = _inputs(300, rank, 9, hidden, dtype).to(dev)
        view = wide[:, hidden:]
        assert view.stride(0) == 2 * hidden
        y_v = ops.peer_allreduce_norm(group, view, weight, EPS)
        want_v = ops.rmsnorm_splitk(_expected_sum(300, tp, 9, hidden, dtype).to(dev).float()[None].contiguous(), weight, EPS)
        assert torch.equal(y_v, want_v), "strided rows: norm differs"
        digest.append(y_v.cpu())

        # a chain of exchanges captured in a graph and replayed: the tags advance on the device
        x0 = _inputs(500, rank, 16, hidden, dtype).to(dev)
        res0 = _inputs(501, 99, 16, hidden, dtype).to(dev)

        def chain(x, res):
            for _ in range(6):
                y, res = ops.peer_allreduce_norm(group, x, weight, EPS, residual=res)
                x = (y * (0.5 + 0.25 * rank)).to(dtype)      # the ranks' partial results differ again
            return y, res

        eager = [chain(x0, res0) for _ in range(3)]
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain(x0, res0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        dist.barrier()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gy, gres = chain(x0, res0)
        for rep in range(3):
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(gy, eager[0][0]) and torch.equal(gres, eager[0][1]), f"replay {rep} differs from the eager chain"
        assert all(torch.equal(e[0], eager[0][0]) for e in eager)
        assert group.status() == 0
        digest.append(gy.cpu())
        dist.barrier()
        group.close()
        dist.destroy_process_group()
        q.put((rank, [d.view(torch.int16).numpy() for d in digest], None))
    except Exception:   # noqa: BLE001
        q.put((rank, None, traceback.format_exc()))


def _spawn(target, tp, extra=()):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, tp, port, q) + tuple(extra)) for r in range(tp)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        rank, out, err = q.get(timeout=240)
        assert err is None, f"rank {rank} failed:\n{err}"
        results[rank] = out
    for p in procs:
        p.join(timeout=60)
    return results


@pytest.mark.parametrize("tp,dtype_name,hidden", [(2, "bfloat16", 1024), (4, "bfloat16", 1024), (2, "float16", 1024),
                                                  (2, "bfloat16", 4096), (4, "bfloat16", 5120), (8, "bfloat16", 1024)])
def test_peer_allreduce_norm_matches_the_sum_in_rank_order(tp, dtype_name, hidden):
    results = _spawn(_peer_worker, tp, (dtype_name, hidden))
    for r in range(1, tp):        # every rank holds the same bits
        assert len(results[r]) == len(results[0])
        for a, b in zip(results[0], results[r]):
            assert np.array_equal(a, b), f"rank {r} and rank 0 hold different bits"


def _timeout_worker(rank, tp, port, q):
    try:
        import time

        from nano_vllm_x_amd import ops

        group, dist = _setup(tp, rank, port, HIDDEN, 8, timeout_ms=300)
        dev = torch.device("cuda", 0)
        x = torch.ones(4, HIDDEN, dtype=torch.bfloat16, device=dev)
        ops.peer_allreduce_norm(group, x)           # one good exchange
        torch.cuda.synchronize()
        ok_first = group.status() == 0 and bool((x == tp).all())
        # shapes the group was not made for are refused on the host, before any launch (a kernel handed a wrong
        # shape would write outside the landing buffers)
        for bad in (torch.ones(9, HIDDEN, dtype=torch.bfloat16, device=dev), torch.ones(4, HIDDEN // 2, dtype=torch.bfloat16, device=dev),
                    torch.ones(4, HIDDEN, dtype=torch.float32, device=dev)):
            try:
                ops.peer_allreduce_norm(group, bad)
            except RuntimeError:
                pass
            else:
                raise AssertionError(f"a tensor of shape {tuple(bad.shape)} {bad.dtype} was accepted")
        import ctypes
        from nano_vllm_x_amd import _lib
        lib = _lib.load()
        rc = lib.nvx_peer_allreduce_norm(group.comm, x.data_ptr(), HIDDEN, None, None, x.data_ptr(), None, 9, HIDDEN, ctypes.c_float(0.0), None)
        assert rc == -4 and b"9 rows" in lib.nvx_last_error(), (rc, lib.nvx_last_error())      # NVX_ENOSPACE from the C ABI itself
        dist.barrier()
        took = 0.0
        if rank == 0:                               # rank 1 never makes the second call: rank 0's wait must run out
            t0 = time.perf_counter()
            ops.peer_allreduce_norm(group, x)
            torch.cuda.synchronize()
            took = time.perf_counter() - t0
        status = group.status()
        dist.barrier()
        group.close()
        dist.destroy_process_group()
        q.put((rank, (ok_first, status, took), None))
    except Exception:   # noqa: BLE001
        q.put((rank, None, traceback.format_exc()))


def test_a_missing_peer_ends_the_wait_and_raises_the_status_word():
    """No wave may spin forever: a rank whose peer never arrives gives up after the group's time-out, finishes its
    launch and leaves 1 in the status word (which the engine turns into an error)."""
    results = _spawn(_timeout_worker, 2)
    ok0, status0, took0 = results[0]
    ok1, status1, _ = results[1]
    assert ok0 and ok1
    assert status0 == 1 and status1 == 0
    assert 0.25 <= took0 < 5.0, f"the wait took {took0:.2f} s for a 0.3 s time-out"


def _routing_worker(rank, tp, port, q):
    """bench.py's own configuration at 8 ranks (max_bs 256, hidden 1024): the group must come up, and exchanges route
    by size -- decode-sized ones peer to peer, taller ones through the library."""
    try:
        from nano_vllm_x_amd import distributed as tpd

        torch.cuda.set_device(0)
        tpd.init_tensor_parallel(tp, rank=rank, backend="gloo", port=port)
        dev = torch.device("cuda", 0)
        ok = tpd.init_peer_allreduce(256, HIDDEN, torch.bfloat16, dev)
        state = tpd.PEER_STATE
        group = tpd._PEER
        facts = {"ok": ok, "state": state}
        if ok:
            cap = 4 * 1024 * 1024 // ((tp - 1) * HIDDEN * 4)
            facts["max_rows"], facts["cap"] = group.max_rows, cap
            small = torch.full((min(29, group.max_rows), HIDDEN), float(rank + 1), dtype=torch.bfloat16, device=dev)
            tall = torch.full((256, HIDDEN), float(rank + 1), dtype=torch.bfloat16, device=dev)
            a = tpd.tp_reduce_later(small)
            b = tpd.tp_reduce_later(tall)
            facts["small_is_deferred"] = isinstance(a, tpd.Unreduced)
            facts["tall_is_reduced_now"] = torch.is_tensor(b) and bool((b == tp * (tp + 1) // 2).all())
            summed = tpd.tp_finish_reduce(a)
            torch.cuda.synchronize()
            facts["small_sum_ok"] = bool((summed == tp * (tp + 1) // 2).all())
            facts["accepts_not_prefers"] = (not group.prefer(tall)) and (not group.accepts(tall)) and group.accepts(small)
            facts["status"] = group.status()
        tpd.shutdown()
        q.put((rank, facts, None))
    except Exception:   # noqa: BLE001
        q.put((rank, None, traceback.format_exc()))


@pytest.mark.parametrize("as_on_own_gpus", [True, False])
def test_the_group_comes_up_at_the_bench_configuration_and_routes_by_size(as_on_own_gpus, monkeypatch):
    """Sized as for eight GPUs (NVX_PEER_SHARED_DEVICE=0: 146 rows fit the push limit), and as the eight ranks of a rehearsal
    on this one GPU get it by themselves: 128 / 7 rows, so that the waiting waves of seven peers leave every SIMD room for
    this rank's kernels."""
    tp = 8
    if as_on_own_gpus:
        monkeypatch.setenv("NVX_PEER_SHARED_DEVICE", "0")
    else:
        monkeypatch.delenv("NVX_PEER_SHARED_DEVICE", raising=False)
    results = _spawn(_routing_worker, tp)
    for r in range(tp):
        f = results[r]
        assert f["ok"] and f["state"].startswith("direct-peer"), f["state"]
        assert f["cap"] == 146 and f["max_rows"] == (146 if as_on_own_gpus else 18)
        assert f["small_is_deferred"] and f["tall_is_reduced_now"] and f["small_sum_ok"] and f["accepts_not_prefers"]
        assert f["status"] == 0

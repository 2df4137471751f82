"""Time of one direct-peer all-reduce + RMSNorm (nvx_peer_allreduce_norm) inside a replayed graph.

    python tools/peer_bench.py [--tp 2 4 8] [--rows 1 29 123 256] [--hidden 1024]

The ranks are processes that share GPU 0 (a one-GPU box has no second device): what is timed is the launch,
the push of the row into the peers' landing buffers, the poll of the local one, the sum and the norm -- not a
link.  On separate GPUs the pushes cross xGMI (posted writes) and the polls stay local.  For scale, the same
chain with the one-rank kernel that sums K-split partials instead (nvx_rmsnorm_splitk, 4 splits) is timed too.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHAIN = 56      # exchanges per replay: the all-reduces of a 28-layer decode step


def _time_graph(fn, reps=30):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * CHAIN)


def _worker(rank, tp, port, q, rows_list, hidden):
    import torch.distributed as dist

    from nano_vllm_x_amd import ops

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=tp, rank=rank)
    group = ops.PeerGroup(tp, rank, max(rows_list), hidden)
    handles = [None] * tp
    dist.all_gather_object(handles, group.export_handle())
    for p in range(tp):
        if p != rank:
            group.import_handle(p, handles[p])
    dist.barrier()
    dev = torch.device("cuda", 0)
    w = torch.ones(hidden, dtype=torch.bfloat16, device=dev)
    out = {}
    for rows in rows_list:
        x = torch.randn(rows, hidden, device=dev).to(torch.bfloat16)
        res = torch.randn(rows, hidden, device=dev).to(torch.bfloat16)

        def chain():
            r = res
            for _ in range(CHAIN):
                _, r = ops.peer_allreduce_norm(group, x, w, 1e-6, residual=r)

        dist.barrier()
        out[rows] = round(_time_graph(chain), 2)
        assert group.status() == 0
    dist.barrier()
    group.close()
    dist.destroy_process_group()
    q.put((rank, out))


def _one_rank_reference(rows_list, hidden):
    from nano_vllm_x_amd import ops

    dev = torch.device("cuda", 0)
    w = torch.ones(hidden, dtype=torch.bfloat16, device=dev)
    out = {}
    for rows in rows_list:
        part = torch.randn(4, rows, hidden, device=dev)
        res = torch.randn(rows, hidden, device=dev).to(torch.bfloat16)

        def chain():
            r = res
            for _ in range(CHAIN):
                _, r = ops.rmsnorm_splitk(part, w, 1e-6, residual=r)

        out[rows] = round(_time_graph(chain), 2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tp", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--rows", type=int, nargs="+", default=[1, 29, 123, 256])
    ap.add_argument("--hidden", type=int, default=1024)
    args = ap.parse_args()
    result = {"hidden": args.hidden, "chain": CHAIN, "unit": "us per exchange+norm, graph replay, ranks sharing one GPU",
              "one_rank_splitk_norm_us": _one_rank_reference(args.rows, args.hidden)}
    ctx = mp.get_context("spawn")
    for tp in args.tp:
        rows = [r for r in args.rows if r * tp <= 1024]      # all ranks' workgroups must be resident together on the one GPU
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, tp, port, q, rows, args.hidden)) for r in range(tp)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(timeout=60)
        result[f"tp{tp}_us"] = {str(r): max(got[k][r] for k in got) for r in rows}
        print(json.dumps({f"tp{tp}_us": result[f"tp{tp}_us"]}), flush=True)
    print(json.dumps(result))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Micro-benchmarks of single kernels on the shapes of the bench workload (run on the GPU box).

  python tools/kernel_bench.py decode  [--every 8]     paged decode attention over the workload's decode batches
  python tools/kernel_bench.py prefill                 paged prefill attention on the workload's 256 prompts
  python tools/kernel_bench.py sampler                 sampler on (B, 151936) logits

Timing: HIP events on torch's current stream (the stream the kernels are launched on) around a run of
back-to-back launches that rotate through the layers of a KV pool far larger than the Infinity Cache.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench
from nano_vllm_x_amd import ops

DEV = "cuda:0"
HQ, HKV, LAYERS = 16, 8, 28


def timed(fn, reps, graph=True):
    """Seconds per call of fn.  With graph=True the `reps` calls are captured into one hipGraph and the
    replay is timed, so short kernels are not paced by the host's launch rate."""
    fn()
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(reps):
                    fn()
        torch.cuda.current_stream().wait_stream(s)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def cmd_decode(args):
    global HQ, HKV
    if args.heads:
        HQ, HKV = (int(x) for x in args.heads.split(","))
    prompts, max_tokens = bench.workload(256)
    slots = 200_000
    batches, n_dec, kv_rows = bench.decode_batches_of_workload(prompts, max_tokens, slots, 256, 4096, args.every)
    layers = LAYERS if HKV >= 8 else 28 * 8 // HKV        # keep the pool larger than the Infinity Cache at few kv heads
    if args.hkv1:
        # same bytes, but every (sequence, kv head) becomes its own sequence of contiguous 256-byte rows:
        # isolates what the 2 KB row stride / partial-row reads cost
        HQ, HKV = 2, 1
        nb = []
        for lens, flat in batches:
            lens8 = np.repeat(lens, 8).astype(np.int32)
            flat8 = np.arange(int(lens8.sum()), dtype=np.int32)
            nb.append((lens8, flat8))
        batches = nb
        slots = 1_600_000
    kc = torch.randn(layers, slots, HKV, 128, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(layers, slots, HKV, 128, device=DEV, dtype=torch.bfloat16)
    ws = ops.DecodeWorkspace(2048 if args.hkv1 else 256, HQ, HKV, DEV)
    tot_b = tot_t = 0.0
    rows_out = []
    buckets = {}
    layer = [0]
    for i, (lens, flat) in enumerate(batches):
        b = len(lens)
        kv_len = torch.from_numpy(lens).to(DEV)
        kv_start = torch.from_numpy(np.concatenate(([0], np.cumsum(lens[:-1]))).astype(np.int64)).to(DEV)
        idx = torch.from_numpy(flat).to(DEV)
        q = torch.randn(b, HQ * 128, device=DEV, dtype=torch.bfloat16)
        out = torch.empty_like(q)
        if args.unfused:   # what the fused variant replaces: the q/k-norm + rotary + scatter kernel, then attention on finished queries
            from nano_vllm_x_amd.layers.rotary_embedding import build_cos_sin_cache
            qkv = torch.randn(b, (HQ + 2 * HKV) * 128, device=DEV, dtype=torch.bfloat16)
            pos = (kv_len - 1).to(torch.int64)
            new_slots = idx[(kv_start + kv_len - 1)].to(torch.int64)
            table = build_cos_sin_cache(128, 4096, 1e6).to(DEV)
            nw = torch.ones(128, device=DEV, dtype=torch.bfloat16)
            qbuf = torch.empty(b, HQ * 128, device=DEV, dtype=torch.bfloat16)
            ops.decode_plan(kv_len, kv_start, ws)

            def run():
                ops.qknorm_rope_store(qkv, pos, table, nw, nw, 1e-6, new_slots, kc[layer[0]], vc[layer[0]], HQ, HKV, q_out=qbuf)
                ops.paged_decode(qbuf, kc[layer[0]], vc[layer[0]], idx, b, ws, 128 ** -0.5, out=out)
                layer[0] = (layer[0] + 1) % layers
        elif args.qkv:   # the variant fed with the raw fused projection (norm + rotary + scatter inside)
            from nano_vllm_x_amd.layers.rotary_embedding import build_cos_sin_cache
            qkv = torch.randn(b, (HQ + 2 * HKV) * 128, device=DEV, dtype=torch.bfloat16)
            pos = (kv_len - 1).to(torch.int64)
            new_slots = idx[(kv_start + kv_len - 1)].to(torch.int64)
            table = build_cos_sin_cache(128, 4096, 1e6).to(DEV)
            nw = torch.ones(128, device=DEV, dtype=torch.bfloat16)
            ops.decode_plan(kv_len, kv_start, ws, out_slots=new_slots)

            def run():
                ops.paged_decode_qkv(qkv, pos, table, nw, nw, 1e-6, kc[layer[0]], vc[layer[0]], idx, b, ws, 128 ** -0.5, out=out)
                layer[0] = (layer[0] + 1) % layers
        else:
            ops.decode_plan(kv_len, kv_start, ws)

            def run():
                ops.paged_decode(q, kc[layer[0]], vc[layer[0]], idx, b, ws, 128 ** -0.5, out=out)
                layer[0] = (layer[0] + 1) % layers

        t = timed(run, args.reps)
        rows = int(lens.sum())
        alg = rows * 2 * HKV * 128 * 2 + 2 * b * HQ * 128 * 2 + 4 * rows
        tot_b += alg
        tot_t += t
        bucket = next(hi for hi in (2, 4, 8, 16, 32, 64, 128, 256) if b <= hi)
        acc = buckets.setdefault(bucket, [0, 0.0, 0.0])
        acc[0] += 1; acc[1] += t; acc[2] += alg
        if i % max(1, len(batches) // 12) == 0:
            rows_out.append((i * args.every, b, rows, round(t * 1e6, 1), round(alg / t / 1e9), int(ws.plan[0]), int(ws.plan[1])))
    print("step  B  kv_rows  us  GB/s  items chunk")
    for r in rows_out:
        print(*r)
    print(json.dumps({"kernel": "paged_decode(+merge)", "heads": [HQ, HKV], "batches": len(batches), "avg_us": round(tot_t / len(batches) * 1e6, 2),
                      "GBps": round(tot_b / tot_t / 1e9, 1), "frac_of_8TBps": round(tot_b / tot_t / 8e12, 4),
                      "variant": "qkv" if args.qkv else "unfused" if args.unfused else "plain",
                      "env": {k: v for k, v in os.environ.items() if k.startswith("NVX_DEC")},
                      # by decode batch size (sequences per step, upper edge of the bucket): launches sampled, mean us per launch pair, GB/s
                      "by_batch": {f"<={hi}": {"batches": n, "avg_us": round(tt / n * 1e6, 2), "GBps": round(bb / tt / 1e9, 1),
                                               "frac_of_8TBps": round(bb / tt / 8e12, 4)} for hi, (n, tt, bb) in sorted(buckets.items())}}))


def cmd_prefill(args):
    global HQ, HKV
    if args.heads:
        HQ, HKV = (int(x) for x in args.heads.split(","))
    prompts, _ = bench.workload(args.num_seqs)
    lens = [len(p) for p in prompts]
    if args.uniform_len:   # every sequence the same length: separates steady-state rate from per-workgroup overheads
        lens = [args.uniform_len] * max(1, 131072 // args.uniform_len)
    t_tok = sum(lens)
    slots = t_tok + 1024
    kc = torch.randn(4, slots, HKV, 128, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(4, slots, HKV, 128, device=DEV, dtype=torch.bfloat16)
    q = torch.randn(t_tok, HQ * 128, device=DEV, dtype=torch.bfloat16)
    indptr = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).to(DEV)
    idx = torch.arange(t_tok, dtype=torch.int32, device=DEV)
    plan = ops.prefill_plan_ws(len(lens), DEV)
    ops.prefill_plan(indptr, plan)
    out = torch.empty_like(q)
    layer = [0]

    def run():
        ops.paged_prefill(q, kc[layer[0]], vc[layer[0]], indptr, indptr, idx, plan, HQ, HKV, 128 ** -0.5, out=out)
        layer[0] = (layer[0] + 1) % 4

    t = timed(run, args.reps)
    flop = 4 * HQ * 128 * sum(l * (l + 1) // 2 for l in lens)
    print(json.dumps({"kernel": "paged_prefill", "tokens": t_tok, "ms": round(t * 1e3, 3), "TFLOPs": round(flop / t / 1e12, 1),
                      "frac_of_2.5PF": round(flop / t / 2.5e15, 4)}))


def cmd_qknorm(args):
    """Fused q/k norm + rotary + scatter on the bench workload's prefill (142 827 tokens, 16 q / 8 kv heads)."""
    from nano_vllm_x_amd.layers.rotary_embedding import build_cos_sin_cache
    prompts, _ = bench.workload(args.num_seqs)
    t = sum(len(p) for p in prompts)
    qkv = torch.randn(t, (HQ + 2 * HKV) * 128, device=DEV, dtype=torch.bfloat16)
    pos = torch.cat([torch.arange(len(p)) for p in prompts]).to(DEV)
    slots = torch.randperm(t + 1024, device=DEV)[:t]
    kc = torch.zeros(2, t + 1024, HKV, 128, device=DEV, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    table = build_cos_sin_cache(128, 4096, 1e6).to(DEV)
    w = torch.ones(128, device=DEV, dtype=torch.bfloat16)
    q_out = torch.empty(t, HQ * 128, device=DEV, dtype=torch.bfloat16)
    i = [0]

    def run():
        ops.qknorm_rope_store(qkv, pos, table, w, w, 1e-6, slots, kc[i[0]], vc[i[0]], HQ, HKV, q_out=q_out)
        i[0] ^= 1

    tm = timed(run, 8)
    nbytes = t * (HQ + 2 * HKV) * 128 * 2 * 2
    print(json.dumps({"kernel": "qknorm_rope_store", "tokens": t, "us": round(tm * 1e6, 1), "GBps": round(nbytes / tm / 1e9)}))


def cmd_sampler(args):
    v = 151936
    for b in (1, 29, 123, 256):
        logits = (torch.randn(b, v, device=DEV) * 2).to(torch.bfloat16)
        t = torch.full((b,), 0.6, device=DEV)
        k = torch.full((b,), v, dtype=torch.int64, device=DEV)
        p = torch.ones(b, device=DEV)
        mp = torch.zeros(b, device=DEV)
        out = torch.zeros(b, dtype=torch.int64, device=DEV)
        n = [0]

        def run():
            n[0] += 1
            ops.sample(logits, t, k, p, mp, 1, n[0], out=out)

        dt = timed(run, args.reps)
        res = {"kernel": "sample", "B": b, "plain_us": round(dt * 1e6, 1), "GBps": round(b * v * 2 / dt / 1e9, 1)}
        for name, kk, pp, mm in (("topk50_topp0.9", 50, 0.9, 0.0), ("topp0.9", v, 0.9, 0.0), ("topk50", 50, 1.0, 0.0), ("minp0.05", v, 1.0, 0.05)):
            k2 = torch.full((b,), kk, dtype=torch.int64, device=DEV)
            p2 = torch.full((b,), pp, device=DEV)
            m2 = torch.full((b,), mm, device=DEV)

            def run2():
                n[0] += 1
                ops.sample(logits, t, k2, p2, m2, 1, n[0], out=out)

            res[name + "_us"] = round(timed(run2, max(2, args.reps // 4)) * 1e6, 1)
        print(json.dumps(res))


def cmd_lmhead(args):
    """LM head + draw of a decode step (Qwen3-0.6B: 151 936 x 1024): the one-pass kernel against F.linear + the sampler.
    Three copies of the weight are used in turn so that no launch finds it in the 256 MB Infinity Cache."""
    import torch.nn.functional as F
    v, k = 151936, 1024
    ws = [(torch.randn(v, k, device=DEV) * k ** -0.5).to(torch.bfloat16) for _ in range(3)]
    packed = [ops.lmhead_pack(w) for w in ws]
    for b in (1, 8, 16, 29, 32, 48, 64, 96, 123, 128, 192, 256):
        x = torch.randn(b, k, device=DEV).to(torch.bfloat16)
        t = torch.full((b,), 0.6, device=DEV)
        full = torch.full((b,), v, dtype=torch.int64, device=DEV)
        ones, zeros = torch.ones(b, device=DEV), torch.zeros(b, device=DEV)
        out = torch.zeros(b, dtype=torch.int64, device=DEV)
        lws = ops.lmhead_workspace(b, DEV)
        sws = ops.sample_workspace(b, DEV)
        i = [0]

        def fused():
            i[0] += 1
            ops.lmhead_sample(packed[i[0] % 3], v, x, t, 0, v, 1, i[0], ws=lws, out=out)

        def two():
            i[0] += 1
            ops.sample(F.linear(x, ws[i[0] % 3]), t, full, ones, zeros, 1, i[0], out=out, ws=sws)

        def gemm():
            i[0] += 1
            F.linear(x, ws[i[0] % 3])

        tf, tt, tg = timed(fused, args.reps), timed(two, args.reps), timed(gemm, args.reps)
        print(json.dumps({"kernel": "lmhead_sample", "B": b, "fused_us": round(tf * 1e6, 1), "weight_GBps": round(v * k * 2 / tf / 1e9),
                          "linear_plus_sample_us": round(tt * 1e6, 1), "linear_alone_us": round(tg * 1e6, 1)}))


def cmd_stream(args):
    """Decode-step projections of Qwen3-8B (one GPU): the streamed projection + its consumer against torch (hipBLASLt)."""
    import torch.nn.functional as F
    shapes = [("qkv", 6144, 4096, "sum"), ("o", 4096, 4096, "norm"), ("gate_up+silu", 24576, 4096, "silu"), ("down", 4096, 12288, "norm")]
    for m in (1, 8, 16, 29, 48, 64):
        for name, n, k, consumer in shapes:
            # enough copies that no call finds its weight in the 256 MB memory-side cache (in the model the other layers'
            # weights pass through between two uses of one)
            copies = max(3, -(-(768 << 20) // (n * k * 2)))
            ws = [(torch.randn(n, k, device=DEV) * k ** -0.5).to(torch.bfloat16) for _ in range(copies)]
            packed = [ops.lmhead_pack(w) for w in ws]
            x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
            nw = torch.ones(n, device=DEV, dtype=torch.bfloat16)
            res = torch.randn(m, n, device=DEV).to(torch.bfloat16)
            part = torch.empty(k // 1024, m, n, dtype=torch.float32, device=DEV)
            i = [0]

            def ours():
                i[0] += 1
                ops.linear_stream(x, packed[i[0] % copies], n, out=part)
                if consumer == "sum":
                    ops.splitk_sum(part, torch.bfloat16)
                elif consumer == "silu":
                    ops.splitk_sum(part, torch.bfloat16, silu_mul=True)
                else:
                    ops.rmsnorm_splitk(part, nw, 1e-6, residual=res)

            def lib():
                i[0] += 1
                y = F.linear(x, ws[i[0] % copies])
                if consumer == "silu":
                    ops.silu_mul(y)
                elif consumer == "norm":
                    ops.rmsnorm(y, nw, 1e-6, res)

            t_ours, t_lib = timed(ours, args.reps), timed(lib, args.reps)
            print(json.dumps({"kernel": "linear_stream+consumer", "layer": name, "M": m, "N": n, "K": k, "us": round(t_ours * 1e6, 2),
                              "hipblaslt_path_us": round(t_lib * 1e6, 2), "weight_GBps": round(n * k * 2 / t_ours / 1e9)}))
            del ws, packed


def cmd_block(args):
    """Decode-step projections of Qwen3-8B at 17..128 rows: the blocked projection (+ its consumer) against torch (hipBLASLt)."""
    import torch.nn.functional as F
    shapes = [("qkv", 6144, 4096, 2, "sum"), ("o", 4096, 4096, 2, "norm"), ("gate_up+silu", 24576, 4096, 1, "none"), ("down", 4096, 12288, 2, "norm")]
    for m in ((int(os.environ["NVX_BENCH_M"]),) if os.environ.get("NVX_BENCH_M") else (24, 32, 48, 64, 96, 128)):
        for name, n, k, mode, consumer in shapes:
            copies = max(3, -(-(768 << 20) // (n * k * 2)))
            ws = [(torch.randn(n, k, device=DEV) * k ** -0.5).to(torch.bfloat16) for _ in range(copies)]
            packed = [ops.lmhead_pack(w) for w in ws]
            x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
            nw = torch.ones(n, device=DEV, dtype=torch.bfloat16)
            res = torch.randn(m, n, device=DEV).to(torch.bfloat16)
            s = ops.linear_block_slices(n, k, mode)
            out = torch.empty((s, m, n) if mode == 2 else (m, n // 2), dtype=torch.float32 if mode == 2 else torch.bfloat16, device=DEV)
            i = [0]

            def ours():
                i[0] += 1
                ops.linear_block(x, packed[i[0] % copies], n, mode=mode, k_slices=s, out=out)
                if consumer == "sum":
                    ops.splitk_sum(out, torch.bfloat16)
                elif consumer == "norm":
                    ops.rmsnorm_splitk(out, nw, 1e-6, residual=res)

            def kernel_only():
                i[0] += 1
                ops.linear_block(x, packed[i[0] % copies], n, mode=mode, k_slices=s, out=out)

            def lib():
                i[0] += 1
                y = F.linear(x, ws[i[0] % copies])
                if mode == 1:
                    ops.silu_mul(y)
                elif consumer == "norm":
                    ops.rmsnorm(y, nw, 1e-6, res)

            t_ours, t_k, t_lib = timed(ours, args.reps), timed(kernel_only, args.reps), timed(lib, args.reps)
            print(json.dumps({"kernel": "linear_block+consumer", "layer": name, "M": m, "N": n, "K": k, "slices": s, "us": round(t_ours * 1e6, 2),
                              "kernel_alone_us": round(t_k * 1e6, 2), "hipblaslt_path_us": round(t_lib * 1e6, 2),
                              "weight_GBps_kernel": round(n * k * 2 / t_k / 1e9)}), flush=True)
            del ws, packed


def cmd_linear(args):
    """Decode-step projections of Qwen3-0.6B: this library's skinny kernel vs torch (hipBLASLt)."""
    import torch.nn.functional as F
    shapes = [("qkv", 4096, 1024, 0), ("o", 1024, 2048, 2), ("gate_up+silu", 6144, 1024, 1), ("down", 1024, 3072, 2)]
    if os.environ.get("NVX_BENCH_SHAPES"):   # "name,n,k,mode;..." (mode 2 = K-split partials + the norm that sums them)
        shapes = [(a, int(b), int(c), int(d)) for a, b, c, d in (x.split(",") for x in os.environ["NVX_BENCH_SHAPES"].split(";"))]
    if args.tp > 1:     # one rank's share: column-parallel N / tp, row-parallel K / tp with a 16-bit result for the all-reduce
        tp = args.tp
        shapes = [("qkv", 4096 // tp, 1024, 0), ("o", 1024, 2048 // tp, 0), ("gate_up+silu", 6144 // tp, 1024, 1), ("down", 1024, 3072 // tp, 0)]
        shapes = [sh for sh in shapes if sh[2] in (256, 512, 768, 1024, 1536)]
    layers = 28
    for m in ((int(os.environ["NVX_BENCH_M"]),) if os.environ.get("NVX_BENCH_M") else (1, 8, 29, 64, 123, 256)):
        for name, n, k, mode in shapes:
            ws = [(torch.randn(n, k, device=DEV) * k ** -0.5).to(torch.bfloat16) for _ in range(layers)]
            x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
            s = ops.linear_decode_splits(n, k) if mode == 2 else 1
            shape = (s, m, n) if mode == 2 else (m, n // 2 if mode == 1 else n)
            out = torch.empty(shape, dtype=torch.float32 if mode == 2 else torch.bfloat16, device=DEV)
            i = [0]

            with_norm = mode == 2 and os.environ.get("NVX_BENCH_WITH_NORM") == "1"
            nw = torch.ones(n, device=DEV, dtype=torch.bfloat16)
            res = torch.randn(m, n, device=DEV).to(torch.bfloat16)

            def ours():
                ops.linear_decode(x, ws[i[0]], mode=mode, k_splits=s, out=out)
                if with_norm:
                    ops.rmsnorm_splitk(out, nw, 1e-6, residual=res)
                i[0] = (i[0] + 1) % layers

            def lib():
                y = F.linear(x, ws[i[0]])
                if mode == 1:
                    ops.silu_mul(y)
                if with_norm:
                    ops.rmsnorm(y, nw, 1e-6, res)
                i[0] = (i[0] + 1) % layers

            t_ours, t_lib = timed(ours, args.reps), timed(lib, args.reps)
            print(json.dumps({"kernel": "linear_decode", "layer": name, "M": m, "N": n, "K": k, "splits": s, "us": round(t_ours * 1e6, 2),
                              "hipblaslt_us": round(t_lib * 1e6, 2), "weight_GBps": round(n * k * 2 / t_ours / 1e9)}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["decode", "prefill", "sampler", "linear", "qknorm", "lmhead", "stream", "block"])
    ap.add_argument("--every", type=int, default=8)
    ap.add_argument("--tp", type=int, default=1, help="linear: the shapes of one rank of a tensor-parallel job")
    ap.add_argument("--reps", type=int, default=28)
    ap.add_argument("--num-seqs", type=int, default=256)
    ap.add_argument("--hkv1", action="store_true")
    ap.add_argument("--qkv", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="decode: q/k-norm + rotary + scatter kernel followed by attention on finished queries")
    ap.add_argument("--uniform-len", type=int, default=0)
    ap.add_argument("--heads", type=str, default="", help="query,kv head counts, e.g. 32,8 (default 16,8)")
    a = ap.parse_args()
    {"decode": cmd_decode, "prefill": cmd_prefill, "sampler": cmd_sampler, "qknorm": cmd_qknorm, "linear": cmd_linear, "lmhead": cmd_lmhead, "stream": cmd_stream, "block": cmd_block}[a.cmd](a)

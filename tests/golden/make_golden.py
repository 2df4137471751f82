#!/usr/bin/env python3
"""Freeze golden vectors from the reference implementation (build container only).

Run from the repo root:  python tests/golden/make_golden.py

The reference package under /root/reference is imported here -- and only here -- to produce
input/output vectors for the paged-attention + radix KV-cache path.  The resulting files
(tests/golden/*.json, *.npz) are data: inputs and the outputs the reference produced for
them.  No reference source travels with them.

How the reference is made to run on CPU:
  * `flashinfer` is not installed; its two wrapper class names are bound to a local stand-in
    whose `plan()` records the arguments and whose `run()` is this repo's own fp32 attention
    (so attention arithmetic is NOT pinned by these vectors -- everything around it is).
  * TORCH_COMPILE_DISABLE=1 (eager semantics), TRITON_INTERPRET=1 (the scatter kernel runs on
    CPU tensors).
  * `time.monotonic` inside the reference's kv_cache module is replaced by a strictly increasing
    counter so that LRU order is reproducible; one trace is also run with the real clock and
    must give identical results.
"""
import os
import sys

os.environ.setdefault("TORCH_COMPILE_DISABLE", "1")
os.environ.setdefault("TRITON_INTERPRET", "1")

import hashlib
import itertools
import json
import random
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# --------------------------------------------------------------------------------------
# stand-in for the absent third-party attention library
# --------------------------------------------------------------------------------------
class _RecordingWrapper:
    """Duck-type of the two paged wrappers: records plan() kwargs, run() = fp32 attention."""

    def __init__(self, *args, **kwargs):
        self.init_args = args
        self.init_kwargs = kwargs
        self.planned = None
        self._sm_scale = None

    def plan(self, **kw):
        self.planned = kw

    def run(self, q, kv):
        kw = self.planned
        k_cache, v_cache = kv
        kv_indptr = kw.get("paged_kv_indptr", kw.get("indptr"))
        kv_indices = kw.get("paged_kv_indices", kw.get("indices"))
        if "qo_indptr" in kw:
            qo_indptr = kw["qo_indptr"]
        else:
            qo_indptr = torch.arange(kv_indptr.numel(), dtype=torch.int32)
        hq = kw["num_qo_heads"]
        hkv = kw["num_kv_heads"]
        out = torch.empty_like(q)
        for b in range(kv_indptr.numel() - 1):
            q0, q1 = int(qo_indptr[b]), int(qo_indptr[b + 1])
            s0, s1 = int(kv_indptr[b]), int(kv_indptr[b + 1])
            idx = kv_indices[s0:s1].long()
            kk = k_cache[idx, 0].float()  # (Lkv, Hkv, D)
            vv = v_cache[idx, 0].float()
            qq = q[q0:q1].float()  # (Lq, Hq, D)
            lq, lkv = q1 - q0, s1 - s0
            g = hq // hkv
            kk = kk.repeat_interleave(g, dim=1)
            vv = vv.repeat_interleave(g, dim=1)
            s = torch.einsum("qhd,khd->hqk", qq, kk) * self._sm_scale
            qpos = torch.arange(lq).unsqueeze(1) + (lkv - lq)
            kpos = torch.arange(lkv).unsqueeze(0)
            s = s.masked_fill(kpos > qpos, float("-inf"))
            p = torch.softmax(s, dim=-1)
            o = torch.einsum("hqk,khd->qhd", p, vv)
            out[q0:q1] = o.to(q.dtype)
        return out


def _import_reference():
    fi = types.ModuleType("flashinfer")
    fi.BatchDecodeWithPagedKVCacheWrapper = _RecordingWrapper
    fi.BatchPrefillWithPagedKVCacheWrapper = _RecordingWrapper
    sys.modules["flashinfer"] = fi
    sys.path.insert(0, REF)
    import nanovllmx  # noqa: F401


class _Tick:
    """Strictly increasing stand-in for time.monotonic()."""

    def __init__(self):
        self.c = itertools.count(1)

    def monotonic(self):
        return float(next(self.c))


def _use_fake_clock():
    import nanovllmx.core.kv_cache as kvmod

    kvmod.time = _Tick()


def _use_real_clock():
    import time

    import nanovllmx.core.kv_cache as kvmod

    kvmod.time = time


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def dump_tree(tree):
    """Canonical nested dump: children ordered by first key token; access times -> dense ranks."""
    nodes = []

    def walk(n):
        nodes.append(n)
        for k in sorted(n.children):
            walk(n.children[k])

    walk(tree.root)
    order = {t: r for r, t in enumerate(sorted({n.access_time for n in nodes}))}

    def ser(n):
        return {
            "key": list(n.key),
            "value": list(n.value),
            "ref": n.ref_count,
            "rank": order[n.access_time],
            "children": [ser(n.children[k]) for k in sorted(n.children)],
        }

    return ser(tree.root)


def digest(obj) -> str:
    return hashlib.sha1(json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def save_json(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def save_npz(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


# --------------------------------------------------------------------------------------
# G1: allocator / radix tree op traces
# --------------------------------------------------------------------------------------
def gen_radix_program(seed, n_ops, vocab, pool):
    """A seeded op program; arguments only (results are filled in by running the reference)."""
    rng = random.Random(seed)
    ops = []
    for _ in range(n_ops):
        r = rng.random()
        if r < 0.45:
            ln = rng.randint(1, 14)
            ops.append({"op": "insert", "key": [rng.randrange(vocab) for _ in range(ln)]})
        elif r < 0.70:
            ln = rng.randint(1, 14)
            ops.append({"op": "match", "key": [rng.randrange(vocab) for _ in range(ln)]})
        elif r < 0.80:
            ops.append({"op": "pin"})  # inc_ref the node returned by the latest match/insert
        elif r < 0.90:
            ops.append({"op": "unpin", "which": rng.randrange(1 << 30)})  # dec_ref a pinned node
        else:
            ops.append({"op": "evict", "n": rng.randint(1, 20)})
    return {"seed": seed, "vocab": vocab, "pool": pool, "ops": ops}


def run_radix_program(prog):
    from nanovllmx.core.kv_cache import KVCacheAllocator, RadixTree

    alloc = KVCacheAllocator(prog["pool"])
    tree = RadixTree(alloc)
    pinned = []
    last_node = None
    results = []
    for op in prog["ops"]:
        kind = op["op"]
        res = {}
        if kind == "insert":
            key = list(op["key"])
            slots = alloc.alloc(len(key))
            if slots is None:
                tree.evict(len(key))
                slots = alloc.alloc(len(key))
            res["slots"] = slots
            if slots is not None:
                plen, node = tree.insert(key, slots)
                res["prefix_len"] = plen
                res["node_key"] = list(node.key)
                last_node = node
        elif kind == "match":
            idx, node = tree.match_prefix(list(op["key"]))
            res["indices"] = list(idx)
            res["node_key"] = list(node.key)
            last_node = node
        elif kind == "pin":
            if last_node is not None and last_node is not tree.root:
                tree.inc_ref(last_node)
                pinned.append(last_node)
                res["pinned"] = True
            else:
                res["pinned"] = False
        elif kind == "unpin":
            if pinned:
                node = pinned.pop(op["which"] % len(pinned))
                tree.dec_ref(node)
                res["unpinned_key"] = list(node.key)
            else:
                res["unpinned_key"] = None
        elif kind == "evict":
            tree.evict(op["n"])
        res["free"] = len(alloc.free_slots)
        res["free_tail"] = list(alloc.free_slots[-4:])
        res["tree"] = digest(dump_tree(tree))
        results.append(res)
    return results, dump_tree(tree), list(alloc.free_slots)


def make_g1():
    out = []
    for seed, n_ops, vocab, pool in [(1, 400, 3, 160), (2, 600, 4, 96), (3, 300, 2, 4000)]:
        prog = gen_radix_program(seed, n_ops, vocab, pool)
        _use_fake_clock()
        results, final_tree, free_list = run_radix_program(prog)
        _use_real_clock()
        results2, final_tree2, free_list2 = run_radix_program(prog)
        assert results == results2 and final_tree == final_tree2 and free_list == free_list2, \
            "reference trace depends on clock granularity"
        out.append({"program": prog, "results": results, "final_tree": final_tree,
                    "final_free": free_list})
    # hand-written corner cases
    from nanovllmx.core.kv_cache import KVCacheAllocator, RadixTree

    _use_fake_clock()
    corner = {}
    a = KVCacheAllocator(8)
    corner["alloc_order"] = [a.alloc(3), a.alloc(2)]
    a.free([1, 0])
    corner["after_free"] = [a.alloc(2), a.alloc(4), a.alloc(1)]
    t = RadixTree(KVCacheAllocator(8))
    try:
        t.match_prefix([])
        corner["match_empty"] = "ok"
    except IndexError:
        corner["match_empty"] = "IndexError"
    try:
        t.insert([], [])
        corner["insert_empty"] = "ok"
    except IndexError:
        corner["insert_empty"] = "IndexError"
    # key one longer than value (finished sequence shape) and empty value (aborted waiting seq)
    t.insert([5, 6, 7], [0, 1])
    t.insert([5, 6, 8, 9], [])
    corner["ragged_tree"] = dump_tree(t)
    corner["ragged_match"] = [list(t.match_prefix([5, 6, 8])[0]), list(t.match_prefix([5, 6, 7, 1])[0])]
    corner["ragged_tree_after_match"] = dump_tree(t)
    save_json("g1_radix.json", {"traces": out, "corner": corner})


# --------------------------------------------------------------------------------------
# G2: scheduler / KV manager traces
# --------------------------------------------------------------------------------------
def bench_workload(num_seqs=256, max_in=1024, max_out=1024, seed=0):
    """The request stream of the reference's bench.py (bench.py:13-31), same RNG call order."""
    rnd = random.Random()
    rnd.seed(seed)
    prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(100, max_in))] for _ in range(num_seqs)]
    max_tokens = [rnd.randint(100, max_out) for _ in range(num_seqs)]
    return prompts, max_tokens


def _engine_shell(kv_size, max_bs, context_len):
    from nanovllmx.core.engine import Engine
    from nanovllmx.core.scheduler import Scheduler

    eng = Engine.__new__(Engine)
    eng.context_len = context_len
    eng.scheduler = Scheduler(kv_cache_size=kv_size, max_bs=max_bs)
    eng.pp_queue = None
    return eng


def _step_record(batch, full):
    ids = [s.seq_id for s in batch.seqs]
    kv = [list(s.kv_indices) for s in batch.seqs]
    rec = {
        "mode": batch.forward_mode.name,
        "n": batch.num_seqs,
        "kv_sum": sum(len(x) for x in kv),
        "new_sum": sum(len(s.token_ids) - s.cached_kv_len for s in batch.seqs),
        "h": digest([ids, kv])[:16],
    }
    if full:
        rec["ids"] = ids
        rec["kv"] = kv
        rec["cached"] = [s.cached_kv_len for s in batch.seqs]
    return rec


def make_g2():
    import contextlib
    import io

    from nanovllmx.core.common import SamplingParams

    _use_fake_clock()
    # --- bench workload, fake model emits token 7 -------------------------------------
    prompts, max_tokens = bench_workload()
    eng = _engine_shell(400_000, 256, 4096)
    with contextlib.redirect_stdout(io.StringIO()):
        for i, (p, m) in enumerate(zip(prompts, max_tokens)):
            eng.add_sequence(f"r{i}", list(p), SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=m))
    steps = []
    generated = 0
    while eng.scheduler.has_unfinished_sequences():
        batch = eng.scheduler.schedule()
        assert batch is not None
        steps.append(_step_record(batch, full=False))
        outs = eng.update_from_output(batch, [7] * batch.num_seqs)
        generated += len(outs)
    mgr = eng.scheduler.kv_manager
    tree = dump_tree(mgr.radix_tree)

    def count_vals(n):
        return len(n["value"]) + sum(count_vals(c) for c in n["children"])

    in_tree = count_vals(tree)
    free = len(mgr.kv_cache_allocator.free_slots)
    bench = {
        "kv_size": 400_000, "max_bs": 256, "context_len": 4096, "token": 7,
        "sum_prompt": sum(len(p) for p in prompts), "sum_max_tokens": sum(max_tokens),
        "generated": generated, "steps": steps,
        "final_free": free, "final_in_tree": in_tree, "final_leaked": 400_000 - free - in_tree,
        "final_tree_digest": digest(tree),
        "final_free_digest": digest(list(mgr.kv_cache_allocator.free_slots)),
    }
    save_json("g2_sched_bench.json", bench)

    # --- small pool: eviction, eos, abort, explicit preempt() ---------------------------
    # NB: the reference cannot survive KV exhaustion inside schedule(): its in-loop preemption
    # pops the victim from `running` before calling preempt(), whose membership test then fails,
    # so nothing is freed and the loop spins forever.  The pool below is therefore sized so that
    # eviction of finished sequences always suffices; a watchdog aborts generation otherwise.
    import signal

    def _stuck(signum, frame):
        raise RuntimeError("reference scheduler did not return (KV exhaustion livelock)")

    signal.signal(signal.SIGALRM, _stuck)
    rng = random.Random(11)
    n_req = 48
    reqs = []
    for i in range(n_req):
        # small vocabulary so that finished sequences share prefixes in the tree
        plen = rng.randint(6, 40)
        base = [rng.randrange(3) for _ in range(plen)]
        kind = rng.randrange(3)
        sp = dict(temperature=1.0, ignore_eos=(kind != 0), eos_token_id=3)
        if kind == 1:
            sp["max_tokens"] = plen + rng.randint(1, 30)
        else:
            sp["max_new_tokens"] = rng.randint(1, 30)
        reqs.append({"id": f"s{i}", "prompt": base, "sp": sp, "arrive": rng.randint(0, 90)})
    aborts = {25: "s7", 40: "s20", 41: "s33"}  # step -> seq id to abort (if still live)
    preempts = {30: -1, 55: 0}  # step -> index into the running queue to preempt() explicitly
    kv_size, max_bs, context_len = 640, 6, 32
    eng = _engine_shell(kv_size=kv_size, max_bs=max_bs, context_len=context_len)

    def fake_token(seq_id, n_tok):
        return (int(seq_id[1:]) * 7 + n_tok * 3) % 5  # 3 == eos for some sequences

    steps = []
    step = 0
    pending = sorted(reqs, key=lambda r: r["arrive"])
    pi = 0
    min_free = kv_size
    with contextlib.redirect_stdout(io.StringIO()):
        while pi < len(pending) or eng.scheduler.has_unfinished_sequences():
            while pi < len(pending) and pending[pi]["arrive"] <= step:
                r = pending[pi]
                eng.add_sequence(r["id"], list(r["prompt"]), SamplingParams(**r["sp"]))
                pi += 1
            if step in aborts:
                eng.abort_sequence(aborts[step])
            if step in preempts and eng.scheduler.running:
                eng.scheduler.preempt(eng.scheduler.running[preempts[step]])
            signal.alarm(5)
            batch = eng.scheduler.schedule()
            signal.alarm(0)
            if batch is None:
                steps.append({"mode": "IDLE"})
            else:
                rec = _step_record(batch, full=True)
                toks = [fake_token(s.seq_id, len(s.token_ids)) for s in batch.seqs]
                outs = eng.update_from_output(batch, toks)
                rec["tokens"] = toks
                rec["finished"] = [[o.seq_id, o.finish_reason.name] for o in outs if o.is_finished]
                rec["waiting"] = [s.seq_id for s in eng.scheduler.waiting]
                rec["running"] = [s.seq_id for s in eng.scheduler.running]
                rec["free"] = len(eng.scheduler.kv_manager.kv_cache_allocator.free_slots)
                min_free = min(min_free, rec["free"])
                rec["tree"] = digest(dump_tree(eng.scheduler.kv_manager.radix_tree))
                steps.append(rec)
            step += 1
            assert step < 5000
    mgr = eng.scheduler.kv_manager
    small = {
        "kv_size": kv_size, "max_bs": max_bs, "context_len": context_len, "requests": reqs,
        "aborts": {str(k): v for k, v in aborts.items()},
        "preempts": {str(k): v for k, v in preempts.items()},
        "steps": steps, "final_tree": dump_tree(mgr.radix_tree),
        "final_free": list(mgr.kv_cache_allocator.free_slots),
    }
    n_pre = sum(1 for s in steps if s.get("mode") == "PREFILL")
    print("small trace:", len(steps), "steps,", n_pre, "prefill steps, min free", min_free)
    save_json("g2_sched_small.json", small)


# --------------------------------------------------------------------------------------
# G3: block-table metadata
# --------------------------------------------------------------------------------------
def make_g3():
    from nanovllmx.core.common import ForwardBatch, ForwardMode, Sequence
    from nanovllmx.core.cuda_graph import CUDAGraph
    from nanovllmx.core.kv_cache import KVCachePool
    from nanovllmx.layers.attention import AttentionBackend

    pool = KVCachePool(torch.bfloat16, torch.device("cpu"), 64, 1, 2, 8)
    be = AttentionBackend.__new__(AttentionBackend)
    be.kv_cache = pool
    be.num_qo_heads, be.num_kv_heads, be.head_dim = 4, 2, 8
    be.dtype, be.device = torch.bfloat16, torch.device("cpu")
    be.workspace_buffer = None
    be.prefill_wrapper = _RecordingWrapper()
    be.decode_wrapper = _RecordingWrapper()
    be.decode_wrappers_cuda_graph = {}

    def seq(i, ntok, kv, cached):
        return Sequence(seq_id=f"q{i}", token_ids=list(range(100 * i, 100 * i + ntok)),
                        num_tokens=ntok, prompt_len=ntok, kv_indices=list(kv), cached_kv_len=cached)

    def tolist(t):
        return [int(x) for x in t.tolist()]

    cases = []
    layouts = {
        "prefill_fresh": (ForwardMode.PREFILL, [(5, [0, 1, 2, 3, 4], 0), (3, [9, 8, 7], 0), (1, [30], 0)]),
        "prefill_with_cached_prefix": (ForwardMode.PREFILL, [(6, [10, 11, 12, 40, 41, 42], 3), (4, [20, 21, 22, 50], 2)]),
        "decode": (ForwardMode.DECODE, [(6, [0, 1, 2, 3, 4, 60], 5), (4, [9, 8, 7, 61], 3), (2, [30, 62], 1)]),
        "decode_single": (ForwardMode.DECODE, [(3, [5, 6, 7], 2)]),
    }
    for name, (mode, rows) in layouts.items():
        seqs = [seq(i, *r) for i, r in enumerate(rows)]
        md = be.build_metadata(ForwardBatch(forward_mode=mode, num_seqs=len(seqs), seqs=seqs))
        kw = md.wrapper.planned
        rec = {"name": name, "mode": mode.name,
               "seqs": [{"ntok": r[0], "kv": r[1], "cached": r[2]} for r in rows],
               "output_kv_indices": tolist(md.output_kv_indices),
               "output_kv_indices_dtype": str(md.output_kv_indices.dtype),
               "plan": {k: (tolist(v) if isinstance(v, torch.Tensor) else str(v) if isinstance(v, torch.dtype) else v)
                        for k, v in kw.items()},
               "plan_dtypes": {k: str(v.dtype) for k, v in kw.items() if isinstance(v, torch.Tensor)}}
        cases.append(rec)

    # graph-replay variant (static buffers, padded rows)
    g = CUDAGraph()
    be.prepare_for_cuda_graph_capture(g, max_bs=8, context_len=16)
    cap = {}
    for bs in (8, 4, 2, 1):
        md = be.build_metadata_for_cuda_graph_capture(g, bs, 16)
        kw = md.wrapper.planned
        cap[str(bs)] = {"indptr": tolist(kw["indptr"]), "indices_len": int(kw["indices"].numel()),
                        "last_page_len": tolist(kw["last_page_len"]),
                        "output_kv_indices_len": int(md.output_kv_indices.numel())}
    rows = [(6, [0, 1, 2, 3, 4, 60], 5), (4, [9, 8, 7, 61], 3), (2, [30, 62], 1)]
    seqs = [seq(i, *r) for i, r in enumerate(rows)]
    md = be.build_metadata_for_cuda_graph_replay(
        g, ForwardBatch(forward_mode=ForwardMode.DECODE, num_seqs=3, seqs=seqs), padded_bs=4)
    kw = md.wrapper.planned
    replay = {
        "max_bs": 8, "context_len": 16, "padded_bs": 4,
        "seqs": [{"ntok": r[0], "kv": r[1], "cached": r[2]} for r in rows],
        "indptr": tolist(kw["indptr"]), "indices": tolist(kw["indices"]),
        "last_page_len": tolist(kw["last_page_len"]),
        "output_kv_indices": tolist(md.output_kv_indices),
        "buf_indptr": tolist(g.get_input_buffer("paged_kv_indptr_buffer")),
        "buf_indices_head": tolist(g.get_input_buffer("paged_kv_indices_buffer")[:20]),
        "buf_output_kv_indices": tolist(g.get_input_buffer("output_kv_indices_buffer")),
    }
    save_json("g3_metadata.json", {"cases": cases, "capture": cap, "replay": replay})


# --------------------------------------------------------------------------------------
# G4: KV scatter
# --------------------------------------------------------------------------------------
def make_g4():
    from nanovllmx.core.kv_cache import KVCachePool

    g = torch.Generator().manual_seed(4)
    hq, hkv, d, t, slots, layers = 4, 2, 128, 7, 24, 2
    qkv = torch.randn(t, (hq + 2 * hkv) * d, generator=g).to(torch.bfloat16)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)  # strided views like the model's
    pool = KVCachePool(torch.bfloat16, torch.device("cpu"), slots, layers, hkv, d, start_layer=3)
    pool.k_cache.copy_(torch.randn(pool.k_cache.shape, generator=g).to(torch.bfloat16))
    pool.v_cache.copy_(torch.randn(pool.v_cache.shape, generator=g).to(torch.bfloat16))
    k0, v0 = pool.k_cache.clone(), pool.v_cache.clone()
    idx = torch.tensor([5, -1, 23, 0, 7, -1, 11], dtype=torch.long)
    pool.set_kv_cache(4, idx, k.view(-1, hkv, d), v.view(-1, hkv, d))
    kk, vv = pool.get_kv_cache(4)
    assert torch.equal(kk[5], k.view(-1, hkv, d)[0])
    save_npz("g4_store_kv.npz", qkv=bf16_bits(qkv), idx=idx.numpy(),
             k_before=bf16_bits(k0), v_before=bf16_bits(v0),
             k_after=bf16_bits(pool.k_cache), v_after=bf16_bits(pool.v_cache),
             meta=np.array([hq, hkv, d, t, slots, layers, 3, 4], dtype=np.int64))


# --------------------------------------------------------------------------------------
# G5: RMSNorm / RoPE / SiluAndMul
# --------------------------------------------------------------------------------------
def make_g5():
    from nanovllmx.layers.activation import SiluAndMul
    from nanovllmx.layers.layernorm import RMSNorm
    from nanovllmx.layers.rotary_embedding import RotaryEmbedding

    g = torch.Generator().manual_seed(5)
    out = {}
    # hidden-size RMSNorm, bf16 weight (model dtype), with and without residual
    for tag, rows, h in (("h1024", 6, 1024), ("h4096", 3, 4096), ("hd128", 6 * 16, 128)):
        norm = RMSNorm(h, eps=1e-6)
        w = (torch.randn(h, generator=g) * 0.2 + 1.0).to(torch.bfloat16)
        norm.weight.data = w
        x = (torch.randn(rows, h, generator=g) * 3.0).to(torch.bfloat16)
        r = (torch.randn(rows, h, generator=g) * 3.0).to(torch.bfloat16)
        y = norm(x.clone())
        y2, r2 = norm(x.clone(), r.clone())
        out[f"rms_{tag}_w"] = bf16_bits(w)
        out[f"rms_{tag}_x"] = bf16_bits(x)
        out[f"rms_{tag}_res"] = bf16_bits(r)
        out[f"rms_{tag}_y"] = bf16_bits(y)
        out[f"rms_{tag}_y_res"] = bf16_bits(y2)
        out[f"rms_{tag}_res_out"] = bf16_bits(r2)
    # RoPE: Qwen3 geometry, theta 1e6
    rope = RotaryEmbedding(128, rotary_dim=128, max_position_embeddings=40960, base=1000000)
    pos = torch.tensor([0, 1, 2, 17, 1023, 2047, 4095, 40959], dtype=torch.long)
    q = torch.randn(pos.numel(), 16 * 128, generator=g).to(torch.bfloat16)
    k = torch.randn(pos.numel(), 8 * 128, generator=g).to(torch.bfloat16)
    q2, k2 = rope(pos, q.clone(), k.clone())
    out["rope_pos"] = pos.numpy()
    out["rope_q"] = bf16_bits(q)
    out["rope_k"] = bf16_bits(k)
    out["rope_q_out"] = bf16_bits(q2)
    out["rope_k_out"] = bf16_bits(k2)
    out["rope_cache_rows"] = rope.cos_sin_cache[pos].numpy()  # fp32 (8, 128) = [cos | sin]
    # SiluAndMul
    x = (torch.randn(5, 2 * 384, generator=g) * 2.0).to(torch.bfloat16)
    out["silu_x"] = bf16_bits(x)
    out["silu_y"] = bf16_bits(SiluAndMul()(x))
    save_npz("g5_layers.npz", **out)


# --------------------------------------------------------------------------------------
# G6: sampler masks
# --------------------------------------------------------------------------------------
def make_g6():
    from nanovllmx.layers.sampler import Sampler, _apply_min_p, _apply_top_k_top_p

    g = torch.Generator().manual_seed(6)
    b, v = 6, 2048
    logits = (torch.randn(b, v, generator=g) * 2.5).to(torch.bfloat16)
    logits[4, 100:110] = logits[4, 100]  # ties around a top-k boundary
    temps = torch.tensor([0.6, 1.0, 0.8, 1.3, 0.7, 1.0])
    top_k = torch.tensor([v, 50, 5, v, 7, 1], dtype=torch.long)
    top_p = torch.tensor([1.0, 0.9, 0.5, 0.95, 1.0, 1.0])
    min_p = torch.tensor([0.0, 0.05, 0.0, 0.1, 0.0, 0.0])
    x = logits.to(torch.float)
    x.div_(temps.unsqueeze(1))
    after_kp = _apply_top_k_top_p(x.clone(), top_p.clone(), top_k.clone())
    after_minp = _apply_min_p(after_kp.clone(), min_p.clone())
    probs = torch.softmax(after_minp, dim=-1, dtype=torch.float)
    # the sampler itself: with top_k = 1 the draw is deterministic
    torch.manual_seed(0)
    ids = Sampler()(logits.clone(), temps.clone(), min_p.clone(), top_p.clone(), top_k.clone())
    save_npz("g6_sampler.npz", logits=bf16_bits(logits), temps=temps.numpy(), top_k=top_k.numpy(),
             top_p=top_p.numpy(), min_p=min_p.numpy(),
             keep_after_kp=torch.isfinite(after_kp).numpy(), keep_after_minp=torch.isfinite(after_minp).numpy(),
             probs=probs.numpy(), greedy_row5=np.array([int(ids[5])]),
             argmax=torch.argmax(logits.float(), dim=-1).numpy())


# --------------------------------------------------------------------------------------
# G7: tiny Qwen3 logits (model wiring around the attention op)
# --------------------------------------------------------------------------------------
TINY = dict(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
            num_attention_heads=2, num_key_value_heads=1, head_dim=128, rms_norm_eps=1e-6,
            max_position_embeddings=512, rope_theta=1000000.0, tie_word_embeddings=False)


def formula_weight(shape, salt: int, scale: float) -> torch.Tensor:
    """Deterministic pseudo-random weights that do not depend on any RNG implementation."""
    n = 1
    for s in shape:
        n *= s
    i = torch.arange(n, dtype=torch.float64)
    w = torch.sin(i * 0.61803398875 + salt * 1.37) * torch.cos(i * 0.0137 + salt)
    return (w * scale).reshape(shape).to(torch.bfloat16)


def tiny_state_dict():
    c = TINY
    h, d, hq, hkv, it = c["hidden_size"], c["head_dim"], c["num_attention_heads"], c["num_key_value_heads"], c["intermediate_size"]
    sd = {"model.embed_tokens.weight": formula_weight((c["vocab_size"], h), 1, 1.0),
          "model.norm.weight": formula_weight((h,), 2, 0.2) + 1.0,
          "lm_head.weight": formula_weight((c["vocab_size"], h), 3, 0.3)}
    for l in range(c["num_hidden_layers"]):
        p = f"model.layers.{l}."
        s = 10 * (l + 1)
        sd[p + "self_attn.q_proj.weight"] = formula_weight((hq * d, h), s + 1, 0.15)
        sd[p + "self_attn.k_proj.weight"] = formula_weight((hkv * d, h), s + 2, 0.15)
        sd[p + "self_attn.v_proj.weight"] = formula_weight((hkv * d, h), s + 3, 0.15)
        sd[p + "self_attn.o_proj.weight"] = formula_weight((h, hq * d), s + 4, 0.1)
        sd[p + "self_attn.q_norm.weight"] = formula_weight((d,), s + 5, 0.2) + 1.0
        sd[p + "self_attn.k_norm.weight"] = formula_weight((d,), s + 6, 0.2) + 1.0
        sd[p + "mlp.gate_proj.weight"] = formula_weight((it, h), s + 7, 0.15)
        sd[p + "mlp.up_proj.weight"] = formula_weight((it, h), s + 8, 0.15)
        sd[p + "mlp.down_proj.weight"] = formula_weight((h, it), s + 9, 0.1)
        sd[p + "input_layernorm.weight"] = formula_weight((h,), s + 10, 0.2) + 1.0
        sd[p + "post_attention_layernorm.weight"] = formula_weight((h,), s + 11, 0.2) + 1.0
    return {k: v.to(torch.bfloat16) for k, v in sd.items()}


def make_g7():
    import tempfile

    from transformers.models.qwen3 import Qwen3Config

    from nanovllmx.core.common import ForwardBatch, ForwardMode, Sequence
    from nanovllmx.core.kv_cache import KVCachePool
    from nanovllmx.distributed.parallel_state import init_distributed_environment, initialize_model_parallel
    from nanovllmx.layers.attention import AttentionBackend, attention_kv_cache
    from nanovllmx.models.qwen3 import Qwen3ForCausalLM

    with tempfile.NamedTemporaryFile() as f:
        init_distributed_environment(1, 0, "gloo", init_method=f"file://{f.name}_rdv")
    initialize_model_parallel(1, 1, 0, 0)
    cfg = Qwen3Config(**TINY)
    torch.set_default_dtype(torch.bfloat16)
    model = Qwen3ForCausalLM(cfg)
    torch.set_default_dtype(torch.float32)
    model.load_weights(tiny_state_dict().items())
    model.eval()

    pool = KVCachePool(torch.bfloat16, torch.device("cpu"), 64, TINY["num_hidden_layers"], 1, 128)
    pool.k_cache.zero_()
    pool.v_cache.zero_()
    be = AttentionBackend.__new__(AttentionBackend)
    be.kv_cache = pool
    be.num_qo_heads, be.num_kv_heads, be.head_dim = 2, 1, 128
    be.dtype, be.device = torch.bfloat16, torch.device("cpu")
    be.prefill_wrapper = _RecordingWrapper()
    be.decode_wrapper = _RecordingWrapper()

    prompts = [[3, 17, 200, 45, 9], [250, 1, 1, 8, 77, 31, 5, 120, 64], [99, 100, 101]]
    slots = [[10, 11, 12, 13, 14], [20, 21, 22, 23, 24, 25, 26, 27, 28], [40, 41, 42]]
    seqs = [Sequence(seq_id=f"t{i}", token_ids=list(p), num_tokens=len(p), prompt_len=len(p), kv_indices=list(s))
            for i, (p, s) in enumerate(zip(prompts, slots))]
    next_slot = [15, 29, 43]
    feed = [[11, 222, 33], [4, 5, 6]]  # tokens appended after each step (not sampled: deterministic)
    logits_steps, hidden_steps = [], []
    with torch.inference_mode():
        for step in range(3):
            mode = ForwardMode.PREFILL if step == 0 else ForwardMode.DECODE
            batch = ForwardBatch(forward_mode=mode, num_seqs=3, seqs=seqs)
            ids, pos = [], []
            for s in seqs:
                ids += s.token_ids[s.cached_kv_len:]
                pos += list(range(s.cached_kv_len, len(s.token_ids)))
            md = be.build_metadata(batch)
            with attention_kv_cache(model, md):
                hs = model(torch.tensor(ids), torch.tensor(pos))
            last, cu = [], 0
            for s in seqs:
                cu += len(s.token_ids) - s.cached_kv_len
                last.append(cu - 1)
            hl = hs[last]
            logits = model.compute_logits(hl)
            logits_steps.append(logits.float().numpy())
            hidden_steps.append(bf16_bits(hl))
            if step < 2:
                for i, s in enumerate(seqs):
                    s.cached_kv_len = len(s.kv_indices)
                    s.token_ids.append(feed[step][i])
                    s.kv_indices.append(next_slot[i] + step)
    save_npz("g7_tiny_qwen3.npz",
             logits0=logits_steps[0], logits1=logits_steps[1], logits2=logits_steps[2],
             hidden0=hidden_steps[0], hidden1=hidden_steps[1], hidden2=hidden_steps[2],
             k_cache=bf16_bits(pool.k_cache), v_cache=bf16_bits(pool.v_cache))
    save_json("g7_tiny_qwen3.json", {"config": TINY, "prompts": prompts, "slots": slots,
                                      "next_slot": next_slot, "feed": feed, "pool_slots": 64})


def main():
    _import_reference()
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7"]
    for w in which:
        globals()["make_" + w]()


if __name__ == "__main__":
    main()

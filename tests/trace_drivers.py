"""Replay the golden op traces against an implementation through a small adapter.

Two adapters exist: one for the oracle (oracle/*.py) and one for the product classes
(nano_vllm_x_amd.core.*), so both are held to the same frozen reference outputs.
"""
from helpers import digest


# ---------------------------------------------------------------------------------------
# adapters
# ---------------------------------------------------------------------------------------
class OracleRadixAdapter:
    def __init__(self, pool):
        from oracle.kv_index_oracle import RadixOracle, SlotStackOracle

        self.a = SlotStackOracle(pool)
        self.t = RadixOracle(self.a)
        self.root = 0

    def alloc(self, n):
        return self.a.alloc(n)

    def free_list(self):
        return self.a.stack

    def insert(self, key, slots):
        return self.t.insert(key, slots)

    def match(self, key):
        return self.t.match_prefix(key)

    def node_key(self, h):
        return list(self.t.key[h])

    def is_root(self, h):
        return h == 0

    def inc_ref(self, h):
        self.t.inc_ref(h)

    def dec_ref(self, h):
        self.t.dec_ref(h)

    def evict(self, n):
        self.t.evict(n)

    def dump(self):
        return self.t.dump()


def dump_node_tree(root):
    """Dump for reference-shaped node objects (.key .value .children .ref_count .access_time)."""
    nodes = []

    def walk(n):
        nodes.append(n)
        for k in sorted(n.children):
            walk(n.children[k])

    walk(root)
    order = {t: r for r, t in enumerate(sorted({n.access_time for n in nodes}))}

    def ser(n):
        return {"key": list(n.key), "value": list(n.value), "ref": n.ref_count,
                "rank": order[n.access_time],
                "children": [ser(n.children[k]) for k in sorted(n.children)]}

    return ser(root)


class ProductRadixAdapter:
    def __init__(self, pool):
        from nano_vllm_x_amd.core.kv_cache import KVCacheAllocator, RadixTree

        self.a = KVCacheAllocator(pool)
        self.t = RadixTree(self.a)

    def alloc(self, n):
        return self.a.alloc(n)

    def free_list(self):
        return self.a.free_slots

    def insert(self, key, slots):
        return self.t.insert(key, slots)

    def match(self, key):
        return self.t.match_prefix(key)

    def node_key(self, h):
        return list(h.key)

    def is_root(self, h):
        return h is self.t.root

    def inc_ref(self, h):
        self.t.inc_ref(h)

    def dec_ref(self, h):
        self.t.dec_ref(h)

    def evict(self, n):
        self.t.evict(n)

    def dump(self):
        return dump_node_tree(self.t.root)


# ---------------------------------------------------------------------------------------
# G1 replay
# ---------------------------------------------------------------------------------------
def replay_radix_program(adapter_cls, trace):
    prog, want = trace["program"], trace["results"]
    ad = adapter_cls(prog["pool"])
    pinned = []
    last = None
    for i, (op, exp) in enumerate(zip(prog["ops"], want)):
        kind = op["op"]
        got = {}
        if kind == "insert":
            key = list(op["key"])
            slots = ad.alloc(len(key))
            if slots is None:
                ad.evict(len(key))
                slots = ad.alloc(len(key))
            got["slots"] = slots
            if slots is not None:
                plen, node = ad.insert(key, slots)
                got["prefix_len"] = plen
                got["node_key"] = ad.node_key(node)
                last = node
        elif kind == "match":
            idx, node = ad.match(list(op["key"]))
            got["indices"] = list(idx)
            got["node_key"] = ad.node_key(node)
            last = node
        elif kind == "pin":
            if last is not None and not ad.is_root(last):
                ad.inc_ref(last)
                pinned.append(last)
                got["pinned"] = True
            else:
                got["pinned"] = False
        elif kind == "unpin":
            if pinned:
                node = pinned.pop(op["which"] % len(pinned))
                ad.dec_ref(node)
                got["unpinned_key"] = ad.node_key(node)
            else:
                got["unpinned_key"] = None
        elif kind == "evict":
            ad.evict(op["n"])
        fl = ad.free_list()
        got["free"] = len(fl)
        got["free_tail"] = list(fl[-4:])
        got["tree"] = digest(ad.dump())
        assert got == exp, f"op {i} {op}: got {got}, want {exp}"
    assert ad.dump() == trace["final_tree"]
    assert list(ad.free_list()) == trace["final_free"]


# ---------------------------------------------------------------------------------------
# G2 replay (scheduler / KV manager)
# ---------------------------------------------------------------------------------------
class OracleEngineAdapter:
    def __init__(self, kv_size, max_bs, context_len):
        from oracle.sched_oracle import EngineOracle

        self.e = EngineOracle(kv_size, max_bs, context_len)

    def add(self, seq_id, prompt, sp: dict):
        from oracle.sched_oracle import SamplingOracle

        self.e.add(seq_id, prompt, SamplingOracle(**sp))

    def abort(self, seq_id):
        self.e.abort(seq_id)

    def preempt_running(self, index):
        if self.e.sched.running:
            self.e.sched.preempt(self.e.sched.running[index])

    def busy(self):
        return self.e.sched.busy()

    def schedule(self):
        b = self.e.sched.schedule()
        if b is None:
            return None
        return b.mode, b.seqs, b

    def absorb(self, handle, tokens):
        return [(o[0], o[2], o[3]) for o in self.e.absorb(handle, tokens)]

    def queues(self):
        return [s.seq_id for s in self.e.sched.waiting], [s.seq_id for s in self.e.sched.running]

    def free_list(self):
        return self.e.sched.kv.slots.stack

    def dump(self):
        return self.e.sched.kv.tree.dump()


class ProductEngineAdapter:
    def __init__(self, kv_size, max_bs, context_len):
        from nano_vllm_x_amd.core.engine import EngineCore

        self.e = EngineCore(kv_cache_size=kv_size, max_bs=max_bs, context_len=context_len)

    def add(self, seq_id, prompt, sp: dict):
        from nano_vllm_x_amd.core.common import SamplingParams

        self.e.add_sequence(seq_id, list(prompt), SamplingParams(**sp))

    def abort(self, seq_id):
        self.e.abort_sequence(seq_id)

    def preempt_running(self, index):
        s = self.e.scheduler
        if s.running:
            s.preempt(s.running[index])

    def busy(self):
        return self.e.has_unfinished_sequences()

    def schedule(self):
        b = self.e.scheduler.schedule()
        if b is None:
            return None
        return b.forward_mode.name, b.seqs, b

    def absorb(self, handle, tokens):
        outs = self.e.update_from_output(handle, tokens)
        return [(o.seq_id, o.is_finished, o.finish_reason.name if o.finish_reason else None) for o in outs]

    def queues(self):
        s = self.e.scheduler
        return [x.seq_id for x in s.waiting], [x.seq_id for x in s.running]

    def free_list(self):
        return self.e.scheduler.kv_manager.kv_cache_allocator.free_slots

    def dump(self):
        return dump_node_tree(self.e.scheduler.kv_manager.radix_tree.root)


def _check_step(rec, mode, seqs, full):
    ids = [s.seq_id for s in seqs]
    kv = [list(s.kv_indices) for s in seqs]
    assert rec["mode"] == mode
    assert rec["n"] == len(seqs)
    assert rec["kv_sum"] == sum(len(x) for x in kv)
    assert rec["new_sum"] == sum(len(s.token_ids) - s.cached_kv_len for s in seqs)
    assert rec["h"] == digest([ids, kv])[:16]
    if full:
        assert rec["ids"] == ids and rec["kv"] == kv
        assert rec["cached"] == [s.cached_kv_len for s in seqs]


def replay_bench_trace(adapter_cls, g, prompts, max_tokens):
    ad = adapter_cls(g["kv_size"], g["max_bs"], g["context_len"])
    for i, (p, m) in enumerate(zip(prompts, max_tokens)):
        ad.add(f"r{i}", p, dict(temperature=0.6, ignore_eos=True, max_tokens=m))
    generated = 0
    for n, rec in enumerate(g["steps"]):
        assert ad.busy()
        mode, seqs, h = ad.schedule()
        _check_step(rec, mode, seqs, full=False)
        generated += len(ad.absorb(h, [g["token"]] * len(seqs)))
    assert not ad.busy()
    assert generated == g["generated"]
    tree = ad.dump()

    def count(n):
        return len(n["value"]) + sum(count(c) for c in n["children"])

    assert len(ad.free_list()) == g["final_free"]
    assert count(tree) == g["final_in_tree"]
    assert digest(tree) == g["final_tree_digest"]
    assert digest(list(ad.free_list())) == g["final_free_digest"]


def replay_small_trace(adapter_cls, g):
    ad = adapter_cls(g["kv_size"], g["max_bs"], g["context_len"])
    pending = sorted(g["requests"], key=lambda r: r["arrive"])
    aborts = {int(k): v for k, v in g["aborts"].items()}
    preempts = {int(k): v for k, v in g["preempts"].items()}

    def fake_token(seq_id, n_tok):
        return (int(seq_id[1:]) * 7 + n_tok * 3) % 5

    pi = 0
    for step, rec in enumerate(g["steps"]):
        while pi < len(pending) and pending[pi]["arrive"] <= step:
            r = pending[pi]
            ad.add(r["id"], r["prompt"], r["sp"])
            pi += 1
        if step in aborts:
            ad.abort(aborts[step])
        if step in preempts:
            ad.preempt_running(preempts[step])
        got = ad.schedule()
        if rec["mode"] == "IDLE":
            assert got is None, f"step {step}"
            continue
        assert got is not None, f"step {step}"
        mode, seqs, h = got
        _check_step(rec, mode, seqs, full=True)
        toks = [fake_token(s.seq_id, len(s.token_ids)) for s in seqs]
        assert toks == rec["tokens"]
        outs = ad.absorb(h, toks)
        assert [[o[0], o[2]] for o in outs if o[1]] == rec["finished"], f"step {step}"
        w, r = ad.queues()
        assert w == rec["waiting"] and r == rec["running"], f"step {step}"
        assert len(ad.free_list()) == rec["free"], f"step {step}"
        assert digest(ad.dump()) == rec["tree"], f"step {step}"
    assert pi == len(pending) and not ad.busy()
    assert ad.dump() == g["final_tree"]
    assert list(ad.free_list()) == g["final_free"]

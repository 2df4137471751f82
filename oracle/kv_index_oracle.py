"""Oracle: slot allocator, radix prefix tree and KV manager (integer, bit-exact).

TEST INFRASTRUCTURE.  Restates nanovllmx/core/kv_cache.py:101-369 with a flat node table
and a logical clock.  Pinned by tests/golden/g1_radix.json and g2_sched_*.json, which were
produced by running the reference itself.

Where the reference stamps `time.monotonic()` this file ticks an integer counter at the
same program points (node construction kv_cache.py:135, touch during descent :166/:193),
so every comparison between two stamps has the same outcome.
"""
from __future__ import annotations

import heapq


class SlotStackOracle:
    """kv_cache.py:101-117 -- LIFO stack of token slots; first pops give 0,1,2,..."""

    def __init__(self, size: int):
        self.size = size
        self.stack = list(range(size - 1, -1, -1))

    def alloc(self, n: int):
        if n > len(self.stack):
            return None
        got = []
        for _ in range(n):
            got.append(self.stack.pop())
        return got

    def free(self, slots):
        self.stack.extend(slots)


ROOT = 0


class RadixOracle:
    """kv_cache.py:119-310.  Nodes live in parallel lists indexed by an integer id."""

    def __init__(self, slots: SlotStackOracle):
        self.slots = slots
        self.clock = 0
        self.key: list[tuple] = []
        self.val: list[tuple] = []
        self.kids: list[dict] = []
        self.up: list[int] = []
        self.ref: list[int] = []
        self.stamp: list[int] = []
        self._new(parent=-1, key=(), val=(), ref=1, kids=None)  # root, kv_cache.py:156

    # -- plumbing ----------------------------------------------------------------------
    def _tick(self) -> int:
        self.clock += 1
        return self.clock

    def _new(self, parent, key, val, ref, kids):
        nid = len(self.key)
        self.key.append(tuple(key))
        self.val.append(tuple(val))
        self.kids.append(dict(kids) if kids else {})
        self.up.append(parent)
        self.ref.append(ref)
        self.stamp.append(self._tick())
        return nid

    @staticmethod
    def _common(a, b) -> int:
        n = min(len(a), len(b))
        i = 0
        while i < n and a[i] == b[i]:
            i += 1
        return i

    def _split(self, nid: int, at: int) -> int:
        """kv_cache.py:293-310: parent -> nid becomes parent -> top -> nid; returns top."""
        k, v = self.key[nid], self.val[nid]
        top = self._new(parent=self.up[nid], key=k[:at], val=v[:at], ref=self.ref[nid], kids={k[at]: nid})
        self.up[nid] = top
        self.key[nid] = k[at:]
        self.val[nid] = v[at:]
        if self.up[top] >= 0:
            self.kids[self.up[top]][self.key[top][0]] = top
        return top

    # -- public ------------------------------------------------------------------------
    def match_prefix(self, key):
        """kv_cache.py:159-184.  NB: splits a node when the match ends inside it."""
        key = list(key)
        head = key[0]  # IndexError on an empty key, as in the reference
        cur = ROOT
        found: list[int] = []
        while key and head in self.kids[cur]:
            child = self.kids[cur][head]
            self.stamp[child] = self._tick()
            m = self._common(self.key[child], key)
            if m < len(self.key[child]):
                cur = self._split(child, m)
                found.extend(self.val[cur])
                break
            found.extend(self.val[child])
            cur = child
            key = key[m:]
            head = key[0] if key else -1
        return found, cur

    def insert(self, key, value):
        """kv_cache.py:186-220 -> (matched_len, last_node)."""
        key = list(key)
        value = list(value)
        head = key[0]
        cur = ROOT
        matched = 0
        while key and head in self.kids[cur]:
            child = self.kids[cur][head]
            self.stamp[child] = self._tick()
            m = self._common(self.key[child], key)
            matched += m
            full = m >= len(self.key[child])
            key = key[m:]
            value = value[m:]
            head = key[0] if key else -1
            if not full:
                cur = self._split(child, m)
                break
            cur = child
        last = cur
        if key:
            last = self._new(parent=cur, key=key, val=value, ref=0, kids=None)
            self.kids[cur][key[0]] = last
        return matched, last

    def inc_ref(self, nid: int):
        while nid != ROOT:
            self.ref[nid] += 1
            nid = self.up[nid] if self.up[nid] >= 0 else ROOT

    def dec_ref(self, nid: int):
        while nid != ROOT:
            self.ref[nid] -= 1
            nid = self.up[nid] if self.up[nid] >= 0 else ROOT

    def _leaves(self):
        out = []
        todo = [ROOT]
        while todo:
            n = todo.pop()
            if not self.kids[n]:
                out.append(n)
            else:
                todo.extend(self.kids[n].values())
        return out

    def evict(self, want: int):
        """kv_cache.py:233-256: pop least-recently-touched unreferenced leaves."""
        heap = [(self.stamp[n], n) for n in self._leaves()]
        heapq.heapify(heap)
        freed = 0
        while freed < want and heap:
            _, n = heapq.heappop(heap)
            if n == ROOT:
                break
            if self.ref[n] > 0:
                continue
            self.slots.free(self.val[n])
            freed += len(self.val[n])
            par = self.up[n]
            if par >= 0:
                del self.kids[par][self.key[n][0]]
                if not self.kids[par]:
                    heapq.heappush(heap, (self.stamp[par], par))

    # -- inspection (same shape as tests/golden/make_golden.py:dump_tree) ---------------
    def dump(self):
        order_nodes = []

        def walk(n):
            order_nodes.append(n)
            for k in sorted(self.kids[n]):
                walk(self.kids[n][k])

        walk(ROOT)
        rank = {t: r for r, t in enumerate(sorted({self.stamp[n] for n in order_nodes}))}

        def ser(n):
            return {
                "key": list(self.key[n]),
                "value": list(self.val[n]),
                "ref": self.ref[n],
                "rank": rank[self.stamp[n]],
                "children": [ser(self.kids[n][k]) for k in sorted(self.kids[n])],
            }

        return ser(ROOT)


class KVManagerOracle:
    """kv_cache.py:312-369."""

    def __init__(self, size: int):
        self.slots = SlotStackOracle(size)
        self.tree = RadixOracle(self.slots)
        self.open: dict[str, tuple[int, int]] = {}

    def alloc_slots(self, n: int):
        got = self.slots.alloc(n)
        if got is None:
            self.tree.evict(n)  # the full request, not the shortfall (kv_cache.py:323)
            got = self.slots.alloc(n)
        if got is None:
            raise RuntimeError(f"Failed to allocate {n} slots, KV cache is full!")
        return got

    def free_slots(self, slots):
        self.slots.free(slots)

    def cache_sequence(self, seq, finished: bool):
        """kv_cache.py:333-369.  `seq` needs .seq_id, .token_ids, .kv_indices, .cached_kv_len."""
        new_len, new_last = self.tree.insert(seq.token_ids, seq.kv_indices)
        old_len, old_last = self.open.get(seq.seq_id, (0, ROOT))
        shared, _ = self.tree.match_prefix(seq.token_ids)
        if new_len > old_len and old_len != 0:
            self.slots.free(seq.kv_indices[old_len:new_len])
            seq.kv_indices[old_len:new_len] = shared[old_len:new_len]
        self.open[seq.seq_id] = (len(shared), new_last)
        self.tree.dec_ref(old_last)
        self.tree.inc_ref(new_last)
        if finished:
            self.tree.dec_ref(new_last)
            self.open.pop(seq.seq_id, None)
            seq.kv_indices.clear()
            seq.cached_kv_len = 0

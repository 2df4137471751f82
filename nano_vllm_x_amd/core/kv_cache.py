"""KV-cache side of the hot path: device pool, slot allocator, radix prefix tree, manager.

Same class names and call signatures as nanovllmx/core/kv_cache.py.  The device part
(`KVCachePool`, `store_kvcache`) runs the HIP scatter kernel; the index part is host integer logic
whose observable behaviour (slot numbers handed out, tree shape, eviction order) is bit-exact with
the reference and is pinned by the golden traces under tests/golden.

Recency is tracked with a logical clock instead of `time.monotonic()` (kv_cache.py:135,166,193):
it ticks at the same program points, so every comparison between two stamps has the same result,
and traces become reproducible.
"""
from __future__ import annotations

import heapq
from collections import abc
from typing import Optional

import torch

from .. import ops
from .common import Sequence, SequenceStatus

store_kvcache = ops.store_kvcache


class KVCachePool:
    """Per-worker K and V tensors of shape (layers, tokens, kv_heads, head_dim) (kv_cache.py:47-99).

    288 GB of HBM3E hold about two million Qwen3-0.6B tokens; the pool is one allocation per tensor so
    that a slot id is a plain row index for every layer."""

    def __init__(self, dtype: torch.dtype, device: torch.device, num_tokens: int, num_layers: int,
                 num_heads: int, head_dim: int, start_layer: Optional[int] = None, end_layer: Optional[int] = None):
        self.dtype, self.device = dtype, device
        self.num_tokens, self.num_layers = num_tokens, num_layers
        self.num_heads, self.head_dim = num_heads, head_dim
        self.start_layer = start_layer or 0
        self.end_layer = end_layer or num_layers - 1
        shape = (num_layers, num_tokens, num_heads, head_dim)
        # zero-filled: a padded or masked key contributes p = 0, and 0 * NaN from stale memory is NaN
        self.k_cache = torch.zeros(shape, dtype=dtype, device=device)
        self.v_cache = torch.zeros(shape, dtype=dtype, device=device)

    def _layer(self, layer: int) -> int:
        return layer - self.start_layer

    def get_kv_cache(self, layer: int, index: Optional[torch.Tensor] = None):
        k, v = self.k_cache[self._layer(layer)], self.v_cache[self._layer(layer)]
        if index is not None:
            k, v = k[index], v[index]
        return k, v

    def set_kv_cache(self, layer: int, index: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor):
        i = self._layer(layer)
        store_kvcache(k_cache, v_cache, self.k_cache[i], self.v_cache[i], index)


class KVCacheAllocator:
    """Stack of free token slots (kv_cache.py:101-117): a fresh pool hands out 0, 1, 2, ...; freed
    slots are reissued last-in first-out.

    The reference materialises the whole stack as a Python list of `size` integers; 288 GB of HBM
    hold millions of slots per rank, so here the never-yet-used tail of the stack is kept as a
    counter (`_fresh`: slots _fresh .. size-1 are untouched and lie BELOW every freed slot) and only
    freed slots are stored.  Pops come out in exactly the same order."""

    def __init__(self, size: int):
        self.size = size
        self._fresh = 0          # next never-used slot
        self._returned: list = []   # freed slots, top of the stack at the end

    @property
    def free_slots(self) -> list:
        """The stack as the reference stores it (bottom first).  O(size): for inspection and tests."""
        return list(range(self.size - 1, self._fresh - 1, -1)) + self._returned

    def num_free(self) -> int:
        return (self.size - self._fresh) + len(self._returned)

    def alloc(self, need_size: int):
        if need_size > self.num_free():
            return None
        got = []
        ret = self._returned
        take = min(need_size, len(ret))
        if take:
            got = ret[-take:]
            del ret[-take:]
            got.reverse()
        rest = need_size - take
        if rest:
            got.extend(range(self._fresh, self._fresh + rest))
            self._fresh += rest
        return got

    def free(self, free_index: abc.Iterable):
        self._returned.extend(free_index)


class _Clock:
    __slots__ = ("t",)

    def __init__(self):
        self.t = 0

    def tick(self) -> int:
        self.t += 1
        return self.t


class RadixTreeNode:
    """Edge-compressed trie node: `key` is the token run on the edge from the parent, `value` the KV
    slots of those tokens (it may be shorter than `key`: the last token of a finished sequence has no
    KV row), `children` maps the first token of each child edge to the child."""
    __slots__ = ("children", "parent", "key", "value", "access_time", "ref_count")

    def __init__(self, stamp: int, parent: Optional["RadixTreeNode"] = None, key: tuple = (), value: tuple = (),
                 ref_count: int = 0, children: Optional[dict] = None):
        self.children = children if children is not None else {}
        self.parent = parent
        self.key = key
        self.value = value
        self.access_time = stamp
        self.ref_count = ref_count

    def __lt__(self, other: "RadixTreeNode"):
        return self.access_time < other.access_time


def get_prefix_len(key1: abc.Sequence, key2: abc.Sequence) -> int:
    n = min(len(key1), len(key2))
    for i in range(n):
        if key1[i] != key2[i]:
            return i
    return n


class RadixTree:
    """Prefix tree over token ids whose nodes own KV slots (kv_cache.py:151-310).

    Behaviour kept from the reference, because slot numbers depend on it: a lookup that ends inside an
    edge SPLITS that edge (lookups mutate the tree); every node on the walked path is re-stamped; a
    split-off parent inherits the child's reference count; eviction frees whole least-recently-stamped
    unreferenced leaves until at least the requested number of slots came back."""

    def __init__(self, kv_cache_allocator: KVCacheAllocator):
        self._clock = _Clock()
        self.root = RadixTreeNode(self._clock.tick(), ref_count=1)
        self.kv_cache_allocator = kv_cache_allocator

    # -- descent shared by lookup and insertion ---------------------------------------------------
    def _walk(self, key, on_full, on_partial):
        """Follow `key` from the root.  `on_full(child, n)` is called for every edge matched in full,
        `on_partial(top, n)` once if the walk stops inside an edge (after the split).  Returns
        (last node reached, unmatched suffix of key)."""
        node = self.root
        pos = 0
        n_key = len(key)
        while pos < n_key:
            child = node.children.get(key[pos])
            if child is None:
                break
            child.access_time = self._clock.tick()
            edge = child.key
            lim = min(len(edge), n_key - pos)
            if edge[:lim] == tuple(key[pos:pos + lim]):   # whole-edge comparison at C speed (the common case)
                m = lim
            else:
                m = 0
                while edge[m] == key[pos + m]:      # a mismatch exists below lim
                    m += 1
            pos += m
            if m < len(edge):
                node = self._split_node(child, m)
                on_partial(node, m)
                break
            on_full(child, m)
            node = child
        return node, pos

    def match_prefix(self, key: list):
        if len(key) == 0:
            raise IndexError("match_prefix needs a non-empty key")
        slots: list = []
        node, _ = self._walk(key, lambda n, m: slots.extend(n.value), lambda n, m: slots.extend(n.value))
        return slots, node

    def insert(self, key: list, value: list):
        if len(key) == 0:
            raise IndexError("insert needs a non-empty key")
        node, matched = self._walk(key, lambda n, m: None, lambda n, m: None)
        last = node
        if matched < len(key):
            last = RadixTreeNode(self._clock.tick(), parent=node, key=tuple(key[matched:]), value=tuple(value[matched:]))
            node.children[key[matched]] = last
        return matched, last

    def _split_node(self, node: RadixTreeNode, split_len: int) -> RadixTreeNode:
        edge, slots = node.key, node.value
        top = RadixTreeNode(self._clock.tick(), parent=node.parent, key=edge[:split_len], value=slots[:split_len],
                            ref_count=node.ref_count, children={edge[split_len]: node})
        node.parent = top
        node.key = edge[split_len:]
        node.value = slots[split_len:]
        if top.parent is not None:
            top.parent.children[top.key[0]] = top
        return top

    # -- reference counts ------------------------------------------------------------------------
    def inc_ref(self, node: RadixTreeNode):
        root = self.root
        while node is not root and node is not None:
            node.ref_count += 1
            node = node.parent

    def dec_ref(self, node: RadixTreeNode):
        root = self.root
        while node is not root and node is not None:
            node.ref_count -= 1
            node = node.parent

    # -- eviction -------------------------------------------------------------------------------
    def _get_leaf_nodes(self):
        leaves, todo = [], [self.root]
        while todo:
            n = todo.pop()
            if n.children:
                todo.extend(n.children.values())
            else:
                leaves.append(n)
        return leaves

    def evict(self, num_tokens: int):
        heap = self._get_leaf_nodes()
        heapq.heapify(heap)
        freed = 0
        while freed < num_tokens and heap:
            victim = heapq.heappop(heap)
            if victim is self.root:
                break
            if victim.ref_count > 0:
                continue
            self.kv_cache_allocator.free(victim.value)
            freed += len(victim.value)
            parent = victim.parent
            if parent is not None:
                del parent.children[victim.key[0]]
                if not parent.children:
                    heapq.heappush(heap, parent)


class KVCacheManager:
    """Slot accounting for the scheduler (kv_cache.py:312-369)."""

    def __init__(self, size: int):
        self.kv_cache_allocator = KVCacheAllocator(size)
        self.radix_tree = RadixTree(self.kv_cache_allocator)
        self.unfinished_sequences: dict = {}   # seq_id -> (prefix_len, last_node)
        self.adopted: dict = {}                # prefix-reuse mode: seq_id -> (shared_len, tree node held)
        self.prefix_hit_tokens = 0             # prompt tokens whose KV rows came from the tree
        self.prefix_query_tokens = 0           # prompt tokens looked up

    def alloc_slots(self, num_slots: int):
        got = self.kv_cache_allocator.alloc(num_slots)
        if got is None:
            # evict the whole request, not just the shortfall (kv_cache.py:323)
            self.radix_tree.evict(num_slots)
            got = self.kv_cache_allocator.alloc(num_slots)
            if got is None:
                raise RuntimeError(f"Failed to allocate {num_slots} slots, KV cache is full!")
        return got

    def free_slots(self, indices: abc.Iterable):
        self.kv_cache_allocator.free(indices)

    def cache_sequence(self, seq: Sequence):
        """Hand a sequence's slots over to the tree (called when it finishes, scheduler.py:189)."""
        tree = self.radix_tree
        new_len, new_last = tree.insert(seq.token_ids, seq.kv_indices)
        old_len, old_last = self.unfinished_sequences.get(seq.seq_id, (0, tree.root))
        shared, _ = tree.match_prefix(seq.token_ids)
        if new_len > old_len and old_len != 0:
            # slots duplicated by a prefix that another sequence cached meanwhile
            self.kv_cache_allocator.free(seq.kv_indices[old_len:new_len])
            seq.kv_indices[old_len:new_len] = shared[old_len:new_len]
        self.unfinished_sequences[seq.seq_id] = (len(shared), new_last)
        tree.dec_ref(old_last)
        tree.inc_ref(new_last)
        if seq.status == SequenceStatus.FINISHED:
            tree.dec_ref(new_last)
            self.unfinished_sequences.pop(seq.seq_id, None)
            seq.kv_indices.clear()   # the tree owns them now
            seq.cached_kv_len = 0

    # -- admission-time prefix reuse (not in the reference: its tree is never consulted at admission) --
    def adopt_cached_prefix(self, seq: Sequence) -> int:
        """Point the head of `seq.kv_indices` at the slots of the longest cached prefix of its tokens and
        pin that path in the tree.  At least one token is left to compute (its logits are needed).
        Returns the number of tokens adopted."""
        assert not seq.kv_indices and seq.seq_id not in self.adopted
        n = len(seq.token_ids)
        self.prefix_query_tokens += n
        if n < 2:
            return 0
        slots, node = self.radix_tree.match_prefix(seq.token_ids[:-1])
        if not slots:
            return 0
        self.radix_tree.inc_ref(node)
        self.adopted[seq.seq_id] = (len(slots), node)
        seq.kv_indices.extend(slots)
        seq.cached_kv_len = len(slots)
        self.prefix_hit_tokens += len(slots)
        return len(slots)

    def release_sequence(self, seq: Sequence):
        """Give back what a sequence owns (preemption, failed admission): its private slots go to the
        allocator, the adopted prefix stays in the tree and is unpinned."""
        shared, node = self.adopted.pop(seq.seq_id, (0, None))
        if node is not None:
            self.radix_tree.dec_ref(node)
        self.kv_cache_allocator.free(seq.kv_indices[shared:])
        seq.kv_indices.clear()
        seq.cached_kv_len = 0

    def cache_sequence_shared(self, seq: Sequence):
        """Prefix-reuse counterpart of cache_sequence: only tokens that have a KV row are keyed (so
        every node has as many slots as tokens), slots that duplicate rows the tree already holds are
        freed instead of leaked, and the adopted prefix is unpinned."""
        tree = self.radix_tree
        shared, node = self.adopted.pop(seq.seq_id, (0, None))
        n = len(seq.kv_indices)
        if n:
            matched, _ = tree.insert(seq.token_ids[:n], seq.kv_indices)
            # [0, shared) are the tree's own slots; [shared, matched) were computed here but another
            # sequence cached the same tokens first; [matched, n) now belong to the new leaf
            if matched > shared:
                self.kv_cache_allocator.free(seq.kv_indices[shared:matched])
        if node is not None:
            tree.dec_ref(node)
        seq.kv_indices.clear()
        seq.cached_kv_len = 0

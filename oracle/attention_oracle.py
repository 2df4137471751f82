"""Oracle: block-table metadata and paged attention (CPU).

TEST INFRASTRUCTURE.

Block-table metadata restates nanovllmx/layers/attention.py:62-149 (eager path) and
:219-295 (graph replay); it is integer work, pinned bit-exactly by
tests/golden/g3_metadata.json.

PARITY UNPINNED for `paged_attention_oracle`: the arithmetic lives in the third-party
`flashinfer-python>=0.2.0` (pyproject.toml:19, no lock file), which is absent from
/root/reference and from this image, and the reference holds no test or vector at that
boundary.  The function restates the published semantics of the two calls the reference
makes (attention.py:106-132 plan, :354-369 run): NHD layout, page size 1, causal with the
queries right-aligned to the keys, grouped-query heads, softmax in fp32 with scale
head_dim**-0.5, output rounded to the input dtype.
"""
from __future__ import annotations

import numpy as np
import torch


def block_table_oracle(seqs):
    """attention.py:62-101,134-141.  `seqs`: iterable of (n_tokens, kv_indices, cached_kv_len).

    Returns dict of numpy arrays: qo_indptr, kv_indptr (int32), kv_indices (int32),
    last_page_len (int32, all ones at page size 1), out_slots (int64)."""
    q_lens = [n - c for n, kv, c in seqs]
    kv_lens = [len(kv) for n, kv, c in seqs]
    flat = [s for n, kv, c in seqs for s in kv]
    fresh = [s for n, kv, c in seqs for s in kv[c:]]
    return {
        "qo_indptr": np.concatenate(([0], np.cumsum(q_lens))).astype(np.int32),
        "kv_indptr": np.concatenate(([0], np.cumsum(kv_lens))).astype(np.int32),
        "kv_indices": np.asarray(flat, dtype=np.int32),
        "last_page_len": np.ones(len(kv_lens), dtype=np.int32),
        "out_slots": np.asarray(fresh, dtype=np.int64),
    }


def block_table_replay_oracle(seqs, padded_bs: int, max_bs: int, context_len: int):
    """attention.py:151-168,219-275: static buffers of a captured decode graph after a replay fill.

    Rows >= len(seqs) are padding: one key each, pointing at slot 0, and -1 in the scatter list."""
    bs = len(seqs)
    kv_lens = [len(kv) for n, kv, c in seqs] + [1] * (padded_bs - bs)
    indptr_buf = (np.arange(max_bs + 1, dtype=np.int64) * context_len).astype(np.int32)
    indptr_buf[:padded_bs + 1] = np.concatenate(([0], np.cumsum(kv_lens))).astype(np.int32)
    indices_buf = np.zeros(max_bs * context_len, dtype=np.int32)
    flat = [s for n, kv, c in seqs for s in kv]
    indices_buf[:len(flat)] = flat
    out_buf = np.full(max_bs, -1, dtype=np.int64)
    fresh = [s for n, kv, c in seqs for s in kv[c:]]
    out_buf[:bs] = fresh
    return {"indptr_buf": indptr_buf, "indices_buf": indices_buf, "out_slots_buf": out_buf,
            "n_indices": len(flat)}


def paged_attention_oracle(q, k_cache, v_cache, qo_indptr, kv_indptr, kv_indices, sm_scale, causal=True):
    """q (T, Hq, D); k_cache/v_cache (slots, Hkv, D); returns (T, Hq, D) in q's dtype.

    For sequence b, queries qo_indptr[b]:qo_indptr[b+1] attend to the keys held in slots
    kv_indices[kv_indptr[b]:kv_indptr[b+1]], the last query seeing all of them."""
    t, hq, d = q.shape
    hkv = k_cache.shape[-2]
    g = hq // hkv
    out = torch.zeros_like(q)
    qo = [int(x) for x in qo_indptr]
    kp = [int(x) for x in kv_indptr]
    idx_all = torch.as_tensor(np.asarray(kv_indices), dtype=torch.long)
    for b in range(len(qo) - 1):
        q0, q1, s0, s1 = qo[b], qo[b + 1], kp[b], kp[b + 1]
        lq, lk = q1 - q0, s1 - s0
        if lq == 0 or lk == 0:
            continue
        idx = idx_all[s0:s1]
        kk = k_cache[idx].to(torch.float32).repeat_interleave(g, dim=1)
        vv = v_cache[idx].to(torch.float32).repeat_interleave(g, dim=1)
        qq = q[q0:q1].to(torch.float32)
        s = torch.einsum("qhd,khd->hqk", qq, kk) * sm_scale
        if causal:
            qpos = torch.arange(lq)[:, None] + (lk - lq)
            s = s.masked_fill(torch.arange(lk)[None, :] > qpos, float("-inf"))
        p = torch.softmax(s, dim=-1)
        out[q0:q1] = torch.einsum("hqk,khd->qhd", p, vv).to(q.dtype)
    return out

"""Oracle: the caller of the hot path -- a Qwen3 decoder forward on CPU, and a CPU engine.

TEST INFRASTRUCTURE (and the `cpu_baseline` leg of bench.py).  Restates the op order of
nanovllmx/models/qwen3.py:117-136 (attention block), :167-188 (decoder layer), :249-268
(stack + final norm), :295-300 (logits) and the step loop of core/model_runner.py:175-189,
221-227,262-272 with every op taken from the other oracle modules.  Pinned by
tests/golden/g7_tiny_qwen3.* (reference model run on CPU around this repo's fp32 attention).
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn.functional as F

from .attention_oracle import block_table_oracle, paged_attention_oracle
from .layers_oracle import rmsnorm_oracle, rope_rows_oracle, rope_table_oracle, silu_mul_oracle


def pack_weights_oracle(sd: dict, n_layers: int):
    """HF-named tensors -> per-layer fused matrices (qwen3.py:310-348 stacking order q|k|v, gate|up)."""
    out = {"embed": sd["model.embed_tokens.weight"], "final_norm": sd["model.norm.weight"],
           "lm_head": sd.get("lm_head.weight", sd["model.embed_tokens.weight"]), "layers": []}
    for l in range(n_layers):
        p = f"model.layers.{l}."
        out["layers"].append({
            "qkv": torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"],
                              sd[p + "self_attn.v_proj.weight"]], dim=0),
            "o": sd[p + "self_attn.o_proj.weight"],
            "q_norm": sd[p + "self_attn.q_norm.weight"], "k_norm": sd[p + "self_attn.k_norm.weight"],
            "gate_up": torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], dim=0),
            "down": sd[p + "mlp.down_proj.weight"],
            "ln1": sd[p + "input_layernorm.weight"], "ln2": sd[p + "post_attention_layernorm.weight"],
        })
    return out


def _linear_oracle(x, weight, fp32: bool):
    """linear.py:73,169,276: F.linear in the tensors' dtype -- or, `fp32`, the product in fp32 rounded once to the
    activations' dtype (what a 16-bit GEMM with fp32 accumulation computes, minus its summation order)."""
    return F.linear(x.float(), weight.float()).to(x.dtype) if fp32 else F.linear(x, weight)


def qwen3_forward_oracle(w, cfg, input_ids, positions, k_cache, v_cache, table, rope_table, linear_fp32=False):
    """One forward over the flattened new tokens of a batch.

    k_cache/v_cache: (layers, slots, Hkv, D), updated in place.  `table`: block_table_oracle()."""
    hq, hkv, d = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    eps = cfg["rms_norm_eps"]
    h = F.embedding(torch.as_tensor(input_ids, dtype=torch.long), w["embed"])
    cs = rope_table[torch.as_tensor(positions, dtype=torch.long)]
    out_slots = torch.as_tensor(table["out_slots"], dtype=torch.long)
    live = out_slots >= 0
    res = None
    for l, lw in enumerate(w["layers"]):
        if res is None:
            res, x = h, rmsnorm_oracle(h, lw["ln1"], eps)
        else:
            x, res = rmsnorm_oracle(h, lw["ln1"], eps, res)
        qkv = _linear_oracle(x, lw["qkv"], linear_fp32)
        q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
        t = q.shape[0]
        q = rmsnorm_oracle(q.reshape(t, hq, d), lw["q_norm"], eps).reshape(t, hq * d)
        k = rmsnorm_oracle(k.reshape(t, hkv, d), lw["k_norm"], eps).reshape(t, hkv * d)
        q = rope_rows_oracle(q, cs, d)
        k = rope_rows_oracle(k, cs, d)
        k_cache[l][out_slots[live]] = k.reshape(t, hkv, d)[live]
        v_cache[l][out_slots[live]] = v.reshape(t, hkv, d)[live]
        a = paged_attention_oracle(q.reshape(t, hq, d), k_cache[l], v_cache[l], table["qo_indptr"],
                                   table["kv_indptr"], table["kv_indices"], d ** -0.5)
        h = _linear_oracle(a.reshape(t, hq * d).to(x.dtype), lw["o"], linear_fp32)
        x, res = rmsnorm_oracle(h, lw["ln2"], eps, res)
        h = _linear_oracle(silu_mul_oracle(_linear_oracle(x, lw["gate_up"], linear_fp32)), lw["down"], linear_fp32)
    y, _ = rmsnorm_oracle(h, w["final_norm"], eps, res)
    return y


def last_rows_oracle(table):
    """model_runner.py:221-227: index of each sequence's final new token."""
    return (np.asarray(table["qo_indptr"][1:], dtype=np.int64) - 1).tolist()


def logits_oracle(w, hidden_last, linear_fp32=False):
    if linear_fp32:      # fp32 logits from the 16-bit hidden states
        return F.linear(hidden_last.float(), w["lm_head"].float())
    return F.linear(hidden_last, w["lm_head"])


def random_weights_oracle(cfg, dtype, seed=0):
    """Random-init weights of a Qwen3 geometry (no checkpoints exist offline)."""
    g = torch.Generator().manual_seed(seed)
    h, d = cfg["hidden_size"], cfg["head_dim"]
    hq, hkv, it, v = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["intermediate_size"], cfg["vocab_size"]

    def mat(o, i):
        return (torch.randn(o, i, generator=g) * (i ** -0.5)).to(dtype)

    def vec(n):
        return (1.0 + 0.1 * torch.randn(n, generator=g)).to(dtype)

    emb = (torch.randn(v, h, generator=g) * 0.05).to(dtype)
    w = {"embed": emb, "final_norm": vec(h), "lm_head": emb if cfg.get("tie_word_embeddings", True) else mat(v, h),
         "layers": []}
    for _ in range(cfg["num_hidden_layers"]):
        w["layers"].append({"qkv": mat((hq + 2 * hkv) * d, h), "o": mat(h, hq * d), "q_norm": vec(d),
                            "k_norm": vec(d), "gate_up": mat(2 * it, h), "down": mat(h, it),
                            "ln1": vec(h), "ln2": vec(h)})
    return w


def cpu_engine_run_oracle(cfg, prompts, max_tokens, kv_slots, max_bs, context_len, dtype=torch.float32,
                          temperature=0.6, seed=0, max_steps=None, log=None):
    """The whole serving loop on CPU: oracle scheduler + oracle model + softmax/multinomial sampling.

    Returns dict(wall_s, steps, generated, nominal_tokens) where `nominal_tokens` counts what the
    reference's bench.py counts (sum of max_tokens of the requests that were run to completion)."""
    from .sched_oracle import EngineOracle, SamplingOracle

    w = random_weights_oracle(cfg, dtype, seed)
    eng = EngineOracle(kv_slots, max_bs, context_len)
    for i, (p, m) in enumerate(zip(prompts, max_tokens)):
        eng.add(f"r{i}", p, SamplingOracle(temperature=temperature, ignore_eos=True, max_tokens=m))
    nl, hkv, d = cfg["num_hidden_layers"], cfg["num_key_value_heads"], cfg["head_dim"]
    k_cache = torch.zeros(nl, kv_slots, hkv, d, dtype=dtype)
    v_cache = torch.zeros_like(k_cache)
    rope = rope_table_oracle(d, cfg["max_position_embeddings"], cfg["rope_theta"])
    gen = torch.Generator().manual_seed(seed)
    t0 = time.perf_counter()
    steps = generated = 0
    with torch.inference_mode():
        while eng.sched.busy():
            if max_steps is not None and steps >= max_steps:
                break
            b = eng.sched.schedule()
            rows = [(len(s.token_ids), s.kv_indices, s.cached_kv_len) for s in b.seqs]
            table = block_table_oracle(rows)
            ids = [t for s in b.seqs for t in s.token_ids[s.cached_kv_len:]]
            pos = [p for s in b.seqs for p in range(s.cached_kv_len, len(s.token_ids))]
            hid = qwen3_forward_oracle(w, cfg, ids, pos, k_cache, v_cache, table, rope)
            lg = logits_oracle(w, hid[last_rows_oracle(table)]).float() / temperature
            toks = torch.multinomial(torch.softmax(lg, dim=-1), 1, generator=gen)[:, 0].tolist()
            generated += len(eng.absorb(b, toks))
            steps += 1
            if log:
                log(steps, b.mode, len(b.seqs))
    return {"wall_s": time.perf_counter() - t0, "steps": steps, "generated": generated,
            "nominal_tokens": int(sum(max_tokens))}


def cpu_engine_greedy_oracle(cfg, state_dict, prompts, max_new_tokens, kv_slots, max_bs, context_len, dtype=torch.float32,
                             keep_logits=True, forced=None, linear_fp32=False):
    """The serving loop on CPU with arg-max sampling (what temperature -> 0 or top_k = 1 gives): oracle
    scheduler + oracle model over the weights of `state_dict` (Hugging Face names, as checkpoints and
    nano_vllm_x_amd.qwen3.random_checkpoint give them), cast to `dtype`.

    Returns dict(streams {request index: [token ids]}, logits {request index: [fp32 (vocab) row per step]} when
    keep_logits, wall_s, prefill_s, decode_s, steps, generated, prompt_tokens).  `forced` (optional,
    {request index: [ids]}): tokens fed instead of the arg-max ones (teacher forcing), so that the logits of two
    implementations can be compared step by step on identical histories."""
    from .sched_oracle import PREFILL, EngineOracle, SamplingOracle

    w = pack_weights_oracle({k: v.to(dtype) for k, v in state_dict.items()}, cfg["num_hidden_layers"])
    eng = EngineOracle(kv_slots, max_bs, context_len)
    for i, (p, n) in enumerate(zip(prompts, max_new_tokens)):
        eng.add(str(i), p, SamplingOracle(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=n))
    nl, hkv, d = cfg["num_hidden_layers"], cfg["num_key_value_heads"], cfg["head_dim"]
    k_cache = torch.zeros(nl, kv_slots, hkv, d, dtype=dtype)
    v_cache = torch.zeros_like(k_cache)
    rope = rope_table_oracle(d, cfg["max_position_embeddings"], cfg["rope_theta"])
    streams = {i: [] for i in range(len(prompts))}
    logits = {i: [] for i in range(len(prompts))}
    t0 = time.perf_counter()
    steps = generated = 0
    split = {"prefill_s": 0.0, "decode_s": 0.0}
    with torch.inference_mode():
        while eng.sched.busy():
            ts = time.perf_counter()
            b = eng.sched.schedule()
            rows = [(len(s.token_ids), s.kv_indices, s.cached_kv_len) for s in b.seqs]
            table = block_table_oracle(rows)
            ids = [t for s in b.seqs for t in s.token_ids[s.cached_kv_len:]]
            pos = [p for s in b.seqs for p in range(s.cached_kv_len, len(s.token_ids))]
            hid = qwen3_forward_oracle(w, cfg, ids, pos, k_cache, v_cache, table, rope, linear_fp32)
            lg = logits_oracle(w, hid[last_rows_oracle(table)], linear_fp32).float()
            toks = lg.argmax(dim=-1).tolist()
            for j, s in enumerate(b.seqs):
                i = int(s.seq_id)
                if forced is not None:
                    toks[j] = forced[i][len(streams[i])]
                streams[i].append(toks[j])
                if keep_logits:
                    logits[i].append(lg[j].clone())
            generated += len(eng.absorb(b, toks))
            steps += 1
            split["prefill_s" if b.mode == PREFILL else "decode_s"] += time.perf_counter() - ts
    return {"streams": streams, "logits": logits if keep_logits else None, "wall_s": time.perf_counter() - t0,
            "steps": steps, "generated": generated, "prompt_tokens": int(sum(len(p) for p in prompts)), **split}

"""Oracle vs golden vectors frozen from the reference (eager CPU): metadata, scatter, layers, sampler, tiny model."""
import numpy as np
import torch

from helpers import bf16_ulp_diff, from_bits, golden_json, golden_npz, to_bits
from oracle.attention_oracle import block_table_oracle, block_table_replay_oracle
from oracle.layers_oracle import (rmsnorm_oracle, rope_rows_oracle, rope_table_oracle, sampler_masks_oracle,
                                  silu_mul_oracle, store_kv_oracle)


def test_block_table_matches_reference():
    g = golden_json("g3_metadata.json")
    for case in g["cases"]:
        rows = [(s["ntok"], s["kv"], s["cached"]) for s in case["seqs"]]
        t = block_table_oracle(rows)
        plan = case["plan"]
        indptr = plan.get("paged_kv_indptr", plan.get("indptr"))
        indices = plan.get("paged_kv_indices", plan.get("indices"))
        lpl = plan.get("paged_kv_last_page_len", plan.get("last_page_len"))
        assert t["kv_indptr"].tolist() == indptr and t["kv_indptr"].dtype == np.int32
        assert t["kv_indices"].tolist() == indices and t["kv_indices"].dtype == np.int32
        assert t["last_page_len"].tolist() == lpl
        if "qo_indptr" in plan:
            assert t["qo_indptr"].tolist() == plan["qo_indptr"]
        assert t["out_slots"].tolist() == case["output_kv_indices"] and t["out_slots"].dtype == np.int64
        assert case["output_kv_indices_dtype"] == "torch.int64"


def test_block_table_replay_matches_reference():
    r = golden_json("g3_metadata.json")["replay"]
    rows = [(s["ntok"], s["kv"], s["cached"]) for s in r["seqs"]]
    t = block_table_replay_oracle(rows, r["padded_bs"], r["max_bs"], r["context_len"])
    assert t["indptr_buf"].tolist() == r["buf_indptr"]
    assert t["indptr_buf"][:r["padded_bs"] + 1].tolist() == r["indptr"]
    assert t["indices_buf"][:20].tolist() == r["buf_indices_head"]
    assert t["indices_buf"][:t["n_indices"]].tolist() == r["indices"]
    assert t["out_slots_buf"].tolist() == r["buf_output_kv_indices"]
    assert t["out_slots_buf"][:r["padded_bs"]].tolist() == r["output_kv_indices"]


def test_store_kv_matches_reference():
    g = golden_npz("g4_store_kv.npz")
    hq, hkv, d, t, slots, layers, start_layer, layer = g["meta"].tolist()
    qkv = g["qkv"]
    k = qkv[:, hq * d:(hq + hkv) * d].reshape(t, hkv, d)
    v = qkv[:, (hq + hkv) * d:].reshape(t, hkv, d)
    kc, vc = g["k_before"].copy(), g["v_before"].copy()
    store_kv_oracle(k, v, kc[layer - start_layer], vc[layer - start_layer], g["idx"])
    assert np.array_equal(kc, g["k_after"]) and np.array_equal(vc, g["v_after"])


def test_rmsnorm_matches_reference():
    g = golden_npz("g5_layers.npz")
    for tag in ("h1024", "h4096", "hd128"):
        w, x, r = (from_bits(g[f"rms_{tag}_{n}"]) for n in ("w", "x", "res"))
        assert np.array_equal(to_bits(rmsnorm_oracle(x, w, 1e-6)), g[f"rms_{tag}_y"])
        y, ro = rmsnorm_oracle(x, w, 1e-6, r)
        assert np.array_equal(to_bits(y), g[f"rms_{tag}_y_res"])
        assert np.array_equal(to_bits(ro), g[f"rms_{tag}_res_out"])


def test_rope_matches_reference():
    g = golden_npz("g5_layers.npz")
    table = rope_table_oracle(128, 40960, 1000000)
    pos = torch.from_numpy(g["rope_pos"])
    assert np.array_equal(table[pos].numpy(), g["rope_cache_rows"])
    for name in ("q", "k"):
        out = rope_rows_oracle(from_bits(g[f"rope_{name}"]), table[pos], 128)
        assert np.array_equal(to_bits(out), g[f"rope_{name}_out"])


def test_silu_mul_matches_reference():
    g = golden_npz("g5_layers.npz")
    assert np.array_equal(to_bits(silu_mul_oracle(from_bits(g["silu_x"]))), g["silu_y"])


def test_sampler_masks_match_reference():
    g = golden_npz("g6_sampler.npz")
    logits = from_bits(g["logits"]).float().numpy()
    k1, k2, probs = sampler_masks_oracle(logits, g["temps"], g["top_k"], g["top_p"], g["min_p"])
    assert np.array_equal(k1, g["keep_after_kp"]) and np.array_equal(k2, g["keep_after_minp"])
    assert np.array_equal(probs, g["probs"])
    assert int(np.argmax(probs[5])) == int(g["greedy_row5"][0]) == int(g["argmax"][5])


def test_tiny_qwen3_matches_reference():
    from oracle.qwen3_oracle import (last_rows_oracle, logits_oracle, pack_weights_oracle,
                                     qwen3_forward_oracle)
    from tiny_model import tiny_state_dict

    meta = golden_json("g7_tiny_qwen3.json")
    g = golden_npz("g7_tiny_qwen3.npz")
    cfg = meta["config"]
    w = pack_weights_oracle(tiny_state_dict(cfg), cfg["num_hidden_layers"])
    rope = rope_table_oracle(cfg["head_dim"], cfg["max_position_embeddings"], cfg["rope_theta"])
    kc = torch.zeros(cfg["num_hidden_layers"], meta["pool_slots"], 1, 128, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    toks = [list(p) for p in meta["prompts"]]
    kv = [list(s) for s in meta["slots"]]
    cached = [0, 0, 0]
    for step in range(3):
        rows = [(len(t), k, c) for t, k, c in zip(toks, kv, cached)]
        table = block_table_oracle(rows)
        ids = [x for t, c in zip(toks, cached) for x in t[c:]]
        pos = [p for t, c in zip(toks, cached) for p in range(c, len(t))]
        hid = qwen3_forward_oracle(w, cfg, ids, pos, kc, vc, table, rope)
        hl = hid[last_rows_oracle(table)]
        assert bf16_ulp_diff(hl, from_bits(g[f"hidden{step}"])) == 0
        assert np.array_equal(logits_oracle(w, hl).float().numpy(), g[f"logits{step}"])
        if step < 2:
            for i in range(3):
                cached[i] = len(kv[i])
                toks[i].append(meta["feed"][step][i])
                kv[i].append(meta["next_slot"][i] + step)
    assert np.array_equal(to_bits(kc), g["k_cache"]) and np.array_equal(to_bits(vc), g["v_cache"])

"""Tensor-parallel path at world size 2 on CPU (gloo): collective wrappers, shard rules, SPMD engine.

The device kernels need a GPU, so the checks here are on what tensor parallelism adds around them:
the wrappers of distributed.py, the shard rules of Qwen3ForCausalLM.load_weights (qkv by head,
gate/up by column, o/down by row, vocabulary by row) and the rule that every rank replays the same
scheduler with rank 0's sampled ids.  Reference: nanovllmx/layers/linear.py:387-408,488-498,
vocab_parallel_embedding.py:93-108,147-151, distributed/communication_op.py:9-52.
"""
from __future__ import annotations

import os
import socket
import sys
import traceback

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORLD = 2
CFG = dict(vocab_size=200, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
           num_key_value_heads=2, head_dim=128, max_position_embeddings=64, tie_word_embeddings=False)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, port: int, fn_name: str, q):
    try:
        torch.set_num_threads(1)
        from nano_vllm_x_amd import distributed as tpd

        tpd.init_tensor_parallel(WORLD, rank=rank, backend="gloo", port=port)
        with torch.no_grad():   # the engine runs its forward passes without autograd
            globals()[fn_name](rank)
        tpd.shutdown()
        q.put((rank, None))
    except Exception:   # noqa: BLE001 - reported to the parent
        q.put((rank, traceback.format_exc()))


def _run(fn_name: str):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, fn_name, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, err in results:
        assert err is None, f"rank {rank} failed:\n{err}"


# ---------------------------------------------------------------------------------------------
def _case_collectives(rank: int):
    from nano_vllm_x_amd import distributed as tpd

    assert tpd.tp_size() == WORLD and tpd.tp_rank() == rank
    x = torch.full((3, 4), float(rank + 1))
    assert torch.equal(tpd.tp_all_reduce(x), torch.full((3, 4), 3.0))
    shard = torch.arange(6, dtype=torch.float32).view(2, 3) + 100 * rank
    full = tpd.tp_all_gather_last_dim(shard)
    want = torch.cat([torch.arange(6, dtype=torch.float32).view(2, 3) + 100 * r for r in range(WORLD)], dim=-1)
    assert torch.equal(full, want)
    ids = torch.tensor([5, 6, 7]) if rank == 0 else torch.zeros(3, dtype=torch.int64)
    assert tpd.tp_broadcast(ids).tolist() == [5, 6, 7]
    assert tpd.tp_all_gather_min_int(1000 - 10 * rank, "cpu") == 990
    # the ranks' per-row keys of a sharded sampling step, rank by rank
    keys = torch.tensor([10 * rank + 1, 10 * rank + 2, 10 * rank + 3], dtype=torch.int64)
    assert tpd.tp_all_gather_rows(keys).tolist() == [[1, 2, 3], [11, 12, 13]]
    # without a GPU there is no direct-peer exchange: it reports so, and a row-parallel output is reduced at once
    assert tpd.init_peer_allreduce(8, 64, torch.bfloat16, "cpu") is False and not tpd.peer_allreduce_active()
    # gloo moves tensors through host memory: without the direct-peer exchange no decode size of this group can be recorded
    assert tpd.decode_graph_capturable(1) is False and tpd.ranks_on_my_device("cpu") == 1
    y = tpd.tp_reduce_later(torch.full((2, 64), float(rank + 1)))
    assert isinstance(y, torch.Tensor) and torch.equal(y, torch.full((2, 64), 3.0))
    assert tpd.tp_finish_reduce(y) is y
    tpd.check_peer_status()        # nothing to report


def test_collective_wrappers():
    _run("_case_collectives")


def _full_weights(cfg):
    gen = torch.Generator().manual_seed(11)
    h, d = cfg["hidden_size"], cfg["head_dim"]
    hq, hkv, inter, v = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["intermediate_size"], cfg["vocab_size"]
    w = {"model.embed_tokens.weight": torch.randn(v, h, generator=gen), "lm_head.weight": torch.randn(v, h, generator=gen),
         "model.norm.weight": torch.randn(h, generator=gen)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        w[p + "self_attn.q_proj.weight"] = torch.randn(hq * d, h, generator=gen)
        w[p + "self_attn.k_proj.weight"] = torch.randn(hkv * d, h, generator=gen)
        w[p + "self_attn.v_proj.weight"] = torch.randn(hkv * d, h, generator=gen)
        w[p + "self_attn.o_proj.weight"] = torch.randn(h, hq * d, generator=gen)
        w[p + "self_attn.q_norm.weight"] = torch.randn(d, generator=gen)
        w[p + "self_attn.k_norm.weight"] = torch.randn(d, generator=gen)
        w[p + "mlp.gate_proj.weight"] = torch.randn(inter, h, generator=gen)
        w[p + "mlp.up_proj.weight"] = torch.randn(inter, h, generator=gen)
        w[p + "mlp.down_proj.weight"] = torch.randn(h, inter, generator=gen)
        w[p + "input_layernorm.weight"] = torch.randn(h, generator=gen)
        w[p + "post_attention_layernorm.weight"] = torch.randn(h, generator=gen)
    return w


def _case_shards(rank: int):
    """Sharded projections + collectives give the unsharded result (fp32 on CPU, so exact up to the
    order of one two-term sum)."""
    import torch.nn.functional as F

    from nano_vllm_x_amd import distributed as tpd
    from nano_vllm_x_amd.qwen3 import Qwen3Config, Qwen3ForCausalLM

    cfg = Qwen3Config.from_dict(CFG)
    model = Qwen3ForCausalLM(cfg)
    w = _full_weights(CFG)
    model.load_weights(w.items())
    h, d, hq, hkv = cfg.hidden_size, cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(5, h, generator=gen)
    layer = model.model.layers[1]
    p = "model.layers.1."

    # qkv: this rank holds heads [rank*hq/2, ...) of q and [rank*hkv/2, ...) of k and v
    a = layer.self_attn
    assert (a.num_heads, a.num_kv_heads) == (hq // WORLD, hkv // WORLD)
    q, k, v = a.qkv_proj(x).split([a.q_size, a.kv_size, a.kv_size], dim=-1)
    q_full = F.linear(x, w[p + "self_attn.q_proj.weight"])
    k_full = F.linear(x, w[p + "self_attn.k_proj.weight"])
    v_full = F.linear(x, w[p + "self_attn.v_proj.weight"])
    assert torch.equal(q, q_full[:, rank * a.q_size:(rank + 1) * a.q_size])
    assert torch.equal(k, k_full[:, rank * a.kv_size:(rank + 1) * a.kv_size])
    assert torch.equal(v, v_full[:, rank * a.kv_size:(rank + 1) * a.kv_size])

    # o_proj: row parallel, partial products summed by all-reduce
    attn_full = torch.randn(5, hq * d, generator=gen)
    part = a.o_proj(attn_full[:, rank * a.q_size:(rank + 1) * a.q_size].contiguous())
    got = tpd.tp_all_reduce(part)
    torch.testing.assert_close(got, F.linear(attn_full, w[p + "self_attn.o_proj.weight"]), rtol=1e-5, atol=1e-4)

    # MLP: gate/up column shards, down row shard (activation computed here in torch: the HIP op needs a GPU)
    gu = layer.mlp.gate_up_proj(x)
    inter = cfg.intermediate_size // WORLD
    g_full = F.linear(x, w[p + "mlp.gate_proj.weight"])
    u_full = F.linear(x, w[p + "mlp.up_proj.weight"])
    assert torch.equal(gu[:, :inter], g_full[:, rank * inter:(rank + 1) * inter])
    assert torch.equal(gu[:, inter:], u_full[:, rank * inter:(rank + 1) * inter])
    act = F.silu(gu[:, :inter]) * gu[:, inter:]
    got = tpd.tp_all_reduce(layer.mlp.down_proj(act))
    want = F.linear(F.silu(g_full) * u_full, w[p + "mlp.down_proj.weight"])
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-4)

    # vocabulary: masked embedding + all-reduce; LM head shards gathered and cut back to vocab_size
    ids = torch.tensor([0, 1, 99, 100, 127, 128, 199])
    assert torch.equal(model.model.embed_tokens(ids), w["model.embed_tokens.weight"][ids])
    logits = model.compute_logits(x)
    assert logits.shape == (5, cfg.vocab_size)
    torch.testing.assert_close(logits, F.linear(x, w["lm_head.weight"]), rtol=1e-5, atol=1e-4)


def test_shard_rules_match_unsharded_model():
    _run("_case_shards")


def _case_kv_head_replication(rank: int):
    """Fewer KV heads than ranks: every rank keeps one KV head, ranks r and r+1 share head r // replicas
    (nanovllmx/layers/linear.py:333-338,403)."""
    from nano_vllm_x_amd.qwen3 import Qwen3Config, Qwen3ForCausalLM

    cfg_d = dict(CFG, num_key_value_heads=1, num_attention_heads=2, num_hidden_layers=1)
    cfg = Qwen3Config.from_dict(cfg_d)
    model = Qwen3ForCausalLM(cfg)
    w = _full_weights(cfg_d)
    model.load_weights(w.items())
    a = model.model.layers[0].self_attn
    assert (a.num_heads, a.num_kv_heads, a.kv_replicas) == (1, 1, 2)
    wk = a.qkv_proj.weight[a.q_size:a.q_size + a.kv_size]
    assert torch.equal(wk, w["model.layers.0.self_attn.k_proj.weight"])   # the single KV head, on both ranks
    wq = a.qkv_proj.weight[:a.q_size]
    assert torch.equal(wq, w["model.layers.0.self_attn.q_proj.weight"][rank * 128:(rank + 1) * 128])


def test_kv_head_replication_when_tp_exceeds_kv_heads():
    _run("_case_kv_head_replication")


def _case_spmd_engine(rank: int):
    """Every rank drives its own copy of the scheduler; only rank 0's sampled ids are used.  After the
    run all ranks hold identical slot assignments and radix trees."""
    import hashlib

    from nano_vllm_x_amd import distributed as tpd
    from nano_vllm_x_amd.core.common import SamplingParams
    from nano_vllm_x_amd.core.engine import EngineCore

    core = EngineCore(kv_cache_size=700, max_bs=4, context_len=256)
    gen = torch.Generator().manual_seed(5)
    for i in range(9):
        n = int(torch.randint(5, 60, (1,), generator=gen))
        prompt = torch.randint(0, 50, (n,), generator=gen).tolist()
        core.add_sequence(f"s{i}", prompt, SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=n + 7 + i))
    local = torch.Generator().manual_seed(100 + rank)    # ranks would sample differently on their own
    digest = hashlib.sha256()
    steps = 0
    while core.has_unfinished_sequences():
        batch = core.scheduler.schedule()
        ids = torch.randint(0, 50, (batch.num_seqs,), generator=local)
        ids = tpd.tp_broadcast(ids, src=0)
        for s in batch.seqs:
            digest.update(repr((s.seq_id, s.kv_indices[-3:], len(s.token_ids))).encode())
        core.update_from_output(batch, ids.tolist())
        steps += 1
    mgr = core.scheduler.kv_manager
    state = torch.tensor([steps, len(mgr.kv_cache_allocator.free_slots),
                          int.from_bytes(digest.digest()[:7], "big")], dtype=torch.int64)
    other = state.clone()
    tpd.tp_broadcast(other, src=0)
    assert torch.equal(state, other), f"rank {rank} diverged from rank 0: {state.tolist()} vs {other.tolist()}"


def test_spmd_engine_ranks_stay_identical():
    _run("_case_spmd_engine")

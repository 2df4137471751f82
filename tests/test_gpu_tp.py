"""Tensor parallelism and pipeline stages on the device path: two ranks sharing the one GPU of the test box.

RCCL refuses two ranks on one device, so the ranks talk through gloo (device tensors are staged through
the host) and decode runs eagerly; what is checked is everything tensor parallelism changes on the
device side: per-rank head counts in the attention / norm / rotary / scatter kernels, column- and
row-sharded projections (skinny kernel and library GEMM), vocab-parallel embedding, the LM-head gather
and the broadcast of the sampled ids.  The logits of the two-rank model must equal the one-rank model's
to bf16 working precision (the all-reduce sums two bf16 halves instead of one fp32 accumulation).
"""
from __future__ import annotations

import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from model_checks import check_logits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

CFG = dict(vocab_size=1000, hidden_size=512, intermediate_size=1536, num_hidden_layers=2, num_attention_heads=8,
           num_key_value_heads=2, head_dim=128, max_position_embeddings=256, tie_word_embeddings=False)
PROMPT_LENS = (5, 19, 33, 8)


def _run_model(tp: int, rank: int, port: int, pp: int = 1, eager: bool = True, peer: str = "1"):
    if tp > 1:      # (a child process of the test: nothing leaks; the one-rank reference runs inside the test process)
        os.environ["NVX_PEER_AR"] = peer
    from nano_vllm_x_amd import distributed as tpd
    from nano_vllm_x_amd.core.common import ForwardBatch, ForwardMode, SamplingParams, Sequence
    from nano_vllm_x_amd.core.model_runner import ModelRunner
    from nano_vllm_x_amd.qwen3 import Qwen3Config

    if tp * pp > 1:
        tpd.init_tensor_parallel(tp, rank=rank, backend="gloo", port=port, pp_size=pp)
    r = ModelRunner(model="tp-test", max_bs=8, tp_size=tp, enforce_eager=eager, context_len=128,
                    config=Qwen3Config.from_dict(CFG), seed=5, pp_size=pp)
    r.load_model()           # seeded random weights: every rank draws the full matrices and keeps its shard
    r.initialize_kv_cache(512)
    rng = np.random.default_rng(9)
    seqs, slot = [], 0
    for i, n in enumerate(PROMPT_LENS):
        p = rng.integers(0, CFG["vocab_size"], size=n).tolist()
        seqs.append(Sequence(seq_id=f"s{i}", token_ids=p, num_tokens=n, prompt_len=n, kv_indices=list(range(slot, slot + n)),
                             sampling_params=SamplingParams(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=8)))
        slot += n + 8
    logits, sampled = [], []
    for step in range(3):
        mode = ForwardMode.PREFILL if step == 0 else ForwardMode.DECODE
        batch = ForwardBatch(forward_mode=mode, num_seqs=len(seqs), seqs=seqs)
        lg = r.compute_logits(batch)
        logits.append(None if lg is None else lg.float().cpu().numpy())      # None: not the last pipeline stage
        sampled.append(r.execute_model(batch) if step == 2 else None)    # the sampler + broadcast path, once
        for s in seqs:
            s.cached_kv_len = len(s.kv_indices)
            s.token_ids.append(int(rng.integers(0, CFG["vocab_size"])))
            s.kv_indices.append(s.kv_indices[-1] + 1)
    heads = (r.num_heads, r.num_kv_heads, r.num_layers)
    LAST_RUN.update(peer_state=tpd.PEER_STATE, stats=dict(r.stats))
    r.shutdown()
    return logits, sampled[2], heads


LAST_RUN: dict = {}     # how the last _run_model of this process ran (all-reduce path, step counters)


def _worker(rank, port, q, tp, pp, eager=True, peer="1"):
    try:
        out = _run_model(tp, rank, port, pp, eager, peer)
        q.put((rank, out + (dict(LAST_RUN),), None))
    except Exception:   # noqa: BLE001
        q.put((rank, None, traceback.format_exc()))


def _run_ranks(tp: int, pp: int, eager: bool = True, peer: str = "1") -> dict:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q, tp, pp, eager, peer)) for r in range(tp * pp)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        rank, out, err = q.get(timeout=240)
        assert err is None, f"rank {rank} failed:\n{err}"
        results[rank] = out
    for p in procs:
        p.join(timeout=60)
    return results


def test_two_pipeline_stages_on_one_gpu_match_one_rank():
    """pp_size 2: stage 0 holds the embedding and layer 0, stage 1 layer 1, the final norm and the LM head;
    (hidden, residual) cross between them once per step.  Same arithmetic in the same order as one rank,
    except that a decode step's K-split partial products are summed and rounded to bf16 before they leave
    stage 0 (one rank adds them to the residual in fp32), hence the tolerance."""
    results = _run_ranks(1, 2)
    want, want_ids, heads1 = _run_model(1, 0, 0)
    assert heads1 == (8, 2, 2) and results[0][2] == (8, 2, 1) and results[1][2] == (8, 2, 1)
    for step in range(3):
        assert results[0][0][step] is None
        check_logits("pp2_vs_one_rank", results[1][0][step], want[step], f"step {step}")
    assert results[0][1] == results[1][1] == want_ids      # greedy ids, broadcast from the last stage


def test_two_stages_of_two_tensor_parallel_ranks_match_one_rank():
    """world 4 = 2 pipeline stages x 2 tensor-parallel ranks (the layout of the 32B configuration, scaled down): each
    stage's TP group has its own direct-peer exchange; a stage that is not the last finishes its last all-reduce
    itself (no norm of its own follows) before the pair goes to the next stage."""
    results = _run_ranks(2, 2)
    want, want_ids, _ = _run_model(1, 0, 0)
    for r in range(4):
        assert results[r][2] == (4, 1, 1) and results[r][3]["peer_state"].startswith("direct-peer"), results[r][2:]
    for step in range(3):
        assert results[0][0][step] is None and results[1][0][step] is None          # first stage: no logits
        assert np.array_equal(results[2][0][step], results[3][0][step]), "the last stage's ranks disagree"
        check_logits("tp2_pp2_vs_one_rank", results[2][0][step], want[step], f"step {step}")
    assert all(results[r][1] == want_ids for r in range(4))


@pytest.mark.parametrize("peer", ["1", "0"])
def test_two_ranks_on_one_gpu_match_one_rank(peer):
    """peer "1": the all-reduces of the row-parallel projections and of the embedding go peer to peer (fused with
    the norms); "0": through the process group (gloo, staged through the host)."""
    results = _run_ranks(2, 1, peer=peer)
    want, want_ids, heads1 = _run_model(1, 0, 0)
    assert heads1 == (8, 2, 2) and results[0][2] == (4, 1, 2)
    for r in range(2):
        assert results[r][3]["peer_state"].startswith("direct-peer" if peer == "1" else "off"), results[r][3]
    for step in range(3):
        a0, a1, b = results[0][0][step], results[1][0][step], want[step]
        assert np.array_equal(a0, a1), "ranks disagree after the LM-head gather"
        check_logits("tp2_vs_one_rank", a0, b, f"step {step}")
    assert results[0][1] == results[1][1] == want_ids      # greedy ids, broadcast from rank 0


def test_four_ranks_with_replicated_kv_heads_match_one_rank():
    """Four ranks over two kv heads: each kv head lives on two ranks (linear.py:333-338); direct-peer all-reduce."""
    results = _run_ranks(4, 1)
    want, want_ids, _ = _run_model(1, 0, 0)
    assert results[0][2] == (2, 1, 2) and results[3][3]["peer_state"].startswith("direct-peer")
    for step in range(3):
        for r in range(1, 4):
            assert np.array_equal(results[0][0][step], results[r][0][step]), "ranks disagree after the LM-head gather"
        check_logits("tp4_vs_one_rank", results[0][0][step], want[step], f"step {step}")
    assert all(results[r][1] == want_ids for r in range(4))


def test_tensor_parallel_decode_graph_with_direct_peer_allreduce():
    """With the all-reduces peer to peer, a tensor-parallel rank's decode step holds no process-group call and is
    captured as a hipGraph even where the process group cannot be (gloo ranks sharing one GPU): both decode steps
    must replay from the graph and give the eager run's logits bit for bit, on every rank."""
    graph = _run_ranks(2, 1, eager=False)
    eager = _run_ranks(2, 1, eager=True)
    for r in range(2):
        st = graph[r][3]["stats"]
        assert st["graph_steps"] == st["decode_steps"] >= 2 and st["eager_decode_steps"] == 0, st
        assert eager[r][3]["stats"]["graph_steps"] == 0
        for step in range(3):
            assert np.array_equal(graph[r][0][step], eager[r][0][step]), f"rank {r} step {step}: graph replay differs from eager launches"
    assert graph[0][1] == graph[1][1] == eager[0][1]


# -- run-ahead on tensor-parallel ranks -----------------------------------------------------------------
def _run_tp_engine(rank, tp, port, q, run_ahead, sample_mode="shard"):
    try:
        os.environ["NVX_PEER_AR"] = "1"
        os.environ["NVX_TP_SAMPLE"] = sample_mode
        from nano_vllm_x_amd import distributed as tpd
        from nano_vllm_x_amd.core.common import SamplingParams
        from nano_vllm_x_amd.core.engine import Engine
        from nano_vllm_x_amd.qwen3 import PRESETS

        PRESETS["tp-engine-test"] = dict(CFG)
        tpd.init_tensor_parallel(tp, rank=rank, backend="gloo", port=port)
        eng = Engine("tp-engine-test", 0.5, max_bs=8, tp_size=tp, enforce_eager=False, context_len=128, kv_cache_size=1024, seed=5,
                     run_ahead=run_ahead)
        rng = np.random.default_rng(4)
        want = {}
        for i in range(6):
            n_new = int(rng.integers(6, 14))
            want[f"q{i}"] = n_new
            eng.add_sequence(f"q{i}", rng.integers(0, CFG["vocab_size"], size=int(rng.integers(4, 30))).tolist(),
                             SamplingParams(temperature=0.8, ignore_eos=True, max_new_tokens=n_new))
        got = {k: [] for k in want}
        for _ in range(400):
            if not (eng.has_unfinished_sequences() or eng.step_in_flight):
                break
            for o in eng.step_overlapped():
                got[o.seq_id].append(o.new_token_id)
        stats = dict(eng.model_runner.stats)
        ahead = eng.ahead_steps
        eng.shutdown()
        q.put((rank, (got, want, stats, ahead), None))
    except Exception:   # noqa: BLE001
        q.put((rank, None, traceback.format_exc()))


def test_sharded_sampling_and_run_ahead_on_tensor_parallel_ranks_generate_the_same_streams():
    """Two ranks, decode steps replayed from the graph with direct-peer all-reduces inside, step N+1 queued before
    step N's ids have reached the host: the decision to queue ahead depends on the scheduler's state alone, which
    is the same on every rank, so the ranks stay in lockstep and the sampled streams equal those without it."""
    runs = {}
    # the reference's way (gather the logits, sample whole rows, broadcast the ids), then each rank drawing over its own
    # shard of the vocabulary, then that with run-ahead: the sampled streams must be the same in all three
    for ra, mode in ((False, "gather"), (False, "shard"), (True, "shard")):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_run_tp_engine, args=(r, 2, port, q, ra, mode)) for r in range(2)]
        for p in procs:
            p.start()
        res = {}
        for _ in procs:
            rank, out, err = q.get(timeout=240)
            assert err is None, f"rank {rank} failed:\n{err}"
            res[rank] = out
        for p in procs:
            p.join(timeout=60)
        assert res[0][0] == res[1][0], "ranks generated different tokens"
        for k, n in res[0][1].items():
            assert len(res[0][0][k]) == n
        runs[(ra, mode)] = res[0]
    base, shard, ahead = runs[(False, "gather")], runs[(False, "shard")], runs[(True, "shard")]
    assert shard[0] == base[0], "sampling over vocabulary shards changed the sampled streams"
    assert ahead[0] == base[0], "run-ahead changed the sampled streams"
    assert ahead[3] > 0 and base[3] == 0 and shard[3] == 0
    assert ahead[2]["graph_steps"] > 0 and ahead[2]["eager_decode_steps"] == 0
    assert base[2].get("sharded_sampling_steps", 0) == 0 and shard[2]["sharded_sampling_steps"] > 0 and ahead[2]["sharded_sampling_steps"] > 0


# -- several batches in flight across the stages --------------------------------------------------
def _run_engine(pp: int, rank: int, port: int):
    from nano_vllm_x_amd import distributed as tpd
    from nano_vllm_x_amd.core.common import SamplingParams
    from nano_vllm_x_amd.core.engine import Engine
    from nano_vllm_x_amd.qwen3 import PRESETS

    PRESETS["pp-engine-test"] = dict(CFG)
    if pp > 1:
        tpd.init_tensor_parallel(1, rank=rank, backend="gloo", port=port, pp_size=pp)
    eng = Engine("pp-engine-test", 0.5, max_bs=3, pp_size=pp, enforce_eager=True, context_len=128, kv_cache_size=1024, seed=5)
    rng = np.random.default_rng(4)
    want = {}
    for i in range(8):      # more requests than a batch holds: the scheduler always has a batch for an idle first stage
        n_new = int(rng.integers(3, 9))
        want[f"q{i}"] = n_new
        eng.add_sequence(f"q{i}", rng.integers(0, CFG["vocab_size"], size=int(rng.integers(4, 30))).tolist(),
                         SamplingParams(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=n_new))
    got = {k: [] for k in want}
    finished = set()
    aborted = False
    for _ in range(400):
        if not (eng.has_unfinished_sequences() or eng.step_in_flight):
            break
        for o in eng.step():
            assert o.seq_id not in finished
            got[o.seq_id].append(o.new_token_id)
            if o.is_finished:
                finished.add(o.seq_id)
        if not aborted and len(got["q1"]) == 1:      # an abort while the sequence may be in flight
            eng.abort_sequence("q1")
            aborted = True
    free = eng.scheduler.kv_manager.kv_cache_allocator.num_free()
    stats = dict(eng.pp_stats)
    eng.shutdown()
    return got, want, sorted(finished), stats, free


def _engine_worker(rank, port, q, pp):
    try:
        q.put((rank, _run_engine(pp, rank, port), None))
    except Exception:   # noqa: BLE001
        q.put((rank, None, traceback.format_exc()))


def test_pipeline_engine_keeps_batches_in_flight_and_ranks_agree():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, port, q, 2)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        rank, out, err = q.get(timeout=240)
        assert err is None, f"rank {rank} failed:\n{err}"
        results[rank] = out
    for p in procs:
        p.join(timeout=60)
    got0, want, fin0, stats0, free0 = results[0]
    got1, _, fin1, stats1, free1 = results[1]
    assert got0 == got1 and fin0 == fin1 and free0 == free1         # every rank saw the same tokens and made the same decisions
    for k, n in want.items():
        if k == "q1":
            assert 1 <= len(got0[k]) <= 2 and k not in fin0         # aborted: at most the step that was in flight came back
        else:
            assert len(got0[k]) == n and k in fin0
    # both stages worked on most ticks: batches really overlapped (one batch at a time would give 50 %)
    for st in (stats0, stats1):
        assert st["stage_busy_ticks"] >= 0.75 * st["ticks"], st


# -- RCCL inside the captured decode graph, on the one GPU of the test box -------------------------------
def _run_forced(rank, port, q, eager):
    """One rank with NVX_FORCE_COLLECTIVES=1: a one-rank RCCL group, the tensor-parallel code path (bf16
    row-parallel outputs through all-reduce, masked embedding, LM-head gather, id broadcast)."""
    try:
        os.environ["NVX_FORCE_COLLECTIVES"] = "1"
        import torch.distributed as dist

        from nano_vllm_x_amd import distributed as tpd
        from nano_vllm_x_amd.core.common import ForwardBatch, ForwardMode, SamplingParams, Sequence
        from nano_vllm_x_amd.core.model_runner import ModelRunner
        from nano_vllm_x_amd.qwen3 import Qwen3Config

        counts = {"all_reduce": 0, "all_gather": 0, "capturing": 0}
        real_ar, real_ag = dist.all_reduce, dist.all_gather_into_tensor

        def ar(t, *a, **k):
            counts["all_reduce"] += 1
            counts["capturing"] += bool(t.is_cuda and torch.cuda.is_current_stream_capturing())
            return real_ar(t, *a, **k)

        def ag(o, t, *a, **k):
            counts["all_gather"] += 1
            return real_ag(o, t, *a, **k)

        dist.all_reduce, dist.all_gather_into_tensor = ar, ag
        r = ModelRunner(model="forced", max_bs=8, tp_size=1, nccl_port=port, enforce_eager=eager, context_len=128,
                        config=Qwen3Config.from_dict(CFG), seed=5)
        assert tpd.collectives_active() and dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        r.load_model()
        r.initialize_kv_cache(512)     # captures the decode graphs unless eager
        captured = counts["capturing"]
        rng = np.random.default_rng(9)
        seqs, slot = [], 0
        for i, n in enumerate(PROMPT_LENS):
            p = rng.integers(0, CFG["vocab_size"], size=n).tolist()
            seqs.append(Sequence(seq_id=f"s{i}", token_ids=p, num_tokens=n, prompt_len=n, kv_indices=list(range(slot, slot + n)),
                                 sampling_params=SamplingParams(temperature=1.0, top_k=1, ignore_eos=True, max_new_tokens=8)))
            slot += n + 8
        logits = []
        for step in range(3):
            mode = ForwardMode.PREFILL if step == 0 else ForwardMode.DECODE
            lg = r.compute_logits(ForwardBatch(forward_mode=mode, num_seqs=len(seqs), seqs=seqs))
            logits.append(lg.float().cpu().numpy())
            for s in seqs:
                s.cached_kv_len = len(s.kv_indices)
                s.token_ids.append(int(rng.integers(0, CFG["vocab_size"])))
                s.kv_indices.append(s.kv_indices[-1] + 1)
        stats = dict(r.stats)
        r.shutdown()
        q.put((eager, (logits, stats, dict(counts), captured), None))
    except Exception:   # noqa: BLE001
        q.put((eager, None, traceback.format_exc()))


def test_rccl_collectives_are_captured_in_the_decode_graph_on_one_rank():
    """The decode hipGraph of a tensor-parallel rank holds RCCL all-reduces (two per layer).  With one GPU the
    only RCCL group that can exist has one rank; NVX_FORCE_COLLECTIVES=1 makes the model issue every collective
    on it.  Capture must succeed with all-reduce calls made while the stream is capturing, the replayed graph
    must give the eager steps' logits bit for bit, and both must agree with the ordinary one-rank model (which
    sums its row-parallel projections in fp32 inside the next norm instead of rounding them to bf16 first)."""
    ctx = mp.get_context("spawn")
    out = {}
    for eager in (False, True):        # one after the other: each is a process with its own one-rank group
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = ctx.Queue()
        p = ctx.Process(target=_run_forced, args=(0, port, q, eager))
        p.start()
        key, res, err = q.get(timeout=240)
        p.join(timeout=60)
        assert err is None, f"forced-collectives run (eager={eager}) failed:\n{err}"
        out[key] = res
    g_logits, g_stats, g_counts, g_captured = out[False]
    e_logits, e_stats, e_counts, _ = out[True]
    layers = CFG["num_hidden_layers"]
    assert g_captured >= 2 * layers, f"only {g_captured} all-reduce calls were made under stream capture"
    assert g_stats["graph_steps"] == 2 and g_stats["eager_decode_steps"] == 0 and e_stats["graph_steps"] == 0
    assert e_counts["all_reduce"] >= 3 * (2 * layers + 1) and e_counts["all_gather"] >= 3
    for step in range(3):
        assert np.array_equal(g_logits[step], e_logits[step]), f"step {step}: graph replay differs from eager launches"
    want, _, _ = _run_model(1, 0, 0)
    for step in range(3):
        check_logits("forced_collectives_vs_one_rank", g_logits[step], want[step], f"step {step}")

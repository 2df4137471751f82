"""Pipeline stages on CPU (gloo): rank layout, stage-to-stage transfers, layer ranges, per-stage weights.

The device kernels need a GPU, so the checks here are on what pipeline parallelism adds around them:
world = tp * pp with rank = pp_rank * tp + tp_rank and one TP group per stage, the (hidden, residual)
hand-over between neighbouring stages (1/tp slices + all-gather), the broadcast of the sampled ids from
the last stage, and the slice of the module graph and of a checkpoint that each stage keeps.
Reference: nanovllmx/core/worker.py:40-41,84-104, distributed/communication_op.py:236-346,
models/qwen3.py:191-206.
"""
from __future__ import annotations

import os
import socket
import sys
import traceback

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CFG = dict(vocab_size=200, hidden_size=64, intermediate_size=96, num_hidden_layers=3, num_attention_heads=4,
           num_key_value_heads=2, head_dim=128, max_position_embeddings=64, tie_word_embeddings=False)


def _worker(rank: int, port: int, tp: int, pp: int, fn_name: str, q):
    try:
        torch.set_num_threads(1)
        from nano_vllm_x_amd import distributed as tpd

        tpd.init_tensor_parallel(tp, rank=rank, backend="gloo", port=port, pp_size=pp)
        with torch.no_grad():
            globals()[fn_name](rank, tp, pp)
        tpd.barrier()
        tpd.shutdown()
        q.put((rank, None))
    except Exception:   # noqa: BLE001 - reported to the parent
        q.put((rank, traceback.format_exc()))


def _run(fn_name: str, tp: int, pp: int):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, port, tp, pp, fn_name, q)) for r in range(tp * pp)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, err in results:
        assert err is None, f"rank {rank} failed:\n{err}"


def _ref_pp_indices(num_layers, pp_rank, pp_size):
    """nanovllmx/distributed/utils.py:10-15, restated: floor(n/pp) layers each, the last stage the rest."""
    per = num_layers // pp_size
    return pp_rank * per, (num_layers if pp_rank == pp_size - 1 else (pp_rank + 1) * per)


def test_stage_layer_ranges_cover_every_layer_once():
    from nano_vllm_x_amd.qwen3 import stage_layer_range

    for layers in (1, 2, 3, 11, 28, 36, 64, 65):
        for pp in (1, 2, 3, 4, 8):
            if pp > layers:
                continue
            for balanced in (False, True):
                spans = [stage_layer_range(layers, r, pp, balanced=balanced) for r in range(pp)]
                assert spans[0][0] == 0 and spans[-1][1] == layers
                assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
                sizes = [e - s for s, e in spans]
                assert sizes == sorted(sizes)   # the first stage (embedding) is never the larger
                if balanced:
                    assert max(sizes) - min(sizes) <= 1
                else:   # the default is the reference's split, stage by stage
                    assert spans == [_ref_pp_indices(layers, r, pp) for r in range(pp)]
    assert [stage_layer_range(11, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 11)]
    assert [stage_layer_range(11, r, 3, balanced=True) for r in range(3)] == [(0, 3), (3, 7), (7, 11)]
    assert stage_layer_range(64, 1, 2) == (32, 64)


# ---------------------------------------------------------------------------------------------
def _case_transfers(rank: int, tp: int, pp: int):
    from nano_vllm_x_amd import distributed as tpd

    assert (tpd.tp_rank(), tpd.pp_rank(), tpd.tp_size(), tpd.pp_size()) == (rank % tp, rank // tp, tp, pp)
    assert tpd.pp_is_first() == (rank < tp) and tpd.pp_is_last() == (rank >= (pp - 1) * tp)
    # the TP group is the stage's: a sum over it sees only the stage's ranks
    x = tpd.tp_all_reduce(torch.tensor([float(rank)]))
    stage = rank // tp
    assert x.item() == sum(range(stage * tp, (stage + 1) * tp))
    # hand-over down the pipeline: each stage adds its number; odd sizes travel whole, even ones in 1/tp slices
    for shape in ((2, 5, 64), (3, 7)):
        if tpd.pp_is_first():
            t = torch.arange(float(torch.Size(shape).numel())).view(shape).to(torch.bfloat16)
        else:
            t = tpd.pp_recv_prev(shape, torch.bfloat16, "cpu")
        if not tpd.pp_is_last():
            tpd.pp_send_next(t + 1)
        want = torch.arange(float(torch.Size(shape).numel())).view(shape).to(torch.bfloat16)
        for _ in range(stage):
            want = want + 1
        assert torch.equal(t, want), (rank, shape)
    # sampled ids: from TP rank 0 of the last stage to everyone
    ids = torch.tensor([9, 8, 7]) if rank == (pp - 1) * tp else torch.zeros(3, dtype=torch.int64)
    assert tpd.broadcast_sampled_ids(ids).tolist() == [9, 8, 7]
    assert tpd.all_ranks_min_int(500 + rank, "cpu") == 500
    # broadcast inside the stage takes a TP rank as source
    y = tpd.tp_broadcast(torch.tensor([rank]), src=0)
    assert y.item() == stage * tp


@pytest.mark.parametrize("tp,pp", [(1, 2), (2, 2), (1, 3)])
def test_rank_layout_and_stage_transfers(tp, pp):
    _run("_case_transfers", tp, pp)


def _checkpoint(cfg):
    g = torch.Generator().manual_seed(3)
    h, d = cfg.hidden_size, cfg.head_dim
    w = {"model.embed_tokens.weight": torch.randn(cfg.vocab_size, h, generator=g), "model.norm.weight": torch.randn(h, generator=g),
         "lm_head.weight": torch.randn(cfg.vocab_size, h, generator=g)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        w[p + "self_attn.q_proj.weight"] = torch.randn(cfg.num_attention_heads * d, h, generator=g)
        w[p + "self_attn.k_proj.weight"] = torch.randn(cfg.num_key_value_heads * d, h, generator=g)
        w[p + "self_attn.v_proj.weight"] = torch.randn(cfg.num_key_value_heads * d, h, generator=g)
        w[p + "self_attn.o_proj.weight"] = torch.randn(h, cfg.num_attention_heads * d, generator=g)
        w[p + "self_attn.q_norm.weight"] = torch.randn(d, generator=g)
        w[p + "self_attn.k_norm.weight"] = torch.randn(d, generator=g)
        w[p + "mlp.gate_proj.weight"] = torch.randn(cfg.intermediate_size, h, generator=g)
        w[p + "mlp.up_proj.weight"] = torch.randn(cfg.intermediate_size, h, generator=g)
        w[p + "mlp.down_proj.weight"] = torch.randn(h, cfg.intermediate_size, generator=g)
        w[p + "input_layernorm.weight"] = torch.randn(h, generator=g)
        w[p + "post_attention_layernorm.weight"] = torch.randn(h, generator=g)
    return w


def _case_stage_weights(rank: int, tp: int, pp: int):
    from nano_vllm_x_amd import distributed as tpd
    from nano_vllm_x_amd.qwen3 import MissingLayer, Qwen3Config, Qwen3ForCausalLM

    for tied in (False, True):
        cfg = Qwen3Config.from_dict(dict(CFG, tie_word_embeddings=tied))
        model = Qwen3ForCausalLM(cfg)
        model.finalize()
        w = _checkpoint(cfg)
        model.load_weights(w.items())
        m = model.model
        first, last = tpd.pp_is_first(), tpd.pp_is_last()
        assert (m.start_layer, m.end_layer) == ((0, 1), (1, 3))[tpd.pp_rank()]      # 3 layers over 2 stages: 1 + 2
        for i, layer in enumerate(m.layers):
            assert isinstance(layer, MissingLayer) == (not m.start_layer <= i < m.end_layer)
        assert isinstance(m.norm, MissingLayer) == (not last)
        assert isinstance(model.lm_head, MissingLayer) == (not last)
        assert isinstance(m.embed_tokens, MissingLayer) == (not (first or (last and tied)))
        names = {n for n, _ in model.named_parameters()}
        assert all(int(n.split("layers.")[1].split(".")[0]) in range(m.start_layer, m.end_layer) for n in names if "layers." in n)
        # the stage's tensors hold the checkpoint's values (TP shard of them)
        r, t = tpd.tp_rank(), tp
        for i in range(m.start_layer, m.end_layer):
            p = f"model.layers.{i}."
            a = m.layers[i].self_attn
            nq = a.num_heads * 128
            assert torch.equal(a.qkv_proj.weight[:nq], w[p + "self_attn.q_proj.weight"][r * nq:(r + 1) * nq])
            per = cfg.intermediate_size // t
            assert torch.equal(m.layers[i].mlp.down_proj.weight, w[p + "mlp.down_proj.weight"][:, r * per:(r + 1) * per])
            assert torch.equal(m.layers[i].input_layernorm.weight, w[p + "input_layernorm.weight"])
        if last:
            assert torch.equal(m.norm.weight, w["model.norm.weight"])
            src = w["model.embed_tokens.weight" if tied else "lm_head.weight"]
            per = model.lm_head.weight.shape[0]
            rows = src[r * per:(r + 1) * per]
            assert torch.equal(model.lm_head.weight[:rows.shape[0]], rows)
        if first:
            per = m.embed_tokens.weight.shape[0]
            rows = w["model.embed_tokens.weight"][r * per:(r + 1) * per]
            assert torch.equal(m.embed_tokens.weight[:rows.shape[0]], rows)


@pytest.mark.parametrize("tp,pp", [(1, 2), (2, 2)])
def test_each_stage_keeps_its_slice_of_the_model(tp, pp):
    _run("_case_stage_weights", tp, pp)


# -- tick bookkeeping of the engine, with a stand-in for the device side ---------------------------
class _FakeRunner:
    """Records which batch each tick gives this stage and which batch's ids come back; the "sampled" id
    of a sequence is a function of its length, so every stage's replica of the scheduler agrees."""

    def __init__(self):
        self.ran, self.retired = [], []

    def launch_tick(self, mine, out):
        self.ran.append(None if mine is None else tuple(s.seq_id for s in mine.seqs))
        self.retired.append(None if out is None else tuple(s.seq_id for s in out.seqs))
        return (out, None, 0.0)

    def finish_step(self, handle):
        return [] if handle[0] is None else [100 + len(s.token_ids) for s in handle[0].seqs]

    def release_sequence(self, seq):
        pass


def _fake_engine(pp_rank: int, pp: int, monkeypatch):
    from collections import deque

    from nano_vllm_x_amd import distributed as tpd
    from nano_vllm_x_amd.core.engine import Engine, EngineCore

    monkeypatch.setattr(tpd, "_PP_RANK", pp_rank)
    monkeypatch.setattr(tpd, "_PP_SIZE", pp)
    eng = Engine.__new__(Engine)
    EngineCore.__init__(eng, kv_cache_size=4096, max_bs=2, context_len=256)
    eng.model_runner = _FakeRunner()
    eng._inflight, eng.pp_size, eng._pp_window, eng._pp_aborts = None, pp, deque(maxlen=pp), []
    eng.run_ahead, eng._ahead, eng._placeholders, eng._late_aborts, eng.ahead_steps, eng.ahead_retired = False, None, None, [], 0, 0
    eng.pp_stats = {"ticks": 0, "stage_busy_ticks": 0}
    return eng


@pytest.mark.parametrize("pp", [2, 3])
def test_every_stage_runs_every_batch_once_in_schedule_order(pp, monkeypatch):
    from nano_vllm_x_amd.core.common import SamplingParams

    logs = []
    for r in range(pp):
        eng = _fake_engine(r, pp, monkeypatch)
        for i in range(7):
            eng.add_sequence(f"s{i}", list(range(3 + i)), SamplingParams(ignore_eos=True, max_new_tokens=2 + i % 3))
        outs = []
        for _ in range(200):
            if not (eng.has_unfinished_sequences() or eng.step_in_flight):
                break
            outs.extend((o.seq_id, o.new_token_id, o.is_finished) for o in eng.step())
            if len(outs) == 3:
                eng.abort_sequence("s5")        # in flight or not, the stages must still agree
        assert not eng.has_unfinished_sequences() and not eng.step_in_flight
        run = eng.model_runner
        logs.append((outs, run.ran, run.retired))
        done = [b for b in run.retired if b is not None]
        assert [b for b in run.ran if b is not None] == done               # each batch once, in retirement (= schedule) order
        # stage r sees a batch r ticks after it was scheduled, and the ids pp-1 ticks after
        first = next(i for i, b in enumerate(run.ran) if b is not None)
        assert first == r and next(i for i, b in enumerate(run.retired) if b is not None) == pp - 1
        assert sum(b is not None for b in run.ran) >= 0.6 * len(run.ran)     # the stages overlap
    for r in range(1, pp):
        assert logs[r][0] == logs[0][0]                       # same tokens, same finish decisions on every stage
        assert logs[r][2] == logs[0][2]
        # the stage's own sequence of batches is stage 0's, shifted by r ticks
        assert [b for b in logs[r][1] if b is not None] == [b for b in logs[0][1] if b is not None]


def test_draining_moves_batches_in_flight_through_every_stage_without_admitting(monkeypatch):
    from nano_vllm_x_amd.core.common import SamplingParams

    for r in range(3):
        eng = _fake_engine(r, 3, monkeypatch)
        for i in range(6):
            eng.add_sequence(f"s{i}", list(range(4 + i)), SamplingParams(ignore_eos=True, max_new_tokens=50))
        for _ in range(5):
            eng.step()
        run = eng.model_runner
        scheduled_so_far = 5                                    # one batch per tick: max_bs 2, six sequences
        while eng._begin_tick(admit=False):
            eng.end_step()
        assert not eng.step_in_flight
        assert sum(b is not None for b in run.ran) == scheduled_so_far      # nothing new was admitted ...
        assert [b for b in run.ran if b is not None] == [b for b in run.retired if b is not None]   # ... and nothing was left behind

"""Block-table metadata of the PRODUCT against the reference's own tensors (tests/golden/g3_metadata.json,
recorded from nanovllmx/layers/attention.py:62-149 eager and :151-295 graph capture / replay).

Three paths are pinned, all integer work and bit-exact, dtype included:
  * AttentionBackend.build_metadata: the tensors uploaded to the device and the keyword set handed to plan();
  * the reference's static-buffer protocol (prepare_for_cuda_graph_capture / build_metadata_for_cuda_graph_capture
    / build_metadata_for_cuda_graph_replay);
  * the engine's own replay path -- a device-resident table updated by nvx_block_table_write_rows and
    nvx_decode_step_setup -- which must describe the same rows the reference's replay buffers describe.
"""
import numpy as np
import pytest
import torch

from helpers import golden_json
from oracle.attention_oracle import block_table_replay_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_TORCH_DTYPE = {"torch.int32": torch.int32, "torch.int64": torch.int64}


def _seqs(rows, full=True):
    """Sequence objects of a fixture case; full=False: as they stood before the step's new slots were added."""
    from nano_vllm_x_amd.core.common import Sequence

    out = []
    for i, s in enumerate(rows):
        kv = list(s["kv"]) if full else list(s["kv"][:s["cached"]])
        out.append(Sequence(seq_id=f"g{i}", token_ids=[1] * s["ntok"], num_tokens=s["ntok"], prompt_len=s["ntok"],
                            kv_indices=kv, cached_kv_len=s["cached"]))
    return out


def _backend(hq=4, hkv=2, slots=128):
    from nano_vllm_x_amd.core.kv_cache import KVCachePool
    from nano_vllm_x_amd.layers.attention import AttentionBackend

    pool = KVCachePool(dtype=torch.bfloat16, device=torch.device(DEV), num_tokens=slots, num_layers=1, num_heads=hkv, head_dim=128)
    return AttentionBackend(kv_cache=pool, num_qo_heads=hq, num_kv_heads=hkv, head_dim=128, dtype=torch.bfloat16,
                            device=torch.device(DEV))


class _Recorder:
    """Stands in front of a wrapper's plan(): keeps the keyword arguments, then plans for real."""

    def __init__(self, wrapper):
        self.wrapper, self.kw, self._plan = wrapper, None, wrapper.plan
        wrapper.plan = self

    def __call__(self, **kw):
        self.kw = kw
        return self._plan(**kw)


def test_build_metadata_uploads_the_reference_tensors():
    from nano_vllm_x_amd.core.common import ForwardBatch, ForwardMode

    g = golden_json("g3_metadata.json")
    assert len(g["cases"]) == 4
    for case in g["cases"]:
        be = _backend()
        rec_p, rec_d = _Recorder(be.prefill_wrapper), _Recorder(be.decode_wrapper)
        mode = ForwardMode[case["mode"]]
        seqs = _seqs(case["seqs"])
        md = be.build_metadata(ForwardBatch(forward_mode=mode, num_seqs=len(seqs), seqs=seqs))
        torch.cuda.synchronize()
        rec = rec_p if mode == ForwardMode.PREFILL else rec_d
        assert rec.kw is not None and (rec_d if rec is rec_p else rec_p).kw is None, case["name"]
        assert md.forward_mode == mode and md.wrapper is rec.wrapper and md.kv_cache is be.kv_cache
        plan = case["plan"]
        # the keyword SET is the duck type of the wrapper slot (attention.py:106-119 prefill, :122-132 decode)
        assert set(rec.kw) == set(plan), (case["name"], set(rec.kw) ^ set(plan))
        for name, want in plan.items():
            got = rec.kw[name]
            if isinstance(want, list):
                assert got.is_cuda and got.dtype == _TORCH_DTYPE[case["plan_dtypes"][name]], (case["name"], name, got.dtype)
                assert got.cpu().tolist() == want, (case["name"], name)
            elif name in ("q_data_type", "kv_data_type"):
                assert str(got) == want
            elif name.startswith("head_dim"):
                assert got == 128      # the fixture's toy head size is 8; the kernels are built for 128
            else:
                assert got == want, (case["name"], name)
        assert md.output_kv_indices.is_cuda and md.output_kv_indices.dtype == _TORCH_DTYPE[case["output_kv_indices_dtype"]]
        assert md.output_kv_indices.cpu().tolist() == case["output_kv_indices"]


def test_graph_buffer_protocol_matches_the_reference_buffers():
    """attention.py:151-168 (static buffers), :171-217 (capture-time metadata), :219-295 (replay fill)."""
    from nano_vllm_x_amd.core.common import ForwardBatch, ForwardMode
    from nano_vllm_x_amd.core.hip_graph import HIPGraph

    g = golden_json("g3_metadata.json")
    r = g["replay"]
    max_bs, ctx, padded = r["max_bs"], r["context_len"], r["padded_bs"]
    be = _backend(slots=max_bs * ctx)
    graph = HIPGraph()
    be.prepare_for_cuda_graph_capture(graph, max_bs=max_bs, context_len=ctx)
    for name, dt in (("paged_kv_indptr_buffer", torch.int32), ("paged_kv_indices_buffer", torch.int32),
                     ("paged_kv_last_page_len_buffer", torch.int32), ("output_kv_indices_buffer", torch.int64)):
        assert graph.get_input_buffer(name).dtype == dt and graph.get_input_buffer(name).is_cuda
    assert graph.get_input_buffer("paged_kv_indices_buffer").numel() == max_bs * ctx
    for bs_s, want in g["capture"].items():
        bs = int(bs_s)
        md = be.build_metadata_for_cuda_graph_capture(graph, bs, ctx)
        w = be.decode_wrappers_cuda_graph[bs]
        assert md.wrapper is w and md.forward_mode == ForwardMode.DECODE
        assert graph.get_input_buffer("paged_kv_indptr_buffer")[:bs + 1].cpu().tolist() == want["indptr"]
        assert graph.get_input_buffer("paged_kv_last_page_len_buffer")[:bs].cpu().tolist() == want["last_page_len"]
        assert bs * ctx == want["indices_len"] and md.output_kv_indices.numel() == want["output_kv_indices_len"]
    seqs = _seqs(r["seqs"])
    rec = _Recorder(be.decode_wrappers_cuda_graph[padded])
    md = be.build_metadata_for_cuda_graph_replay(graph, ForwardBatch(forward_mode=ForwardMode.DECODE, num_seqs=len(seqs), seqs=seqs), padded)
    torch.cuda.synchronize()
    assert md.wrapper is be.decode_wrappers_cuda_graph[padded]
    assert graph.get_input_buffer("paged_kv_indptr_buffer").cpu().tolist() == r["buf_indptr"]
    assert graph.get_input_buffer("paged_kv_indices_buffer")[:20].cpu().tolist() == r["buf_indices_head"]
    assert graph.get_input_buffer("output_kv_indices_buffer").cpu().tolist() == r["buf_output_kv_indices"]
    assert rec.kw["indptr"].cpu().tolist() == r["indptr"] and rec.kw["indptr"].dtype == torch.int32
    assert rec.kw["indices"].cpu().tolist() == r["indices"] and rec.kw["indices"].dtype == torch.int32
    assert rec.kw["last_page_len"].cpu().tolist() == r["last_page_len"]
    assert md.output_kv_indices.cpu().tolist() == r["output_kv_indices"] and md.output_kv_indices.dtype == torch.int64
    # and the oracle's restatement of the same fill agrees with what is now on the device, whole buffers
    rows = [(s["ntok"], s["kv"], s["cached"]) for s in r["seqs"]]
    t = block_table_replay_oracle(rows, padded, max_bs, ctx)
    assert np.array_equal(graph.get_input_buffer("paged_kv_indptr_buffer").cpu().numpy(), t["indptr_buf"])
    assert np.array_equal(graph.get_input_buffer("paged_kv_indices_buffer").cpu().numpy(), t["indices_buf"])
    assert np.array_equal(graph.get_input_buffer("output_kv_indices_buffer").cpu().numpy(), t["out_slots_buf"])


def test_graph_protocol_wrapper_attends_like_the_eager_wrapper():
    """The wrapper planned through the static buffers runs the same kernel over the same rows: its output for
    the live rows equals the eager wrapper's, bit for bit; padding rows attend to slot 0 only."""
    from nano_vllm_x_amd.core.common import ForwardBatch, ForwardMode
    from nano_vllm_x_amd.core.hip_graph import HIPGraph

    r = golden_json("g3_metadata.json")["replay"]
    max_bs, ctx, padded = r["max_bs"], r["context_len"], r["padded_bs"]
    be = _backend(slots=max_bs * ctx)
    gen = torch.Generator().manual_seed(5)
    be.kv_cache.k_cache.copy_(torch.randn(be.kv_cache.k_cache.shape, generator=gen).to(torch.bfloat16))
    be.kv_cache.v_cache.copy_(torch.randn(be.kv_cache.v_cache.shape, generator=gen).to(torch.bfloat16))
    graph = HIPGraph()
    be.prepare_for_cuda_graph_capture(graph, max_bs=max_bs, context_len=ctx)
    be.build_metadata_for_cuda_graph_capture(graph, padded, ctx)
    seqs = _seqs(r["seqs"])
    batch = ForwardBatch(forward_mode=ForwardMode.DECODE, num_seqs=len(seqs), seqs=seqs)
    md_g = be.build_metadata_for_cuda_graph_replay(graph, batch, padded)
    q = torch.randn(padded, 4, 128, generator=gen).to(torch.bfloat16).to(DEV)
    kv = be.kv_cache.get_kv_cache(0)
    out_g = md_g.wrapper.run(q, kv).clone()
    md_e = be.build_metadata(batch)
    out_e = md_e.wrapper.run(q[:len(seqs)].contiguous(), kv)
    torch.cuda.synchronize()
    assert torch.equal(out_g[:len(seqs)], out_e)
    v0 = kv[1][0].float()                                    # slot 0: (Hkv, 128); one key => output = its value row
    want_pad = v0.repeat_interleave(2, dim=0).to(torch.bfloat16)
    assert torch.equal(out_g[len(seqs)].cpu(), want_pad.cpu())


def test_device_block_table_replay_describes_the_same_rows():
    """The engine's replay path: rows written after the prefill (nvx_block_table_write_rows), then one decode
    step staged and applied on the device (nvx_decode_step_setup).  Row by row it must list the slots the
    reference's replay buffers list, with the same lengths and the same scatter targets."""
    from nano_vllm_x_amd import ops
    from nano_vllm_x_amd.layers.attention import BlockTable, block_table_from_batch

    r = golden_json("g3_metadata.json")["replay"]
    max_bs, ctx, padded = r["max_bs"], r["context_len"], r["padded_bs"]
    rows = [(s["ntok"], s["kv"], s["cached"]) for s in r["seqs"]]
    want = block_table_replay_oracle(rows, padded, max_bs, ctx)
    bt = BlockTable(max_bs, n_rows=6, row_capacity=ctx, device=torch.device(DEV))
    before = _seqs(r["seqs"], full=False)       # after their prefill: rows for the cached tokens only
    t = block_table_from_batch(before)
    bt.write_rows(before, torch.from_numpy(t["paged_kv_indices"]).to(DEV), torch.from_numpy(t["paged_kv_indptr"]).to(DEV))
    now = _seqs(r["seqs"])                       # the scheduler has added this step's slot to every sequence
    bt.stage_decode_step(now)
    ops.decode_step_setup(bt.step_dev, bt.max_bs, padded, bt.table, bt.pad_start, bt.kv_start, bt.kv_len, bt.out_slots,
                          bt.positions, bt.input_ids)
    torch.cuda.synchronize()
    table = bt.table.cpu().view(-1)
    indptr = want["indptr_buf"][:padded + 1]
    for i in range(padded):
        n = int(indptr[i + 1] - indptr[i])
        assert int(bt.kv_len[i]) == n
        start = int(bt.kv_start[i])
        assert table[start:start + n].tolist() == want["indices_buf"][indptr[i]:indptr[i + 1]].tolist(), i
    assert bt.out_slots[:padded].cpu().tolist() == want["out_slots_buf"][:padded].tolist() == r["output_kv_indices"]
    assert bt.out_slots.dtype == torch.int64 and bt.kv_len.dtype == torch.int32 and bt.table.dtype == torch.int32
    assert bt.positions[:padded].cpu().tolist() == [s["ntok"] - 1 for s in r["seqs"]] + [0] * (padded - len(rows))
    assert [int(bt.kv_start[i]) for i in range(len(rows), padded)] == [bt.pad_start] * (padded - len(rows))


def test_block_table_rows_are_decided_per_sequence():
    """A chunk view gets no row; a sequence that cannot have one (no free row, or a list longer than a row)
    loses any row it held, so that fits() is False and its decode steps run from a CSR table."""
    from nano_vllm_x_amd.core.common import Sequence
    from nano_vllm_x_amd.layers.attention import BlockTable, block_table_from_batch

    bt = BlockTable(4, n_rows=3, row_capacity=8, device=torch.device(DEV))      # two usable rows + the padding row

    def seq(name, n):
        return Sequence(seq_id=name, token_ids=[1] * n, num_tokens=n, prompt_len=n, kv_indices=list(range(100, 100 + n)))

    def write(seqs):
        t = block_table_from_batch(seqs)
        bt.write_rows(seqs, torch.from_numpy(t["paged_kv_indices"]).to(DEV), torch.from_numpy(t["paged_kv_indptr"]).to(DEV))

    a, b, c = seq("a", 5), seq("b", 3), seq("c", 4)
    view = Sequence(seq_id="b", token_ids=[1] * 2, num_tokens=2, prompt_len=3, kv_indices=b.kv_indices[:2])
    view.partial_of = b
    write([a, view])
    assert "a" in bt.row_of and "b" not in bt.row_of          # the chunk view took no row
    write([b, c])                                              # last chunk of b, and c: one free row left
    assert "b" in bt.row_of and "c" not in bt.row_of and not bt.fits([c]) and bt.fits([a, b])
    torch.cuda.synchronize()
    assert bt.table[bt.row_of["b"], :3].cpu().tolist() == b.kv_indices
    a.kv_indices.extend(range(200, 204))                       # 9 slots: longer than a row
    write([a])
    assert "a" not in bt.row_of and len(bt.free_rows) == 1
    write([c])                                                 # the freed row goes to the next sequence that asks
    assert bt.fits([b, c])

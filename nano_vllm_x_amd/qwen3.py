"""Qwen3 decoder stack: the caller that hosts the hot-path ops.

Op order, shard rules and weight-name mapping follow nanovllmx/models/qwen3.py (attention block
:117-136, decoder layer :167-188, stack :249-268, logits :295-300, stacked weights :310-348) and
the parallel linears of nanovllmx/layers/linear.py (:387-408 qkv shards, :488-498 merged-column
shards) and vocab_parallel_embedding.py (:93-108 masked embedding + all-reduce, :147-151 LM head
all-gather).  Dense projections stay library GEMMs (torch -> hipBLASLt); everything between them
is a HIP kernel of this package: RMSNorm(+residual), fused q/k-norm + rotary + KV scatter, paged
attention, SiluAndMul.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, fields
from typing import Iterable, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import distributed as tpd
from . import ops
from .layers.activation import SiluAndMul
from .core.common import ForwardMode
from .layers.attention import Attention
from .layers.layernorm import RMSNorm
from .layers.rotary_embedding import RotaryEmbedding, build_cos_sin_cache


@dataclass
class Qwen3Config:
    vocab_size: int = 151936
    hidden_size: int = 1024
    intermediate_size: int = 3072
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    max_position_embeddings: int = 40960
    rope_theta: float = 1000000.0
    tie_word_embeddings: bool = True
    torch_dtype: str = "bfloat16"

    @classmethod
    def from_dict(cls, d: dict) -> "Qwen3Config":
        known = {f.name for f in fields(cls)}
        kw = {k: v for k, v in d.items() if k in known}
        if "torch_dtype" not in kw and "dtype" in d:
            kw["torch_dtype"] = d["dtype"]
        if "head_dim" not in kw and "hidden_size" in kw and "num_attention_heads" in kw:
            kw["head_dim"] = kw["hidden_size"] // kw["num_attention_heads"]
        return cls(**kw)


# public model-card geometry (no config.json is reachable offline)
PRESETS = {
    "Qwen3-0.6B": dict(hidden_size=1024, intermediate_size=3072, num_hidden_layers=28, num_attention_heads=16,
                       num_key_value_heads=8, tie_word_embeddings=True),
    "Qwen3-8B": dict(hidden_size=4096, intermediate_size=12288, num_hidden_layers=36, num_attention_heads=32,
                     num_key_value_heads=8, tie_word_embeddings=False),
    "Qwen3-32B": dict(hidden_size=5120, intermediate_size=25600, num_hidden_layers=64, num_attention_heads=64,
                      num_key_value_heads=8, tie_word_embeddings=False),
}


# NVX_EXTRA_PRESETS: a JSON object {name: {field: value}} of further geometries by name -- how a test or a rehearsal gives every
# rank of a job (the variable is inherited by the ranks it starts) a small model without a directory on disk
PRESETS.update(json.loads(os.environ.get("NVX_EXTRA_PRESETS") or "{}"))


def load_config(model: str) -> Qwen3Config:
    """`model` is a directory holding config.json, or the name of a preset (possibly as the last
    path component, e.g. "../Qwen3-0.6B" as the reference's bench.py passes it)."""
    cfg_file = os.path.join(model, "config.json")
    if os.path.isfile(cfg_file):
        with open(cfg_file) as f:
            return Qwen3Config.from_dict(json.load(f))
    name = os.path.basename(os.path.normpath(model))
    if name in PRESETS:
        return Qwen3Config(**PRESETS[name])
    raise FileNotFoundError(f"{cfg_file} not found and {name!r} is not one of {sorted(PRESETS)}")


# Largest decode batch for which the skinny projection kernel (ops.linear_decode) beats the library
# GEMM, per projection; measured on MI355X with the Qwen3-0.6B shapes (tools/kernel_bench.py linear).
# Above these, and for every prefill, the projection is a library GEMM.
DECODE_LINEAR_MAX_ROWS = {"qkv": 256, "o": 256, "gate_up": 256, "down": 256}
# ... and largest weight matrix.  The skinny kernel wins where a projection is latency-bound (a few MB of
# weights: one slice per workgroup, fetched in one go); a matrix of tens of MB is a bandwidth-bound stream,
# which the library GEMM sustains better (Qwen3-8B o_proj, 33 MB: 28 us here vs ~12 us of streaming).
DECODE_LINEAR_MAX_WEIGHT_BYTES = 16 << 20


# NVX_FP32_LINEAR=1 (read when a model is built): every dense projection, the LM head included, multiplies in fp32 and
# rounds its result to the model's 16-bit format once (the LM head's logits stay fp32).  Not a serving mode: it takes
# the library GEMMs' and the skinny kernels' summation order out of a comparison, so that what differs from an fp32
# restatement of the model is the hand-written path alone (norms, rotary, KV scatter, attention).
FP32_LINEAR = False


def linear_fp32(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    return F.linear(x.float(), weight.float()).to(x.dtype)


# The blocked projection (ops.linear_block: weight streamed from a tiled copy, the batch's rows through LDS in chunks of k)
# takes the MLP weights of a decode step when they are at least this large: gate/up (with SiluAndMul in its epilogue) up to
# BLOCK_GATEUP_MAX_ROWS rows, down (fp32 partials per k-slice for the RMSNorm that follows) from BLOCK_DOWN_MIN_ROWS to
# BLOCK_DOWN_MAX_ROWS rows, on a single rank (a tensor-parallel rank hands its partial result to the all-reduce in 16 bits).
# Qwen3-8B, us per layer with the weight coming from HBM (tools/kernel_bench.py block; library = GEMM + its consumer):
# gate/up 38.7 / 39.0 / 40.9 / 44.9 against 49.6 / 48.3 / 49.2 / 50.7 at 24 / 32 / 64 / 96 rows (55 against 54 at 128);
# down + norm 26 / 29 / 35 / 39 / 50 / 52 against 29 / 41 / 58 / 70 / 80 / 94 at 24 / 64 / 96 / 128 / 160 / 192 rows (below 17
# rows the library's 24 us stand; from 208 rows its 54-59 us do: there the products, not the stream, set this kernel's pace).  qkv and o (<= 50 MB) are level with the library once their consumer is counted and stay with it.
# NVX_LINEAR_BLOCK=0 keeps the library projections.
BLOCK_MIN_WEIGHT_BYTES = 64 << 20
BLOCK_GATEUP_MAX_ROWS = 96
BLOCK_DOWN_MIN_ROWS, BLOCK_DOWN_MAX_ROWS = 17, 192


def decode_linear(x: torch.Tensor, weight: torch.Tensor, kind: str, packed: Optional[torch.Tensor] = None, k_slices: int = 0) -> torch.Tensor:
    """Projection of a decode batch.  Returns bf16 (rows, n) -- or, for the row-parallel projections
    ("o", "down") on a single rank, fp32 partial products (splits, rows, n) that the RMSNorm which
    follows sums (under tensor parallelism they stay 16-bit: the sum over ranks comes first);
    "gate_up" returns SiluAndMul of the projection, (rows, n/2)."""
    if FP32_LINEAR:
        y = linear_fp32(x, weight)
        return ops.silu_mul(y) if kind == "gate_up" else y
    rows, k = x.shape
    n = weight.shape[0]
    use = (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and rows <= DECODE_LINEAR_MAX_ROWS[kind] and x.stride(1) == 1
           and n * k * 2 <= DECODE_LINEAR_MAX_WEIGHT_BYTES)
    if kind in ("o", "down"):
        if tpd.collectives_active():
            # a tensor-parallel rank hands its partial result to the all-reduce in the 16-bit format (linear.py:277-278);
            # its share of K is 1/tp of the full one, which the skinny kernel takes whole where it is one of these
            if use and k in (256, 512, 768, 1024, 1536) and n % 16 == 0:
                return ops.linear_decode(x, weight, mode=0)
            return F.linear(x, weight)
        splits = ops.linear_decode_splits(n, k) if use and n % 16 == 0 else 0
        if splits:
            return ops.linear_decode(x, weight, mode=2, k_splits=splits)
        if packed is not None and k_slices and BLOCK_DOWN_MIN_ROWS <= rows <= BLOCK_DOWN_MAX_ROWS and x.stride(1) == 1:
            return ops.linear_block(x, packed, n, mode=2, k_slices=k_slices)
        return F.linear(x, weight)
    fits = use and k in (512, 768, 1024)
    if kind == "gate_up":
        if fits and (n // 2) % 16 == 0:
            return ops.linear_decode(x, weight, mode=1)
        if packed is not None and rows <= BLOCK_GATEUP_MAX_ROWS and x.stride(1) == 1:
            return ops.linear_block(x, packed, n, mode=1)
        return ops.silu_mul(F.linear(x, weight))
    if fits and n % 16 == 0:
        return ops.linear_decode(x, weight, mode=0)
    return F.linear(x, weight)


def _shard(n: int, parts: int) -> int:
    assert n % parts == 0, f"{n} is not divisible by tensor-parallel size {parts}"
    return n // parts


class Qwen3Attention(nn.Module):
    def __init__(self, cfg: Qwen3Config, layer_id: int, rope_table: torch.Tensor):
        super().__init__()
        tp = tpd.tp_size()
        self.total_heads, self.total_kv_heads, self.head_dim = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        self.num_heads = _shard(self.total_heads, tp)
        if self.total_kv_heads >= tp:
            self.num_kv_heads, self.kv_replicas = _shard(self.total_kv_heads, tp), 1
        else:   # fewer kv heads than ranks: replicate (linear.py:333-338)
            self.num_kv_heads, self.kv_replicas = 1, _shard(tp, self.total_kv_heads)
        self.q_size, self.kv_size = self.num_heads * self.head_dim, self.num_kv_heads * self.head_dim
        self.eps = cfg.rms_norm_eps
        self.qkv_proj = nn.Linear(cfg.hidden_size, self.q_size + 2 * self.kv_size, bias=False)
        self.o_proj = nn.Linear(self.q_size, cfg.hidden_size, bias=False)
        self.q_norm = RMSNorm(self.head_dim, eps=cfg.rms_norm_eps)
        self.k_norm = RMSNorm(self.head_dim, eps=cfg.rms_norm_eps)
        self.attn = Attention(self.num_heads, self.head_dim, self.head_dim ** -0.5, num_kv_heads=self.num_kv_heads,
                              layer_id=layer_id)
        self.rope_table = rope_table     # shared fp32 (max_pos, 128) buffer, not a parameter
        self.rotary_emb: Optional[RotaryEmbedding] = None   # built on demand for the unfused path

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor, fused: bool = True) -> torch.Tensor:
        md = self.attn.attention_metadata
        if fused and md.forward_mode == ForwardMode.DECODE:
            qkv = decode_linear(hidden_states, self.qkv_proj.weight, "qkv")
            k_cache, v_cache = md.kv_cache.get_kv_cache(self.attn.layer_id)
            if getattr(md.wrapper, "qkv_fused", False):   # norm + rotary + scatter happen inside the attention kernel
                md.wrapper._sm_scale = self.attn.scaling
                attn_out = md.wrapper.run_qkv(qkv, positions, self.rope_table, self.q_norm.weight, self.k_norm.weight, self.eps,
                                              (k_cache, v_cache))
            else:
                q = ops.qknorm_rope_store(qkv, positions, self.rope_table, self.q_norm.weight, self.k_norm.weight, self.eps,
                                          md.output_kv_indices, k_cache, v_cache, self.num_heads, self.num_kv_heads)
                attn_out = self.attn(q, None, None)
            out = decode_linear(attn_out, self.o_proj.weight, "o")
            return out if out.dtype == torch.float32 else tpd.tp_reduce_later(out)
        qkv = linear_fp32(hidden_states, self.qkv_proj.weight) if FP32_LINEAR else self.qkv_proj(hidden_states)
        if fused:
            k_cache, v_cache = md.kv_cache.get_kv_cache(self.attn.layer_id)
            q = ops.qknorm_rope_store(qkv, positions, self.rope_table, self.q_norm.weight, self.k_norm.weight, self.eps,
                                      md.output_kv_indices, k_cache, v_cache, self.num_heads, self.num_kv_heads)
            attn_out = self.attn(q, None, None)
        else:   # op by op, as the reference's module graph runs it (qwen3.py:122-134)
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
            t = q.shape[0]
            q = self.q_norm(q.reshape(t, self.num_heads, self.head_dim)).view(t, self.q_size)
            k = self.k_norm(k.reshape(t, self.num_kv_heads, self.head_dim)).view(t, self.kv_size)
            q, k = ops.rope(positions, q, k, self.rope_table)
            attn_out = self.attn(q, k, v.contiguous())
        out = linear_fp32(attn_out, self.o_proj.weight) if FP32_LINEAR else self.o_proj(attn_out)
        return tpd.tp_reduce_later(out)


class Qwen3MLP(nn.Module):
    def __init__(self, cfg: Qwen3Config):
        super().__init__()
        inter = _shard(cfg.intermediate_size, tpd.tp_size())
        self.gate_up_proj = nn.Linear(cfg.hidden_size, 2 * inter, bias=False)
        self.down_proj = nn.Linear(inter, cfg.hidden_size, bias=False)
        self.act_fn = SiluAndMul()
        # tiled copies for ops.linear_block (large weights only; see BLOCK_MIN_WEIGHT_BYTES)
        self.gate_up_packed: Optional[torch.Tensor] = None
        self.down_packed: Optional[torch.Tensor] = None
        self.down_slices = 0

    def pack_decode_weights(self):
        if os.environ.get("NVX_LINEAR_BLOCK", "1") == "0":
            return

        def wants(w):
            return w.is_cuda and w.dtype in (torch.bfloat16, torch.float16) and w.numel() * w.element_size() >= BLOCK_MIN_WEIGHT_BYTES

        w = self.gate_up_proj.weight
        if wants(w) and ops.linear_block_slices(w.shape[0], w.shape[1], 1) > 0:
            self.gate_up_packed = ops.lmhead_pack(w.detach())
        w = self.down_proj.weight
        if wants(w) and not tpd.collectives_active():
            self.down_slices = ops.linear_block_slices(w.shape[0], w.shape[1], 2)
            if self.down_slices > 0:
                self.down_packed = ops.lmhead_pack(w.detach())

    def forward(self, x, decode: bool = False):
        if decode:
            out = decode_linear(decode_linear(x, self.gate_up_proj.weight, "gate_up", self.gate_up_packed), self.down_proj.weight, "down",
                                self.down_packed, self.down_slices)
            return out if out.dtype == torch.float32 else tpd.tp_reduce_later(out)
        if FP32_LINEAR:
            return tpd.tp_reduce_later(linear_fp32(self.act_fn(linear_fp32(x, self.gate_up_proj.weight)), self.down_proj.weight))
        return tpd.tp_reduce_later(self.down_proj(self.act_fn(self.gate_up_proj(x))))


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, cfg: Qwen3Config, layer_id: int, rope_table: torch.Tensor):
        super().__init__()
        self.self_attn = Qwen3Attention(cfg, layer_id, rope_table)
        self.mlp = Qwen3MLP(cfg)
        self.input_layernorm = RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)

    def forward(self, positions, hidden_states, residual, fused: bool = True):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, fused)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        md = self.self_attn.attn.attention_metadata
        return self.mlp(hidden_states, decode=fused and md.forward_mode == ForwardMode.DECODE), residual


class VocabParallelEmbedding(nn.Module):
    """Rows [rank*V/tp, (rank+1)*V/tp) of the (64-padded) vocabulary; other ids contribute zeros and the
    ranks' partial rows are summed (vocab_parallel_embedding.py:56-108)."""

    def __init__(self, vocab: int, hidden: int):
        super().__init__()
        tp = tpd.tp_size()
        self.vocab = vocab
        padded = ((vocab + 63) // 64) * 64
        self.per_rank = _shard(padded, tp)
        self.start = self.per_rank * tpd.tp_rank()
        self.weight = nn.Parameter(torch.empty(self.per_rank, hidden))

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        if not tpd.collectives_active():
            return F.embedding(ids, self.weight)
        local = ids - self.start
        mine = (local >= 0) & (local < self.per_rank)
        out = F.embedding(torch.where(mine, local, torch.zeros_like(local)), self.weight)
        out.masked_fill_(~mine.unsqueeze(-1), 0)
        return tpd.tp_all_reduce(out)


def stage_layer_range(num_layers: int, pp_rank: int, pp_size: int, balanced: bool = False):
    """[start, end) of the decoder layers a pipeline stage holds.  Default: the reference's split
    (distributed/utils.py:10-15) -- num_layers // pp_size layers per stage, the LAST stage taking the whole
    remainder (11 layers over 3 stages: 3, 3, 5).  `balanced=True` (env NVX_PP_BALANCED=1 when the model is
    built) spreads the remainder one layer each over the last stages instead (3, 4, 4): the longest stage
    sets the pipeline's step time.  Both agree whenever pp_size divides num_layers."""
    base, extra = divmod(num_layers, pp_size)
    if not balanced:
        start = pp_rank * base
        return start, (num_layers if pp_rank == pp_size - 1 else start + base)
    sizes = [base + (1 if s >= pp_size - extra else 0) for s in range(pp_size)]
    start = sum(sizes[:pp_rank])
    return start, start + sizes[pp_rank]


class MissingLayer(nn.Module):
    """Stands where another pipeline stage's module would be, so that layer numbers (and with them
    checkpoint names and KV-pool layer ids) stay global (qwen3.py:191-206)."""

    def forward(self, *args, **kwargs):
        raise RuntimeError("this module lives on another pipeline stage")


class Qwen3Model(nn.Module):
    def __init__(self, cfg: Qwen3Config):
        super().__init__()
        self.config = cfg
        first, last = tpd.pp_is_first(), tpd.pp_is_last()
        self.start_layer, self.end_layer = stage_layer_range(cfg.num_hidden_layers, tpd.pp_rank(), tpd.pp_size(),
                                                             balanced=os.environ.get("NVX_PP_BALANCED") == "1")
        # tied weights: the last stage needs the embedding matrix for its LM head
        self.embed_tokens = VocabParallelEmbedding(cfg.vocab_size, cfg.hidden_size) \
            if first or (last and cfg.tie_word_embeddings) else MissingLayer()
        self.register_buffer("rope_table", build_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta),
                             persistent=False)
        self.layers = nn.ModuleList(Qwen3DecoderLayer(cfg, i, self.rope_table) if self.start_layer <= i < self.end_layer
                                    else MissingLayer() for i in range(cfg.num_hidden_layers))
        self.norm = RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps) if last else MissingLayer()

    def local_layers(self):
        return self.layers[self.start_layer:self.end_layer]

    def forward(self, input_ids, positions, fused: bool = True):
        hidden_states, residual = self.embed_tokens(input_ids), None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual, fused)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states

    def forward_stage(self, input_ids, positions, hidden_states, residual, fused: bool = True):
        """One pipeline stage.  The first stage embeds `input_ids` (hidden_states/residual are None), the
        others continue from the previous stage's pair; the last returns (normed hidden states, None),
        the others the pair to hand on (qwen3.py:249-268)."""
        if tpd.pp_is_first():
            hidden_states, residual = self.embed_tokens(input_ids), None
        for layer in self.local_layers():
            hidden_states, residual = layer(positions, hidden_states, residual, fused)
        if not tpd.pp_is_last():
            hidden_states = tpd.tp_finish_reduce(hidden_states)    # an all-reduce no norm of this stage will do
            if hidden_states.dtype == torch.float32:   # K-split partial products of a skinny projection: sum them
                hidden_states = hidden_states.sum(0).to(residual.dtype)
            return hidden_states, residual
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states, None


def random_checkpoint(cfg: Qwen3Config, seed: int = 0):
    """(name, fp32 tensor) pairs of a seeded random Qwen3 checkpoint under the Hugging Face tensor names, full
    (unsharded) sizes, one tensor at a time.  Model.init_random loads exactly this stream, so anything else
    that consumes it (a CPU restatement of the model in the tests) holds the same weights."""
    gen = torch.Generator(device="cpu").manual_seed(seed)

    def full(rows, cols, std):
        return torch.randn(rows, cols, generator=gen) * std

    def vec(n):
        return 1.0 + 0.1 * torch.randn(n, generator=gen)

    h, d = cfg.hidden_size, cfg.head_dim
    hq, hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    yield "model.embed_tokens.weight", full(cfg.vocab_size, h, 0.05)
    yield "model.norm.weight", vec(h)
    if not cfg.tie_word_embeddings:
        yield "lm_head.weight", full(cfg.vocab_size, h, h ** -0.5)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        yield p + "self_attn.q_proj.weight", full(hq * d, h, h ** -0.5)
        yield p + "self_attn.k_proj.weight", full(hkv * d, h, h ** -0.5)
        yield p + "self_attn.v_proj.weight", full(hkv * d, h, h ** -0.5)
        yield p + "self_attn.o_proj.weight", full(h, hq * d, (hq * d) ** -0.5)
        yield p + "self_attn.q_norm.weight", vec(d)
        yield p + "self_attn.k_norm.weight", vec(d)
        yield p + "mlp.gate_proj.weight", full(cfg.intermediate_size, h, h ** -0.5)
        yield p + "mlp.up_proj.weight", full(cfg.intermediate_size, h, h ** -0.5)
        yield p + "mlp.down_proj.weight", full(h, cfg.intermediate_size, cfg.intermediate_size ** -0.5)
        yield p + "input_layernorm.weight", vec(h)
        yield p + "post_attention_layernorm.weight", vec(h)


class Qwen3ForCausalLM(nn.Module):
    # (fused parameter, checkpoint name, shard id) -- qwen3.py:312-319
    STACKED = [("qkv_proj", "q_proj", "q"), ("qkv_proj", "k_proj", "k"), ("qkv_proj", "v_proj", "v"),
               ("gate_up_proj", "gate_proj", 0), ("gate_up_proj", "up_proj", 1)]

    def __init__(self, cfg: Qwen3Config):
        super().__init__()
        self.config = cfg
        self.model = Qwen3Model(cfg)
        self.lm_head = VocabParallelEmbedding(cfg.vocab_size, cfg.hidden_size) if tpd.pp_is_last() else MissingLayer()
        self.fused = True
        global FP32_LINEAR
        FP32_LINEAR = os.environ.get("NVX_FP32_LINEAR") == "1"

    def finalize(self):
        """Call after .to(device, dtype): restore the fp32 rotary table and share it with every layer
        (module-wide casts touch buffers), tie the LM head when the checkpoint says so."""
        m = self.model
        dev = next(self.parameters()).device
        m.rope_table = build_cos_sin_cache(self.config.head_dim, self.config.max_position_embeddings,
                                           self.config.rope_theta).to(dev)
        for layer in m.local_layers():
            layer.self_attn.rope_table = m.rope_table
        if self.config.tie_word_embeddings and tpd.pp_is_last():
            self.lm_head.weight = m.embed_tokens.weight

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        return self.model(input_ids, positions, self.fused)

    def pack_decode_weights(self):
        """Call once the weights are in place: tiled copies of the weights that decode steps stream (large gate/up projections)."""
        for layer in self.model.local_layers():
            layer.mlp.pack_decode_weights()

    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        if FP32_LINEAR:
            return tpd.tp_all_gather_last_dim(F.linear(hidden_states.float(), self.lm_head.weight.float()))[..., :self.config.vocab_size]
        logits = tpd.tp_all_gather_last_dim(F.linear(hidden_states, self.lm_head.weight))
        return logits[..., :self.config.vocab_size]

    def compute_logits_shard(self, hidden_states: torch.Tensor):
        """This rank's columns of the logits, not gathered: ((B, vocab / tp) logits, id of their first token)."""
        return F.linear(hidden_states, self.lm_head.weight), self.lm_head.start

    # -- weights ---------------------------------------------------------------------------------
    @torch.no_grad()
    def init_random(self, seed: int = 0):
        """Seeded random weights of the right shapes (no checkpoint exists offline).  Each full matrix
        is drawn from the same stream on every rank and then sharded, so TP=N computes the same
        function as TP=1."""
        self.load_weights(random_checkpoint(self.config, seed))

    @torch.no_grad()
    def load_weights(self, weights: Iterable):
        """Checkpoint tensors (HF names, full size) -> this rank's fused, sharded parameters."""
        params = dict(self.named_parameters())
        tp, rank = tpd.tp_size(), tpd.tp_rank()
        for name, w in weights:
            for fused_name, ckpt_name, shard_id in self.STACKED:
                if ckpt_name in name:
                    target = params.get(name.replace(ckpt_name, fused_name))
                    if target is not None:   # else: a layer of another pipeline stage
                        self._load_stacked(target, w, shard_id, name, tp, rank)
                    break
            else:
                if name not in params:   # another stage's tensor, or lm_head of a tied checkpoint
                    continue
                p = params[name]
                if name.endswith("embed_tokens.weight") or name == "lm_head.weight":
                    per = p.shape[0]
                    rows = w[rank * per:(rank + 1) * per]
                    p[:rows.shape[0]].copy_(rows)
                    p[rows.shape[0]:].zero_()
                elif name.endswith("o_proj.weight") or name.endswith("down_proj.weight"):
                    per = p.shape[1]   # row-parallel: shard the input dimension
                    p.copy_(w[:, rank * per:(rank + 1) * per])
                else:
                    p.copy_(w)

    def _load_stacked(self, target, w, shard_id, name, tp, rank):
        layer = self.model.layers[int(name.split("layers.")[1].split(".")[0])]
        if shard_id in ("q", "k", "v"):
            a = layer.self_attn
            d = a.head_dim
            if shard_id == "q":
                off, size, idx = 0, a.num_heads * d, rank
            elif shard_id == "k":
                off, size, idx = a.num_heads * d, a.num_kv_heads * d, rank // a.kv_replicas
            else:
                off, size, idx = (a.num_heads + a.num_kv_heads) * d, a.num_kv_heads * d, rank // a.kv_replicas
        else:
            size = target.shape[0] // 2
            off, idx = shard_id * size, rank
        target[off:off + size].copy_(w[idx * size:(idx + 1) * size])

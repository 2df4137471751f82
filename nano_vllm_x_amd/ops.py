"""Tensor-level wrappers over the C ABI (include/nvx.h).

PyTorch is used here only as the owner of device memory and of the current HIP stream; every
function validates shapes on the host (a kernel that is handed a wrong shape faults the GPU), then
makes exactly one call into libnvx_hip.so on `torch.cuda.current_stream()`, which makes all of them
capturable into a hipGraph.  `nano_vllm_x_amd.torch_ops` registers the stateless ones as
`torch.ops.nvx.*` custom ops.
"""
from __future__ import annotations

import torch

from . import _lib

BF16 = torch.bfloat16
F16 = torch.float16
H16 = (BF16, F16)   # the 16-bit formats the library is built for: the kernels of a call all run in the format of its activations


def _fn(lib, name: str, dtype):
    """Entry point `name` in the build for `dtype`: bfloat16 is the unsuffixed symbol, float16 `name_f16`."""
    return getattr(lib, name + "_f16" if dtype == F16 else name)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need(cond: bool, msg: str):
    if not cond:
        raise RuntimeError(msg)


def _dev(*ts):
    for t in ts:
        if t is not None:
            _need(t.is_cuda, "nvx ops need device tensors (the HIP path has no CPU fallback)")


def _ptr(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------
def store_kvcache(k, v, k_cache, v_cache, kv_indices):
    """Same contract as nanovllmx.core.kv_cache.store_kvcache (kv_cache.py:36-44)."""
    _dev(k, v, k_cache, v_cache, kv_indices)
    num_tokens, num_heads, head_dim = k.shape
    row = num_heads * head_dim
    assert k.stride(-1) == 1 and v.stride(-1) == 1
    assert k.stride(-2) == head_dim and v.stride(-2) == head_dim
    assert k_cache.stride(-3) == row and v_cache.stride(-3) == row
    assert kv_indices.numel() == num_tokens
    _need(k.dtype == v.dtype == k_cache.dtype == v_cache.dtype and k.element_size() == 2, "store_kvcache: 16-bit dtypes only")
    _need(kv_indices.dtype == torch.int64 and kv_indices.is_contiguous(), "store_kvcache: kv_indices must be contiguous int64")
    lib = _lib.load()
    _lib.check(lib.nvx_store_kv(_ptr(k), _ptr(v), k.stride(0), v.stride(0), _ptr(k_cache), _ptr(v_cache), _ptr(kv_indices),
                                num_tokens, row, k_cache.shape[-3], _stream()), "nvx_store_kv")


def rmsnorm(x, weight, eps, residual=None):
    """RMSNorm.forward (layernorm.py:17-36): returns y, or (y, new_residual) when residual is given."""
    _dev(x, weight, residual)
    _need(x.dtype in H16, "rmsnorm: bfloat16 or float16 activations")
    hidden = x.shape[-1]
    if x.dim() == 2 and x.stride(1) == 1:
        x2 = x  # rows may be strided (a split view of a fused projection)
    else:
        x = x.contiguous()
        x2 = x.view(-1, hidden)
    rows = x2.shape[0]
    _need(weight.numel() == hidden and weight.is_contiguous(), "rmsnorm: weight shape")
    _need(weight.dtype in (x.dtype, torch.float32), "rmsnorm: weight must have the activations' dtype or be fp32")
    y = torch.empty((rows, hidden), dtype=x.dtype, device=x.device)
    res_out = None
    if residual is not None:
        _need(residual.dtype == x.dtype and residual.shape == x.shape, "rmsnorm: residual shape/dtype")
        residual = residual.contiguous()
        res_out = torch.empty((rows, hidden), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_rmsnorm", x.dtype)(_ptr(x2), x2.stride(0), _ptr(residual), _ptr(weight), int(weight.dtype == torch.float32),
                               _ptr(y), _ptr(res_out), rows, hidden, float(eps), _stream()), "nvx_rmsnorm")
    y = y.view(x.shape)
    if residual is None:
        return y
    return y, res_out.view(x.shape)


def rope(positions, q, k, cos_sin_cache, head_size=128):
    """RotaryEmbedding.forward (rotary_embedding.py:74-102), rotary_dim == head_size == 128."""
    _dev(positions, q, k, cos_sin_cache)
    _need(head_size == 128 and cos_sin_cache.shape[-1] == 128 and cos_sin_cache.dtype == torch.float32 and
          cos_sin_cache.is_contiguous(), "rope: needs an fp32 (max_pos, 128) table")
    _need(q.dtype in H16 and k.dtype == q.dtype, "rope: bfloat16 or float16, q and k alike")
    positions = positions.flatten()
    _need(positions.dtype == torch.int64 and positions.is_contiguous(), "rope: positions must be int64")
    t = positions.numel()
    q2, k2 = q.reshape(t, -1) if q.is_contiguous() else q.view(t, -1), k.reshape(t, -1) if k.is_contiguous() else k.view(t, -1)
    _need(q2.stride(1) == 1 and k2.stride(1) == 1 and q2.shape[1] % 128 == 0 and k2.shape[1] % 128 == 0, "rope: row layout")
    q_out = torch.empty((t, q2.shape[1]), dtype=q.dtype, device=q.device)
    k_out = torch.empty((t, k2.shape[1]), dtype=k.dtype, device=k.device)
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_rope", q.dtype)(_ptr(positions), _ptr(cos_sin_cache), cos_sin_cache.shape[0], _ptr(q2), q2.stride(0), _ptr(k2),
                            k2.stride(0), _ptr(q_out), _ptr(k_out), t, q2.shape[1] // 128, k2.shape[1] // 128, _stream()),
               "nvx_rope")
    return q_out.view(q.shape), k_out.view(k.shape)


def qknorm_rope_store(qkv, positions, cos_sin_cache, q_norm_weight, k_norm_weight, eps, slots, k_cache, v_cache,
                      n_q_heads, n_kv_heads, q_out=None):
    """qwen3.py:123-133 + attention.py:344-352 in one launch.  Returns q (T, n_q_heads*128)."""
    _dev(qkv, positions, cos_sin_cache, q_norm_weight, k_norm_weight, slots, k_cache, v_cache)
    t = qkv.shape[0]
    width = (n_q_heads + 2 * n_kv_heads) * 128
    _need(qkv.dim() == 2 and qkv.shape[1] == width and qkv.stride(1) == 1 and qkv.dtype in H16, "qknorm_rope_store: qkv layout")
    _need(positions.dtype == torch.int64 and positions.numel() >= t and positions.is_contiguous(), "qknorm_rope_store: positions")
    _need(slots.dtype == torch.int64 and slots.numel() >= t and slots.is_contiguous(), "qknorm_rope_store: slots")
    _need(k_cache.shape[-2:] == (n_kv_heads, 128) and v_cache.shape == k_cache.shape and k_cache.is_contiguous() and
          v_cache.is_contiguous() and k_cache.dtype == qkv.dtype and v_cache.dtype == qkv.dtype, "qknorm_rope_store: cache layout")
    _need(q_norm_weight.dtype == qkv.dtype and k_norm_weight.dtype == qkv.dtype and q_norm_weight.numel() == 128 and
          k_norm_weight.numel() == 128, "qknorm_rope_store: norm weights must be bf16 (128)")
    _need(cos_sin_cache.dtype == torch.float32 and cos_sin_cache.shape[-1] == 128 and cos_sin_cache.is_contiguous(), "qknorm_rope_store: table")
    if q_out is None:
        q_out = torch.empty((t, n_q_heads * 128), dtype=qkv.dtype, device=qkv.device)
    else:
        _need(q_out.shape == (t, n_q_heads * 128) and q_out.is_contiguous() and q_out.dtype == qkv.dtype, "qknorm_rope_store: q_out")
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_qknorm_rope_store", qkv.dtype)(_ptr(qkv), qkv.stride(0), _ptr(positions), _ptr(cos_sin_cache), cos_sin_cache.shape[0],
                                         _ptr(q_norm_weight), _ptr(k_norm_weight), float(eps), _ptr(slots), _ptr(q_out),
                                         _ptr(k_cache), _ptr(v_cache), k_cache.shape[0], t, n_q_heads, n_kv_heads, _stream()),
               "nvx_qknorm_rope_store")
    return q_out


def linear_decode_splits(n: int, k: int) -> int:
    return int(_lib.load().nvx_linear_decode_splits(n, k))


def linear_decode(x, w, mode: int = 0, k_splits: int = 1, out=None):
    """x (m <= 256, k) bf16 times w (n, k)^T.  mode 0 -> bf16 (m, n); mode 1 -> SiluAndMul of the
    [gate; up] projection, bf16 (m, n/2); mode 2 -> fp32 partials (k_splits, m, n) for rmsnorm_splitk."""
    _dev(x, w, out)
    _need(x.dim() == 2 and w.dim() == 2 and x.dtype in H16 and w.dtype == x.dtype, "linear_decode: 2-D bfloat16 or float16 operands")
    _need(x.stride(1) == 1 and w.is_contiguous() and x.shape[1] == w.shape[1], "linear_decode: layouts")
    m, k = x.shape
    n = w.shape[0]
    if mode == 0:
        shape, dtype = (m, n), x.dtype
    elif mode == 1:
        shape, dtype = (m, n // 2), x.dtype
    else:
        shape, dtype = (k_splits, m, n), torch.float32
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=x.device)
    else:
        _need(out.dtype == dtype and out.is_contiguous() and out.numel() >= shape[0] * shape[1] * (shape[2] if len(shape) == 3 else 1),
              "linear_decode: out")
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_linear_decode", x.dtype)(_ptr(x), x.stride(0), _ptr(w), _ptr(out), m, n, k, mode, k_splits, _stream()), "nvx_linear_decode")
    return out


def rmsnorm_splitk(partials, weight, eps, residual=None):
    """RMSNorm over bf16(sum of fp32 partials (splits, rows, hidden)); returns y or (y, new_residual)."""
    _dev(partials, weight, residual)
    _need(partials.dim() == 3 and partials.dtype == torch.float32 and partials.is_contiguous(), "rmsnorm_splitk: partials")
    _need(weight.dtype in H16 and weight.is_contiguous(), "rmsnorm_splitk: bfloat16 or float16 weight (it sets the output format)")
    s, rows, hidden = partials.shape
    y = torch.empty((rows, hidden), dtype=weight.dtype, device=partials.device)
    res_out = None
    if residual is not None:
        _need(residual.dtype == weight.dtype and residual.is_contiguous() and residual.shape == (rows, hidden), "rmsnorm_splitk: residual")
        res_out = torch.empty_like(residual)
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_rmsnorm_splitk", weight.dtype)(_ptr(partials), s, _ptr(residual), _ptr(weight), _ptr(y), _ptr(res_out), rows, hidden,
                                      float(eps), _stream()), "nvx_rmsnorm_splitk")
    return y if residual is None else (y, res_out)


def silu_mul(x, out=None):
    """SiluAndMul.forward (activation.py:15-18)."""
    _dev(x)
    _need(x.dtype in H16 and x.is_contiguous(), "silu_mul: contiguous bfloat16 or float16")
    d = x.shape[-1] // 2
    rows = x.numel() // (2 * d)
    if out is None:
        out = torch.empty(x.shape[:-1] + (d,), dtype=x.dtype, device=x.device)
    else:
        _need(out.is_contiguous() and out.numel() == rows * d and out.dtype == x.dtype, "silu_mul: out")
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_silu_mul", x.dtype)(_ptr(x), _ptr(out), rows, d, _stream()), "nvx_silu_mul")
    return out


# ---------------------------------------------------------------------------------------------
class DecodeWorkspace:
    """Caller-owned scratch of the decode attention (plan list + split partials), sized once."""

    def __init__(self, max_batch: int, n_q_heads: int, n_kv_heads: int, device):
        lib = _lib.load()
        self.max_batch, self.n_q_heads, self.n_kv_heads = max_batch, n_q_heads, n_kv_heads
        self.plan = torch.zeros(lib.nvx_decode_plan_ws_ints(max_batch, n_kv_heads), dtype=torch.int32, device=device)
        self.partial = torch.empty(lib.nvx_decode_partial_ws_floats(max_batch, n_q_heads, n_kv_heads), dtype=torch.float32,
                                   device=device)

    @classmethod
    def from_tensors(cls, plan: torch.Tensor, partial, n_q_heads: int, n_kv_heads: int) -> "DecodeWorkspace":
        """A workspace over tensors the caller already holds (how the custom ops receive it).  The batch it can
        serve is what both tensors have room for, by the library's own sizing rules."""
        lib = _lib.load()
        _need(plan.dtype == torch.int32 and plan.is_contiguous(), "decode workspace: plan must be contiguous int32")
        self = cls.__new__(cls)
        self.n_q_heads, self.n_kv_heads, self.plan, self.partial = n_q_heads, n_kv_heads, plan, partial
        # nvx_decode_plan_ws_ints(b, hkv) = 8 + 8 * (b + 2048 / hkv + 8) + 2 * b, nvx_decode_partial_ws_floats likewise linear in b
        fixed = int(lib.nvx_decode_plan_ws_ints(0, n_kv_heads))
        mb = (plan.numel() - fixed) // 10
        if partial is not None:
            _need(partial.dtype == torch.float32 and partial.is_contiguous(), "decode workspace: partials must be contiguous fp32")
            per = int(lib.nvx_decode_partial_ws_floats(1, n_q_heads, n_kv_heads)) - int(lib.nvx_decode_partial_ws_floats(0, n_q_heads, n_kv_heads))
            mb = min(mb, (partial.numel() - int(lib.nvx_decode_partial_ws_floats(0, n_q_heads, n_kv_heads))) // max(1, per))
        self.max_batch = max(0, int(mb))
        return self


def decode_plan(kv_len, kv_start, ws: DecodeWorkspace, out_slots=None):
    """Cut the batch's sequences into chunks (device side); must precede paged_decode of the same step.
    `out_slots` (int64 per sequence, -1 = none) is needed by paged_decode_qkv: the slot the step's new
    K/V row goes to."""
    _dev(kv_len, kv_start, out_slots, ws.plan)
    b = kv_len.numel()
    if out_slots is not None:
        _need(out_slots.dtype == torch.int64 and out_slots.is_contiguous() and out_slots.numel() == b, "decode_plan: out_slots")
    _need(kv_len.dtype == torch.int32 and kv_len.is_contiguous() and 0 < b <= ws.max_batch, "decode_plan: kv_len")
    _need(kv_start.dtype == torch.int64 and kv_start.is_contiguous() and kv_start.numel() == b, "decode_plan: kv_start")
    lib = _lib.load()
    _lib.check(lib.nvx_decode_plan(_ptr(kv_len), _ptr(kv_start), _ptr(out_slots), b, ws.n_kv_heads, _ptr(ws.plan), ws.plan.numel(),
                                   _stream()), "nvx_decode_plan")


def paged_decode_qkv(qkv, positions, cos_sin_cache, q_weight, k_weight, eps, k_cache, v_cache, kv_indices, batch: int,
                     ws: DecodeWorkspace, sm_scale, out=None):
    """Decode attention fed with the raw fused projection (B, (Hq + 2*Hkv)*128): per-head RMSNorm + rotary
    of q and of the new k, scatter of the new k/v row (to the slots given to decode_plan) and attention,
    in one launch.  Equivalent to qknorm_rope_store followed by paged_decode."""
    _dev(qkv, positions, cos_sin_cache, q_weight, k_weight, k_cache, v_cache, kv_indices)
    b = batch
    hq, hkv = ws.n_q_heads, ws.n_kv_heads
    _need(qkv.dim() == 2 and qkv.shape[0] == b and qkv.shape[1] == (hq + 2 * hkv) * 128 and qkv.dtype in H16 and qkv.stride(1) == 1,
          "paged_decode_qkv: qkv layout")
    _need(positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == b, "paged_decode_qkv: positions")
    _need(cos_sin_cache.dtype == torch.float32 and cos_sin_cache.is_contiguous() and cos_sin_cache.shape[-1] == 128,
          "paged_decode_qkv: cos/sin table")
    _need(q_weight.dtype == qkv.dtype and k_weight.dtype == qkv.dtype and q_weight.numel() == 128 and k_weight.numel() == 128,
          "paged_decode_qkv: norm weights")
    _need(k_cache.dtype == qkv.dtype and v_cache.dtype == qkv.dtype and k_cache.is_contiguous() and v_cache.is_contiguous() and
          tuple(k_cache.shape[-2:]) == (hkv, 128) and v_cache.shape == k_cache.shape, "paged_decode_qkv: cache layout")
    _need(kv_indices.dtype == torch.int32 and kv_indices.is_contiguous(), "paged_decode_qkv: kv_indices must be int32")
    _need(0 < b <= ws.max_batch, "paged_decode_qkv: batch exceeds the workspace")
    if out is None:
        out = torch.empty((b, hq * 128), dtype=qkv.dtype, device=qkv.device)
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_paged_decode_qkv", qkv.dtype)(_ptr(qkv), qkv.stride(0), _ptr(positions), _ptr(cos_sin_cache), cos_sin_cache.shape[0],
                                        _ptr(q_weight), _ptr(k_weight), float(eps), _ptr(k_cache), _ptr(v_cache), k_cache.shape[-3],
                                        _ptr(kv_indices), _ptr(ws.plan), _ptr(ws.partial), _ptr(out), b, hq, hkv, float(sm_scale),
                                        _stream()), "nvx_paged_decode_qkv")
    return out


def paged_decode(q, k_cache, v_cache, kv_indices, batch: int, ws: DecodeWorkspace, sm_scale, out=None):
    """q (B, Hq*128) or (B, Hq, 128); caches (slots, Hkv, 128) of one layer; ws.plan must hold the plan
    of this step (decode_plan with the same kv_indices layout)."""
    _dev(q, k_cache, v_cache, kv_indices)
    b = batch
    hq, hkv = ws.n_q_heads, ws.n_kv_heads
    q2 = q.view(b, hq * 128)
    _need(q2.dtype in H16 and q2.stride(1) == 1, "paged_decode: q layout")
    _need(k_cache.dtype == q2.dtype and v_cache.dtype == q2.dtype and k_cache.is_contiguous() and v_cache.is_contiguous() and
          tuple(k_cache.shape[-2:]) == (hkv, 128) and v_cache.shape == k_cache.shape, "paged_decode: cache layout")
    _need(kv_indices.dtype == torch.int32 and kv_indices.is_contiguous(), "paged_decode: kv_indices must be int32")
    _need(0 < b <= ws.max_batch, "paged_decode: batch exceeds the workspace")
    if out is None:
        out = torch.empty((b, hq * 128), dtype=q2.dtype, device=q.device)
    else:
        _need(out.is_contiguous() and out.numel() == b * hq * 128 and out.dtype == q2.dtype, "paged_decode: out")
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_paged_decode", q2.dtype)(_ptr(q2), q2.stride(0), _ptr(k_cache), _ptr(v_cache), k_cache.shape[0], _ptr(kv_indices),
                                    _ptr(ws.plan), _ptr(ws.partial), _ptr(out), b, hq, hkv, float(sm_scale), _stream()),
               "nvx_paged_decode")
    return out


def paged_decode_noalloc(q, k_cache, v_cache, kv_indices, kv_len, kv_start, n_q_heads: int, n_kv_heads: int, sm_scale, out=None):
    """decode_plan + paged_decode without a workspace: plan and partials live in an array built into the library
    (one per GPU; batch <= 256; calls that overlap on one GPU race).  For callers that cannot allocate."""
    _dev(q, k_cache, v_cache, kv_indices, kv_len, kv_start)
    b = kv_len.numel()
    q2 = q.view(b, n_q_heads * 128)
    _need(q2.dtype == BF16 and q2.stride(1) == 1, "paged_decode_noalloc: q layout")
    _need(k_cache.dtype == q2.dtype and v_cache.dtype == q2.dtype and k_cache.is_contiguous() and v_cache.is_contiguous() and
          tuple(k_cache.shape[-2:]) == (n_kv_heads, 128) and v_cache.shape == k_cache.shape, "paged_decode_noalloc: cache layout")
    _need(kv_indices.dtype == torch.int32 and kv_indices.is_contiguous(), "paged_decode_noalloc: kv_indices must be int32")
    _need(kv_len.dtype == torch.int32 and kv_len.is_contiguous() and kv_start.dtype == torch.int64 and kv_start.is_contiguous()
          and kv_start.numel() == b, "paged_decode_noalloc: kv_len int32 / kv_start int64, one per sequence")
    if out is None:
        out = torch.empty((b, n_q_heads * 128), dtype=BF16, device=q.device)
    lib = _lib.load()
    _lib.check(lib.nvx_paged_decode_noalloc(_ptr(q2), q2.stride(0), _ptr(k_cache), _ptr(v_cache), k_cache.shape[0], _ptr(kv_indices),
                                            _ptr(kv_len), _ptr(kv_start), _ptr(out), b, n_q_heads, n_kv_heads, float(sm_scale),
                                            _stream()), "nvx_paged_decode_noalloc")
    return out


def prefill_plan_ws(batch: int, device):
    lib = _lib.load()
    return torch.zeros(lib.nvx_prefill_plan_ws_ints(batch), dtype=torch.int32, device=device)


def prefill_plan(qo_indptr, plan_ws):
    _dev(qo_indptr, plan_ws)
    b = qo_indptr.numel() - 1
    _need(qo_indptr.dtype == torch.int32 and qo_indptr.is_contiguous() and b > 0, "prefill_plan: qo_indptr")
    lib = _lib.load()
    _lib.check(lib.nvx_prefill_plan(_ptr(qo_indptr), b, _ptr(plan_ws), plan_ws.numel(), _stream()), "nvx_prefill_plan")


def paged_prefill(q, k_cache, v_cache, qo_indptr, kv_indptr, kv_indices, plan_ws, n_q_heads, n_kv_heads, sm_scale, out=None):
    """q (T, Hq*128); causal varlen attention over paged K/V (attention.py:360-364)."""
    _dev(q, k_cache, v_cache, qo_indptr, kv_indptr, kv_indices, plan_ws)
    t = q.shape[0]
    b = qo_indptr.numel() - 1
    q2 = q.view(t, n_q_heads * 128)
    _need(q2.dtype in H16 and q2.stride(1) == 1, "paged_prefill: q layout")
    _need(k_cache.dtype == q2.dtype and v_cache.dtype == q2.dtype and k_cache.is_contiguous() and v_cache.is_contiguous() and
          tuple(k_cache.shape[-2:]) == (n_kv_heads, 128) and v_cache.shape == k_cache.shape, "paged_prefill: cache layout")
    for name, x in (("qo_indptr", qo_indptr), ("kv_indptr", kv_indptr), ("kv_indices", kv_indices)):
        _need(x.dtype == torch.int32 and x.is_contiguous(), f"paged_prefill: {name} must be contiguous int32")
    _need(kv_indptr.numel() == b + 1, "paged_prefill: indptr lengths differ")
    if out is None:
        out = torch.empty((t, n_q_heads * 128), dtype=q2.dtype, device=q.device)
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_paged_prefill", q2.dtype)(_ptr(q2), q2.stride(0), _ptr(k_cache), _ptr(v_cache), k_cache.shape[0], _ptr(qo_indptr),
                                     _ptr(kv_indptr), _ptr(kv_indices), _ptr(plan_ws), _ptr(out), t, b, n_q_heads, n_kv_heads,
                                     float(sm_scale), _stream()), "nvx_paged_prefill")
    return out


# ---------------------------------------------------------------------------------------------
def _sampler_args(logits, temperatures, top_ks, top_ps, min_ps):
    _dev(logits, temperatures, top_ks, top_ps, min_ps)
    _need(logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype in (BF16, torch.float32), "sampler: logits layout")
    b, v = logits.shape
    _need(temperatures.dtype == torch.float32 and top_ps.dtype == torch.float32 and min_ps.dtype == torch.float32 and
          top_ks.dtype == torch.int64, "sampler: parameter dtypes (float32, int64)")
    for x in (temperatures, top_ks, top_ps, min_ps):
        _need(x.numel() == b and x.is_contiguous(), "sampler: one parameter per row")
    return b, v


_SAMPLE_WS: dict = {}   # device -> int64 workspace (slices' winners), grown on demand outside graph capture


def sample_workspace(batch: int, device) -> torch.Tensor:
    """Workspace that lets nvx_sample cut rows into slices.  Allocate it before capturing a graph that
    samples: the captured launch keeps the pointer."""
    lib = _lib.load()
    need = int(lib.nvx_sample_ws_bytes(batch)) // 8
    key = torch.device(device)
    ws = _SAMPLE_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, int(lib.nvx_sample_ws_bytes(256)) // 8), dtype=torch.int64, device=device)
        _SAMPLE_WS[key] = ws
    return ws


def sample(logits, temperatures, top_ks, top_ps, min_ps, seed: int, offset: int, out=None, ws=None):
    """Sampler.forward (sampler.py:39-55): one token id per row, Philox(seed, offset) Gumbel-max draw."""
    b, v = _sampler_args(logits, temperatures, top_ks, top_ps, min_ps)
    if ws is None:
        ws = sample_workspace(b, logits.device)
    if out is None:
        out = torch.empty((b,), dtype=torch.int64, device=logits.device)
    else:
        _need(out.dtype == torch.int64 and out.numel() >= b and out.is_contiguous(), "sample: out")
    lib = _lib.load()
    _lib.check(lib.nvx_sample(_ptr(logits), int(logits.dtype == torch.float32), logits.stride(0), _ptr(temperatures),
                              _ptr(top_ks), _ptr(top_ps), _ptr(min_ps), b, v, seed & (2**64 - 1), offset & (2**64 - 1),
                              _ptr(out), _ptr(ws), ws.numel() * 8, _stream()), "nvx_sample")
    return out


def sample_shard(logits, temperatures, vocab_offset: int, vocab: int, seed: int, offset: int, out=None, ws=None):
    """One tensor-parallel rank's part of an unfiltered sampling step: the draw over its shard of the vocabulary
    (logits (B, shard), token j = token vocab_offset + j of the whole row), one 64-bit (score, token) key per row."""
    _dev(logits, temperatures)
    _need(logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype in (BF16, torch.float32), "sample_shard: logits layout")
    b, v = logits.shape
    _need(temperatures.dtype == torch.float32 and temperatures.numel() == b and temperatures.is_contiguous(), "sample_shard: temperatures")
    _need(vocab_offset >= 0 and vocab_offset % 4 == 0 and vocab > 0, "sample_shard: the shard must start at a multiple of 4 tokens")
    if ws is None:
        ws = sample_workspace(b, logits.device)
    if out is None:
        out = torch.empty((b,), dtype=torch.int64, device=logits.device)
    else:
        _need(out.dtype == torch.int64 and out.numel() >= b and out.is_contiguous(), "sample_shard: out")
    lib = _lib.load()
    _lib.check(lib.nvx_sample_shard(_ptr(logits), int(logits.dtype == torch.float32), logits.stride(0), _ptr(temperatures), b, v,
                                    vocab_offset, vocab, seed & (2**64 - 1), offset & (2**64 - 1), _ptr(out), _ptr(ws), ws.numel() * 8,
                                    _stream()), "nvx_sample_shard")
    return out


def lmhead_max_rows() -> int:
    return int(_lib.load().nvx_lmhead_max_rows())


def lmhead_pack(weight):
    """LM-head weight (rows, k) 16-bit -> the tiled copy lmhead_sample streams (built once, when the model is loaded)."""
    _dev(weight)
    _need(weight.dim() == 2 and weight.is_contiguous() and weight.dtype in H16 and weight.shape[1] % 32 == 0, "lmhead_pack: weight layout")
    lib = _lib.load()
    n, k = weight.shape
    packed = torch.empty(int(lib.nvx_lmhead_packed_elems(n, k)), dtype=weight.dtype, device=weight.device)
    _lib.check(lib.nvx_lmhead_pack(_ptr(weight), n, k, _ptr(packed), _stream()), "nvx_lmhead_pack")
    return packed


def lmhead_workspace(batch: int, device):
    return torch.empty(int(_lib.load().nvx_lmhead_ws_bytes(batch)) // 8, dtype=torch.int64, device=device)


def lmhead_sample(packed, n_rows: int, x, temperatures, vocab_offset: int, vocab: int, seed: int, offset: int, ws=None, keys=False,
                  out=None, logits_out=None):
    """hidden states x (B, k) -> one token per row (or, `keys`, the 64-bit (score, token) keys as int64 bit patterns, for
    sample_pick over tensor-parallel ranks): the LM head and the unfiltered Gumbel-max draw in one pass over the packed
    weight.  `logits_out` (B, >= n_rows padded to 16) receives the 16-bit logits if given."""
    _dev(packed, x, temperatures, logits_out)
    _need(x.dim() == 2 and x.stride(1) == 1 and x.dtype in H16 and packed.dtype == x.dtype, "lmhead_sample: x layout")
    b, k = x.shape
    _need(0 < b <= lmhead_max_rows(), "lmhead_sample: batch exceeds what the kernel keeps on a compute unit")
    _need(temperatures.dtype == torch.float32 and temperatures.numel() == b and temperatures.is_contiguous(), "lmhead_sample: temperatures")
    _need(packed.is_contiguous() and packed.numel() == ((n_rows + 15) // 16) * 16 * k, "lmhead_sample: packed weight size")
    if ws is None:
        ws = lmhead_workspace(b, x.device)
    if out is None:
        out = torch.empty((b,), dtype=torch.int64, device=x.device)
    else:
        _need(out.dtype == torch.int64 and out.numel() >= b and out.is_contiguous(), "lmhead_sample: out")
    if logits_out is not None:
        _need(logits_out.dim() == 2 and logits_out.dtype == x.dtype and logits_out.shape[0] >= b and logits_out.stride(1) == 1 and
              logits_out.shape[1] >= ((n_rows + 15) // 16) * 16, "lmhead_sample: logits_out")
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_lmhead_sample", x.dtype)(_ptr(packed), n_rows, k, _ptr(x), x.stride(0), b, _ptr(temperatures), vocab_offset, vocab,
                                                     seed & (2**64 - 1), offset & (2**64 - 1), _ptr(ws), ws.numel() * 8,
                                                     _ptr(out) if keys else None, None if keys else _ptr(out), _ptr(logits_out),
                                                     logits_out.stride(0) if logits_out is not None else 0, _stream()), "nvx_lmhead_sample")
    return out


def linear_stream(x, packed, n: int, out=None):
    """x (m, k) 16-bit, m <= 64, k a multiple of 1024; `packed` = lmhead_pack of the (n, k) weight -> fp32 partial products
    (k / 1024, m, n), one per 1024-wide slice of k, for rmsnorm_splitk or splitk_sum to add up."""
    _dev(x, packed, out)
    _need(x.dim() == 2 and x.stride(1) == 1 and x.dtype in H16 and packed.dtype == x.dtype and packed.is_contiguous(), "linear_stream: x / packed layout")
    m, k = x.shape
    _need(0 < m <= lmhead_rows_per_pass() and k % 1024 == 0, "linear_stream: at most 64 rows, k a multiple of 1024")
    _need(packed.numel() == ((n + 15) // 16) * 16 * k, "linear_stream: packed weight size")
    s = k // 1024
    if out is None:
        out = torch.empty((s, m, n), dtype=torch.float32, device=x.device)
    else:
        _need(out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (s, m, n), "linear_stream: out")
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_linear_stream", x.dtype)(_ptr(packed), n, k, _ptr(x), x.stride(0), m, _ptr(out), _stream()), "nvx_linear_stream")
    return out


def linear_block_max_rows() -> int:
    return int(_lib.load().nvx_linear_block_max_rows())


def linear_block_slices(n: int, k: int, mode: int) -> int:
    """k-slices linear_block cuts a weight of n rows over k into (mode 2), 1 for modes 0 / 1; 0 = shape not supported."""
    return int(_lib.load().nvx_linear_block_slices(n, k, mode))


def linear_block(x, packed, n: int, mode: int = 0, k_slices: int = 1, out=None):
    """x (m, k) 16-bit, m <= 256; `packed` = lmhead_pack of the (n, k) weight.  mode 0: y (m, n); mode 1: the weight is
    [gate rows; up rows], y (m, n / 2) = SiluAndMul of the 16-bit-rounded projection; mode 2: fp32 partial products
    (k_slices, m, n) for rmsnorm_splitk / splitk_sum to add up."""
    _dev(x, packed, out)
    _need(x.dim() == 2 and x.stride(1) == 1 and x.dtype in H16 and packed.dtype == x.dtype and packed.is_contiguous(), "linear_block: x / packed layout")
    m, k = x.shape
    _need(0 < m <= linear_block_max_rows(), "linear_block: 1..256 rows")
    _need(n % 16 == 0 and packed.numel() == n * k, "linear_block: packed weight size")
    _need(mode in (0, 1, 2) and k_slices >= 1 and (mode == 2 or k_slices == 1), "linear_block: mode / k_slices")
    shape = (k_slices, m, n) if mode == 2 else (m, n // 2 if mode == 1 else n)
    dtype = torch.float32 if mode == 2 else x.dtype
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=x.device)
    else:
        _need(out.dtype == dtype and out.is_contiguous() and tuple(out.shape) == shape, "linear_block: out")
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_linear_block", x.dtype)(_ptr(packed), n, k, _ptr(x), x.stride(0), m, mode, k_slices, _ptr(out), _stream()),
               "nvx_linear_block")
    return out


def lmhead_rows_per_pass() -> int:
    return int(_lib.load().nvx_lmhead_rows_per_pass())


def splitk_sum(partials, dtype, silu_mul: bool = False):
    """(splits, rows, n) fp32 -> (rows, n) in `dtype`, or SiluAndMul of the summed [gate | up] projection, (rows, n / 2)."""
    _dev(partials)
    _need(partials.dim() == 3 and partials.dtype == torch.float32 and partials.is_contiguous() and dtype in H16, "splitk_sum: partials")
    s, rows, n = partials.shape
    y = torch.empty((rows, n // 2 if silu_mul else n), dtype=dtype, device=partials.device)
    lib = _lib.load()
    _lib.check(_fn(lib, "nvx_splitk_sum", dtype)(_ptr(partials), s, rows, n, int(silu_mul), _ptr(y), _stream()), "nvx_splitk_sum")
    return y


def sample_pick(winners, out=None):
    """Token ids from the ranks' keys: winners (n_sets, B) int64 (bit patterns of the 64-bit keys) -> (B,) int64."""
    _dev(winners)
    _need(winners.dim() == 2 and winners.dtype == torch.int64 and winners.is_contiguous(), "sample_pick: winners must be contiguous int64 (sets, B)")
    n, b = winners.shape
    if out is None:
        out = torch.empty((b,), dtype=torch.int64, device=winners.device)
    else:
        _need(out.dtype == torch.int64 and out.numel() >= b and out.is_contiguous(), "sample_pick: out")
    lib = _lib.load()
    _lib.check(lib.nvx_sample_pick(_ptr(winners), n, b, _ptr(out), _stream()), "nvx_sample_pick")
    return out


def sampler_probs(logits, temperatures, top_ks, top_ps, min_ps):
    """The distribution the sampler draws from (after temperature, top-k, top-p, min-p), fp32 (B, V)."""
    b, v = _sampler_args(logits, temperatures, top_ks, top_ps, min_ps)
    probs = torch.empty((b, v), dtype=torch.float32, device=logits.device)
    lib = _lib.load()
    _lib.check(lib.nvx_sampler_probs(_ptr(logits), int(logits.dtype == torch.float32), logits.stride(0), _ptr(temperatures),
                                     _ptr(top_ks), _ptr(top_ps), _ptr(min_ps), b, v, _ptr(probs), _stream()),
               "nvx_sampler_probs")
    return probs


# ---------------------------------------------------------------------------------------------
def block_table_write_rows(kv_indices, kv_indptr, rows, table):
    _dev(kv_indices, kv_indptr, rows, table)
    b = rows.numel()
    _need(table.dim() == 2 and table.is_contiguous() and table.dtype == torch.int32, "block_table_write_rows: table")
    for x in (kv_indices, kv_indptr, rows):
        _need(x.dtype == torch.int32 and x.is_contiguous(), "block_table_write_rows: int32 inputs")
    _need(kv_indptr.numel() == b + 1, "block_table_write_rows: indptr length")
    lib = _lib.load()
    _lib.check(lib.nvx_block_table_write_rows(_ptr(kv_indices), _ptr(kv_indptr), _ptr(rows), b, _ptr(table), table.shape[0],
                                              table.shape[1], _stream()), "nvx_block_table_write_rows")


def decode_step_setup(step, stride, padded_batch, table, pad_start, kv_start, kv_len, out_slots, positions, input_ids,
                      prev_ids=None):
    """`prev_ids` (int64, at least `stride` entries): where a negative token -1-j of the step buffer is looked up."""
    _dev(step, table, kv_start, kv_len, out_slots, positions, input_ids)
    if prev_ids is not None:
        _dev(prev_ids)
        _need(prev_ids.dtype == torch.int64 and prev_ids.numel() >= stride and prev_ids.is_contiguous(), "decode_step_setup: prev_ids")
    _need(step.dtype == torch.int32 and step.numel() >= 4 + 4 * stride and step.is_contiguous(), "decode_step_setup: step buffer")
    _need(table.dim() == 2 and table.is_contiguous() and table.dtype == torch.int32, "decode_step_setup: table")
    _need(kv_len.dtype == torch.int32 and all(x.dtype == torch.int64 for x in (kv_start, out_slots, positions, input_ids)),
          "decode_step_setup: output dtypes")
    for x in (kv_start, kv_len, out_slots, positions, input_ids):
        _need(x.numel() >= padded_batch and x.is_contiguous(), "decode_step_setup: output sizes")
    lib = _lib.load()
    _lib.check(lib.nvx_decode_step_setup(_ptr(step), stride, padded_batch, _ptr(table), table.shape[0], table.shape[1], pad_start,
                                         _ptr(kv_start), _ptr(kv_len), _ptr(out_slots), _ptr(positions), _ptr(input_ids),
                                         _ptr(prev_ids) if prev_ids is not None else None, _stream()), "nvx_decode_step_setup")


# ---------------------------------------------------------------------------------------------
class PeerGroup:
    """One rank's end of the direct-peer all-reduce of a tensor-parallel group (include/nvx.h, nvx_peer_*): a landing
    buffer in uncached device memory that every peer has mapped, into which the peers' kernels write their rows.
    Created once at start-up; `export_handle` / `import_handle` carry the 64-byte inter-process handles, which the
    caller exchanges by its own means.  `_registry` lets the custom ops address a group by an integer."""

    _registry: dict = {}

    def __init__(self, tp_size: int, rank: int, max_rows: int, hidden: int, timeout_ms: int = 0):
        import ctypes
        lib = _lib.load()
        comm = ctypes.c_void_p()
        _lib.check(lib.nvx_peer_create(tp_size, rank, max_rows, hidden, timeout_ms, ctypes.byref(comm)), "nvx_peer_create")
        self.comm, self.tp_size, self.rank, self.max_rows, self.hidden = comm.value, tp_size, rank, max_rows, hidden
        self.exchange_bytes = int(lib.nvx_peer_exchange_bytes(tp_size, max_rows, hidden))
        PeerGroup._registry[self.comm] = self

    def export_handle(self) -> bytes:
        import ctypes
        buf = ctypes.create_string_buffer(64)
        _lib.check(_lib.load().nvx_peer_export(self.comm, buf), "nvx_peer_export")
        return buf.raw

    def import_handle(self, peer_rank: int, handle: bytes):
        _need(len(handle) == 64, "PeerGroup.import_handle: a handle is 64 bytes")
        _lib.check(_lib.load().nvx_peer_import(self.comm, peer_rank, handle), "nvx_peer_import")

    def status(self) -> int:
        """0, or 1 once a wait inside an exchange ran out (a peer died or fell behind by more than the time-out)."""
        return int(_lib.load().nvx_peer_status(self.comm)) if self.comm else -1

    def close(self):
        if self.comm:
            PeerGroup._registry.pop(self.comm, None)
            _lib.check(_lib.load().nvx_peer_destroy(self.comm), "nvx_peer_destroy")
            self.comm = None

    # Every rank writes its whole tensor to every peer (one hop, no second exchange): the latency of one posted write,
    # but (tp - 1) x rows x hidden x 4 bytes leave each GPU (2 bytes of tag beside every 2 bytes of data), against
    # 2 x (tp - 1) / tp x rows x hidden x 2 for a reduce-scatter + all-gather.  Past a few MiB per exchange the links'
    # time outweighs the saved hops and launches, and the library's bandwidth-oriented all-reduce takes over.
    # This is a ROUTING policy (`prefer`), not a limit of the kernel (`accepts`); the ranks of a group must route alike,
    # so distributed.init_peer_allreduce agrees on one value for all of them (`max_push_bytes`).
    MAX_PUSH_BYTES = int(__import__("os").environ.get("NVX_PEER_MAX_PUSH_BYTES", str(4 << 20)))

    @staticmethod
    def rows_within_push_cap(tp_size: int, hidden: int, max_push_bytes: int) -> int:
        """Most rows of one exchange whose pushes stay within `max_push_bytes` per GPU (at least one)."""
        return max(1, max_push_bytes // (max(tp_size - 1, 1) * hidden * 4))

    max_push_bytes = MAX_PUSH_BYTES

    def accepts(self, x) -> bool:
        """What the kernel can take at all: the group's row width and format, at most the rows its landing buffer holds."""
        return self.comm is not None and x.dim() == 2 and x.shape[0] <= self.max_rows and x.shape[1] == self.hidden and x.dtype in H16 \
            and x.stride(1) == 1

    def prefer(self, x) -> bool:
        """Whether this exchange should go peer to peer rather than through the library all-reduce."""
        return self.accepts(x) and (self.tp_size - 1) * x.shape[0] * self.hidden * 4 <= self.max_push_bytes

    fits = prefer     # (older name)


def peer_allreduce_norm(group: PeerGroup, x, weight=None, eps: float = 0.0, residual=None):
    """Sum of the ranks' `x` (rows, hidden), rounded to the 16-bit format; then, with `weight`, RMSNorm of it (of it plus
    `residual`).  Returns the sum written over `x` (weight None), y, or (y, new_residual).  Every rank of the group
    must make the same calls in the same order."""
    _dev(x, weight, residual)
    _need(group.accepts(x), f"peer_allreduce_norm: x {tuple(x.shape)} {x.dtype} does not fit the group ({group.max_rows} rows x {group.hidden})")
    rows, hidden = x.shape
    lib = _lib.load()
    if weight is None:
        _need(residual is None, "peer_allreduce_norm: a residual needs the norm")
        _need(x.stride(0) == hidden, "peer_allreduce_norm: the in-place sum needs contiguous rows")
        _lib.check(_fn(lib, "nvx_peer_allreduce_norm", x.dtype)(group.comm, _ptr(x), x.stride(0), None, None, _ptr(x), None, rows, hidden, 0.0,
                                                                 _stream()), "nvx_peer_allreduce_norm")
        return x
    _need(weight.dtype == x.dtype and weight.is_contiguous() and weight.numel() == hidden, "peer_allreduce_norm: weight")
    y = torch.empty((rows, hidden), dtype=x.dtype, device=x.device)
    res_out = None
    if residual is not None:
        _need(residual.dtype == x.dtype and residual.is_contiguous() and residual.shape == (rows, hidden), "peer_allreduce_norm: residual")
        res_out = torch.empty_like(residual)
    _lib.check(_fn(lib, "nvx_peer_allreduce_norm", x.dtype)(group.comm, _ptr(x), x.stride(0), _ptr(residual), _ptr(weight), _ptr(y), _ptr(res_out),
                                                             rows, hidden, float(eps), _stream()), "nvx_peer_allreduce_norm")
    return y if residual is None else (y, res_out)

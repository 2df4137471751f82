"""CPU oracle for the paged-attention + radix KV-cache path.  TEST INFRASTRUCTURE ONLY.

Nothing under this directory is part of the product: it may be imported by `tests/`,
by `__graft_entry__.smoke()` and by the `cpu_baseline` leg of `bench.py`, always as the
checker (or as the timed CPU baseline), never as the thing that is shipped or measured
as the GPU result.  The product package fails loudly when its HIP library is missing; it
never falls back to this code.

Pinning status (see each module's header):
  * integer paths (slot allocator, radix tree, KV manager, scheduler, block-table
    metadata, KV scatter) -- pinned bit-exactly by golden traces frozen from the
    reference (tests/golden/g1..g4).
  * RMSNorm / RoPE / SiluAndMul / sampler masks / tiny-Qwen3 wiring -- pinned by golden
    vectors frozen from the reference run in eager mode on CPU (g5..g7).
  * the attention arithmetic itself lives in a third-party library that the reference
    only declares as `flashinfer-python>=0.2.0` (pyproject.toml:19); it is absent here
    and the reference holds no tests or vectors at that boundary: PARITY UNPINNED for
    that one function; `attention_oracle.py` restates the published semantics of the
    call sites (nanovllmx/layers/attention.py:106-132, 354-369).
"""

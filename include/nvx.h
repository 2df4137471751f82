/*
 * nvx.h -- C ABI of libnvx_hip.so: the MI355X (gfx950) kernels behind the paged-attention +
 * radix KV-cache path of nano-vllm-x.
 *
 * Conventions (every entry point):
 *   - plain pointers and sizes only; device pointers unless the name says `host`;
 *   - the caller owns all memory; nothing here allocates, frees or synchronises;
 *   - work is launched on `stream` (a hipStream_t passed as void*), so every call can be
 *     captured into a hipGraph;
 *   - return 0 on success, a negative NVX_E* code otherwise; nvx_last_error() gives the text.
 *     The Python host raises RuntimeError on a non-zero code, which is how the reference
 *     reports failures on this path (nanovllmx/core/kv_cache.py:327, asserts at :39-42).
 *   - 16-bit floating point means bfloat16 (the dtype of the Qwen3 checkpoints the reference
 *     serves, nanovllmx/core/model_runner.py:38-39).
 *   - head_dim is 128 (all Qwen3 sizes); other values are rejected with NVX_EUNSUPPORTED.
 *
 * The reference has no FFI of its own: its operator surface is the set of Python classes in
 * nanovllmx/core/kv_cache.py and nanovllmx/layers/{attention,layernorm,rotary_embedding,
 * sampler,activation}.py.  Each entry point below names the reference call it stands behind.
 */
#ifndef NVX_H
#define NVX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVX_OK 0
#define NVX_EINVAL (-1)       /* bad argument (null pointer, negative size, misalignment) */
#define NVX_EUNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define NVX_ELAUNCH (-3)      /* the HIP runtime refused the launch */
#define NVX_ENOSPACE (-4)     /* caller-provided workspace too small */

typedef void* nvx_stream_t;

/* The one definition of the ABI version: nvx_abi_version() returns it, and the Python binding
 * (nano_vllm_x_amd/_lib.py) reads this line and refuses a library built from another header.
 * Bump it whenever a signature below changes.
 *   2: nvx_decode_step_setup gained `prev_ids`.
 *   3: float16 entry points (`*_f16`) added.
 *   4: direct-peer all-reduce (`nvx_peer_*`) and sharded sampling (`nvx_sample_shard`, `nvx_sample_pick`) added. */
#define NVX_ABI_VERSION 7

/* Library identification and diagnostics. */
int nvx_abi_version(void);
const char* nvx_last_error(void);

/* ---------------------------------------------------------------------------------------
 * KV scatter.  Replaces store_kvcache / store_kvcache_kernel
 * (nanovllmx/core/kv_cache.py:12-44, called from layers/attention.py:347-352).
 * Row i of k and v (row_elems = num_kv_heads*head_dim contiguous 16-bit elements, rows
 * k_row_stride / v_row_stride elements apart) is copied to row slots[i] of k_cache / v_cache
 * (rows row_elems apart).  slots[i] == -1 skips the row.  Bit-exact copy.
 * row_elems must be a multiple of 8; all row bases 16-byte aligned. */
int nvx_store_kv(const void* k, const void* v, int64_t k_row_stride, int64_t v_row_stride,
                 void* k_cache, void* v_cache, const int64_t* slots, int64_t n_tokens,
                 int64_t row_elems, int64_t n_slots, nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * RMSNorm.  Replaces RMSNorm.forward (nanovllmx/layers/layernorm.py:17-36).
 * x: (rows, hidden) bf16, rows x_row_stride elements apart; weight: (hidden) bf16 or fp32
 * (weight_is_f32); y: (rows, hidden) bf16 contiguous.
 * With residual != NULL: acc = x + residual in fp32, residual_out = bf16(acc), and the
 * normalisation consumes the unrounded acc (layernorm.py:26-31).  residual_out may alias
 * residual, y may alias x.  hidden must be a multiple of 8 and <= 8192. */
int nvx_rmsnorm(const void* x, int64_t x_row_stride, const void* residual, const void* weight,
                int weight_is_f32, void* y, void* residual_out, int64_t rows, int64_t hidden,
                float eps, nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Rotary embedding.  Replaces RotaryEmbedding.forward
 * (nanovllmx/layers/rotary_embedding.py:74-102) for rotary_dim == head_dim == 128.
 * cos_sin: fp32 (max_pos, 128) = [cos(64) | sin(64)]; positions int64 (n_tokens).
 * Input halves (j, j+64) are paired and the result is written interleaved (2j, 2j+1); cos/sin
 * are rounded to bf16 first and every product and sum is rounded to bf16, as eager bf16
 * arithmetic does (rotary_embedding.py:16-24).  q: (n_tokens, n_q_heads*128) rows q_row_stride
 * elements apart, likewise k; outputs contiguous.  Outputs may alias inputs. */
int nvx_rope(const int64_t* positions, const float* cos_sin, int64_t max_pos,
             const void* q, int64_t q_row_stride, const void* k, int64_t k_row_stride,
             void* q_out, void* k_out, int64_t n_tokens, int n_q_heads, int n_kv_heads,
             nvx_stream_t stream);

/* Fused per-head RMSNorm(q), RMSNorm(k), rotary, and KV scatter: what
 * Qwen3Attention.forward does between the QKV projection and the attention call
 * (nanovllmx/models/qwen3.py:123-133 followed by layers/attention.py:344-352), one launch.
 * qkv: (n_tokens, (n_q_heads + 2*n_kv_heads)*128) bf16, rows qkv_row_stride apart.
 * q_out: (n_tokens, n_q_heads*128) contiguous.  Normalised+rotated k and raw v of token i go
 * to row slots[i] of k_cache / v_cache ((n_slots, n_kv_heads, 128)); slots[i] == -1 skips.
 * Intermediate roundings are those of the unfused reference ops. */
int nvx_qknorm_rope_store(const void* qkv, int64_t qkv_row_stride, const int64_t* positions,
                          const float* cos_sin, int64_t max_pos, const void* q_norm_weight,
                          const void* k_norm_weight, float eps, const int64_t* slots,
                          void* q_out, void* k_cache, void* v_cache, int64_t n_slots,
                          int64_t n_tokens, int n_q_heads, int n_kv_heads, nvx_stream_t stream);

/* SiluAndMul.  Replaces SiluAndMul.forward (nanovllmx/layers/activation.py:15-18).
 * x: (rows, 2*d) bf16 contiguous -> y: (rows, d) = bf16(bf16(silu(x[:, :d])) * x[:, d:]). */
int nvx_silu_mul(const void* x, void* y, int64_t rows, int64_t d, nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Linear layer of a decode step: y = x * w^T for m <= 256 activation rows.  Stands where the
 * reference's parallel linears call F.linear on decode batches (nanovllmx/layers/linear.py:292,
 * 467-470 under models/qwen3.py:121,135,161-163), fused with the elementwise op that follows it.
 * x: (m, k) bf16 rows x_row_stride apart; w: (n, k) bf16 row-major (an nn.Linear weight).
 * mode 0: y bf16 (m, n).
 * mode 1: w = [gate rows; up rows] (n/2 each); y bf16 (m, n/2) = SiluAndMul of the bf16-rounded
 *         projection (nanovllmx/layers/activation.py:15-18).
 * mode 2: K is cut into k_splits slices handled by different workgroups; y fp32 (k_splits, m, n)
 *         partial products, to be summed by nvx_rmsnorm_splitk.  nvx_linear_decode_splits(n, k)
 *         proposes k_splits.
 * fp32 accumulation on MFMA; k / k_splits must be 512, 768 or 1024 (nvx_linear_decode_splits
 * returns 0 when no split of k gives one of them: use a library GEMM then); mode 0 also takes
 * k = 256 and k = 1536 whole (a tensor-parallel rank's share of K in the row-parallel projections,
 * nanovllmx/layers/linear.py:266-280); output width a multiple of 16. */
int nvx_linear_decode_splits(int64_t n, int64_t k);
int nvx_linear_decode(const void* x, int64_t x_row_stride, const void* w, void* y, int64_t m,
                      int64_t n, int64_t k, int mode, int k_splits, nvx_stream_t stream);

/* RMSNorm.forward (nanovllmx/layers/layernorm.py:17-36) on the output of a K-split projection:
 * x = bf16(sum of the k_splits fp32 partials (n_splits, rows, hidden)), then as nvx_rmsnorm with a
 * bf16 weight.  residual / residual_out as in nvx_rmsnorm. */
int nvx_rmsnorm_splitk(const float* partials, int n_splits, const void* residual, const void* weight,
                       void* y, void* residual_out, int64_t rows, int64_t hidden, float eps,
                       nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Paged attention, decode: one query token per sequence.  Replaces
 * BatchDecodeWithPagedKVCacheWrapper.plan/run as called from
 * nanovllmx/layers/attention.py:120-132,197-207,277-288 (plan) and :365-369 (run),
 * page size 1, NHD layout.
 *
 * Block table: sequence b owns the slots kv_indices[kv_start[b] .. kv_start[b]+kv_len[b]).
 * A CSR table (attention.py:84-101) is the case kv_start = indptr[:-1], kv_len = diff(indptr).
 *
 * nvx_decode_plan splits every sequence into chunks of one common length so that the launch
 * fills the chip, and writes the work list (one 32-byte record per chunk, holding the chunk's
 * absolute position in kv_indices) into plan_ws (int32, >= nvx_decode_plan_ws_ints(), 16-byte
 * aligned).  It runs on the device (no host sync) and is part of the captured graph; call it
 * once per step, then nvx_paged_decode once per layer with the same kv_indices.
 *
 * nvx_paged_decode: q (batch, n_q_heads, 128) bf16 rows q_row_stride apart; k_cache/v_cache
 * (n_slots, n_kv_heads, 128) bf16 of one layer; out (batch, n_q_heads, 128) bf16 contiguous.
 * softmax(q.k * sm_scale) in fp32.  partial_ws: fp32 scratch for split sequences,
 * >= nvx_decode_partial_ws_floats() floats.  Rows with kv_len <= 0 produce zeros; slot ids outside
 * [0, n_slots) are treated as masked keys.  Two kernels stand behind it: up to 4 query heads per kv head rows go
 * pool -> registers and the products are fp32 vector arithmetic; from 8 (and at 4 over one or two kv heads) rows go
 * pool -> LDS and the products run on MFMA with the probabilities rounded to bf16, as in the prefill kernel.
 *
 * nvx_paged_decode_qkv is the same attention fed with the raw fused projection of the step instead of
 * finished queries: qkv (batch, (n_q_heads + 2*n_kv_heads)*128) as in nvx_qknorm_rope_store.  The
 * kernel applies the per-head RMSNorm and rotary embedding to the query heads itself and, for every
 * sequence whose out_slots[b] >= 0 was given to nvx_decode_plan, does the same for the new key,
 * writes the new k/v row to slot out_slots[b] and attends to it from registers; that row must be
 * the last entry of the sequence's slot list.  One launch replaces nvx_qknorm_rope_store +
 * nvx_paged_decode (nanovllmx/models/qwen3.py:123-133 + layers/attention.py:344-369).
 * out_slots may be NULL in nvx_decode_plan when only nvx_paged_decode is used. */
int64_t nvx_decode_plan_ws_ints(int64_t max_batch, int n_kv_heads);
int64_t nvx_decode_partial_ws_floats(int64_t max_batch, int n_q_heads, int n_kv_heads);
int nvx_decode_plan(const int32_t* kv_len, const int64_t* kv_start, const int64_t* out_slots,
                    int64_t batch, int n_kv_heads, int32_t* plan_ws, int64_t plan_ws_ints,
                    nvx_stream_t stream);
int nvx_paged_decode(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                     int64_t n_slots, const int32_t* kv_indices, const int32_t* plan_ws,
                     float* partial_ws, void* out, int64_t batch, int n_q_heads, int n_kv_heads,
                     float sm_scale, nvx_stream_t stream);
int nvx_paged_decode_qkv(const void* qkv, int64_t qkv_row_stride, const int64_t* positions,
                         const float* cos_sin, int64_t max_pos, const void* q_norm_weight,
                         const void* k_norm_weight, float eps, void* k_cache, void* v_cache,
                         int64_t n_slots, const int32_t* kv_indices, const int32_t* plan_ws,
                         float* partial_ws, void* out, int64_t batch, int n_q_heads, int n_kv_heads,
                         float sm_scale, nvx_stream_t stream);

/* nvx_paged_decode for callers that cannot allocate device memory (code under hipGraph capture that has no
 * workspace to hand over, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  Same arguments and the same result, bit for bit, as nvx_decode_plan(kv_len, kv_start, NULL, ...)
 * followed by nvx_paged_decode(...), but the plan and the split partials live in an array built into the
 * library (12.1 MiB of device memory per GPU the library runs on, held until the process ends) instead of
 * plan_ws / partial_ws.  Limits: batch <= nvx_paged_decode_noalloc_max_batch() (256) and
 * nvx_decode_partial_ws_floats(batch, n_q_heads, n_kv_heads) <= 3 Mi floats (met at batch 256 by 32/8, 16/2,
 * 8/1 query/kv heads; not by 64/8); larger calls return NVX_ENOSPACE.  Calls no allocator.  Every call on a GPU
 * uses that GPU's one array: calls that overlap on one GPU (two streams, two threads) race. */
int64_t nvx_paged_decode_noalloc_max_batch(void);
int nvx_paged_decode_noalloc(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                             int64_t n_slots, const int32_t* kv_indices, const int32_t* kv_len,
                             const int64_t* kv_start, void* out, int64_t batch, int n_q_heads, int n_kv_heads,
                             float sm_scale, nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Paged attention, prefill: causal, variable-length, queries right-aligned to the keys.
 * Replaces BatchPrefillWithPagedKVCacheWrapper.plan/run as called from
 * nanovllmx/layers/attention.py:104-119 and :360-364.
 * q: (n_tokens, n_q_heads, 128) rows q_row_stride apart; sequence b owns query rows
 * qo_indptr[b]..qo_indptr[b+1] and the slots kv_indices[kv_indptr[b]..kv_indptr[b+1]);
 * query i of a sequence with Lq queries and Lkv keys sees keys 0 .. i + (Lkv - Lq).
 * plan_ws: int32 scratch >= nvx_prefill_plan_ws_ints(batch); filled by nvx_prefill_plan
 * (device side, no sync).  MFMA (bf16 in, fp32 accumulate) for QK^T and PV. */
int64_t nvx_prefill_plan_ws_ints(int64_t batch);
int nvx_prefill_plan(const int32_t* qo_indptr, int64_t batch, int32_t* plan_ws,
                     int64_t plan_ws_ints, nvx_stream_t stream);
int nvx_paged_prefill(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                      int64_t n_slots, const int32_t* qo_indptr, const int32_t* kv_indptr,
                      const int32_t* kv_indices, const int32_t* plan_ws, void* out,
                      int64_t n_tokens, int64_t batch, int n_q_heads, int n_kv_heads,
                      float sm_scale, nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Sampler.  Replaces Sampler.forward with all four parameter tensors present
 * (nanovllmx/layers/sampler.py:39-55, as called from core/model_runner.py:266-272).
 * logits: (batch, vocab) bf16 or fp32 (logits_is_f32), rows logits_row_stride apart.
 * Per row: x = logits / temperature; top-k keeps x >= (k-th largest x); top-p drops the
 * longest ascending prefix whose softmax mass is <= 1 - top_p (never the maximum); min-p
 * drops softmax(x) < min_p * max; the token is drawn from softmax of what is left.
 * The draw uses Philox4x32-10(seed, offset) Gumbel noise per (row, token): exact in
 * distribution, reproducible for a given (seed, offset), not bit-comparable with
 * torch.multinomial.  out_ids: int64 (batch).  Equal scores are ordered by token id wherever the
 * reference's sort order decides (top-p boundary), as a stable ascending sort does.
 * No sort: thresholds come from radix selection over the row.
 * ws (optional, may be NULL): >= nvx_sample_ws_bytes(batch) bytes, 8-byte aligned.  With it, rows
 * that have no active filter are cut into slices processed by several workgroups each (the draw is
 * the same token either way); without it every row is one workgroup.
 * nvx_sampler_probs writes the masked, renormalised distribution itself (fp32 (batch, vocab)):
 * the part of the reference computation that is comparable up to fp32 rounding. */
int64_t nvx_sample_ws_bytes(int64_t max_batch);
int nvx_sample(const void* logits, int logits_is_f32, int64_t logits_row_stride,
               const float* temperatures, const int64_t* top_ks, const float* top_ps,
               const float* min_ps, int64_t batch, int64_t vocab, uint64_t seed, uint64_t offset,
               int64_t* out_ids, void* ws, int64_t ws_bytes, nvx_stream_t stream);
int nvx_sampler_probs(const void* logits, int logits_is_f32, int64_t logits_row_stride,
                      const float* temperatures, const int64_t* top_ks, const float* top_ps,
                      const float* min_ps, int64_t batch, int64_t vocab, float* probs_out,
                      nvx_stream_t stream);

/* Sampling on tensor-parallel ranks without gathering the logits.  The reference's LM head gathers every rank's
 * (batch, vocab / tp) logits into full rows before Sampler.forward (nanovllmx/layers/vocab_parallel_embedding.py:147-151,
 * core/model_runner.py:262-272).  For rows without a filter (top_k = all, top_p = 1, min_p = 0) and greedy rows the
 * draw decomposes: nvx_sample_shard runs the draw of nvx_sample over this rank's shard -- logits (batch, shard_vocab),
 * whose token j is token vocab_offset + j of the whole row (vocab_offset a multiple of 4; tokens >= vocab are padding
 * and never drawn) -- and leaves one 64-bit (score, token) key per row in winner_out; the ranks exchange those keys
 * (8 bytes per row) and nvx_sample_pick takes the largest of the n_sets keys of each row (winners is (n_sets, batch)):
 * the same token, bit for bit, that nvx_sample draws from the gathered row with the same (seed, offset).
 * ws as for nvx_sample (required here). */
int nvx_sample_shard(const void* logits, int logits_is_f32, int64_t logits_row_stride,
                     const float* temperatures, int64_t batch, int64_t shard_vocab, int64_t vocab_offset,
                     int64_t vocab, uint64_t seed, uint64_t offset, uint64_t* winner_out, void* ws,
                     int64_t ws_bytes, nvx_stream_t stream);
int nvx_sample_pick(const uint64_t* winners, int64_t n_sets, int64_t batch, int64_t* out_ids,
                    nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Device-resident block table for graph-replayed decode steps.  Replaces the per-step rebuild
 * of the CSR buffers and model inputs from Python lists in
 * AttentionBackend.build_metadata_for_cuda_graph_replay (nanovllmx/layers/attention.py:219-275)
 * and ModelRunner.prepare_input (core/model_runner.py:175-189) with an incremental update
 * driven by one small host-to-device copy per step.
 *
 * table: int32 (n_rows, row_capacity); a running sequence keeps one row for its lifetime.
 *
 * nvx_block_table_write_rows (after a prefill): for b < batch copies
 * kv_indices[kv_indptr[b] .. kv_indptr[b+1]) into table[rows[b]][0 ..).
 *
 * nvx_decode_step_setup (first node of a decode graph): `step` is the int32 buffer the host
 * uploaded: step[0] = batch (live rows, <= padded_batch), then four arrays of `stride` entries
 * starting at step[4]: row, pos, slot, token.  For i < batch:
 *   table[row[i]][pos[i]] = slot[i]; kv_start[i] = row[i]*row_capacity; kv_len[i] = pos[i]+1;
 *   out_slots[i] = slot[i]; positions[i] = pos[i]; input_ids[i] = token[i] -- or, for a negative
 *   token -1-j, prev_ids[j]: the id the step before sampled for its row j (`prev_ids`, nullable, holds
 *   at least `stride` ids and is read when the kernel runs, so a step can be queued before the ids of
 *   the step before have reached the host).
 * For batch <= i < padded_batch (padding rows of a captured graph, attention.py:240-265):
 *   kv_start[i] = pad_start, kv_len[i] = 1, out_slots[i] = -1, positions[i] = input_ids[i] = 0. */
int nvx_block_table_write_rows(const int32_t* kv_indices, const int32_t* kv_indptr,
                               const int32_t* rows, int64_t batch, int32_t* table, int64_t n_rows,
                               int64_t row_capacity, nvx_stream_t stream);
int nvx_decode_step_setup(const int32_t* step, int64_t stride, int64_t padded_batch,
                          int32_t* table, int64_t n_rows, int64_t row_capacity, int64_t pad_start,
                          int64_t* kv_start, int32_t* kv_len, int64_t* out_slots,
                          int64_t* positions, int64_t* input_ids, const int64_t* prev_ids,
                          nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * LM head of a decode step that draws as it computes: logits = x (batch, k) . W^T (n_rows, k), then one token per
 * row, in ONE pass over W.  Replaces, for rows without top-k / top-p / min-p, ParallelLMHead.forward followed by
 * Sampler.forward (nanovllmx/layers/vocab_parallel_embedding.py:147-151, core/model_runner.py:262-272,
 * layers/sampler.py:39-55): the logits never travel to memory and back.
 *
 * nvx_lmhead_pack repacks W (n_rows, k) 16-bit row-major, once at load time, into tiles of 16 rows whose 16-byte
 * pieces are in matrix-core operand order (nvx_lmhead_packed_elems() elements; rows past n_rows are zeros), so that
 * the kernel's loads are whole 1 KB lines that go straight into operand registers.
 *
 * nvx_lmhead_sample: x rows x_row_stride elements apart, batch <= nvx_lmhead_max_rows(), k = 1024.  The batch stays in
 * the LDS of every compute unit, nvx_lmhead_rows_per_pass() rows at a time.  Up to twice that many rows go out as two row
 * groups in one launch, on workgroups paired within an XCD that take the same tiles (fetched from HBM once, found in L2
 * the second time); a taller batch takes one pass over the weight per group of rows, one after the other (measured on
 * Qwen3-0.6B: up to 128 rows this beats the projection + sampler, beyond it does not).  Results do not depend on how the
 * rows were grouped.  Row r of W is token vocab_offset + r of a vocabulary of `vocab` tokens (a
 * tensor-parallel rank passes its shard; tokens >= vocab are padding and never drawn).  Every logit is rounded to the
 * 16-bit format, as the projection would have stored it; the draw is nvx_sample's (same noise per (seed, offset,
 * row, token), same tie rule; temperature <= 0: arg-max), so key_out / out_ids equal what nvx_sample_shard /
 * nvx_sample give on those logits.  key_out (batch) receives the 64-bit (score, token) keys (for nvx_sample_pick
 * over the ranks' keys), out_ids (batch) the token ids; either may be NULL.  logits_out (batch rows of
 * logits_row_stride >= n_rows padded to 16 elements), if not NULL, receives the rounded logits.
 * ws: >= nvx_lmhead_ws_bytes(batch) bytes, 8-byte aligned.  No allocation, no host sync, capturable. */
int64_t nvx_lmhead_max_rows(void);
int64_t nvx_lmhead_rows_per_pass(void);
int64_t nvx_lmhead_packed_elems(int64_t n_rows, int64_t k);
int64_t nvx_lmhead_ws_bytes(int64_t batch);
int nvx_lmhead_pack(const void* w, int64_t n_rows, int64_t k, void* packed, nvx_stream_t stream);
int nvx_lmhead_sample(const void* w_packed, int64_t n_rows, int64_t k, const void* x, int64_t x_row_stride,
                      int64_t batch, const float* temperatures, int64_t vocab_offset, int64_t vocab,
                      uint64_t seed, uint64_t offset, void* ws, int64_t ws_bytes, uint64_t* key_out,
                      int64_t* out_ids, void* logits_out, int64_t logits_row_stride, nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Projection of a decode step over a weight too large for nvx_linear_decode's weight-stationary kernel (tens of MB: the
 * projections of Qwen3-8B and larger; the reference's F.linear, nanovllmx/layers/linear.py:73,169,276): the weight, packed
 * by nvx_lmhead_pack (k a multiple of 1024), is streamed once with the batch's rows (m <= 64) resident in LDS, one
 * 1024-wide slice of k per workgroup.  partials (k / 1024, m, n) fp32 receives each slice's share of x . W^T; the
 * consumer sums them: nvx_rmsnorm_splitk after a row-parallel projection, or nvx_splitk_sum -- (rows, n) in the 16-bit
 * format, or with silu_mul != 0 the SiluAndMul (nanovllmx/layers/activation.py:15-18) of a [gate | up] projection,
 * (rows, n / 2), the sum rounded to the 16-bit format first as the unsplit projection would have stored it. */
int nvx_linear_stream(const void* w_packed, int64_t n, int64_t k, const void* x, int64_t x_row_stride, int64_t m,
                      float* partials, nvx_stream_t stream);
int nvx_splitk_sum(const float* partials, int n_splits, int64_t rows, int64_t n, int silu_mul, void* y,
                   nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The same projections at up to 256 rows (F.linear of the parallel linears on decode batches,
 * nanovllmx/layers/linear.py:167-176,266-280, with SiluAndMul, nanovllmx/layers/activation.py:15-18): the weight, packed by
 * nvx_lmhead_pack, is streamed once by workgroups that own 128 of its rows over one slice of k, while the batch's rows
 * pass through LDS in chunks of 128 k.  k must be a multiple of 512 x k_slices.
 * mode 0: y 16-bit (m, n).   mode 1: the weight is [gate rows; up rows]; y 16-bit (m, n / 2) = SiluAndMul of the
 * 16-bit-rounded projection.   mode 2: y fp32 (k_slices, m, n) partial products for nvx_rmsnorm_splitk / nvx_splitk_sum;
 * nvx_linear_block_slices(n, k, mode) proposes k_slices (1 for modes 0 and 1; 0 = shape not supported). */
int64_t nvx_linear_block_max_rows(void);
int nvx_linear_block_slices(int64_t n, int64_t k, int mode);
int nvx_linear_block(const void* w_packed, int64_t n, int64_t k, const void* x, int64_t x_row_stride, int64_t m,
                     int mode, int k_slices, void* y, nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Tensor-parallel all-reduce of a decode step, over memory the peers have mapped, fused with the
 * RMSNorm that follows it.  Replaces the NCCL all-reduce behind the row-parallel linears and the
 * vocab-parallel embedding (nanovllmx/distributed/communication_op.py:41-52,153-156, called from
 * layers/linear.py:277-278 and layers/vocab_parallel_embedding.py:105) together with the
 * RMSNorm.forward that consumes its result (layers/layernorm.py:17-36, models/qwen3.py:179-186).
 * A ring all-reduce of a decode-sized tensor (<= 2 MiB) is bound by 2(tp-1) link latencies; here
 * every rank writes its rows into every peer's landing buffer at once (the xGMI mesh is fully
 * connected) and then reads only its own memory.
 *
 * A group is one rank's end of the exchange among tp_size <= 8 ranks (one process per GPU):
 *   nvx_peer_create   allocates this rank's landing buffer on the current device (uncached device
 *                     memory, nvx_peer_exchange_bytes() bytes), a per-row exchange counter and a
 *                     status word in pinned host memory.  These three calls are the only ones in
 *                     the library that allocate; they run once, at start-up.  timeout_ms bounds
 *                     every wait inside the kernels (<= 0: 10 s).
 *   nvx_peer_export   writes the 64-byte inter-process handle of the landing buffer; the host sends
 *                     it to every peer by whatever channel it has (torch.distributed's object
 *                     all-gather in this package).
 *   nvx_peer_import   maps peer_rank's landing buffer from its handle (hipIpcOpenMemHandle).  All
 *                     tp_size - 1 peers must be imported before the first exchange.
 *   nvx_peer_status   0, or 1 once any wait of any exchange ran out (a peer died or fell more than
 *                     timeout_ms behind): results since then are not to be trusted.
 *   nvx_peer_destroy  unmaps and frees; every rank must have finished its exchanges (the host
 *                     synchronises the ranks first).
 *
 * nvx_peer_allreduce_norm: x (rows, hidden) 16-bit, rows x_row_stride elements apart, is this
 * rank's partial result; s = the sum over ranks, added in fp32 in rank order (identical bits on
 * every rank) and rounded to the 16-bit format, as the all-reduce of 16-bit tensors hands it over.
 *   weight == NULL:               y = s                      (the all-reduce alone; y may alias x)
 *   weight, residual == NULL:     y = RMSNorm(s) * weight
 *   weight, residual:             acc = s + residual in fp32, residual_out = round(acc),
 *                                 y = RMSNorm(acc) * weight  (as nvx_rmsnorm)
 * rows <= max_rows and hidden as given to nvx_peer_create; every rank of the group must make the
 * same sequence of calls with the same rows (they run the same scheduler).  No allocation, no
 * host synchronisation, hipGraph-capturable: the exchange counter lives in device memory and is
 * advanced by the kernel.  One exchange at a time per group (one stream). */
int64_t nvx_peer_exchange_bytes(int tp_size, int64_t max_rows, int64_t hidden);
int nvx_peer_create(int tp_size, int rank, int64_t max_rows, int64_t hidden, int64_t timeout_ms,
                    void** comm_out);
int nvx_peer_export(void* comm, unsigned char* handle_out_64_bytes);
int nvx_peer_import(void* comm, int peer_rank, const unsigned char* handle_64_bytes);
int nvx_peer_status(void* comm);
int nvx_peer_destroy(void* comm);
int nvx_peer_allreduce_norm(void* comm, const void* x, int64_t x_row_stride, const void* residual,
                            const void* weight, void* y, void* residual_out, int64_t rows,
                            int64_t hidden, float eps, nvx_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * IEEE half (float16) builds of the entry points whose arithmetic depends on the 16-bit format.  The reference
 * resolves the dtype from the checkpoint and defaults to float16 (nanovllmx/core/model_runner.py:14-23,38-39).
 * Same arguments and semantics as the entry point without the suffix, with "16-bit floating point" meaning
 * float16 for activations, weights, KV rows and outputs: conversions round to nearest even into float16 (the
 * intermediate roundings of the eager float16 ops), q.k by v_dot2_f32_f16, the MFMA products in their f16 forms.
 * Built from the same sources as the bfloat16 entry points (compiled with -DNVX_F16).  The other entry points --
 * KV scatter, the plans and workspace sizes, block table, the sampler (fp32 logits) -- serve both formats;
 * nvx_paged_decode_noalloc exists for bfloat16 only. */
int nvx_rmsnorm_f16(const void* x, int64_t x_row_stride, const void* residual, const void* weight,
                int weight_is_f32, void* y, void* residual_out, int64_t rows, int64_t hidden,
                float eps, nvx_stream_t stream);
int nvx_rope_f16(const int64_t* positions, const float* cos_sin, int64_t max_pos,
             const void* q, int64_t q_row_stride, const void* k, int64_t k_row_stride,
             void* q_out, void* k_out, int64_t n_tokens, int n_q_heads, int n_kv_heads,
             nvx_stream_t stream);
int nvx_qknorm_rope_store_f16(const void* qkv, int64_t qkv_row_stride, const int64_t* positions,
                          const float* cos_sin, int64_t max_pos, const void* q_norm_weight,
                          const void* k_norm_weight, float eps, const int64_t* slots,
                          void* q_out, void* k_cache, void* v_cache, int64_t n_slots,
                          int64_t n_tokens, int n_q_heads, int n_kv_heads, nvx_stream_t stream);
int nvx_silu_mul_f16(const void* x, void* y, int64_t rows, int64_t d, nvx_stream_t stream);
int nvx_linear_decode_f16(const void* x, int64_t x_row_stride, const void* w, void* y, int64_t m,
                      int64_t n, int64_t k, int mode, int k_splits, nvx_stream_t stream);
int nvx_rmsnorm_splitk_f16(const float* partials, int n_splits, const void* residual, const void* weight,
                       void* y, void* residual_out, int64_t rows, int64_t hidden, float eps,
                       nvx_stream_t stream);
int nvx_paged_decode_f16(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                     int64_t n_slots, const int32_t* kv_indices, const int32_t* plan_ws,
                     float* partial_ws, void* out, int64_t batch, int n_q_heads, int n_kv_heads,
                     float sm_scale, nvx_stream_t stream);
int nvx_paged_decode_qkv_f16(const void* qkv, int64_t qkv_row_stride, const int64_t* positions,
                         const float* cos_sin, int64_t max_pos, const void* q_norm_weight,
                         const void* k_norm_weight, float eps, void* k_cache, void* v_cache,
                         int64_t n_slots, const int32_t* kv_indices, const int32_t* plan_ws,
                         float* partial_ws, void* out, int64_t batch, int n_q_heads, int n_kv_heads,
                         float sm_scale, nvx_stream_t stream);
int nvx_paged_prefill_f16(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                      int64_t n_slots, const int32_t* qo_indptr, const int32_t* kv_indptr,
                      const int32_t* kv_indices, const int32_t* plan_ws, void* out,
                      int64_t n_tokens, int64_t batch, int n_q_heads, int n_kv_heads,
                      float sm_scale, nvx_stream_t stream);
int nvx_peer_allreduce_norm_f16(void* comm, const void* x, int64_t x_row_stride, const void* residual,
                            const void* weight, void* y, void* residual_out, int64_t rows,
                            int64_t hidden, float eps, nvx_stream_t stream);
int nvx_linear_stream_f16(const void* w_packed, int64_t n, int64_t k, const void* x, int64_t x_row_stride, int64_t m,
                          float* partials, nvx_stream_t stream);
int nvx_splitk_sum_f16(const float* partials, int n_splits, int64_t rows, int64_t n, int silu_mul, void* y,
                       nvx_stream_t stream);
int nvx_linear_block_f16(const void* w_packed, int64_t n, int64_t k, const void* x, int64_t x_row_stride, int64_t m,
                         int mode, int k_slices, void* y, nvx_stream_t stream);
int nvx_lmhead_sample_f16(const void* w_packed, int64_t n_rows, int64_t k, const void* x, int64_t x_row_stride,
                          int64_t batch, const float* temperatures, int64_t vocab_offset, int64_t vocab,
                          uint64_t seed, uint64_t offset, void* ws, int64_t ws_bytes, uint64_t* key_out,
                          int64_t* out_ids, void* logits_out, int64_t logits_row_stride, nvx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NVX_H */

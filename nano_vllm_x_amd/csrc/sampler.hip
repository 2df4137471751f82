// Sampler for gfx950: temperature, top-k, top-p, min-p, categorical draw -- without sorting the vocabulary.
//
//   * rows with no active filter (top_k >= vocab, top_p >= 1, min_p <= 0): ONE pass,
//     token = argmax_i(logit_i / T + Gumbel_i): an exact categorical draw from softmax(logits / T).
//     The pass is bound by the Philox and log arithmetic of one compute unit, so with a caller
//     workspace a row is cut into up to 16 slices, one 1024-thread workgroup each; every slice
//     leaves its (score, token) winner as one 64-bit key and a second small kernel takes the
//     maximum per row;
//   * filtered rows: one 1024-thread workgroup per row; the row (<= 608 KB) is re-read from L2 on
//     every pass;
//   * otherwise the reference's sort is replaced by radix selection on the order-preserving integer
//     image of the fp32 scores (4 passes of 8 bits, 256-bin LDS histograms):
//       top-k  -> the k-th largest score (counts);
//       top-p  -> the ascending-order boundary of cumulative softmax mass (masses accumulated as
//                 32.32 fixed point in 64-bit LDS atomics, so the result does not depend on the order
//                 in which lanes arrive); equal scores are ordered by token id, as a stable sort does;
//       min-p  -> a threshold on softmax probability.
// Random numbers: Philox4x32-10 keyed by `seed`, counter = (token/4, row, offset).
#include "nvx_dev.h"

#define SMP_THREADS 1024
#define SMP_MAX_SPLIT 16

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }

__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(M0, ctr.x), lo0 = M0 * ctr.x;
        const uint32_t hi1 = mulhi32(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0; key.y += W1;
    }
    return ctr;
}

// Gumbel noise from 23 random bits.  (r >> 9) + 0.5 is exact in fp32 (24 significant bits), so u lies
// strictly inside (0, 1); with 24 bits the top value rounds to u = 1, -log(u) = 0 and the noise becomes
// +infinity: that token would win whatever its logit.  The inner logarithm is clamped at 2^-25 as well, in
// case the fast logarithm returns 0 just below 1 (the noise is then at most 17.3).
__device__ __forceinline__ float gumbel_from_bits(uint32_t r) {
    const float u = ((float)(r >> 9) + 0.5f) * (1.0f / 8388608.0f);
    return -__logf(fmaxf(-__logf(u), 2.98023224e-8f));
}

// order-preserving map fp32 -> uint32 (ascending)
__device__ __forceinline__ uint32_t order_key(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// (score, token) as one 64-bit key: larger score wins, then the lower token id
__device__ __forceinline__ unsigned long long pack_winner(float v, int idx) {
    return ((unsigned long long)order_key(v) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)idx);
}

template <bool F32>
__device__ __forceinline__ float load_logit(const void* row, int64_t i) {
    if (F32) return reinterpret_cast<const float*>(row)[i];
    return bf16bits_to_f32(reinterpret_cast<const uint16_t*>(row)[i]);
}

struct BlockScratch {
    float fred[16];
    unsigned long long ured[16];
    int ired[16];
    unsigned long long hist64[256];
    unsigned int hist32[256];
    unsigned int sel_digit;
    unsigned long long sel_before;
    unsigned long long sel_bucket;
    int found;
};

__device__ __forceinline__ float block_max(float v, BlockScratch& s) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s.fred[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = s.fred[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) r = fmaxf(r, s.fred[i]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, BlockScratch& s) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s.fred[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += s.fred[i];
    return r;
}
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, BlockScratch& s) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s.ured[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += s.ured[i];
    return r;
}

// (value, index) argmax with lowest index on ties
__device__ __forceinline__ void block_argmax(float& v, int& idx, BlockScratch& s) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(v, m, 64);
        const int oi = __shfl_xor(idx, m, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s.fred[threadIdx.x >> 6] = v; s.ired[threadIdx.x >> 6] = idx; }
    __syncthreads();
    v = s.fred[0]; idx = s.ired[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        const float ov = s.fred[i]; const int oi = s.ired[i];
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

__device__ __forceinline__ unsigned long long mass_fx(float x, float m) {
    return (unsigned long long)(__expf(x - m) * 4294967296.0f);
}

template <bool F32, bool PROBS>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(
    const void* __restrict__ logits, int64_t row_stride, const float* __restrict__ temps, const int64_t* __restrict__ top_ks,
    const float* __restrict__ top_ps, const float* __restrict__ min_ps, int vocab, uint64_t seed, uint64_t offset,
    int64_t* __restrict__ out_ids, float* __restrict__ probs_out, unsigned long long* __restrict__ winners, int n_split) {
    __shared__ BlockScratch sc;
    const int row = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
    const void* lrow = F32 ? (const void*)((const float*)logits + row * row_stride)
                           : (const void*)((const uint16_t*)logits + row * row_stride);
    const float T = temps[row];
    int64_t k64 = top_ks[row];
    const int k = (int)(k64 < 1 ? 1 : (k64 > vocab ? vocab : k64));
    const float top_p = top_ps[row], min_p = min_ps[row];
    const bool filtered = k < vocab || top_p < 1.0f || min_p > 0.0f;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));

    if (!filtered && !PROBS) {
        // single pass: Gumbel-max over this workgroup's slice.  Four tokens share one Philox block.
        const int per = ((vocab + n_split - 1) / n_split + 3) & ~3;
        const int lo = slice * per, hi = min(vocab, lo + per);
        const float inv_t = 1.0f / T;
        float best = -INFINITY; int besti = 0x7fffffff;
        for (int base = lo + tid * 4; base < hi; base += SMP_THREADS * 4) {
            const uint4 r = philox4x32_10(make_uint4((uint32_t)(base >> 2), (uint32_t)row, (uint32_t)offset, (uint32_t)(offset >> 32)), key);
            const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = base + j;
                if (i < hi) {
                    const float v = load_logit<F32>(lrow, i) * inv_t + gumbel_from_bits(rr[j]);
                    if (v > best) { best = v; besti = i; }
                }
            }
        }
        block_argmax(best, besti, sc);
        if (tid == 0) {
            if (winners) winners[(int64_t)row * n_split + slice] = besti == 0x7fffffff ? 0ull : pack_winner(best, besti);
            else out_ids[row] = besti == 0x7fffffff ? 0 : besti;
        }
        return;
    }
    if (slice > 0) {   // a filtered row is handled whole by its first workgroup
        if (tid == 0) winners[(int64_t)row * n_split + slice] = 0ull;
        return;
    }

    // ---- pass: row maximum of x = logit / T ----
    float mloc = -INFINITY;
    for (int i = tid; i < vocab; i += SMP_THREADS) mloc = fmaxf(mloc, load_logit<F32>(lrow, i) / T);
    const float M = block_max(mloc, sc);

    // ---- top-k: key of the k-th largest score = ascending rank vocab - k ----
    uint32_t kth_key = 0;
    if (k < vocab) {
        uint32_t prefix = 0;
        unsigned int rank = (unsigned int)(vocab - k);
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) sc.hist32[tid] = 0;
            __syncthreads();
            for (int i = tid; i < vocab; i += SMP_THREADS) {
                const uint32_t ky = order_key(load_logit<F32>(lrow, i) / T);
                if (pass == 0 || (ky >> (shift + 8)) == prefix) atomicAdd(&sc.hist32[(ky >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned int cum = 0; int d = 0;
                for (; d < 255; ++d) { if (cum + sc.hist32[d] > rank) break; cum += sc.hist32[d]; }
                sc.sel_digit = (unsigned int)d;
                sc.sel_before = cum;
            }
            __syncthreads();
            prefix = (prefix << 8) | sc.sel_digit;
            rank -= (unsigned int)sc.sel_before;
            __syncthreads();
        }
        kth_key = prefix;
    }

    // ---- top-p: boundary of cumulative mass in ascending order ----
    uint32_t p_key = 0;       // scores with key < p_key are dropped
    int p_cut = -1;           // among scores with key == p_key, token ids < p_cut are dropped
    if (top_p < 1.0f) {
        unsigned long long zloc = 0;
        for (int i = tid; i < vocab; i += SMP_THREADS) {
            const float x = load_logit<F32>(lrow, i) / T;
            if (order_key(x) >= kth_key) zloc += mass_fx(x, M);
        }
        const unsigned long long Z = block_sum_u64(zloc, sc);
        double thr_d = (double)(1.0f - top_p) * (double)Z;
        unsigned long long thr = thr_d <= 0.0 ? 0ull : (unsigned long long)thr_d;
        if (thr >= Z) thr = Z > 0 ? Z - 1 : 0;   // the largest score always survives
        uint32_t prefix = 0;
        unsigned long long before = 0;   // mass of scores strictly below the current prefix
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) sc.hist64[tid] = 0;
            __syncthreads();
            for (int i = tid; i < vocab; i += SMP_THREADS) {
                const float x = load_logit<F32>(lrow, i) / T;
                const uint32_t ky = order_key(x);
                if (ky >= kth_key && (pass == 0 || (ky >> (shift + 8)) == prefix))
                    atomicAdd(&sc.hist64[(ky >> shift) & 255u], mass_fx(x, M));
            }
            __syncthreads();
            if (tid == 0) {
                unsigned long long cum = before; int d = 0;
                for (; d < 255; ++d) { if (cum + sc.hist64[d] > thr) break; cum += sc.hist64[d]; }
                sc.sel_digit = (unsigned int)d;
                sc.sel_before = cum;
                sc.sel_bucket = sc.hist64[d];
            }
            __syncthreads();
            prefix = (prefix << 8) | sc.sel_digit;
            before = sc.sel_before;
            __syncthreads();
        }
        p_key = prefix;
        // ties at the boundary score: drop the lowest token ids while the running mass stays <= thr
        const unsigned long long bucket = sc.sel_bucket;
        int cnt_loc = 0; unsigned long long one = 0;
        for (int i = tid; i < vocab; i += SMP_THREADS) {
            const float x = load_logit<F32>(lrow, i) / T;
            if (order_key(x) == p_key) { ++cnt_loc; one = mass_fx(x, M); }
        }
        // every tie has the same mass; take it from any thread that saw one
        unsigned long long one_all = one;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(one_all, m, 64); one_all = one_all > o ? one_all : o; }
        __syncthreads();
        if ((tid & 63) == 0) sc.ured[tid >> 6] = one_all;
        __syncthreads();
        unsigned long long pt = 0;
        for (int i = 0; i < 16; ++i) pt = pt > sc.ured[i] ? pt : sc.ured[i];
        (void)bucket; (void)cnt_loc;
        const unsigned long long room = thr >= before ? thr - before : 0;
        const long long n_drop = pt > 0 ? (long long)(room / pt) : 0;
        if (n_drop > 0) {
            // token id of the tie with ascending-id rank n_drop: walk the row in id order
            if (tid == 0) { sc.found = -1; sc.sel_before = 0; }
            __syncthreads();
            for (int base = 0; base < vocab; base += SMP_THREADS) {
                const int i = base + tid;
                const bool tie = i < vocab && order_key(load_logit<F32>(lrow, i) / T) == p_key;
                const unsigned long long bal = __ballot(tie);
                const int within = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
                __syncthreads();
                if ((tid & 63) == 0) sc.ired[tid >> 6] = __popcll(bal);
                __syncthreads();
                int wave_before = 0, total = 0;
                for (int w = 0; w < 16; ++w) { if (w < (tid >> 6)) wave_before += sc.ired[w]; total += sc.ired[w]; }
                const long long seen = (long long)sc.sel_before;
                if (tie && seen + wave_before + within == n_drop) sc.found = i;
                __syncthreads();
                if (tid == 0) sc.sel_before = (unsigned long long)(seen + total);
                __syncthreads();
                if (sc.found >= 0) break;
            }
            p_cut = sc.found >= 0 ? sc.found : vocab;
            __syncthreads();
        }
    }

    auto kept_kp = [&](float x, int i) -> bool {
        const uint32_t ky = order_key(x);
        if (ky < kth_key) return false;
        if (top_p < 1.0f) { if (ky < p_key) return false; if (ky == p_key && i < p_cut) return false; }
        return true;
    };

    // ---- min-p on softmax of what survived ----
    float minp_thr = -1.0f, Z2 = 1.0f;
    if (min_p > 0.0f) {
        float zl = 0.f;
        for (int i = tid; i < vocab; i += SMP_THREADS) {
            const float x = load_logit<F32>(lrow, i) / T;
            if (kept_kp(x, i)) zl += __expf(x - M);
        }
        Z2 = block_sum(zl, sc);
        minp_thr = min_p * (1.0f / Z2);
    }
    auto kept = [&](float x, int i) -> bool {
        if (!kept_kp(x, i)) return false;
        if (min_p > 0.0f && (__expf(x - M) / Z2) < minp_thr) return false;
        return true;
    };

    if (PROBS) {
        float zl = 0.f;
        for (int i = tid; i < vocab; i += SMP_THREADS) {
            const float x = load_logit<F32>(lrow, i) / T;
            if (kept(x, i)) zl += __expf(x - M);
        }
        const float Z3 = block_sum(zl, sc);
        float* prow = probs_out + (int64_t)row * vocab;
        for (int i = tid; i < vocab; i += SMP_THREADS) {
            const float x = load_logit<F32>(lrow, i) / T;
            prow[i] = kept(x, i) ? __expf(x - M) / Z3 : 0.0f;
        }
        return;
    }

    float best = -INFINITY; int besti = 0x7fffffff;
    for (int base = tid * 4; base < vocab; base += SMP_THREADS * 4) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)(base >> 2), (uint32_t)row, (uint32_t)offset, (uint32_t)(offset >> 32)), key);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + j;
            if (i < vocab) {
                const float x = load_logit<F32>(lrow, i) / T;
                if (kept(x, i)) {
                    const float v = x + gumbel_from_bits(rr[j]);
                    if (v > best || (v == best && i < besti)) { best = v; besti = i; }
                }
            }
        }
    }
    block_argmax(best, besti, sc);
    if (tid == 0) {
        if (winners) winners[(int64_t)row * n_split] = besti == 0x7fffffff ? 0ull : pack_winner(best, besti);
        else out_ids[row] = besti == 0x7fffffff ? 0 : besti;
    }
}

// Per row: the best of the slices' winners.
__global__ __launch_bounds__(256) void sample_finish_kernel(const unsigned long long* __restrict__ winners, int n_split, int batch,
                                                            int64_t* __restrict__ out_ids) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= batch) return;
    unsigned long long w = 0;
    for (int s = 0; s < n_split; ++s) { const unsigned long long o = winners[(int64_t)row * n_split + s]; w = o > w ? o : w; }
    out_ids[row] = w == 0 ? 0 : (int64_t)(0xffffffffu - (uint32_t)(w & 0xffffffffull));
}

static int sampler_check(const void* logits, int64_t stride, const float* t, const int64_t* k, const float* p, const float* mp,
                         int64_t batch, int64_t vocab, const char* who) {
    NVX_REQUIRE(batch >= 0 && vocab > 0 && vocab < 0x7fffff00, NVX_EINVAL, "%s: bad size", who);
    NVX_REQUIRE(batch == 0 || (logits && t && k && p && mp), NVX_EINVAL, "%s: null pointer", who);
    NVX_REQUIRE(stride >= vocab, NVX_EINVAL, "%s: row stride smaller than the vocabulary", who);
    return NVX_OK;
}

extern "C" int64_t nvx_sample_ws_bytes(int64_t max_batch) { return max_batch * SMP_MAX_SPLIT * 8; }

extern "C" int nvx_sample(const void* logits, int logits_is_f32, int64_t logits_row_stride, const float* temperatures,
                          const int64_t* top_ks, const float* top_ps, const float* min_ps, int64_t batch, int64_t vocab,
                          uint64_t seed, uint64_t offset, int64_t* out_ids, void* ws, int64_t ws_bytes, nvx_stream_t stream) {
    int rc = sampler_check(logits, logits_row_stride, temperatures, top_ks, top_ps, min_ps, batch, vocab, "nvx_sample");
    if (rc) return rc;
    if (batch == 0) return NVX_OK;
    NVX_REQUIRE(out_ids, NVX_EINVAL, "nvx_sample: null pointer");
    NVX_REQUIRE(batch <= 65535, NVX_EUNSUPPORTED, "nvx_sample: more than 65535 rows");
    // slices per row: enough workgroups for two per compute unit, none without a workspace
    int n_split = 1;
    if (ws) {
        NVX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7u) == 0, NVX_EINVAL, "nvx_sample: workspace must be 8-byte aligned");
        n_split = (int)(512 / batch);
        n_split = n_split < 1 ? 1 : (n_split > SMP_MAX_SPLIT ? SMP_MAX_SPLIT : n_split);
        NVX_REQUIRE(ws_bytes >= batch * n_split * 8, NVX_ENOSPACE, "nvx_sample: workspace too small (%lld < %lld bytes)",
                    (long long)ws_bytes, (long long)(batch * n_split * 8));
    }
    unsigned long long* winners = (unsigned long long*)ws;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n_split, (unsigned)batch);
    if (logits_is_f32)
        hipLaunchKernelGGL((sample_kernel<true, false>), grid, dim3(SMP_THREADS), 0, st, logits, logits_row_stride,
                           temperatures, top_ks, top_ps, min_ps, (int)vocab, seed, offset, out_ids, (float*)nullptr, winners, n_split);
    else
        hipLaunchKernelGGL((sample_kernel<false, false>), grid, dim3(SMP_THREADS), 0, st, logits, logits_row_stride,
                           temperatures, top_ks, top_ps, min_ps, (int)vocab, seed, offset, out_ids, (float*)nullptr, winners, n_split);
    rc = nvx_check_launch("nvx_sample");
    if (rc || !ws) return rc;
    hipLaunchKernelGGL(sample_finish_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, winners, n_split, (int)batch, out_ids);
    return nvx_check_launch("nvx_sample(finish)");
}

extern "C" int nvx_sampler_probs(const void* logits, int logits_is_f32, int64_t logits_row_stride, const float* temperatures,
                                 const int64_t* top_ks, const float* top_ps, const float* min_ps, int64_t batch, int64_t vocab,
                                 float* probs_out, nvx_stream_t stream) {
    int rc = sampler_check(logits, logits_row_stride, temperatures, top_ks, top_ps, min_ps, batch, vocab, "nvx_sampler_probs");
    if (rc) return rc;
    if (batch == 0) return NVX_OK;
    NVX_REQUIRE(probs_out, NVX_EINVAL, "nvx_sampler_probs: null pointer");
    NVX_REQUIRE(batch <= 65535, NVX_EUNSUPPORTED, "nvx_sampler_probs: more than 65535 rows");
    hipStream_t st = (hipStream_t)stream;
    if (logits_is_f32)
        hipLaunchKernelGGL((sample_kernel<true, true>), dim3(1, (unsigned)batch), dim3(SMP_THREADS), 0, st, logits, logits_row_stride,
                           temperatures, top_ks, top_ps, min_ps, (int)vocab, 0ull, 0ull, (int64_t*)nullptr, probs_out,
                           (unsigned long long*)nullptr, 1);
    else
        hipLaunchKernelGGL((sample_kernel<false, true>), dim3(1, (unsigned)batch), dim3(SMP_THREADS), 0, st, logits, logits_row_stride,
                           temperatures, top_ks, top_ps, min_ps, (int)vocab, 0ull, 0ull, (int64_t*)nullptr, probs_out,
                           (unsigned long long*)nullptr, 1);
    return nvx_check_launch("nvx_sampler_probs");
}

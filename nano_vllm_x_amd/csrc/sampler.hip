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
#include "sampler_dev.h"

namespace NVX_NS {

#define SMP_THREADS 1024
#define SMP_MAX_SPLIT 16

template <bool F32>
__device__ __forceinline__ float load_logit(const void* row, int64_t i) {
    if (F32) return reinterpret_cast<const float*>(row)[i];
    return h16bits_to_f32(reinterpret_cast<const uint16_t*>(row)[i]);
}

struct BlockScratch {
    float fred[16];
    unsigned long long ured[16];
    int ired[16];
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

// ---------------------------------------------------------------------------------------------
// Rows with no active filter (and rows with temperature <= 0, which are decoded greedily whatever
// their filters say: an extension, the reference divides by the temperature unguarded).
template <bool F32>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(
    const void* __restrict__ logits, int64_t row_stride, const float* __restrict__ temps, const int64_t* __restrict__ top_ks,
    const float* __restrict__ top_ps, const float* __restrict__ min_ps, int vocab, uint64_t seed, uint64_t offset,
    int64_t* __restrict__ out_ids, unsigned long long* __restrict__ winners, int n_split, int tok0, int tok_end) {
    // tok0 / tok_end: the row is the shard [tok0, tok0 + vocab) of a longer row whose tokens >= tok_end do not exist
    // (padding of a vocabulary split over tensor-parallel ranks); noise and winner carry the token's id in the whole
    // row, so the best of the shards' winners is the token a pass over the whole row draws.  0 / vocab otherwise.
    // top_ks == nullptr (shards): no row has a filter.
    __shared__ BlockScratch sc;
    const int row = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
    const void* lrow = F32 ? (const void*)((const float*)logits + row * row_stride)
                           : (const void*)((const uint16_t*)logits + row * row_stride);
    const float T = temps[row];
    const bool greedy = !(T > 0.0f);
    const bool filtered = top_ks && !greedy && ((top_ks[row] < vocab) || top_ps[row] < 1.0f || min_ps[row] > 0.0f);
    if (filtered) {   // drawn by sample_filtered_kernel, which runs after this one and fills slice 0
        if (tid == 0 && winners) winners[(int64_t)row * n_split + slice] = 0ull;
        return;
    }
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    // single pass: Gumbel-max over this workgroup's slice.  Four tokens share one Philox block.
    const int per = ((vocab + n_split - 1) / n_split + 3) & ~3;
    const int lo = slice * per, hi = min(min(vocab, lo + per), tok_end - tok0);
    const float inv_t = greedy ? 1.0f : 1.0f / T;
    float best = -INFINITY; int besti = 0x7fffffff;
    for (int base = lo + tid * 4; base < hi; base += SMP_THREADS * 4) {
        uint4 r = make_uint4(0, 0, 0, 0);
        if (!greedy) r = philox4x32_10(make_uint4((uint32_t)((tok0 + base) >> 2), (uint32_t)row, (uint32_t)offset, (uint32_t)(offset >> 32)), key);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + j;
            if (i < hi) {
                const float v = load_logit<F32>(lrow, i) * inv_t + (greedy ? 0.0f : gumbel_from_bits(rr[j]));
                if (v > best) { best = v; besti = tok0 + i; }
            }
        }
    }
    block_argmax(best, besti, sc);
    if (tid == 0) {
        if (winners) winners[(int64_t)row * n_split + slice] = besti == 0x7fffffff ? 0ull : pack_winner(best, besti);
        else out_ids[row] = besti == 0x7fffffff ? 0 : besti;
    }
}

// ---------------------------------------------------------------------------------------------
// Filtered rows: one 1024-thread workgroup per row.  A bf16 row of up to 155 648 tokens (a multiple of 8) is loaded ONCE, 16 bytes
// per lane, and stays on the compute unit for every pass (19 vectors of 8 tokens per thread: 12 in registers, 7 in LDS); other rows (fp32 logits,
// longer or unaligned rows) are re-read from memory per pass.  Token i belongs to thread (i / 8) % 1024 either way.
//
// Order statistics are taken on integer keys: for bf16 logits the 16-bit order-preserving image of the RAW logit
// (x = logit / T is strictly increasing in the logit for T > 0 over bf16 inputs, so the k-th largest x is the k-th
// largest logit: two radix passes of 8 bits), for fp32 logits the 32-bit image of x itself (four passes; distinct
// fp32 logits can share a quotient).  Histograms live in LDS replicated per lane -- [bin][lane mod 32] counters,
// [bin][lane mod 16] 64-bit masses -- so that the lanes of one atomic instruction never meet on a bank, however
// the scores cluster; bins are then summed and scanned by one wave.
#define SMP_NV 19        // vectors of 8 tokens per thread: 19 * 1024 * 8 = 155 648 tokens
#define SMP_NV_REG 12    // ... of which this many live in registers (48 VGPRs) and the rest in LDS (112 KB)

struct FiltScratch {
    union {
        unsigned int h32[256 * 32];
        unsigned long long h64[256 * 16];
        uint4 words[2048];
    } hist;                                  // 32 KB
    unsigned long long tot[256];
    BlockScratch blk;
    unsigned long long thr, z_total;
    unsigned long long sel_before, sel_bucket;
    unsigned int sel_digit;
    int found;
    int wave_cnt[16];
};

__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int d) {
    const unsigned int lo = __shfl_up((unsigned int)v, d, 64), hi = __shfl_up((unsigned int)(v >> 32), d, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// tot[0..255] -> the first bin d with start + tot[0] + ... + tot[d] > target (255 if none): sel_digit = d,
// sel_before = start + tot[0..d-1], sel_bucket = tot[d].  With want_total, z_total = sum of all bins and the
// target is derived from it by `make_target` semantics below (top-p threshold).  Called by wave 0 only.
__device__ __forceinline__ void select_bin(FiltScratch& s, unsigned long long start, unsigned long long target, int lane) {
    unsigned long long t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = s.tot[4 * lane + q];
    const unsigned long long mine = t[0] + t[1] + t[2] + t[3];
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = shfl_up_u64(incl, d);
        if (lane >= d) incl += o;
    }
    unsigned long long c = start + incl - mine;
    int dsel = -1; unsigned long long bsel = 0, ksel = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (dsel < 0 && c + t[q] > target) { dsel = 4 * lane + q; bsel = c; ksel = t[q]; }
        if (q < 3) c += t[q];
    }
    // c now = mass before bin 4*lane+3
    const unsigned long long found = __ballot(dsel >= 0);
    if (found) {
        if (lane == __ffsll((long long)found) - 1) { s.sel_digit = (unsigned int)dsel; s.sel_before = bsel; s.sel_bucket = ksel; }
    } else if (lane == 63) {
        s.sel_digit = 255u; s.sel_before = c; s.sel_bucket = t[3];
    }
}

__device__ __forceinline__ unsigned int mass_fixed(float l, float c1, float c2m);

template <bool F32, bool PROBS>
__global__ __launch_bounds__(SMP_THREADS) void sample_filtered_kernel(
    const void* __restrict__ logits, int64_t row_stride, const float* __restrict__ temps, const int64_t* __restrict__ top_ks,
    const float* __restrict__ top_ps, const float* __restrict__ min_ps, int vocab, uint64_t seed, uint64_t offset,
    int64_t* __restrict__ out_ids, float* __restrict__ probs_out, unsigned long long* __restrict__ winners, int n_split) {
    __shared__ FiltScratch sc;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const void* lrow = F32 ? (const void*)((const float*)logits + row * row_stride)
                           : (const void*)((const uint16_t*)logits + row * row_stride);
    const float T = temps[row];
    const bool greedy = !(T > 0.0f);
    const int64_t k64 = top_ks[row];
    const int k = (int)(k64 < 1 ? 1 : (k64 > vocab ? vocab : k64));
    const float top_p = top_ps[row], min_p = min_ps[row];
    const bool filtered = !greedy && (k < vocab || top_p < 1.0f || min_p > 0.0f);
    if (!PROBS && !filtered) return;   // drawn by sample_kernel

    // fn(i, logit) for every token of this thread, in ascending token order (token i belongs to thread (i / 8) % 1024)
    auto each = [&](auto&& fn) __attribute__((always_inline)) {
        for (int i0 = tid * 8; i0 < vocab; i0 += SMP_THREADS * 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (i0 + e < vocab) fn(i0 + e, load_logit<F32>(lrow, i0 + e));
        }
    };

    // ---- row maximum (of the raw logits: the maximum of x is its quotient) ----
    float lmax = -INFINITY;
    each([&](int, float l) { lmax = fmaxf(lmax, l); });
    lmax = block_max(lmax, sc.blk);
    if (PROBS && greedy) {   // temperature <= 0: all mass on the first maximum
        int first = 0x7fffffff;
        each([&](int i, float l) { if (l == lmax && i < first) first = i; });
        float has = first == 0x7fffffff ? -INFINITY : 0.f;
        block_argmax(has, first, sc.blk);   // equal values: the lowest index wins
        float* prow = probs_out + (int64_t)row * vocab;
        each([&](int i, float) { prow[i] = i == first ? 1.0f : 0.0f; });
        return;
    }
    const float M = lmax / T;
    constexpr int KEYBITS = F32 ? 32 : 16;
    auto keyf = [&](float l) -> uint32_t { return F32 ? order_key(l / T) : (order_key(l) >> 16); };
    // softmax mass of one token in 32.32 fixed point (integer sums do not depend on the order of the adds)
    const float c1 = 1.4426950408889634f / T, c2m = 32.0f - M * 1.4426950408889634f;
    auto mass_of = [&](float l) -> unsigned long long {
        if (!F32) return mass_fixed(l, c1, c2m);
        return mass_fixed(l / T, 1.4426950408889634f, c2m);   // fp32 logits: the score is the quotient itself
    };
    auto zero_hist = [&]() {
        sc.hist.words[tid] = make_uint4(0, 0, 0, 0);
        sc.hist.words[tid + 1024] = make_uint4(0, 0, 0, 0);
    };

    // ---- top-k: key of the k-th largest score = ascending rank vocab - k ----
    uint32_t k_key = 0;
    if (k < vocab) {
        uint32_t prefix = 0;
        unsigned long long rank = (unsigned long long)(vocab - k);
        for (int pass = 0; pass < KEYBITS / 8; ++pass) {
            const int shift = KEYBITS - 8 - 8 * pass;
            __syncthreads();
            zero_hist();
            __syncthreads();
            each([&](int, float l) {
                const uint32_t ky = keyf(l);
                if (pass == 0 || (ky >> (shift + 8)) == prefix) atomicAdd(&sc.hist.h32[((ky >> shift) & 255u) * 32 + (lane & 31)], 1u);
            });
            __syncthreads();
            {   // bin totals: thread = (bin, quarter of its 32 copies)
                const int bin = tid >> 2, part = tid & 3;
                unsigned int v = 0;
#pragma unroll
                for (int c = 0; c < 8; ++c) v += sc.hist.h32[bin * 32 + part * 8 + ((c + bin) & 7)];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                if (part == 0) sc.tot[bin] = v;
            }
            __syncthreads();
            if (tid < 64) select_bin(sc, 0ull, rank, lane);
            __syncthreads();
            prefix = (prefix << 8) | sc.sel_digit;
            rank -= sc.sel_before;
        }
        k_key = prefix;
    }

    // ---- top-p: boundary of cumulative mass in ascending order ----
    const bool p_on = top_p < 1.0f;
    uint32_t p_key = 0;       // scores with key < p_key are dropped
    int p_cut = -1;           // among scores with key == p_key, token ids < p_cut are dropped
    if (p_on) {
        uint32_t prefix = 0;
        unsigned long long before = 0;   // mass of kept scores strictly below the current prefix
        for (int pass = 0; pass < KEYBITS / 8; ++pass) {
            const int shift = KEYBITS - 8 - 8 * pass;
            __syncthreads();
            zero_hist();
            __syncthreads();
            each([&](int, float l) {
                const uint32_t ky = keyf(l);
                if (ky >= k_key && (pass == 0 || (ky >> (shift + 8)) == prefix))
                    atomicAdd(&sc.hist.h64[((ky >> shift) & 255u) * 16 + (lane & 15)], mass_of(l));
            });
            __syncthreads();
            {
                const int bin = tid >> 2, part = tid & 3;
                unsigned long long v = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) v += sc.hist.h64[bin * 16 + part * 4 + ((c + bin) & 3)];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                if (part == 0) sc.tot[bin] = v;
            }
            __syncthreads();
            if (tid < 64) {
                if (pass == 0) {   // the first pass sees every kept token: its bins add up to the normaliser
                    unsigned long long z = sc.tot[4 * lane] + sc.tot[4 * lane + 1] + sc.tot[4 * lane + 2] + sc.tot[4 * lane + 3];
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) z += __shfl_xor(z, m, 64);
                    const double thr_d = (double)(1.0f - top_p) * (double)z;
                    unsigned long long thr = thr_d <= 0.0 ? 0ull : (unsigned long long)thr_d;
                    if (thr >= z) thr = z > 0 ? z - 1 : 0;   // the largest score always survives
                    if (lane == 0) { sc.thr = thr; sc.z_total = z; }
                    select_bin(sc, 0ull, thr, lane);
                } else {
                    select_bin(sc, before, sc.thr, lane);
                }
            }
            __syncthreads();
            prefix = (prefix << 8) | sc.sel_digit;
            before = sc.sel_before;
        }
        p_key = prefix;
        // ties at the boundary score: drop the lowest token ids while the running mass stays <= thr
        unsigned long long pt;   // mass of one token at the boundary score (ties have equal scores, hence equal masses)
        if (F32) {   // the key is that of x itself: undo the order map
            const uint32_t u = (p_key & 0x80000000u) ? (p_key & 0x7fffffffu) : ~p_key;
            pt = mass_fixed(__uint_as_float(u), 1.4426950408889634f, c2m);
        } else {     // the key is that of the raw bf16 logit
            const uint32_t k16 = p_key & 0xffffu;
            const uint32_t b = (k16 & 0x8000u) ? (k16 & 0x7fffu) : (~k16 & 0xffffu);
            pt = mass_of(__uint_as_float(b << 16));
        }
        const unsigned long long thr = sc.thr;
        const unsigned long long room = thr >= before ? thr - before : 0;
        const long long n_drop = pt > 0 ? (long long)(room / pt) : 0;
        if (n_drop > 0) {
            // token id of the tie with ascending-id rank n_drop.  Tokens are ordered (vector row j, thread, element):
            // ties per vector row -> the row holding that rank -> a scan over its threads -> the element.
            __syncthreads();
            if (tid < 256) sc.tot[tid] = 0;
            if (tid == 0) sc.found = -1;
            __syncthreads();
            {
                int jrow = 0, cnt = 0;
                each([&](int i, float l) {
                    const int j = i / (SMP_THREADS * 8);
                    if (j != jrow) { if (cnt) atomicAdd(&sc.tot[jrow & 255], (unsigned long long)cnt); jrow = j; cnt = 0; }
                    cnt += keyf(l) == p_key;
                });
                if (cnt) atomicAdd(&sc.tot[jrow & 255], (unsigned long long)cnt);
            }
            __syncthreads();
            if (tid < 64) select_bin(sc, 0ull, (unsigned long long)n_drop, lane);
            __syncthreads();
            const int jstar = (int)sc.sel_digit;
            const int r = (int)((unsigned long long)n_drop - sc.sel_before);   // rank inside vector row jstar
            // this thread's eight tokens of that row
            int c = 0;
            unsigned int tie_bits = 0;
            {
                const int i0 = (jstar * SMP_THREADS + tid) * 8;
                each([&](int i, float l) {
                    if (i >= i0 && i < i0 + 8 && keyf(l) == p_key) { tie_bits |= 1u << (i - i0); ++c; }
                });
            }
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
            if (lane == 63) sc.wave_cnt[tid >> 6] = incl;
            __syncthreads();
            int excl = incl - c;
            for (int w = 0; w < (tid >> 6); ++w) excl += sc.wave_cnt[w];
            if (c > 0 && r >= excl && r < excl + c) {
                int need = r - excl;
                for (int e = 0; e < 8; ++e)
                    if ((tie_bits >> e) & 1u) { if (need == 0) { sc.found = (jstar * SMP_THREADS + tid) * 8 + e; break; } --need; }
            }
            __syncthreads();
            p_cut = sc.found >= 0 ? sc.found : vocab;
        }
    }

    auto kept_kp = [&](uint32_t ky, int i) -> bool {
        if (ky < k_key) return false;
        if (p_on) { if (ky < p_key) return false; if (ky == p_key && i < p_cut) return false; }
        return true;
    };

    // ---- min-p on softmax of what survived ----
    float minp_thr = -1.0f, Z2 = 1.0f;
    if (min_p > 0.0f) {
        float zl = 0.f;
        each([&](int i, float l) { if (kept_kp(keyf(l), i)) zl += __expf(l / T - M); });
        Z2 = block_sum(zl, sc.blk);
        minp_thr = min_p * (1.0f / Z2);
    }
    auto kept = [&](float l, int i) -> bool {
        if (!kept_kp(keyf(l), i)) return false;
        if (min_p > 0.0f && (__expf(l / T - M) / Z2) < minp_thr) return false;
        return true;
    };

    if (PROBS) {
        float zl = 0.f;
        each([&](int i, float l) { if (kept(l, i)) zl += __expf(l / T - M); });
        const float Z3 = block_sum(zl, sc.blk);
        float* prow = probs_out + (int64_t)row * vocab;
        each([&](int i, float l) { prow[i] = kept(l, i) ? __expf(l / T - M) / Z3 : 0.0f; });
        return;
    }

    // ---- the draw: Gumbel-max over the kept tokens; four tokens share one Philox block, computed only where a
    // token of the four survived (the counter depends on the token alone, so skipping changes no draw) ----
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    float best = -INFINITY; int besti = 0x7fffffff;
    {
        int gbase = -1; uint32_t rr[4] = {0, 0, 0, 0};
        auto consider = [&](int i, float l) __attribute__((always_inline)) {   // a kept token enters the draw
            if ((i & ~3) != gbase) {
                gbase = i & ~3;
                const uint4 r = philox4x32_10(make_uint4((uint32_t)(gbase >> 2), (uint32_t)row, (uint32_t)offset, (uint32_t)(offset >> 32)), key);
                rr[0] = r.x; rr[1] = r.y; rr[2] = r.z; rr[3] = r.w;
            }
            const uint32_t bits = (i & 3) == 0 ? rr[0] : (i & 3) == 1 ? rr[1] : (i & 3) == 2 ? rr[2] : rr[3];
            const float v = l / T + gumbel_from_bits(bits);
            if (v > best || (v == best && i < besti)) { best = v; besti = i; }
        };
        each([&](int i, float l) { if (kept(l, i)) consider(i, l); });
    }
    block_argmax(best, besti, sc.blk);
    if (tid == 0) {
        if (winners) winners[(int64_t)row * n_split] = besti == 0x7fffffff ? 0ull : pack_winner(best, besti);
        else out_ids[row] = besti == 0x7fffffff ? 0 : besti;
    }
}

// ---------------------------------------------------------------------------------------------
// The same selection for a bf16 row of at most 155 648 tokens (a multiple of 8, 16-byte aligned): the row is read
// from memory ONCE, turned into packed 16-bit order keys and kept on the compute unit for every pass -- 12 vectors
// of 8 keys per thread in registers, 7 in LDS (each thread reads back the 16 bytes it wrote).  Passes are
// branch-free per token (a token that does not take part adds 0) and work on two keys per register.  Token to
// thread mapping, formulas and summation orders are those of sample_filtered_kernel<false, .>: same results.
#define SMP_ROWS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
static_assert(SMP_NV_REG == 12 && SMP_NV == 19, "SMP_ROWS lists the register vectors by name");

// two packed bf16 values -> their two packed order keys (negative: all bits flipped, otherwise the sign bit set)
__device__ __forceinline__ uint32_t keys_of_pair(uint32_t w) {
    const uint32_t neg = ((w >> 15) & 0x00010001u) * 0xffffu;
    return w ^ (neg | 0x80008000u);
}
// ... and back
__device__ __forceinline__ uint32_t pair_of_keys(uint32_t k) {
    const uint32_t pos = ((k >> 15) & 0x00010001u) * 0xffffu;     // 0xffff where the value was not negative
    return k ^ (0xffffffffu ^ (pos & 0x7fff7fffu));
}
__device__ __forceinline__ uint4 keys_of_vec(uint4 v) {
    return make_uint4(keys_of_pair(v.x), keys_of_pair(v.y), keys_of_pair(v.z), keys_of_pair(v.w));
}
// Softmax mass of one token relative to the row maximum, in 0.32 fixed point: exp2(l * c1 + c2m) with
// c1 = log2(e) / T and c2m = 32 - max(x) * log2(e); the conversion saturates, so the maximum itself weighs
// 2^32 - 1.  Integer masses make every sum independent of the order of the adds.
__device__ __forceinline__ unsigned int mass_fixed(float l, float c1, float c2m) {
    return (unsigned int)__builtin_amdgcn_exp2f(fmaf(l, c1, c2m));   // v_cvt_u32_f32 clamps to [0, 2^32 - 1]
}

__device__ __forceinline__ void reduce_counts(FiltScratch& sc, int tid) {   // h32[bin][32 copies] -> tot[bin]
    const int bin = tid >> 2, part = tid & 3;
    unsigned int v = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) v += sc.hist.h32[bin * 32 + part * 8 + ((c + bin) & 7)];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    if (part == 0) sc.tot[bin] = v;
}
__device__ __forceinline__ void reduce_masses(FiltScratch& sc, int tid) {   // h64[bin][16 copies] -> tot[bin]
    const int bin = tid >> 2, part = tid & 3;
    unsigned long long v = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) v += sc.hist.h64[bin * 16 + part * 4 + ((c + bin) & 3)];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    if (part == 0) sc.tot[bin] = v;
}

template <bool PROBS>
__global__ __launch_bounds__(SMP_THREADS) void sample_resident_kernel(
    const uint16_t* __restrict__ logits, int64_t row_stride, const float* __restrict__ temps, const int64_t* __restrict__ top_ks,
    const float* __restrict__ top_ps, const float* __restrict__ min_ps, int vocab, uint64_t seed, uint64_t offset,
    int64_t* __restrict__ out_ids, float* __restrict__ probs_out, unsigned long long* __restrict__ winners, int n_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smp_dyn[];   // FiltScratch, then the LDS part of the row
    FiltScratch& sc = *reinterpret_cast<FiltScratch*>(smp_dyn);
    uint4* const lds_row = reinterpret_cast<uint4*>(smp_dyn + ((sizeof(FiltScratch) + 15) & ~(size_t)15));
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const uint16_t* lrow = logits + row * row_stride;
    const float T = temps[row];
    const bool greedy = !(T > 0.0f);
    const int64_t k64 = top_ks[row];
    const int k = (int)(k64 < 1 ? 1 : (k64 > vocab ? vocab : k64));
    const float top_p = top_ps[row], min_p = min_ps[row];
    const bool filtered = !greedy && (k < vocab || top_p < 1.0f || min_p > 0.0f);
    if (!PROBS && !filtered) return;   // drawn by sample_kernel

    // named registers: an array indexed in unrolled loops was left in scratch memory by the compiler
#define SMP_LOAD(j)                                                                                                     \
    uint4 raw##j = make_uint4(0, 0, 0, 0);                                                                               \
    if ((j * SMP_THREADS + tid) * 8 < vocab)                                                                             \
        raw##j = keys_of_vec(*reinterpret_cast<const uint4*>(lrow + (j * SMP_THREADS + tid) * 8));
    SMP_ROWS(SMP_LOAD)
#undef SMP_LOAD
#pragma unroll
    for (int j = SMP_NV_REG; j < SMP_NV; ++j) {
        const int i0 = (j * SMP_THREADS + tid) * 8;
        lds_row[(j - SMP_NV_REG) * SMP_THREADS + tid] =
            i0 < vocab ? keys_of_vec(*reinterpret_cast<const uint4*>(lrow + i0)) : make_uint4(0, 0, 0, 0);
    }

    // fn(i, kw): kw holds the keys of tokens i (low half) and i + 1, for every pair of this thread's tokens in
    // ascending token order
    auto each_pair = [&](auto&& fn) __attribute__((always_inline)) {
        int voc = vocab;
        asm volatile("" : "+s"(voc));             // no table of bounds tests kept across passes
        auto vec = [&](const int j, uint4& v) __attribute__((always_inline)) {
            // the row's registers are made opaque at the start of every pass: otherwise the compiler hoists what a
            // pass derives from them out of the loops over passes and spills it
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            const int i0 = (j * SMP_THREADS + tid) * 8;
            if (i0 < voc) { fn(i0, v.x); fn(i0 + 2, v.y); fn(i0 + 4, v.z); fn(i0 + 6, v.w); }
            __builtin_amdgcn_sched_barrier(0);    // one vector at a time: interleaving the chains costs registers
        };
#define SMP_VEC(j) vec(j, raw##j);
        SMP_ROWS(SMP_VEC)
#undef SMP_VEC
#pragma unroll
        for (int j = SMP_NV_REG; j < SMP_NV; ++j) {
            uint4 v = lds_row[(j - SMP_NV_REG) * SMP_THREADS + tid];   // own data: no barrier needed
            vec(j, v);
        }
    };
    auto logit_of_key = [](uint32_t k16) -> float { return __uint_as_float((pair_of_keys(k16) & 0xffffu) << 16); };

    // ---- row maximum ----
    uint32_t kmax = 0;
    each_pair([&](int, uint32_t kw) { kmax = max(kmax, max(kw & 0xffffu, kw >> 16)); });
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, m, 64));
    __syncthreads();
    if (lane == 0) sc.blk.ired[tid >> 6] = (int)kmax;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) kmax = max(kmax, (uint32_t)sc.blk.ired[i]);
    const float lmax = logit_of_key(kmax);
    if (PROBS && greedy) {   // temperature <= 0: all mass on the first maximum
        int first = 0x7fffffff;
        each_pair([&](int i, uint32_t kw) {
            if ((kw >> 16) == kmax && i + 1 < first) first = i + 1;
            if ((kw & 0xffffu) == kmax && i < first) first = i;
        });
        float has = first == 0x7fffffff ? -INFINITY : 0.f;
        block_argmax(has, first, sc.blk);   // equal values: the lowest index wins
        float* prow = probs_out + (int64_t)row * vocab;
        each_pair([&](int i, uint32_t) { prow[i] = i == first ? 1.0f : 0.0f; prow[i + 1] = i + 1 == first ? 1.0f : 0.0f; });
        return;
    }
    const float M = lmax / T;
    const float c1 = 1.4426950408889634f / T, c2m = 32.0f - M * 1.4426950408889634f;
    auto zero_hist = [&]() {
        sc.hist.words[tid] = make_uint4(0, 0, 0, 0);
        sc.hist.words[tid + 1024] = make_uint4(0, 0, 0, 0);
    };
    unsigned int* const my32 = sc.hist.h32 + (lane & 31);
    unsigned long long* const my64 = sc.hist.h64 + (lane & 15);

    // ---- top-k: key of the k-th largest score = ascending rank vocab - k ----
    uint32_t k_key = 0;
    if (k < vocab) {
        unsigned long long rank = (unsigned long long)(vocab - k);
        __syncthreads();
        zero_hist();
        __syncthreads();
        each_pair([&](int, uint32_t kw) {
            atomicAdd(my32 + ((kw >> 8) & 255u) * 32, 1u);
            atomicAdd(my32 + (kw >> 24) * 32, 1u);
        });
        __syncthreads();
        reduce_counts(sc, tid);
        __syncthreads();
        if (tid < 64) select_bin(sc, 0ull, rank, lane);
        __syncthreads();
        const uint32_t hi = sc.sel_digit;
        rank -= sc.sel_before;
        __syncthreads();
        zero_hist();
        __syncthreads();
        each_pair([&](int, uint32_t kw) {
            atomicAdd(my32 + (kw & 255u) * 32, ((kw >> 8) & 255u) == hi ? 1u : 0u);
            atomicAdd(my32 + ((kw >> 16) & 255u) * 32, (kw >> 24) == hi ? 1u : 0u);
        });
        __syncthreads();
        reduce_counts(sc, tid);
        __syncthreads();
        if (tid < 64) select_bin(sc, 0ull, rank, lane);
        __syncthreads();
        k_key = (hi << 8) | sc.sel_digit;
    }

    // ---- top-p: boundary of cumulative mass in ascending order ----
    const bool p_on = top_p < 1.0f;
    uint32_t p_key = 0;       // scores with key < p_key are dropped
    int p_cut = -1;           // among scores with key == p_key, token ids < p_cut are dropped
    if (p_on) {
        __syncthreads();
        zero_hist();
        __syncthreads();
        each_pair([&](int, uint32_t kw) {
            const uint32_t k0 = kw & 0xffffu, k1 = kw >> 16;
            if (__any(k0 >= k_key || k1 >= k_key)) {   // (with a small top-k nearly every wave skips the exponentials)
                const uint32_t pr = pair_of_keys(kw);
                const unsigned long long m0 = k0 >= k_key ? mass_fixed(h16lo_to_f32(pr), c1, c2m) : 0u;
                const unsigned long long m1 = k1 >= k_key ? mass_fixed(h16hi_to_f32(pr), c1, c2m) : 0u;
                atomicAdd(my64 + (k0 >> 8) * 16, m0);
                atomicAdd(my64 + (k1 >> 8) * 16, m1);
            }
        });
        __syncthreads();
        reduce_masses(sc, tid);
        __syncthreads();
        if (tid < 64) {   // the first pass sees every kept token: its bins add up to the normaliser
            unsigned long long z = sc.tot[4 * lane] + sc.tot[4 * lane + 1] + sc.tot[4 * lane + 2] + sc.tot[4 * lane + 3];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) z += __shfl_xor(z, m, 64);
            const double thr_d = (double)(1.0f - top_p) * (double)z;
            unsigned long long thr = thr_d <= 0.0 ? 0ull : (unsigned long long)thr_d;
            if (thr >= z) thr = z > 0 ? z - 1 : 0;   // the largest score always survives
            if (lane == 0) { sc.thr = thr; sc.z_total = z; }
            select_bin(sc, 0ull, thr, lane);
        }
        __syncthreads();
        const uint32_t hi = sc.sel_digit;
        unsigned long long before = sc.sel_before;   // mass of kept scores strictly below the current prefix
        const unsigned long long thr = sc.thr;
        __syncthreads();
        zero_hist();
        __syncthreads();
        each_pair([&](int, uint32_t kw) {
            const uint32_t k0 = kw & 0xffffu, k1 = kw >> 16;
            const bool t0 = k0 >= k_key && (k0 >> 8) == hi, t1 = k1 >= k_key && (k1 >> 8) == hi;
            if (__any(t0 || t1)) {
                const uint32_t pr = pair_of_keys(kw);
                const unsigned long long m0 = t0 ? mass_fixed(h16lo_to_f32(pr), c1, c2m) : 0u;
                const unsigned long long m1 = t1 ? mass_fixed(h16hi_to_f32(pr), c1, c2m) : 0u;
                atomicAdd(my64 + (k0 & 255u) * 16, m0);
                atomicAdd(my64 + (k1 & 255u) * 16, m1);
            }
        });
        __syncthreads();
        reduce_masses(sc, tid);
        __syncthreads();
        if (tid < 64) select_bin(sc, before, thr, lane);
        __syncthreads();
        p_key = (hi << 8) | sc.sel_digit;
        before = sc.sel_before;
        // ties at the boundary score: drop the lowest token ids while the running mass stays <= thr
        const unsigned long long pt = mass_fixed(logit_of_key(p_key), c1, c2m);   // equal scores have equal masses
        const unsigned long long room = thr >= before ? thr - before : 0;
        const long long n_drop = pt > 0 ? (long long)(room / pt) : 0;
        if (n_drop > 0) {
            // token id of the tie with ascending-id rank n_drop.  Tokens are ordered (vector row j, thread, element):
            // ties per vector row -> the row holding that rank -> a scan over its threads -> the element.
            __syncthreads();
            if (tid < 256) sc.tot[tid] = 0;
            if (tid == 0) sc.found = -1;
            __syncthreads();
            {
                int jrow = 0, cnt = 0;
                each_pair([&](int i, uint32_t kw) {
                    const int j = i / (SMP_THREADS * 8);
                    if (j != jrow) { if (cnt) atomicAdd(&sc.tot[jrow], (unsigned long long)cnt); jrow = j; cnt = 0; }
                    cnt += ((kw & 0xffffu) == p_key) + ((kw >> 16) == p_key);
                });
                if (cnt) atomicAdd(&sc.tot[jrow], (unsigned long long)cnt);
            }
            __syncthreads();
            if (tid < 64) select_bin(sc, 0ull, (unsigned long long)n_drop, lane);
            __syncthreads();
            const int jstar = (int)sc.sel_digit;
            const int r = (int)((unsigned long long)n_drop - sc.sel_before);   // rank inside vector row jstar
            int c = 0;
            unsigned int tie_bits = 0;   // this thread's eight tokens of that row
            {
                const int i0 = (jstar * SMP_THREADS + tid) * 8;
                each_pair([&](int i, uint32_t kw) {
                    if (i >= i0 && i < i0 + 8) {
                        if ((kw & 0xffffu) == p_key) { tie_bits |= 1u << (i - i0); ++c; }
                        if ((kw >> 16) == p_key) { tie_bits |= 2u << (i - i0); ++c; }
                    }
                });
            }
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
            if (lane == 63) sc.wave_cnt[tid >> 6] = incl;
            __syncthreads();
            int excl = incl - c;
            for (int w = 0; w < (tid >> 6); ++w) excl += sc.wave_cnt[w];
            if (c > 0 && r >= excl && r < excl + c) {
                int need = r - excl;
                for (int e = 0; e < 8; ++e)
                    if ((tie_bits >> e) & 1u) { if (need == 0) { sc.found = (jstar * SMP_THREADS + tid) * 8 + e; break; } --need; }
            }
            __syncthreads();
            p_cut = sc.found >= 0 ? sc.found : vocab;
        }
    }

    auto kept_kp = [&](uint32_t ky, int i) -> bool {
        if (ky < k_key) return false;
        if (p_on) { if (ky < p_key) return false; if (ky == p_key && i < p_cut) return false; }
        return true;
    };

    // ---- min-p on softmax of what survived ----
    float minp_thr = -1.0f, Z2 = 1.0f;
    if (min_p > 0.0f) {
        float zl = 0.f;
        each_pair([&](int i, uint32_t kw) {
            const uint32_t pr = pair_of_keys(kw);
            if (kept_kp(kw & 0xffffu, i)) zl += __expf(h16lo_to_f32(pr) / T - M);
            if (kept_kp(kw >> 16, i + 1)) zl += __expf(h16hi_to_f32(pr) / T - M);
        });
        Z2 = block_sum(zl, sc.blk);
        minp_thr = min_p * (1.0f / Z2);
    }
    auto kept = [&](uint32_t ky, float l, int i) -> bool {
        if (!kept_kp(ky, i)) return false;
        if (min_p > 0.0f && (__expf(l / T - M) / Z2) < minp_thr) return false;
        return true;
    };

    if (PROBS) {
        float zl = 0.f;
        each_pair([&](int i, uint32_t kw) {
            const uint32_t pr = pair_of_keys(kw);
            const float l0 = h16lo_to_f32(pr), l1 = h16hi_to_f32(pr);
            if (kept(kw & 0xffffu, l0, i)) zl += __expf(l0 / T - M);
            if (kept(kw >> 16, l1, i + 1)) zl += __expf(l1 / T - M);
        });
        const float Z3 = block_sum(zl, sc.blk);
        float* prow = probs_out + (int64_t)row * vocab;
        each_pair([&](int i, uint32_t kw) {
            const uint32_t pr = pair_of_keys(kw);
            const float l0 = h16lo_to_f32(pr), l1 = h16hi_to_f32(pr);
            prow[i] = kept(kw & 0xffffu, l0, i) ? __expf(l0 / T - M) / Z3 : 0.0f;
            prow[i + 1] = kept(kw >> 16, l1, i + 1) ? __expf(l1 / T - M) / Z3 : 0.0f;
        });
        return;
    }

    // ---- the draw: Gumbel-max over the kept tokens.  Which of this thread's tokens survived is first recorded as a
    // bit mask (bit = 8 * vector + element); the survivors -- usually few -- are then walked in a loop that holds the
    // Philox code once and reads their logits back from memory.  Four tokens share one Philox block; its counter
    // depends on the token alone, so skipping the blocks without a survivor changes no draw. ----
    uint32_t km[5] = {0, 0, 0, 0, 0};
    each_pair([&](int i, uint32_t kw) {
        const int pos = (i / (SMP_THREADS * 8)) * 8 + (i & 7);
        bool a, b;
        if (min_p > 0.0f) {
            const uint32_t pr = pair_of_keys(kw);
            a = kept(kw & 0xffffu, h16lo_to_f32(pr), i);
            b = kept(kw >> 16, h16hi_to_f32(pr), i + 1);
        } else {
            a = kept_kp(kw & 0xffffu, i);
            b = kept_kp(kw >> 16, i + 1);
        }
        km[pos >> 5] |= ((a ? 1u : 0u) | (b ? 2u : 0u)) << (pos & 31);
    });
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    float best = -INFINITY; int besti = 0x7fffffff;
    int gbase = -1; uint32_t rr[4] = {0, 0, 0, 0};
#pragma unroll
    for (int wd = 0; wd < 5; ++wd) {
        uint32_t m = km[wd];
        while (m) {
            const int pos = wd * 32 + __ffs((int)m) - 1;
            m &= m - 1;
            const int i = ((pos >> 3) * SMP_THREADS + tid) * 8 + (pos & 7);
            const float l = h16bits_to_f32(lrow[i]);
            if ((i & ~3) != gbase) {
                gbase = i & ~3;
                const uint4 r = philox4x32_10(make_uint4((uint32_t)(gbase >> 2), (uint32_t)row, (uint32_t)offset, (uint32_t)(offset >> 32)), key);
                rr[0] = r.x; rr[1] = r.y; rr[2] = r.z; rr[3] = r.w;
            }
            const uint32_t bits = (i & 3) == 0 ? rr[0] : (i & 3) == 1 ? rr[1] : (i & 3) == 2 ? rr[2] : rr[3];
            const float v = l / T + gumbel_from_bits(bits);
            if (v > best || (v == best && i < besti)) { best = v; besti = i; }
        }
    }
    block_argmax(best, besti, sc.blk);
    if (tid == 0) {
        if (winners) winners[(int64_t)row * n_split] = besti == 0x7fffffff ? 0ull : pack_winner(best, besti);
        else out_ids[row] = besti == 0x7fffffff ? 0 : besti;
    }
}
#undef SMP_ROWS

// Per row: the best of n_split winners (winner s of row r at winners[r * row_stride + s * split_stride]), as a token id
// and / or as the winning key itself.
__global__ __launch_bounds__(256) void sample_finish_kernel(const unsigned long long* __restrict__ winners, int n_split, int batch,
                                                            int64_t row_stride, int64_t split_stride, int64_t* __restrict__ out_ids,
                                                            unsigned long long* __restrict__ out_winner) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= batch) return;
    unsigned long long w = 0;
    for (int s = 0; s < n_split; ++s) { const unsigned long long o = winners[(int64_t)row * row_stride + s * split_stride]; w = o > w ? o : w; }
    if (out_ids) out_ids[row] = w == 0 ? 0 : (int64_t)(0xffffffffu - (uint32_t)(w & 0xffffffffull));
    if (out_winner) out_winner[row] = w;
}

static int sampler_check(const void* logits, int64_t stride, const float* t, const int64_t* k, const float* p, const float* mp,
                         int64_t batch, int64_t vocab, const char* who) {
    NVX_REQUIRE(batch >= 0 && vocab > 0, NVX_EINVAL, "%s: bad size", who);
    NVX_REQUIRE(vocab <= 256 * SMP_THREADS * 8, NVX_EUNSUPPORTED, "%s: vocabulary larger than %d tokens", who, 256 * SMP_THREADS * 8);
    NVX_REQUIRE(batch == 0 || (logits && t && k && p && mp), NVX_EINVAL, "%s: null pointer", who);
    NVX_REQUIRE(stride >= vocab, NVX_EINVAL, "%s: row stride smaller than the vocabulary", who);
    return NVX_OK;
}

static size_t resident_lds_bytes() {
    return ((sizeof(FiltScratch) + 15) & ~(size_t)15) + (size_t)(SMP_NV - SMP_NV_REG) * SMP_THREADS * 16;
}

// Launch of the resident-row kernel: it needs more LDS than a kernel gets by default, so the limit is raised first
// (a property of the function, not a stream operation).
template <bool PROBS>
static void launch_resident(unsigned rows, hipStream_t st, const void* logits, int64_t stride, const float* t, const int64_t* k,
                            const float* p, const float* mp, int vocab, uint64_t seed, uint64_t offset, int64_t* out_ids,
                            float* probs_out, unsigned long long* winners, int n_split) {
    const size_t lds = resident_lds_bytes();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_resident_kernel<PROBS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((sample_resident_kernel<PROBS>), dim3(rows), dim3(SMP_THREADS), lds, st, (const uint16_t*)logits, stride, t, k, p,
                       mp, vocab, seed, offset, out_ids, probs_out, winners, n_split);
}

template <bool F32, bool PROBS>
static void launch_filtered(unsigned rows, hipStream_t st, const void* logits, int64_t stride, const float* t, const int64_t* k,
                            const float* p, const float* mp, int vocab, uint64_t seed, uint64_t offset, int64_t* out_ids,
                            float* probs_out, unsigned long long* winners, int n_split) {
    hipLaunchKernelGGL((sample_filtered_kernel<F32, PROBS>), dim3(rows), dim3(SMP_THREADS), 0, st, logits, stride, t, k, p, mp,
                       vocab, seed, offset, out_ids, probs_out, winners, n_split);
}

// A bf16 row that the filtered kernel can load 16 bytes per lane and keep on the compute unit.
static bool row_fits_registers(const void* logits, int64_t stride, int64_t vocab) {
    return vocab <= (int64_t)SMP_NV * SMP_THREADS * 8 && vocab % 8 == 0 && nvx_aligned16(logits) && stride % 8 == 0;
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
    // rows without a filter (and greedy rows), cut into slices ...
    if (logits_is_f32)
        hipLaunchKernelGGL((sample_kernel<true>), grid, dim3(SMP_THREADS), 0, st, logits, logits_row_stride,
                           temperatures, top_ks, top_ps, min_ps, (int)vocab, seed, offset, out_ids, winners, n_split, 0, (int)vocab);
    else
        hipLaunchKernelGGL((sample_kernel<false>), grid, dim3(SMP_THREADS), 0, st, logits, logits_row_stride,
                           temperatures, top_ks, top_ps, min_ps, (int)vocab, seed, offset, out_ids, winners, n_split, 0, (int)vocab);
    rc = nvx_check_launch("nvx_sample");
    if (rc) return rc;
    // ... then the filtered rows, one workgroup each (it returns at once on a row the first kernel drew)
    if (logits_is_f32)
        launch_filtered<true, false>((unsigned)batch, st, logits, logits_row_stride, temperatures, top_ks, top_ps, min_ps,
                                            (int)vocab, seed, offset, out_ids, nullptr, winners, n_split);
    else if (row_fits_registers(logits, logits_row_stride, vocab))
        launch_resident<false>((unsigned)batch, st, logits, logits_row_stride, temperatures, top_ks, top_ps, min_ps,
                                            (int)vocab, seed, offset, out_ids, nullptr, winners, n_split);
    else
        launch_filtered<false, false>((unsigned)batch, st, logits, logits_row_stride, temperatures, top_ks, top_ps, min_ps,
                                             (int)vocab, seed, offset, out_ids, nullptr, winners, n_split);
    rc = nvx_check_launch("nvx_sample(filtered rows)");
    if (rc || !ws) return rc;
    hipLaunchKernelGGL(sample_finish_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, winners, n_split, (int)batch,
                       (int64_t)n_split, (int64_t)1, out_ids, (unsigned long long*)nullptr);
    return nvx_check_launch("nvx_sample(finish)");
}

// One tensor-parallel rank's part of a sampling step whose rows have no filter: the draw over this rank's shard of the
// vocabulary, left as one 64-bit (score, token) key per row.  The largest key over the ranks is the token nvx_sample
// draws from the gathered row (same noise per token, same tie rule), so the ranks exchange 8 bytes per row instead of
// their logits.
extern "C" int nvx_sample_shard(const void* logits, int logits_is_f32, int64_t logits_row_stride, const float* temperatures,
                                int64_t batch, int64_t shard_vocab, int64_t vocab_offset, int64_t vocab, uint64_t seed,
                                uint64_t offset, uint64_t* winner_out, void* ws, int64_t ws_bytes, nvx_stream_t stream) {
    NVX_REQUIRE(batch >= 0 && shard_vocab > 0 && vocab_offset >= 0 && vocab > 0, NVX_EINVAL, "nvx_sample_shard: bad size");
    if (batch == 0) return NVX_OK;
    NVX_REQUIRE(logits && temperatures && winner_out && ws, NVX_EINVAL, "nvx_sample_shard: null pointer");
    NVX_REQUIRE(batch <= 65535 && vocab_offset + shard_vocab <= 0x7fffffffll && vocab <= 0x7fffffffll, NVX_EUNSUPPORTED,
                "nvx_sample_shard: more than 65535 rows or token ids beyond 2^31");
    NVX_REQUIRE(vocab_offset % 4 == 0, NVX_EINVAL, "nvx_sample_shard: a shard must start at a multiple of 4 tokens (four tokens share a noise block)");
    NVX_REQUIRE(logits_row_stride >= shard_vocab, NVX_EINVAL, "nvx_sample_shard: row stride smaller than the shard");
    NVX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7u) == 0 && (reinterpret_cast<uintptr_t>(winner_out) & 7u) == 0, NVX_EINVAL,
                "nvx_sample_shard: workspace and output must be 8-byte aligned");
    int n_split = (int)(512 / batch);
    n_split = n_split < 1 ? 1 : (n_split > SMP_MAX_SPLIT ? SMP_MAX_SPLIT : n_split);
    NVX_REQUIRE(ws_bytes >= batch * n_split * 8, NVX_ENOSPACE, "nvx_sample_shard: workspace too small (%lld < %lld bytes)",
                (long long)ws_bytes, (long long)(batch * n_split * 8));
    unsigned long long* winners = (unsigned long long*)ws;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n_split, (unsigned)batch);
    const int64_t* none_k = nullptr;
    const float* none_f = nullptr;
    if (logits_is_f32)
        hipLaunchKernelGGL((sample_kernel<true>), grid, dim3(SMP_THREADS), 0, st, logits, logits_row_stride, temperatures, none_k, none_f,
                           none_f, (int)shard_vocab, seed, offset, (int64_t*)nullptr, winners, n_split, (int)vocab_offset, (int)vocab);
    else
        hipLaunchKernelGGL((sample_kernel<false>), grid, dim3(SMP_THREADS), 0, st, logits, logits_row_stride, temperatures, none_k, none_f,
                           none_f, (int)shard_vocab, seed, offset, (int64_t*)nullptr, winners, n_split, (int)vocab_offset, (int)vocab);
    int rc = nvx_check_launch("nvx_sample_shard");
    if (rc) return rc;
    hipLaunchKernelGGL(sample_finish_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, winners, n_split, (int)batch,
                       (int64_t)n_split, (int64_t)1, (int64_t*)nullptr, (unsigned long long*)winner_out);
    return nvx_check_launch("nvx_sample_shard(finish)");
}

// The token of every row from the ranks' keys: winners is (n_sets, batch), set by set (what an all-gather of the
// ranks' nvx_sample_shard outputs gives).
extern "C" int nvx_sample_pick(const uint64_t* winners, int64_t n_sets, int64_t batch, int64_t* out_ids, nvx_stream_t stream) {
    NVX_REQUIRE(batch >= 0 && n_sets >= 1 && n_sets <= 4096, NVX_EINVAL, "nvx_sample_pick: bad size");
    if (batch == 0) return NVX_OK;
    NVX_REQUIRE(winners && out_ids, NVX_EINVAL, "nvx_sample_pick: null pointer");
    hipLaunchKernelGGL(sample_finish_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)winners, (int)n_sets, (int)batch, (int64_t)1, batch, out_ids, (unsigned long long*)nullptr);
    return nvx_check_launch("nvx_sample_pick");
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
        launch_filtered<true, true>((unsigned)batch, st, logits, logits_row_stride, temperatures, top_ks, top_ps, min_ps,
                                           (int)vocab, 0ull, 0ull, nullptr, probs_out, nullptr, 1);
    else if (row_fits_registers(logits, logits_row_stride, vocab))
        launch_resident<true>((unsigned)batch, st, logits, logits_row_stride, temperatures, top_ks, top_ps, min_ps,
                                           (int)vocab, 0ull, 0ull, nullptr, probs_out, nullptr, 1);
    else
        launch_filtered<false, true>((unsigned)batch, st, logits, logits_row_stride, temperatures, top_ks, top_ps, min_ps,
                                            (int)vocab, 0ull, 0ull, nullptr, probs_out, nullptr, 1);
    return nvx_check_launch("nvx_sampler_probs");
}

}  // namespace NVX_NS

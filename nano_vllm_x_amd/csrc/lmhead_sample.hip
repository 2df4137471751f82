// LM head of a decode step that draws its tokens as it computes its logits (gfx950).
//
// A decode step ends with logits = hidden (B, K) x W^T (vocab, K) and a categorical draw per row.  For the rows of a
// serving batch that is a stream over W (311 MB for Qwen3-0.6B, the largest single read of the step) followed by a
// pass over B x vocab logits.  Here both are one launch:
//   * W is read ONCE, straight into matrix-core operand registers: it has been repacked (nvx_lmhead_pack) into tiles of
//     16 vocabulary rows in which the 1 KB that one wave instruction loads is exactly the A operand of one
//     v_mfma_f32_16x16x32 step (lane = (row, k-block)), so every load is a fully coalesced 1 KB and nothing passes
//     through LDS;
//   * the batch's hidden states (<= 64 rows) sit in LDS in B-operand order for the whole launch; a workgroup per CU,
//     each wave takes every (4 x workgroups)-th tile with the next tile's 32 KB in flight while it multiplies;
//   * the accumulator layout gives every lane four consecutive tokens of one batch row: one Philox block.  The logit is
//     rounded to the 16-bit format (what the projection would have stored), divided by the temperature, and Gumbel
//     noise keyed by (seed, step, row, token) is added; the lane keeps its best (score, token).  Lanes, waves and
//     workgroups are reduced through 64-bit keys, so the token drawn is the one nvx_sample draws from the same logits
//     with the same seed and step (rows without top-k / top-p / min-p; rows with temperature <= 0 take the arg-max).
// Logits are written out only on request (tests; callers that need them).
#include "nvx_dev.h"
#include "sampler_dev.h"

namespace NVX_NS {

#define LMH_MAX_ROWS 64      // rows of one pass: what the LDS of a compute unit holds as operands
#define LMH_MAX_BATCH 1024   // rows of one call (ceil(batch / 64) passes over the weight)

__device__ __forceinline__ nvx_u32x4 lmh_stream16(const nvx_u32x4* p) { return __builtin_nontemporal_load(p); }

// w (n, k) row-major -> tiles[(n + 15) / 16][k / 32][64 lanes] of 16 bytes: lane (q = lane / 16, r = lane % 16) of
// k-step kk holds w[16 t + r][32 kk + 8 q .. + 7] (zeros for rows >= n).
__global__ __launch_bounds__(256) void lmhead_pack_kernel(const uint16_t* __restrict__ w, int64_t n, int k, nvx_u32x4* __restrict__ out,
                                                          int64_t total) {
    const int ks = k >> 5;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const int64_t blk = i >> 6;
        const int kk = (int)(blk % ks);
        const int64_t t = blk / ks;
        const int64_t row = t * 16 + (lane & 15);
        nvx_u32x4 v = {0u, 0u, 0u, 0u};
        if (row < n) v = *reinterpret_cast<const nvx_u32x4*>(w + row * k + kk * 32 + (lane >> 4) * 8);
        out[i] = v;
    }
}

template <int MT, int K>
__global__ __launch_bounds__(256, 1) void lmhead_sample_kernel(
    const nvx_u32x4* __restrict__ wp, const uint16_t* __restrict__ x, int64_t x_stride, int m_rows, int row0,
    const float* __restrict__ temps, int n_tiles, int tok0, int tok_end, uint64_t seed, uint64_t offset,
    unsigned long long* __restrict__ winners, uint16_t* __restrict__ logits_out, int64_t logits_stride, int n_groups, int group_rows) {
    // x, temps, winners and logits_out point at THIS launch's rows; row0 is the index of the first of them in the batch
    // (the noise of a row is keyed by its index in the whole batch).
    // n_groups = 2: the launch holds two row groups of group_rows rows (the second may be shorter: m_rows counts both).
    // Workgroups b and b + 8 run on the same XCD; they take the SAME tiles, one for each row group, so that a tile is
    // fetched from HBM once and found in that XCD's L2 by the other -- two passes over the weight for the HBM price of one.
    constexpr int KS = K / 32;
    const int group = n_groups == 2 ? (blockIdx.x >> 3) & 1 : 0;
    const int wg = n_groups == 2 ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 4) : blockIdx.x;
    const int wgs = gridDim.x / n_groups;
    if (n_groups == 2) {
        const int first = group * group_rows;
        x += (int64_t)first * x_stride;
        temps += first;
        winners += (int64_t)wgs * first;
        if (logits_out) logits_out += (int64_t)first * logits_stride;
        row0 += first;
        m_rows = group == 0 ? group_rows : m_rows - group_rows;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lmh_smem[];
    nvx_u32x4* xs = reinterpret_cast<nvx_u32x4*>(lmh_smem);   // [MT][KS][64]: the batch rows as B operands
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int quarter = lane >> 4, l16 = lane & 15;

    // The wave's current tile as A operands: 32 steps x 16 bytes per lane.  Each register is refilled with the NEXT
    // tile's piece right after the products that read it, so a whole tile (32 KB per wave) is always in flight and the
    // draw over the finished tile runs under it.  The first tile is requested before anything else: its trip to HBM
    // and the rows' trip to L2 below then overlap instead of following each other.
    const int stride_tiles = wgs * 4;
    int tile = wg * 4 + wave;
    nvx_u32x4 a[KS];
    if (tile < n_tiles) {
        const nvx_u32x4* wt = wp + (int64_t)tile * KS * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a[kk] = lmh_stream16(wt + kk * 64);
    }

    // the batch's rows -> LDS, in operand order (coalesced 16-byte reads along k; rows past the batch are zeros)
    for (int i = tid; i < MT * 16 * (K / 8); i += 256) {
        const int m = i / (K / 8), c = i % (K / 8);
        nvx_u32x4 v = {0u, 0u, 0u, 0u};
        if (m < m_rows) v = *reinterpret_cast<const nvx_u32x4*>(x + (int64_t)m * x_stride + c * 8);
        xs[((m >> 4) * KS + (c >> 2)) * 64 + (c & 3) * 16 + (m & 15)] = v;
    }
    __syncthreads();

    float inv_t[MT], best[MT];
    int besti[MT];
    bool greedy[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + l16;
        const float t = m < m_rows ? temps[m] : 1.0f;
        greedy[mt] = !(t > 0.0f);
        inv_t[mt] = greedy[mt] ? 1.0f : 1.0f / t;
        best[mt] = -INFINITY; besti[mt] = 0x7fffffff;
    }
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    // sched_barrier in the loop keeps the compiler from hoisting every LDS read of a tile to the top (it would need 128
    // registers per 16 batch rows)
    while (tile < n_tiles) {
        const int next = tile + stride_tiles;
        // (past the end: the same tile again, an L2 hit that is never used -- keeps the loop free of branches)
        const nvx_u32x4* wn = wp + (int64_t)(next < n_tiles ? next : tile) * KS * 64 + lane;
        nvx_f32x4 acc[MT];
        nvx_u32x4 b[MT], bn[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt] = nvx_f32x4{0.f, 0.f, 0.f, 0.f}; b[mt] = xs[(mt * KS) * 64 + lane]; }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) bn[mt] = xs[(mt * KS + kk + 1) * 64 + lane];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_16x16x32_h16(a[kk], b[mt], acc[mt]);
            a[kk] = lmh_stream16(wn + kk * 64);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) b[mt] = bn[mt];
            __builtin_amdgcn_sched_barrier(0);
        }
        // this lane: tokens tile * 16 + 4 q .. + 3 (one noise block) of batch row mt * 16 + l16
        const int base = tile * 16 + quarter * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 16 + l16;
            if (m < m_rows) {
                float lg[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) lg[j] = round_h16(acc[mt][j]);
                if (logits_out) {
                    *reinterpret_cast<uint2*>(logits_out + (int64_t)m * logits_stride + base) =
                        make_uint2(pack_h16x2(lg[0], lg[1]), pack_h16x2(lg[2], lg[3]));
                }
                uint4 r = make_uint4(0, 0, 0, 0);
                if (!greedy[mt]) r = philox4x32_10(make_uint4((uint32_t)((tok0 + base) >> 2), (uint32_t)(row0 + m), (uint32_t)offset, (uint32_t)(offset >> 32)), key);
                const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (tok0 + base + j < tok_end) {
                        const float v = lg[j] * inv_t[mt] + (greedy[mt] ? 0.0f : gumbel_from_bits(rr[j]));
                        if (v > best[mt]) { best[mt] = v; besti[mt] = tok0 + base + j; }
                    }
                }
            }
        }
        tile = next;
    }

    // lanes that hold other tokens of the same row (the four quarters), then the four waves through LDS
    __syncthreads();   // every wave is done with the operand rows: the memory is reused for the keys
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(lmh_smem);   // [4 waves][MT * 16]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        unsigned long long k64 = besti[mt] == 0x7fffffff ? 0ull : pack_winner(best[mt], besti[mt]);
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const unsigned long long o = __shfl_xor(k64, sh, 64);
            k64 = o > k64 ? o : k64;
        }
        if (quarter == 0) keys[wave * (MT * 16) + mt * 16 + l16] = k64;
    }
    __syncthreads();
    if (tid < MT * 16 && tid < m_rows) {
        unsigned long long k64 = keys[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const unsigned long long o = keys[w * (MT * 16) + tid]; k64 = o > k64 ? o : k64; }
        winners[(int64_t)wg * m_rows + tid] = k64;
    }
}

// ---------------------------------------------------------------------------------------------
// The same stream for a PROJECTION of a decode step whose weight is too large for the weight-stationary skinny kernel
// (tens of MB: the projections of Qwen3-8B and larger): y (m, n) = x (m, k) . W^T (n, k), k a multiple of 1024.  A
// workgroup keeps ONE 1024-wide slice of the batch's rows in LDS for its whole life and streams the tiles of that slice
// (32 KB each, same packed layout as the LM head: a slice of a tile is contiguous), so what it leaves is its slice's
// share of the products: fp32 partials (k / 1024, m, n) that the consumer sums -- nvx_rmsnorm_splitk after a
// row-parallel projection, nvx_splitk_sum (plain, or with SiluAndMul) otherwise.
template <int MT>
__global__ __launch_bounds__(256, 1) void linear_stream_kernel(
    const nvx_u32x4* __restrict__ wp, const uint16_t* __restrict__ x, int64_t x_stride, int m_rows, int n_tiles, int n_cols,
    int n_slices, float* __restrict__ partials) {
    constexpr int KS = 32;   // k-steps of a 1024-wide slice
    extern __shared__ __attribute__((aligned(16))) unsigned char lmh_smem[];
    nvx_u32x4* xs = reinterpret_cast<nvx_u32x4*>(lmh_smem);   // [MT][KS][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int quarter = lane >> 4, l16 = lane & 15;
    const int slice = blockIdx.x % n_slices, group = blockIdx.x / n_slices, n_groups = gridDim.x / n_slices;

    const int64_t tile_pitch = (int64_t)n_slices * KS * 64;            // 16-byte pieces per tile of 16 rows over all of k
    const nvx_u32x4* base = wp + (int64_t)slice * KS * 64 + lane;
    const int stride_tiles = n_groups * 4;
    int tile = group * 4 + wave;
    float* out = partials + (int64_t)slice * m_rows * n_cols;

    // the first tile is requested before the rows are staged: the two trips overlap
    nvx_u32x4 a[KS];
    if (tile < n_tiles) {
        const nvx_u32x4* wt = base + (int64_t)tile * tile_pitch;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a[kk] = lmh_stream16(wt + kk * 64);
    }
    for (int i = tid; i < MT * 16 * 128; i += 256) {
        const int m = i >> 7, c = i & 127;
        nvx_u32x4 v = {0u, 0u, 0u, 0u};
        if (m < m_rows) v = *reinterpret_cast<const nvx_u32x4*>(x + (int64_t)m * x_stride + slice * 1024 + c * 8);
        xs[((m >> 4) * KS + (c >> 2)) * 64 + (c & 3) * 16 + (m & 15)] = v;
    }
    __syncthreads();
    while (tile < n_tiles) {
        const int next = tile + stride_tiles;
        const nvx_u32x4* wn = base + (int64_t)(next < n_tiles ? next : tile) * tile_pitch;
        nvx_f32x4 acc[MT];
        nvx_u32x4 b[MT], bn[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt] = nvx_f32x4{0.f, 0.f, 0.f, 0.f}; b[mt] = xs[(mt * KS) * 64 + lane]; }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) bn[mt] = xs[(mt * KS + kk + 1) * 64 + lane];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_16x16x32_h16(a[kk], b[mt], acc[mt]);
            a[kk] = lmh_stream16(wn + kk * 64);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) b[mt] = bn[mt];
            __builtin_amdgcn_sched_barrier(0);
        }
        // this lane: columns tile * 16 + 4 q .. + 3 of batch row mt * 16 + l16
        const int col = tile * 16 + quarter * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 16 + l16;
            if (m < m_rows && col < n_cols)
                *reinterpret_cast<nvx_f32x4*>(out + (int64_t)m * n_cols + col) = acc[mt];
        }
        tile = next;
    }
}

// per row: the best of the workgroups' keys, as a key and / or as a token id.  One wave per row: the keys of a row are
// n_sets independent loads (a loop over them in one thread is n_sets memory round trips in a row).
__global__ __launch_bounds__(64) void lmhead_pick_kernel(const unsigned long long* __restrict__ winners, int n_sets, int batch,
                                                         unsigned long long* __restrict__ key_out, int64_t* __restrict__ out_ids,
                                                         int group_rows) {
    // (a launch of two row groups left its keys group by group: n_sets sets of the group's rows each)
    int row = blockIdx.x;
    const int lane = threadIdx.x;
    const int out_row = row;
    if (group_rows < batch) {
        const int g = row / group_rows, first = g * group_rows;
        winners += (int64_t)n_sets * first;
        batch = g == 0 ? group_rows : batch - first;
        row -= first;
    }
    unsigned long long w = 0;
    for (int s0 = 0; s0 < n_sets; s0 += 64 * 4) {
        unsigned long long o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int s = s0 + j * 64 + lane; o[j] = s < n_sets ? winners[(int64_t)s * batch + row] : 0ull; }
#pragma unroll
        for (int j = 0; j < 4; ++j) w = o[j] > w ? o[j] : w;
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) { const unsigned long long o = __shfl_xor(w, sh, 64); w = o > w ? o : w; }
    if (lane == 0) {
        if (key_out) key_out[out_row] = w;
        if (out_ids) out_ids[out_row] = w == 0 ? 0 : (int64_t)(0xffffffffu - (uint32_t)(w & 0xffffffffull));
    }
}

static int lmh_workgroups() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n = cus;
    }
    return n;
}

#ifndef NVX_F16   // format-independent helpers: defined by the bfloat16 build only
extern "C" int64_t nvx_lmhead_max_rows(void) { return LMH_MAX_BATCH; }
extern "C" int64_t nvx_lmhead_rows_per_pass(void) { return LMH_MAX_ROWS; }
extern "C" int64_t nvx_lmhead_packed_elems(int64_t n_rows, int64_t k) { return ((n_rows + 15) / 16) * 16 * k; }
extern "C" int64_t nvx_lmhead_ws_bytes(int64_t batch) { return (int64_t)1024 * (batch > 0 ? batch : 1) * 8; }

extern "C" int nvx_lmhead_pack(const void* w, int64_t n_rows, int64_t k, void* packed, nvx_stream_t stream) {
    NVX_REQUIRE(w && packed && n_rows > 0 && k > 0, NVX_EINVAL, "nvx_lmhead_pack: bad argument");
    NVX_REQUIRE(k % 32 == 0, NVX_EUNSUPPORTED, "nvx_lmhead_pack: k must be a multiple of 32 (got %lld)", (long long)k);
    NVX_REQUIRE(nvx_aligned16(w) && nvx_aligned16(packed), NVX_EINVAL, "nvx_lmhead_pack: pointers must be 16-byte aligned");
    const int64_t total = ((n_rows + 15) / 16) * (k / 32) * 64;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(lmhead_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)w, n_rows, (int)k,
                       (nvx_u32x4*)packed, total);
    return nvx_check_launch("nvx_lmhead_pack");
}
#endif

extern "C" int NVX_ENTRY(nvx_linear_stream)(const void* w_packed, int64_t n, int64_t k, const void* x, int64_t x_row_stride, int64_t m,
                                            float* partials, nvx_stream_t stream) {
    const char* who = "nvx_linear_stream";
    NVX_REQUIRE(m >= 0 && n > 0 && k > 0, NVX_EINVAL, "%s: bad size", who);
    if (m == 0) return NVX_OK;
    NVX_REQUIRE(w_packed && x && partials, NVX_EINVAL, "%s: null pointer", who);
    NVX_REQUIRE(m <= LMH_MAX_ROWS, NVX_ENOSPACE, "%s: %lld rows; at most %d stay in the LDS of a compute unit", who, (long long)m, LMH_MAX_ROWS);
    NVX_REQUIRE(k % 1024 == 0 && k <= 64 * 1024, NVX_EUNSUPPORTED, "%s: k must be a multiple of 1024, at most 65536 (got %lld)", who, (long long)k);
    NVX_REQUIRE(n % 4 == 0, NVX_EUNSUPPORTED, "%s: n must be a multiple of 4 (got %lld)", who, (long long)n);
    NVX_REQUIRE(x_row_stride % 8 == 0 && x_row_stride >= k && nvx_aligned16(x) && nvx_aligned16(w_packed) && nvx_aligned16(partials), NVX_EINVAL,
                "%s: rows must be 16-byte aligned", who);
    const int n_tiles = (int)((n + 15) / 16);
    const int n_slices = (int)(k / 1024);
    // a whole number of slice groups, one workgroup per compute unit where the tiles allow
    int groups = lmh_workgroups() / n_slices;
    if (groups < 1) groups = 1;
    if (groups > (n_tiles + 3) / 4) groups = (n_tiles + 3) / 4;
    const int grid = groups * n_slices;
    hipStream_t st = (hipStream_t)stream;
#define NVX_LST(MTV)                                                                                                                     \
    do {                                                                                                                                 \
        const size_t lds = (size_t)MTV * 32 * 1024;                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_stream_kernel<MTV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((linear_stream_kernel<MTV>), dim3((unsigned)grid), dim3(256), lds, st, (const nvx_u32x4*)w_packed, (const uint16_t*)x, \
                           x_row_stride, (int)m, n_tiles, (int)n, n_slices, partials);                                                   \
    } while (0)
    switch ((int)((m + 15) / 16)) {
        case 1: NVX_LST(1); break;
        case 2: NVX_LST(2); break;
        case 3: NVX_LST(3); break;
        default: NVX_LST(4); break;
    }
#undef NVX_LST
    return nvx_check_launch(who);
}

extern "C" int NVX_ENTRY(nvx_lmhead_sample)(const void* w_packed, int64_t n_rows, int64_t k, const void* x, int64_t x_row_stride,
                                            int64_t batch, const float* temperatures, int64_t vocab_offset, int64_t vocab,
                                            uint64_t seed, uint64_t offset, void* ws, int64_t ws_bytes, uint64_t* key_out,
                                            int64_t* out_ids, void* logits_out, int64_t logits_row_stride, nvx_stream_t stream) {
    const char* who = "nvx_lmhead_sample";
    NVX_REQUIRE(batch >= 0 && n_rows > 0 && vocab > 0 && vocab_offset >= 0, NVX_EINVAL, "%s: bad size", who);
    if (batch == 0) return NVX_OK;
    NVX_REQUIRE(w_packed && x && temperatures && ws && (key_out || out_ids), NVX_EINVAL, "%s: null pointer", who);
    NVX_REQUIRE(batch <= LMH_MAX_BATCH, NVX_ENOSPACE, "%s: %lld rows (at most %d; they are taken %d at a time, one pass over the weight each)",
                who, (long long)batch, LMH_MAX_BATCH, LMH_MAX_ROWS);
    NVX_REQUIRE(k == 1024, NVX_EUNSUPPORTED, "%s: built for hidden size 1024 (got %lld)", who, (long long)k);
    NVX_REQUIRE(vocab_offset % 4 == 0 && vocab_offset + n_rows <= 0x7fffffffll && vocab <= 0x7fffffffll, NVX_EINVAL,
                "%s: the rows must start at a multiple of 4 tokens and token ids stay below 2^31", who);
    NVX_REQUIRE(x_row_stride % 8 == 0 && x_row_stride >= k && nvx_aligned16(x) && nvx_aligned16(w_packed), NVX_EINVAL,
                "%s: rows must be 16-byte aligned", who);
    NVX_REQUIRE(!logits_out || (logits_row_stride % 4 == 0 && logits_row_stride >= ((n_rows + 15) / 16) * 16 &&
                                (reinterpret_cast<uintptr_t>(logits_out) & 7u) == 0), NVX_EINVAL,
                "%s: logits_out rows must hold the rows padded to a multiple of 16 and be 8-byte aligned", who);
    const int n_tiles = (int)((n_rows + 15) / 16);
    int grid = lmh_workgroups();
    if (grid > (n_tiles + 3) / 4) grid = (n_tiles + 3) / 4;
    if (grid > 1024) grid = 1024;
    NVX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7u) == 0 && ws_bytes >= (int64_t)grid * batch * 8, NVX_ENOSPACE,
                "%s: workspace too small (%lld < %lld bytes)", who, (long long)ws_bytes, (long long)((int64_t)grid * batch * 8));
    const int64_t end = vocab_offset + n_rows < vocab ? vocab_offset + n_rows : vocab;
    hipStream_t st = (hipStream_t)stream;
    // more rows than a compute unit holds: passes of equal height (two passes of 62 and 61 rows, not 64 and 59)
    const int passes = (int)((batch + LMH_MAX_ROWS - 1) / LMH_MAX_ROWS);
    const int per_pass = (int)((batch + passes - 1) / passes);
#define NVX_LMH(MTV)                                                                                                                       \
    do {                                                                                                                                   \
        const size_t lds = (size_t)MTV * (1024 / 32) * 1024;                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lmhead_sample_kernel<MTV, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((lmhead_sample_kernel<MTV, 1024>), dim3((unsigned)grid), dim3(256), lds, st, (const nvx_u32x4*)w_packed,         \
                           (const uint16_t*)x + (int64_t)r0 * x_row_stride, x_row_stride, rows, r0, temperatures + r0, n_tiles,             \
                           (int)vocab_offset, (int)end, seed, offset, (unsigned long long*)ws + (int64_t)(grid / groups) * r0,            \
                           logits_out ? (uint16_t*)logits_out + (int64_t)r0 * logits_row_stride : (uint16_t*)nullptr, logits_row_stride,   \
                           groups, per_pass);                                                                                              \
    } while (0)
    // Two passes go out as ONE launch of two row groups on paired workgroups (see the kernel) where the grid allows the
    // pairing; more passes follow one another.
    const int groups = (passes == 2 && grid % 16 == 0) ? 2 : 1;
    for (int r0 = 0; r0 < (int)batch; r0 += per_pass * groups) {
        const int rows = groups == 2 ? (int)batch : ((int)batch - r0 < per_pass ? (int)batch - r0 : per_pass);
        switch ((per_pass < rows ? per_pass + 15 : rows + 15) / 16) {
            case 1: NVX_LMH(1); break;
            case 2: NVX_LMH(2); break;
            case 3: NVX_LMH(3); break;
            default: NVX_LMH(4); break;
        }
        int rc = nvx_check_launch(who);
        if (rc) return rc;
        hipLaunchKernelGGL(lmhead_pick_kernel, dim3((unsigned)rows), dim3(64), 0, st, (const unsigned long long*)ws + (int64_t)(grid / groups) * r0,
                           grid / groups, rows, key_out ? (unsigned long long*)key_out + r0 : (unsigned long long*)nullptr,
                           out_ids ? out_ids + r0 : nullptr, groups == 2 ? per_pass : rows);
        rc = nvx_check_launch("nvx_lmhead_sample(pick)");
        if (rc) return rc;
    }
#undef NVX_LMH
    return NVX_OK;
}

}  // namespace NVX_NS

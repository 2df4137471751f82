// Paged decode attention for gfx950: one query token per sequence, page size 1.
//
// The kernel is HBM-bound (every K and V row of every live sequence is read once per layer) and, at
// serving batch sizes, latency-bound per work unit, so the design goals are bytes in flight and a
// short dependent-load chain, not FLOPs:
//   * plan (device side, once per step): all sequences are cut into chunks of one common length so
//     that (chunks x kv heads) fills the chip at any batch size, without a host round trip and with a
//     fixed grid (graph-capturable).  Each chunk gets one 32-byte record holding everything a
//     workgroup needs, so the chain is record -> slot ids -> K/V rows (three loads deep);
//   * a workgroup (4 waves) owns one (chunk, kv head); each 16-lane quarter-wave owns one token at a
//     time: lane c of the quarter loads the 16-byte piece c of that token's 256-byte K row and of
//     its V row straight into VGPRs (no LDS round trip: each byte is used by exactly one wave); the
//     slot ids of the next tile are fetched while the current tile is being reduced;
//   * q.k by v_dot2c_f32_bf16 on the raw packed words + a 16-lane DPP butterfly; softmax in fp32 in
//     the exp2 domain, one rescale per tile; p.V by fp32 FMAs; the G query heads of a kv head
//     (grouped-query attention) share every loaded row;
//   * quarter-waves are merged by wave shuffles, waves through LDS; sequences split over several
//     chunks leave (m, l, acc) partials that a small second kernel combines.
#include "nvx_dev.h"

#define DEC_TARGET_UNITS 2048   // work units (chunk x kv head) the plan aims for
#define DEC_MIN_CHUNK 128       // tokens
#define DEC_MAX_CHUNK (1 << 24) // tokens: "do not split"
#define DEC_PART_STRIDE 132     // floats per (item, q head): 128 acc + m + l (+2 pad)
#define DEC_REC_INTS 8          // per item: seq, n_tokens, n_chunks_of_seq, chunk_index, abs_lo, abs_hi, 0, 0
#define DEC_LOG2E 1.4426950408889634f

static inline int64_t dec_max_items(int64_t max_batch, int n_kv_heads) {
    return max_batch + 2048 / (n_kv_heads > 0 ? n_kv_heads : 1) + 8;
}

extern "C" int64_t nvx_decode_plan_ws_ints(int64_t max_batch, int n_kv_heads) {
    return 8 + DEC_REC_INTS * dec_max_items(max_batch, n_kv_heads) + 2 * max_batch;
}
extern "C" int64_t nvx_decode_partial_ws_floats(int64_t max_batch, int n_q_heads, int n_kv_heads) {
    return dec_max_items(max_batch, n_kv_heads) * n_q_heads * DEC_PART_STRIDE;
}

// plan_ws layout (int32): [0] n_items, [1] chunk, [2] batch, [3] max_items, [4..7] reserved,
//   records[max_items][8], seq_first[batch], seq_nchunks[batch]
__global__ __launch_bounds__(1024) void decode_plan_kernel(const int32_t* __restrict__ kv_len,
                                                           const int64_t* __restrict__ kv_start, int batch, int hkv,
                                                           int32_t* __restrict__ ws, int max_items, int target_units) {
    __shared__ long long red[16];
    __shared__ int scan[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    long long local = 0;
    for (int b = tid; b < batch; b += 1024) { const int n = kv_len[b]; local += n > 0 ? n : 0; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) local += __shfl_xor(local, m, 64);
    if ((tid & 63) == 0) red[tid >> 6] = local;
    if (tid == 0) carry = 0;
    __syncthreads();
    long long total = 0;
    for (int i = 0; i < 16; ++i) total += red[i];
    // Chunks sized for about `target_units` work units of similar length.  (Leaving sequences whole
    // at large batch saves the merge but was measured slower: units of unequal length leave a tail.)
    long long c = (total * hkv + target_units - 1) / target_units;
    c = ((c + DEC_MIN_CHUNK - 1) / DEC_MIN_CHUNK) * DEC_MIN_CHUNK;
    if (c < DEC_MIN_CHUNK) c = DEC_MIN_CHUNK;
    const int chunk = (int)(c > DEC_MAX_CHUNK ? DEC_MAX_CHUNK : c);   // small enough that len + chunk cannot overflow
    int32_t* recs = ws + 8;
    int32_t* seq_first = recs + (int64_t)DEC_REC_INTS * max_items;
    int32_t* seq_nch = seq_first + batch;
    for (int base = 0; base < batch; base += 1024) {
        const int b = base + tid;
        int nch = 0, len = 0;
        if (b < batch) { len = kv_len[b]; len = len > 0 ? len : 0; nch = len > 0 ? (int)(((long long)len + chunk - 1) / chunk) : 1; }
        scan[tid] = nch;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            const int v = tid >= off ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int first = carry + scan[tid] - nch;
        if (b < batch) {
            seq_first[b] = first;
            seq_nch[b] = nch;
            const int64_t start = kv_start[b];
            for (int i = 0; i < nch && first + i < max_items; ++i) {
                const long long left = (long long)len - (long long)i * chunk;
                const int n = (int)(left > chunk ? chunk : left);
                const int64_t abs = start + (int64_t)i * chunk;
                int32_t* r = recs + (int64_t)DEC_REC_INTS * (first + i);
                r[0] = b; r[1] = n; r[2] = nch; r[3] = i;
                r[4] = (int32_t)(abs & 0xffffffffll); r[5] = (int32_t)(abs >> 32); r[6] = 0; r[7] = 0;
            }
        }
        __syncthreads();
        if (tid == 1023) carry += scan[1023];
        __syncthreads();
    }
    if (tid == 0) {
        ws[0] = carry < max_items ? carry : max_items;
        ws[1] = chunk;
        ws[2] = batch;
        ws[3] = max_items;
    }
}

extern "C" int nvx_decode_plan(const int32_t* kv_len, const int64_t* kv_start, int64_t batch, int n_kv_heads,
                               int32_t* plan_ws, int64_t plan_ws_ints, nvx_stream_t stream) {
    NVX_REQUIRE(batch > 0 && n_kv_heads > 0, NVX_EINVAL, "nvx_decode_plan: bad size");
    NVX_REQUIRE(kv_len && kv_start && plan_ws, NVX_EINVAL, "nvx_decode_plan: null pointer");
    NVX_REQUIRE(plan_ws_ints >= nvx_decode_plan_ws_ints(batch, n_kv_heads), NVX_ENOSPACE,
                "nvx_decode_plan: plan workspace too small (%lld < %lld ints)", (long long)plan_ws_ints,
                (long long)nvx_decode_plan_ws_ints(batch, n_kv_heads));
    hipLaunchKernelGGL(decode_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, kv_len, kv_start, (int)batch,
                       n_kv_heads, plan_ws, (int)dec_max_items(batch, n_kv_heads), DEC_TARGET_UNITS);
    return nvx_check_launch("nvx_decode_plan");
}

// ---------------------------------------------------------------------------------------------
// One wave's tile in registers: TPQ tokens per quarter-wave, K and V piece of each.
template <int TPQ>
struct DecTile {
    nvx_u32x4 k[TPQ];
    nvx_u32x4 v[TPQ];
    unsigned ok;   // bit u: token u of this quarter is a real, in-range key
};

template <int TPQ>
__device__ __forceinline__ void dec_load_slots(int (&slots)[TPQ], const int32_t* __restrict__ idx_base, int tile, int quarter, int n) {
#pragma unroll
    for (int u = 0; u < TPQ; ++u) {
        const int t = tile + u * 4 + quarter;
        slots[u] = t < n ? idx_base[t] : -1;
    }
}

// Issue the row loads of one tile.  Unsigned 32-bit slot arithmetic keeps the address at four VALU ops
// per row (compare, select, 64-bit multiply-add, add).
template <int TPQ>
__device__ __forceinline__ void dec_issue(DecTile<TPQ>& t, const int (&slots)[TPQ], const char* __restrict__ kb,
                                          const char* __restrict__ vb, unsigned row_bytes, unsigned n_slots) {
    t.ok = 0;
#pragma unroll
    for (int u = 0; u < TPQ; ++u) {
        unsigned slot = (unsigned)slots[u];   // -1 -> 0xffffffff: out of range
        const bool ok = slot < n_slots;
        slot = ok ? slot : 0u;
        t.ok |= ok ? (1u << u) : 0u;
        const unsigned long long off = (unsigned long long)slot * row_bytes;
        t.k[u] = *reinterpret_cast<const nvx_u32x4*>(kb + off);
        t.v[u] = *reinterpret_cast<const nvx_u32x4*>(vb + off);
    }
}

template <int G, int TPQ>
__device__ __forceinline__ void dec_compute(const DecTile<TPQ>& t, const nvx_u32x4 (&qv)[G], float (&m)[G], float (&l)[G],
                                            float (&acc)[G][8], float scale_log2) {
    float s[TPQ][G];
#pragma unroll
    for (int u = 0; u < TPQ; ++u) {
        const bool ok = (t.ok >> u) & 1u;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float d = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                // copies first: __builtin_bit_cast straight from a vector-element lvalue reads element 0
                const unsigned qw = qv[g][w], kw = t.k[u][w];
                d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(nvx_bf16x2, qw), __builtin_bit_cast(nvx_bf16x2, kw), d, false);
            }
            d = quarter_sum(d);
            s[u][g] = ok ? d * scale_log2 : -INFINITY;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float mx = m[g];
#pragma unroll
        for (int u = 0; u < TPQ; ++u) mx = fmaxf(mx, s[u][g]);
        const float mref = mx == -INFINITY ? 0.f : mx;
        const float alpha = __builtin_amdgcn_exp2f(m[g] - mref);
        m[g] = mx;
        l[g] *= alpha;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] *= alpha;
#pragma unroll
        for (int u = 0; u < TPQ; ++u) {
            const float p = __builtin_amdgcn_exp2f(s[u][g] - mref);   // masked key: exp2(-inf) = 0
            l[g] += p;
            const bool ok = (t.ok >> u) & 1u;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned vw = ok ? t.v[u][w] : 0u;   // p = 0 times a stale NaN would still be NaN
                acc[g][2 * w] += p * bf16lo_to_f32(vw);
                acc[g][2 * w + 1] += p * bf16hi_to_f32(vw);
            }
        }
    }
}

// G = query heads per kv head, TPQ = tokens per quarter-wave per tile (a wave's tile is 4*TPQ tokens).
template <int G, int TPQ>
__global__ __launch_bounds__(256) void paged_decode_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ k_cache,
    const uint16_t* __restrict__ v_cache, int64_t n_slots, const int32_t* __restrict__ kv_indices,
    const int32_t* __restrict__ plan, float* __restrict__ partial, uint16_t* __restrict__ out, int hkv,
    float scale_log2) {
    constexpr int TILE = 4 * TPQ;
    constexpr int STEP = 4 * TILE;   // four waves interleave their tiles
    __shared__ float lds[4][G][16][10];  // [wave][head in group][piece] -> m, l, acc[8]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int quarter = lane >> 4, piece = lane & 15;
    const int hq_total = hkv * G;
    const unsigned row_bytes = (unsigned)hkv * 256u;
    const unsigned n_slots_u = n_slots > 0xffffffffll ? 0xffffffffu : (unsigned)n_slots;
    const int32_t* recs = plan + 8;

    // The record of this block's first unit is fetched before the header says whether the unit exists:
    // blockIdx.x / hkv < max_items always, so the address is valid either way.
    int unit = blockIdx.x;
    nvx_u32x4 r0 = *reinterpret_cast<const nvx_u32x4*>(recs + (int64_t)DEC_REC_INTS * (unit / hkv));
    nvx_u32x4 r1 = *reinterpret_cast<const nvx_u32x4*>(recs + (int64_t)DEC_REC_INTS * (unit / hkv) + 4);
    const int n_units = plan[0] * hkv;

    while (unit < n_units) {
        const int item = unit / hkv, head = unit % hkv;
        const int b = (int)r0[0], n = (int)r0[1], nch = (int)r0[2];
        const int64_t abs = (int64_t)(((uint64_t)r1[1] << 32) | (uint64_t)r1[0]);
        const int32_t* idx_base = kv_indices + abs;
        const char* kb = reinterpret_cast<const char*>(k_cache) + head * 256 + piece * 16;
        const char* vb = reinterpret_cast<const char*>(v_cache) + head * 256 + piece * 16;

        int slots[TPQ];
        int tile = wave * TILE;
        dec_load_slots<TPQ>(slots, idx_base, tile, quarter, n);

        nvx_u32x4 qv[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
            qv[g] = *reinterpret_cast<const nvx_u32x4*>(q + (int64_t)b * q_stride + (head * G + g) * 128 + piece * 8);

        float m[G], l[G], acc[G][8];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
        }

        // Rows of one tile are in flight while the slot ids of the next are fetched; the tile is then
        // reduced.  (Keeping a second tile of rows in flight per wave was measured slower: the registers
        // it takes halve the number of resident waves.)
        DecTile<TPQ> ta;
        while (tile < n) {
            dec_issue<TPQ>(ta, slots, kb, vb, row_bytes, n_slots_u);
            dec_load_slots<TPQ>(slots, idx_base, tile + STEP, quarter, n);
            dec_compute<G, TPQ>(ta, qv, m, l, acc, scale_log2);
            tile += STEP;
        }

        // merge the four quarter-waves (they hold different tokens, same dims)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mall = fmaxf(m[g], __shfl_xor(m[g], 16, 64));
            mall = fmaxf(mall, __shfl_xor(mall, 32, 64));
            const float mref = mall == -INFINITY ? 0.f : mall;
            const float sc = __builtin_amdgcn_exp2f(m[g] - mref);
            float lv = l[g] * sc;
            lv += __shfl_xor(lv, 16, 64);
            lv += __shfl_xor(lv, 32, 64);
            l[g] = lv; m[g] = mall;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = acc[g][j] * sc;
                a += __shfl_xor(a, 16, 64);
                a += __shfl_xor(a, 32, 64);
                acc[g][j] = a;
            }
            if (quarter == 0) {
                lds[wave][g][piece][0] = m[g];
                lds[wave][g][piece][1] = l[g];
#pragma unroll
                for (int j = 0; j < 8; ++j) lds[wave][g][piece][2 + j] = acc[g][j];
            }
        }
        // the next unit's record travels while the waves meet
        const int next_unit = unit + gridDim.x;
        if (next_unit < n_units) {
            r0 = *reinterpret_cast<const nvx_u32x4*>(recs + (int64_t)DEC_REC_INTS * (next_unit / hkv));
            r1 = *reinterpret_cast<const nvx_u32x4*>(recs + (int64_t)DEC_REC_INTS * (next_unit / hkv) + 4);
        }
        __syncthreads();
        if (tid < 16 * G) {
            const int g = tid >> 4, pc = tid & 15;
            float mall = -INFINITY;
#pragma unroll
            for (int w = 0; w < 4; ++w) mall = fmaxf(mall, lds[w][g][pc][0]);
            const float mref = mall == -INFINITY ? 0.f : mall;
            float lsum = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float sc = __builtin_amdgcn_exp2f(lds[w][g][pc][0] - mref);
                lsum += lds[w][g][pc][1] * sc;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += lds[w][g][pc][2 + j] * sc;
            }
            const int hq = head * G + g;
            if (nch == 1) {
                const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
                uint16_t* dst = out + ((int64_t)b * hq_total + hq) * 128 + pc * 8;
                *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(o[0] * inv, o[1] * inv), pack_bf16x2(o[2] * inv, o[3] * inv),
                                                            pack_bf16x2(o[4] * inv, o[5] * inv), pack_bf16x2(o[6] * inv, o[7] * inv));
            } else {
                float* dst = partial + ((int64_t)item * hq_total + hq) * DEC_PART_STRIDE;
                *reinterpret_cast<float4*>(dst + pc * 8) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(dst + pc * 8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
                if (pc == 0) { dst[128] = mall; dst[129] = lsum; }
            }
        }
        __syncthreads();
        unit = next_unit;
    }
}

// Combine the partials of sequences that were split over several chunks: one 128-thread block per
// (sequence, query head), thread d owns output element d.  Blocks of unsplit sequences exit at once.
__global__ __launch_bounds__(128) void decode_merge_kernel(const int32_t* __restrict__ plan, const float* __restrict__ partial,
                                                           uint16_t* __restrict__ out, int hq_total) {
    const int batch = plan[2], max_items = plan[3];
    const int b = blockIdx.x / hq_total, hq = blockIdx.x % hq_total;
    if (b >= batch) return;
    const int32_t* seq_first = plan + 8 + (int64_t)DEC_REC_INTS * max_items;
    const int nch = seq_first[batch + b];
    if (nch <= 1) return;
    const int first = seq_first[b];
    const int d = threadIdx.x;
    const float* base = partial + ((int64_t)first * hq_total + hq) * DEC_PART_STRIDE;
    const int64_t step = (int64_t)hq_total * DEC_PART_STRIDE;
    float mall = -INFINITY;
    for (int i = 0; i < nch; ++i) mall = fmaxf(mall, base[i * step + 128]);
    const float mref = mall == -INFINITY ? 0.f : mall;
    float lsum = 0.f, o = 0.f;
#pragma unroll 4
    for (int i = 0; i < nch; ++i) {
        const float* p = base + i * step;
        const float sc = __builtin_amdgcn_exp2f(p[128] - mref);
        lsum += p[129] * sc;
        o += p[d] * sc;
    }
    out[((int64_t)b * hq_total + hq) * 128 + d] = f32_to_bf16bits(lsum > 0.f ? o / lsum : 0.f);
}

extern "C" int nvx_paged_decode(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                                int64_t n_slots, const int32_t* kv_indices, const int32_t* plan_ws, float* partial_ws,
                                void* out, int64_t batch, int n_q_heads, int n_kv_heads, float sm_scale,
                                nvx_stream_t stream) {
    NVX_REQUIRE(batch > 0 && n_q_heads > 0 && n_kv_heads > 0 && n_slots > 0, NVX_EINVAL, "nvx_paged_decode: bad size");
    NVX_REQUIRE(q && k_cache && v_cache && kv_indices && plan_ws && partial_ws && out, NVX_EINVAL,
                "nvx_paged_decode: null pointer");
    NVX_REQUIRE(n_q_heads % n_kv_heads == 0, NVX_EINVAL, "nvx_paged_decode: n_q_heads must be a multiple of n_kv_heads");
    NVX_REQUIRE(q_row_stride % 8 == 0 && q_row_stride >= (int64_t)n_q_heads * 128, NVX_EINVAL, "nvx_paged_decode: bad q row stride");
    NVX_REQUIRE(nvx_aligned16(q) && nvx_aligned16(k_cache) && nvx_aligned16(v_cache) && nvx_aligned16(out) &&
                nvx_aligned16(partial_ws) && nvx_aligned16(plan_ws), NVX_EINVAL, "nvx_paged_decode: pointers must be 16-byte aligned");
    const int g = n_q_heads / n_kv_heads;
    const float scale_log2 = sm_scale * DEC_LOG2E;
    hipStream_t st = (hipStream_t)stream;
    const int64_t max_units = dec_max_items(batch, n_kv_heads) * n_kv_heads;
    unsigned grid = (unsigned)(max_units < 2048 ? max_units : 2048);
#define NVX_DEC(GG, TT) hipLaunchKernelGGL((paged_decode_kernel<GG, TT>), dim3(grid), dim3(256), 0, st, (const uint16_t*)q,    \
        q_row_stride, (const uint16_t*)k_cache, (const uint16_t*)v_cache, n_slots, kv_indices, plan_ws, partial_ws,             \
        (uint16_t*)out, n_kv_heads, scale_log2)
    switch (g) {
        case 1: NVX_DEC(1, 8); break;
        case 2: NVX_DEC(2, 8); break;
        case 4: NVX_DEC(4, 4); break;
        case 8: NVX_DEC(8, 2); break;
        default: return nvx_fail(NVX_EUNSUPPORTED, "nvx_paged_decode: query/kv head ratio %d not in {1,2,4,8}", g);
    }
#undef NVX_DEC
    int rc = nvx_check_launch("nvx_paged_decode");
    if (rc) return rc;
    hipLaunchKernelGGL(decode_merge_kernel, dim3((unsigned)(batch * n_q_heads)), dim3(128), 0, st, plan_ws, partial_ws,
                       (uint16_t*)out, n_q_heads);
    return nvx_check_launch("nvx_paged_decode(merge)");
}

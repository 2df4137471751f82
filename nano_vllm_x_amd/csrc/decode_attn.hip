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
#include "rope_dev.h"

#include <mutex>
#include <stdlib.h>

namespace NVX_NS {


#define DEC_TARGET_UNITS 2048   // work units (chunk x kv head) the plan aims for
#define DEC_MIN_CHUNK 128       // tokens
#define DEC_MAX_CHUNK (1 << 24) // tokens: "do not split"
#define DEC_PART_STRIDE 132     // floats per (item, q head): 128 acc + m + l (+2 pad)
#define DEC_REC_INTS 8          // per item: seq, n_tokens, n_chunks_of_seq, chunk_index, abs_lo, abs_hi, 0, new_slot
#define DEC_LOG2E 1.4426950408889634f
#ifndef DEC_TQ_G2
#define DEC_TQ_G2 2
#endif
#ifndef DEC_G4_MIN_WGS
#define DEC_G4_MIN_WGS 2        // workgroups per CU the compiler must leave room for at three or four query heads per kv head
#endif
#ifndef DEC_TQ_G4
#define DEC_TQ_G4 2
#endif
#ifndef DEC_SHARE_Q_MIN_G
#define DEC_SHARE_Q_MIN_G 2     // smallest query group whose heads are prepared once per workgroup (QKV variant, register kernel)
#endif

static inline int64_t dec_max_items(int64_t max_batch, int n_kv_heads) {
    return max_batch + 2048 / (n_kv_heads > 0 ? n_kv_heads : 1) + 8;
}

#ifndef NVX_F16   // independent of the 16-bit format: defined by the bfloat16 build only
extern "C" int64_t nvx_decode_plan_ws_ints(int64_t max_batch, int n_kv_heads) {
    return 8 + DEC_REC_INTS * dec_max_items(max_batch, n_kv_heads) + 2 * max_batch;
}
extern "C" int64_t nvx_decode_partial_ws_floats(int64_t max_batch, int n_q_heads, int n_kv_heads) {
    return dec_max_items(max_batch, n_kv_heads) * n_q_heads * DEC_PART_STRIDE;
}

#endif
#ifndef NVX_F16   // independent of the 16-bit format: defined by the bfloat16 build only
// plan_ws layout (int32): [0] n_items, [1] chunk, [2] batch, [3] max_items, [4..7] reserved,
//   records[max_items][8], seq_info[batch][2] = (first item, number of chunks) of every sequence
__global__ __launch_bounds__(1024) void decode_plan_kernel(const int32_t* __restrict__ kv_len,
                                                           const int64_t* __restrict__ kv_start,
                                                           const int64_t* __restrict__ out_slots, int batch, int hkv,
                                                           int32_t* __restrict__ ws, int max_items, int target_units,
                                                           int min_chunk, int balanced) {
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
    c = ((c + min_chunk - 1) / min_chunk) * min_chunk;
    if (c < min_chunk) c = min_chunk;
    const int chunk = (int)(c > DEC_MAX_CHUNK ? DEC_MAX_CHUNK : c);   // small enough that len + chunk cannot overflow
    int32_t* recs = ws + 8;
    int32_t* seq_info = recs + (int64_t)DEC_REC_INTS * max_items;
    for (int base = 0; base < batch; base += 1024) {
        const int b = base + tid;
        int nch = 0, len = 0, per = chunk;
        if (b < batch) {
            len = kv_len[b]; len = len > 0 ? len : 0; nch = len > 0 ? (int)(((long long)len + chunk - 1) / chunk) : 1;
            if (balanced && nch > 1) {   // chunks of one sequence of equal length (a multiple of 32 tokens) instead of a short last one
                per = (int)((((long long)len + nch - 1) / nch + 31) / 32 * 32);
                nch = (int)(((long long)len + per - 1) / per);
            }
        }
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
            seq_info[2 * b] = first;
            seq_info[2 * b + 1] = nch;
            const int64_t start = kv_start[b];
            // slot the new token's K/V row goes to, recorded on the sequence's last chunk (QKV variant)
            long long new_slot = out_slots ? out_slots[b] : -1;
            if (new_slot < 0 || new_slot > 0x7fffffffll || len == 0) new_slot = -1;
            for (int i = 0; i < nch && first + i < max_items; ++i) {
                const long long left = (long long)len - (long long)i * per;
                const int n = (int)(left > per ? per : left);
                const int64_t abs = start + (int64_t)i * per;
                int32_t* r = recs + (int64_t)DEC_REC_INTS * (first + i);
                r[0] = b; r[1] = n; r[2] = nch; r[3] = i;
                r[4] = (int32_t)(abs & 0xffffffffll); r[5] = (int32_t)(abs >> 32); r[6] = 0; r[7] = i == nch - 1 ? (int32_t)new_slot : -1;
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

extern "C" int nvx_decode_plan(const int32_t* kv_len, const int64_t* kv_start, const int64_t* out_slots, int64_t batch,
                               int n_kv_heads, int32_t* plan_ws, int64_t plan_ws_ints, nvx_stream_t stream) {
    NVX_REQUIRE(batch > 0 && n_kv_heads > 0, NVX_EINVAL, "nvx_decode_plan: bad size");
    NVX_REQUIRE(kv_len && kv_start && plan_ws, NVX_EINVAL, "nvx_decode_plan: null pointer");
    NVX_REQUIRE(plan_ws_ints >= nvx_decode_plan_ws_ints(batch, n_kv_heads), NVX_ENOSPACE,
                "nvx_decode_plan: plan workspace too small (%lld < %lld ints)", (long long)plan_ws_ints,
                (long long)nvx_decode_plan_ws_ints(batch, n_kv_heads));
    // work units per launch the chunking aims at: NVX_DEC_TARGET (64..2048) if set; else fewer, longer chunks where one
    // or two kv heads share the chip (a rank of a tensor-parallel group: 16/2 heads 16.8 -> 15.4 us, 8/2 14.9 -> 13.8 on
    // the bench's batches; no difference at one kv head), DEC_TARGET_UNITS otherwise (16/8: 24.7 us vs 27.3 at 512)
    static int env_target = -1;
    if (env_target < 0) { const char* e = getenv("NVX_DEC_TARGET"); env_target = e ? atoi(e) : 0; if (env_target < 64 || env_target > 2048) env_target = 0; }
    const int target = env_target ? env_target : (n_kv_heads <= 2 ? 512 : DEC_TARGET_UNITS);
    static int env_min = -1, env_bal = -1;
    if (env_min < 0) { const char* e = getenv("NVX_DEC_MIN_CHUNK"); env_min = e ? atoi(e) : 0; if (env_min < 32 || env_min > 4096 || env_min % 32) env_min = DEC_MIN_CHUNK; }
    if (env_bal < 0) { const char* e = getenv("NVX_DEC_BALANCED"); env_bal = e ? (atoi(e) ? 1 : 0) : 0; }
    hipLaunchKernelGGL(decode_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, kv_len, kv_start, out_slots, (int)batch,
                       n_kv_heads, plan_ws, (int)dec_max_items(batch, n_kv_heads), target, env_min, env_bal);
    return nvx_check_launch("nvx_decode_plan");
}

#endif
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
        // every K/V byte is used once per launch: streaming (non-temporal) loads keep them from
        // displacing the block table and the partials in L2 (measured: 29.0 -> 26.2 us per launch)
        t.k[u] = __builtin_nontemporal_load(reinterpret_cast<const nvx_u32x4*>(kb + off));
        t.v[u] = __builtin_nontemporal_load(reinterpret_cast<const nvx_u32x4*>(vb + off));
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
                d = dot2_h16(qw, kw, d);
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
                acc[g][2 * w] += p * h16lo_to_f32(vw);
                acc[g][2 * w + 1] += p * h16hi_to_f32(vw);
            }
        }
    }
}

// What the kernel needs to make its own query and new key from the raw fused projection (QKV variant):
// per-head RMSNorm + rotary embedding of q and k, and the scatter of the new k/v row.
struct DecQkv {
    const int64_t* positions;   // (batch)
    const float* cos_sin;       // (max_pos, 128) fp32
    int64_t max_pos;
    const uint16_t* q_weight;   // (128) bf16
    const uint16_t* k_weight;
    float eps;
    uint16_t* k_cache_w;        // same pools, writable
    uint16_t* v_cache_w;
};

__device__ __forceinline__ nvx_u32x4 dec_new_key(const uint16_t* row, int hq_total, int head, const DecQkv& fx, const float* cs,
                                                 int piece) {
    const HalfQuad x = load_half_quads(row + (hq_total + head) * 128, piece);
    const uint4 o = norm_rotate_interleave(x, half_quad_sumsq(x), fx.k_weight, fx.eps, cs, piece);
    return nvx_u32x4{o.x, o.y, o.z, o.w};
}

// G = query heads per kv head, TPQ = tokens per quarter-wave per tile (a wave's tile is 4*TPQ tokens).
// QKV = false: q holds finished queries (batch, G*hkv, 128).
// QKV = true:  q holds the raw projection rows [q heads | k heads | v heads]; the kernel normalises and
//   rotates its own query heads, and the workgroup that owns a sequence's last chunk does the same for
//   the new key, writes the new k/v row into the pool and attends to it from registers (the row is the
//   last entry of the sequence's slot list and is skipped when reading the pool).
template <int G, int TPQ, bool QKV>
__global__ __launch_bounds__(256, ((G <= 2 && (TPQ <= 2 || (TPQ <= 4 && !QKV))) ? 4 : (G <= 4 ? DEC_G4_MIN_WGS : 2))) void paged_decode_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ k_cache,
    const uint16_t* __restrict__ v_cache, int64_t n_slots, const int32_t* __restrict__ kv_indices,
    const int32_t* __restrict__ plan, float* __restrict__ partial, uint16_t* __restrict__ out, int hkv,
    float scale_log2, DecQkv fx) {
    constexpr int TILE = 4 * TPQ;
    constexpr int STEP = 4 * TILE;   // four waves interleave their tiles
    // QKV variant, wide groups: the G query heads are normalised and rotated ONCE per workgroup (quarter-wave i of the
    // sixteen takes head i) and handed round through LDS, instead of by every quarter-wave for itself
    constexpr bool SHARE_Q = QKV && G >= DEC_SHARE_Q_MIN_G;
    __shared__ float lds[4][G][16][10];  // [wave][head in group][piece] -> m, l, acc[8]
    __shared__ nvx_u32x4 q_shared[SHARE_Q ? G : 1][16];
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

        // QKV variant: the last chunk of a sequence whose new row's slot is in the record attends to that
        // row from registers and reads one token less from the pool
        const int slot_new = QKV ? __builtin_amdgcn_readfirstlane((int)r1[3]) : -1;
        const bool own_new = QKV && slot_new >= 0 && (long long)slot_new < (long long)n_slots && n > 0;
        const int n_pool = own_new ? n - 1 : n;

        // dependent chain: record -> slot ids -> K/V rows.  The query (and, in the QKV variant, its
        // norm and rotation) is fetched beside the slot ids and finished while the first rows fly.
        int slots[TPQ];
        int tile = wave * TILE;
        dec_load_slots<TPQ>(slots, idx_base, tile, quarter, n_pool);

        nvx_u32x4 qv[G];
        HalfQuad qraw[(QKV && !SHARE_Q) ? G : 1];
        const uint16_t* qrow = q + (int64_t)b * q_stride;
        const float* cs = nullptr;
        const int qid = wave * 4 + quarter;   // this quarter-wave among the workgroup's sixteen
        if (QKV) {
            cs = fx.cos_sin + clamp_pos(fx.positions[b], fx.max_pos) * 128;
            if (SHARE_Q) {
                if (qid < G) qraw[0] = load_half_quads(qrow + (head * G + qid) * 128, piece);
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) qraw[g] = load_half_quads(qrow + (head * G + g) * 128, piece);
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) qv[g] = *reinterpret_cast<const nvx_u32x4*>(qrow + (head * G + g) * 128 + piece * 8);
        }

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
        if (QKV) {   // first tile peeled: the query is finished between the issue of its rows and their use
            const bool any = tile < n_pool;
            if (any) {
                dec_issue<TPQ>(ta, slots, kb, vb, row_bytes, n_slots_u);
                dec_load_slots<TPQ>(slots, idx_base, tile + STEP, quarter, n_pool);
            }
            if (SHARE_Q) {
                if (qid < G) {
                    const uint4 o = norm_rotate_interleave(qraw[0], half_quad_sumsq(qraw[0]), fx.q_weight, fx.eps, cs, piece);
                    q_shared[qid][piece] = nvx_u32x4{o.x, o.y, o.z, o.w};
                }
                __syncthreads();   // (the two barriers at the end of a unit keep the next unit's writes behind these reads)
#pragma unroll
                for (int g = 0; g < G; ++g) qv[g] = q_shared[g][piece];
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint4 o = norm_rotate_interleave(qraw[g], half_quad_sumsq(qraw[g]), fx.q_weight, fx.eps, cs, piece);
                    qv[g] = nvx_u32x4{o.x, o.y, o.z, o.w};
                }
            }
            if (any) {
                dec_compute<G, TPQ>(ta, qv, m, l, acc, scale_log2);
                tile += STEP;
            }
        }
        while (tile < n_pool) {
            dec_issue<TPQ>(ta, slots, kb, vb, row_bytes, n_slots_u);
            dec_load_slots<TPQ>(slots, idx_base, tile + STEP, quarter, n_pool);
            dec_compute<G, TPQ>(ta, qv, m, l, acc, scale_log2);
            tile += STEP;
        }

        if (QKV && own_new) {   // the new token: made from the raw projection, stored, and attended to from
                                // registers, counted once (wave 0, quarter 0)
            const bool mine = wave == 0 && quarter == 0;
            const nvx_u32x4 k_new = dec_new_key(qrow, hq_total, head, fx, cs, piece);
            const nvx_u32x4 v_new = *reinterpret_cast<const nvx_u32x4*>(qrow + (hq_total + hkv + head) * 128 + piece * 8);
            if (tid < 16) {
                const int64_t at = ((int64_t)slot_new * hkv + head) * 128 + piece * 8;
                *reinterpret_cast<nvx_u32x4*>(fx.k_cache_w + at) = k_new;
                *reinterpret_cast<nvx_u32x4*>(fx.v_cache_w + at) = v_new;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned qw = qv[g][w], kw = k_new[w];
                    d = dot2_h16(qw, kw, d);
                }
                d = quarter_sum(d);
                const float sc = mine ? d * scale_log2 : -INFINITY;
                const float mx = fmaxf(m[g], sc);
                const float mref = mx == -INFINITY ? 0.f : mx;
                const float alpha = __builtin_amdgcn_exp2f(m[g] - mref);
                const float pr = __builtin_amdgcn_exp2f(sc - mref);
                m[g] = mx;
                l[g] = l[g] * alpha + pr;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned vw = mine ? (unsigned)v_new[w] : 0u;
                    acc[g][2 * w] = acc[g][2 * w] * alpha + pr * h16lo_to_f32(vw);
                    acc[g][2 * w + 1] = acc[g][2 * w + 1] * alpha + pr * h16hi_to_f32(vw);
                }
            }
        }

        // merge the four quarter-waves (they hold different tokens, same dims)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            // (through LDS on purpose: with four workgroups per CU the latency is covered and the vector ALU is the
            // scarce unit; the in-ALU exchange of the ring kernel below measured 0.5 us slower here)
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
                *reinterpret_cast<uint4*>(dst) = make_uint4(pack_h16x2(o[0] * inv, o[1] * inv), pack_h16x2(o[2] * inv, o[3] * inv),
                                                            pack_h16x2(o[4] * inv, o[5] * inv), pack_h16x2(o[6] * inv, o[7] * inv));
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

// ---------------------------------------------------------------------------------------------
// Helpers of the LDS-ring variant below.
typedef __attribute__((address_space(3))) const nvx_u32x4 dec_lds_u4;
__device__ __forceinline__ nvx_u32x4 dec_lds_read16(unsigned addr) { return *reinterpret_cast<dec_lds_u4*>((uintptr_t)addr); }
__device__ __forceinline__ unsigned dec_lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
// 64 lanes x 16 bytes from per-lane global addresses to 1 KB of LDS starting at a wave-uniform address, as a streaming
// (non-temporal) read.  Issued from inline assembly: the compiler's own LDS-DMA builtin makes every later LDS read wait
// for the copy.  m0 has no other user in this kernel: the toolchain refuses m0 in a clobber list (reserved register), so tests/test_abi_exports.py disassembles the library and checks that every access to m0 is one of these moves.
__device__ __forceinline__ void dec_dma16(const char* g, unsigned lds_addr) {
    const unsigned uni = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(g), "s"(uni) : "memory");
}
// at most `n` of this wave's vector-memory operations still in flight (operations complete in order)
template <int N>
__device__ __forceinline__ void dec_wait_copies() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------
// Variant for wide query groups (G >= 8): rows in flight held by LDS, products on the matrix cores.
//
// In the kernel above a wave's rows in flight are registers and every (token, query head) pair costs about 24 vector
// instructions on a quarter-wave, so both the bytes a CU keeps in flight and the vector ALU time scale against it as G
// grows: at 8 query heads per kv head it needs 232 VGPRs (two waves per SIMD, 32 KB in flight per CU where the HBM
// latency wants about 50) and four times the arithmetic of G = 2.  Here each wave owns a ring of NS stages of 16 tokens
// (K rows, then V rows: 8 KB) in LDS and fills it with LDS-DMA copies (pool -> LDS, no registers), NS - 1 stages
// ahead of the one it is reducing; the copies are counted (s_waitcnt vmcnt(8 x stages still wanted in flight)),
// never drained.  Slot ids are wave-uniform scalar loads issued one iteration ahead (their own counter: no wait on
// them touches the copies).  With the rows in LDS they can be read in MFMA operand order, and the G query heads
// become the N side of small matrix products: per stage and wave, S^T[16 tokens x 16 head slots] = K . Q^T is four
// v_mfma_f32_16x16x32_bf16 (K rows read from LDS as the A operand, Q in registers as B, head slots >= G zero); the
// accumulator layout of S IS the B operand layout of v_mfma_f32_16x16x16_bf16 (lane: head l16, tokens 4q..4q+3), so P
// (rounded to bf16, as in the prefill kernel) goes back in without leaving the lane, and O^T[128 dims x 16 head slots]
// += V^T . P is eight of those with V^T read by ds_read_b64_tr_b16.  Vector work per stage is the softmax of four
// scores per lane, whatever G is.
// LDS rows are XOR-swizzled so that the operand reads are conflict free: position p of K row r holds piece
// p ^ (r & 15), of V row r piece p ^ ((r & 7) << 1); the copy writes consecutive positions, so the swizzle is applied
// to the global address a lane fetches from.
// Measured on the bench's decode batches (pair with the merge launch, us, this kernel vs the one above): 64 q / 8 kv
// heads 69.5 vs 115.5 at 123 sequences, 38.7 vs 53.3 at 31, 12.1 vs 16.8 at 2; 16 / 2 heads 28.8 vs 41.0, 16.2 vs 22.5,
// 11.8 vs 16.3.  At G = 4 the two are level (32 / 8 heads: 64.5 vs 69.6, 32.7 vs 30.7, 11.0 vs 12.4) and at G = 2 the
// register kernel is faster (its four workgroups per CU hide the per-unit start-up better than the two that 64 KB of
// rings allow): G <= 4 stays there, except G = 4 with one or two kv heads (4 / 1 heads: 12.2 vs 13.7 us on average).
#define DEC_M_TOKENS 16
#define DEC_M_STAGE_BYTES (DEC_M_TOKENS * 512)
#ifndef DEC_M_NS
#define DEC_M_NS 2
#endif
#ifndef DEC_M_WPS
#define DEC_M_WPS 2
#endif
typedef __attribute__((ext_vector_type(4))) short dec_s16x4;
typedef __attribute__((address_space(3))) dec_s16x4 dec_lds_s16x4;
__device__ __forceinline__ dec_s16x4 dec_lds_read_tr(unsigned addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<dec_lds_s16x4*>((uintptr_t)addr));
}
typedef __attribute__((address_space(3))) nvx_u32x4 dec_lds_w4;
__device__ __forceinline__ void dec_lds_write16(unsigned addr, nvx_u32x4 v) { *reinterpret_cast<dec_lds_w4*>((uintptr_t)addr) = v; }

#if defined(DEC_STAMPS) && !defined(NVX_F16)
// Build-time instrumentation (tools/build_variant.sh ... "-DDEC_STAMPS"): shader-clock stamps of the four waves of
// workgroups 0..15 at four points of their first 16 stages, read back with nvx_debug_dec_stamps.  Never in the shipped library.
__device__ unsigned long long dec_stamps[16 * 4 * 16 * 4];
extern "C" int nvx_debug_dec_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dec_stamps), sizeof(dec_stamps)); }
#define DEC_STAMP(k) do { if (blockIdx.x < 16 && j < 16 && lane == 0) \
    dec_stamps[((blockIdx.x * 4 + wave) * 16 + j) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define DEC_STAMP(k) do { } while (0)
#endif

template <int G, bool QKV, int NS>
__global__ __launch_bounds__(256, DEC_M_WPS) void paged_decode_mfma_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ k_cache,
    const uint16_t* __restrict__ v_cache, int64_t n_slots, const int32_t* __restrict__ kv_indices,
    const int32_t* __restrict__ plan, float* __restrict__ partial, uint16_t* __restrict__ out, int hkv,
    float scale_log2, DecQkv fx) {
    static_assert(G >= 1 && G <= 16, "query heads per kv head fill the 16 columns of the products");
    static_assert(NS >= 2 && NS <= 3, "ring depth");
    extern __shared__ __attribute__((aligned(1024))) unsigned char dec_smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int quarter = lane >> 4, l16 = lane & 15;
    const int hq_total = hkv * G;
    const unsigned row_bytes = (unsigned)hkv * 256u;
    const unsigned n_slots_u = n_slots > 0xffffffffll ? 0xffffffffu : (unsigned)n_slots;
    const int32_t* recs = plan + 8;
    const unsigned ring = dec_lds_addr(dec_smem) + (unsigned)wave * (NS * DEC_M_STAGE_BYTES);
    float (*mine_out)[16][10] = reinterpret_cast<float (*)[16][10]>(dec_smem + (size_t)wave * (NS * DEC_M_STAGE_BYTES));
    static_assert(G * 16 * 10 * 4 <= NS * DEC_M_STAGE_BYTES, "merge rows fit the wave's ring");
    // per-lane parts of the operand addresses inside a stage
    const unsigned k_lane = (unsigned)l16 * 256u;                                   // K row l16; piece XOR applied per k-step
    const int v_row = 4 * quarter + (l16 >> 2);                                     // V^T block: this lane addresses token 4q + (i >> 2)
    const unsigned v_lane = 2048u * 2u + (unsigned)v_row * 256u + 8u * (l16 & 1);   // V rows start 4 KB into the stage
    const int v_pc = (l16 & 3) >> 1, v_sw = (v_row & 7) << 1;

    int unit = blockIdx.x;
    const int n_units = plan[0] * hkv;
    while (unit < n_units) {
        const int item = unit / hkv, head = unit % hkv;
        const int32_t* rec = recs + (int64_t)DEC_REC_INTS * item;
        const int b = __builtin_amdgcn_readfirstlane(rec[0]), n = __builtin_amdgcn_readfirstlane(rec[1]);
        const int nch = __builtin_amdgcn_readfirstlane(rec[2]);
        const int64_t abs = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rec[5]) << 32) |
                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(rec[4]));
        const int32_t* idx_base = kv_indices + abs;
        const char* kb = reinterpret_cast<const char*>(k_cache) + head * 256;
        const char* vb = reinterpret_cast<const char*>(v_cache) + head * 256;
        const int slot_new = QKV ? __builtin_amdgcn_readfirstlane(rec[7]) : -1;
        const bool own_new = QKV && slot_new >= 0 && (long long)slot_new < (long long)n_slots && n > 0;
        const int n_pool = own_new ? n - 1 : n;
        const int n_stages = (n_pool + DEC_M_TOKENS - 1) / DEC_M_TOKENS;
        const int my_stages = n_stages > wave ? (n_stages - wave + 3) >> 2 : 0;   // stages wave, wave + 4, ...

        // One stage: the ids of its 16 tokens travel first (scalar loads, one iteration before the rows they name, so
        // that their round trip is not in front of the copies), then eight copies of 1 KB (K tokens 0-3, V 0-3, K 4-7,
        // ...: quarter-wave q moves the row of token 4a + q, lane p of it the piece that belongs at position p).
        struct StageIds { unsigned id[DEC_M_TOKENS]; unsigned ok; };
        auto load_ids = [&](int stage) -> StageIds {
            StageIds r;
            r.ok = 0;
#pragma unroll
            for (int t = 0; t < DEC_M_TOKENS; ++t) {
                const int at = stage * DEC_M_TOKENS + t;
                const unsigned v = (unsigned)idx_base[at < n_pool ? at : n_pool - 1];
                const bool good = at < n_pool && v < n_slots_u;
                r.ok |= good ? (1u << t) : 0u;
                r.id[t] = good ? v : 0u;
            }
            return r;
        };
        auto issue = [&](const StageIds& ids, unsigned dst) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const unsigned slot = quarter == 0 ? ids.id[4 * a] : quarter == 1 ? ids.id[4 * a + 1] : quarter == 2 ? ids.id[4 * a + 2] : ids.id[4 * a + 3];
                const unsigned long long off = (unsigned long long)slot * row_bytes;
                const int t = 4 * a + quarter;
                dec_dma16(kb + off + ((l16 ^ (t & 15)) << 4), dst + a * 1024);
                dec_dma16(vb + off + ((l16 ^ ((t & 7) << 1)) << 4), dst + 4096 + a * 1024);
            }
        };

        unsigned okq[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) okq[k] = 0;
#pragma unroll
        for (int k = 0; k < NS - 1; ++k)
            if (k < my_stages) { const StageIds ids = load_ids(wave + 4 * k); okq[k] = ids.ok; issue(ids, ring + k * DEC_M_STAGE_BYTES); }
        StageIds ahead_ids;
        ahead_ids.ok = 0;
#pragma unroll
        for (int t = 0; t < DEC_M_TOKENS; ++t) ahead_ids.id[t] = 0;
        if (NS - 1 < my_stages) ahead_ids = load_ids(wave + 4 * (NS - 1));

        // Q^T as the B operand of the score products: lane (q, l16) holds dims 32 ks + 8 q .. +7 of head l16 (zero
        // for l16 >= G).  In the QKV variant the heads are normalised and rotated first, 16 lanes per head as the
        // stand-alone kernel does it, and turned around through the one stage of the ring that is not in flight yet.
        nvx_u32x4 qb[4];
        const uint16_t* qrow = q + (int64_t)b * q_stride;
        const float* cs = nullptr;
        if (QKV) {
            cs = fx.cos_sin + clamp_pos(fx.positions[b], fx.max_pos) * 128;
            const unsigned qarea = ring + (NS - 1) * DEC_M_STAGE_BYTES;
#pragma unroll
            for (int g0 = 0; g0 < G; g0 += 4) {
                const int g = g0 + quarter;
                if (g < G) {
                    const HalfQuad qraw = load_half_quads(qrow + (head * G + g) * 128, l16);
                    const uint4 o = norm_rotate_interleave(qraw, half_quad_sumsq(qraw), fx.q_weight, fx.eps, cs, l16);
                    dec_lds_write16(qarea + g * 256 + l16 * 16, nvx_u32x4{o.x, o.y, o.z, o.w});
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qb[ks] = dec_lds_read16(qarea + (l16 < G ? l16 : 0) * 256 + (4 * ks + quarter) * 16);
                if (l16 >= G) qb[ks] = nvx_u32x4{0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qb[ks] = *reinterpret_cast<const nvx_u32x4*>(qrow + (head * G + (l16 < G ? l16 : 0)) * 128 + 32 * ks + 8 * quarter);
                if (l16 >= G) qb[ks] = nvx_u32x4{0u, 0u, 0u, 0u};
            }
        }
        // Q is complete (and its turn through the ring over) before the loop starts counting copies: a wait the
        // compiler places at the first use inside the loop would drain the ring on every iteration.
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qb[ks]));

        nvx_f32x4 o_acc[8];
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) o_acc[blk] = nvx_f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;   // of head l16; m_run is the same in the four quarter-waves

        // one stage of 16 tokens at LDS address `base` (K rows, then V rows), `okm`: which tokens are real keys
        auto reduce_stage = [&](unsigned base, unsigned okm) {
            nvx_f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const nvx_u32x4 a = dec_lds_read16(base + k_lane + (((4 * ks + quarter) ^ l16) << 4));
                sacc = mfma_16x16x32_h16(a, qb[ks], sacc);
            }
            const unsigned mine4 = (okm >> (4 * quarter)) & 15u;   // tokens 4q .. 4q+3
            float sv[4];
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sv[r] = (mine4 >> r) & 1u ? sacc[r] * scale_log2 : -INFINITY;
                mx = fmaxf(mx, sv[r]);
            }
            mx = fmaxf(mx, xor16(mx));
            mx = fmaxf(mx, xor32(mx));
            const float m_new = fmaxf(m_run, mx);
            const float mref = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - mref);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) o_acc[blk] *= alpha;
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sv[r] - mref); l_run += p[r]; }
            const uint32_t p01 = pack_h16x2(p[0], p[1]), p23 = pack_h16x2(p[2], p[3]);
            const dec_s16x4 pb = __builtin_bit_cast(dec_s16x4, make_uint2(p01, p23));
            // value rows of tokens that are not keys may hold anything (0 x NaN = NaN): cleared in the operand
            const uint32_t vm01 = ((mine4 & 1u) ? 0xffffu : 0u) | ((mine4 & 2u) ? 0xffff0000u : 0u);
            const uint32_t vm23 = ((mine4 & 4u) ? 0xffffu : 0u) | ((mine4 & 8u) ? 0xffff0000u : 0u);
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
                const dec_s16x4 vt = dec_lds_read_tr(base + v_lane + (((2 * blk + v_pc) ^ v_sw) << 4));
                uint2 vw = __builtin_bit_cast(uint2, vt);
                vw.x &= vm01; vw.y &= vm23;
                o_acc[blk] = mfma_16x16x16_h16(__builtin_bit_cast(dec_s16x4, vw), pb, o_acc[blk]);
            }
        };

        int slot_cur = 0, slot_next = (NS - 1) % NS;
        for (int j = 0; j < my_stages; ++j) {
            DEC_STAMP(0);
            const int ahead = j + NS - 1;
            if (ahead < my_stages) { okq[NS - 1] = ahead_ids.ok; issue(ahead_ids, ring + slot_next * DEC_M_STAGE_BYTES); }
            if (ahead + 1 < my_stages) ahead_ids = load_ids(wave + 4 * (ahead + 1));   // lands under the wait below
            DEC_STAMP(1);
            const int later = my_stages - 1 - j < NS - 1 ? my_stages - 1 - j : NS - 1;
            if (later >= 2) dec_wait_copies<16>();
            else if (later == 1) dec_wait_copies<8>();
            else dec_wait_copies<0>();
            DEC_STAMP(2);
            reduce_stage(ring + slot_cur * DEC_M_STAGE_BYTES, okq[0]);
            DEC_STAMP(3);
#pragma unroll
            for (int k = 0; k < NS - 1; ++k) okq[k] = okq[k + 1];
            slot_cur = slot_cur + 1 == NS ? 0 : slot_cur + 1;
            slot_next = slot_next + 1 == NS ? 0 : slot_next + 1;
        }
        dec_wait_copies<0>();

        if (QKV && own_new) {   // the new token: made from the raw projection, stored in the pool, and reduced by wave 0
                                // as a stage of its own with one real key (row 0 of the wave's first stage)
            const nvx_u32x4 k_new = dec_new_key(qrow, hq_total, head, fx, cs, l16);
            const nvx_u32x4 v_new = *reinterpret_cast<const nvx_u32x4*>(qrow + (hq_total + hkv + head) * 128 + l16 * 8);
            if (tid < 16) {
                const int64_t at = ((int64_t)slot_new * hkv + head) * 128 + l16 * 8;
                *reinterpret_cast<nvx_u32x4*>(fx.k_cache_w + at) = k_new;
                *reinterpret_cast<nvx_u32x4*>(fx.v_cache_w + at) = v_new;
            }
            if (wave == 0) {
                if (quarter == 0) {   // row 0: no swizzle
                    dec_lds_write16(ring + l16 * 16, k_new);
                    dec_lds_write16(ring + 4096 + l16 * 16, v_new);
                }
                reduce_stage(ring, 1u);
            }
        }

        // the four quarter-waves saw different tokens: their row sums add up (the maxima already agree)
        l_run += xor16(l_run);
        l_run += xor32(l_run);
        // hand the wave's state to the workgroup through the (dead) ring: [head][16-byte piece] -> m, l, acc[8];
        // this lane holds dims 16 blk + 4 q .. +3 of head l16: half of piece 2 blk + (q >> 1)
        if (l16 < G) {
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
                float* dst = mine_out[l16][2 * blk + (quarter >> 1)];
                if ((quarter & 1) == 0) { dst[0] = m_run; dst[1] = l_run; }
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[2 + 4 * (quarter & 1) + r] = o_acc[blk][r];
            }
        }
        __syncthreads();
        if (tid < 16 * G) {
            const int g = tid >> 4, pc = tid & 15;
            const float* src[4];
#pragma unroll
            for (int w = 0; w < 4; ++w)
                src[w] = reinterpret_cast<const float*>(dec_smem + (size_t)w * (NS * DEC_M_STAGE_BYTES)) + (g * 16 + pc) * 10;
            float mall = -INFINITY;
#pragma unroll
            for (int w = 0; w < 4; ++w) mall = fmaxf(mall, src[w][0]);
            const float mr = mall == -INFINITY ? 0.f : mall;
            float lsum = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float sc = __builtin_amdgcn_exp2f(src[w][0] - mr);
                lsum += src[w][1] * sc;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += src[w][2 + j] * sc;
            }
            const int hq = head * G + g;
            if (nch == 1) {
                const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
                uint16_t* dst = out + ((int64_t)b * hq_total + hq) * 128 + pc * 8;
                *reinterpret_cast<uint4*>(dst) = make_uint4(pack_h16x2(o[0] * inv, o[1] * inv), pack_h16x2(o[2] * inv, o[3] * inv),
                                                            pack_h16x2(o[4] * inv, o[5] * inv), pack_h16x2(o[6] * inv, o[7] * inv));
            } else {
                float* dst = partial + ((int64_t)item * hq_total + hq) * DEC_PART_STRIDE;
                *reinterpret_cast<float4*>(dst + pc * 8) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(dst + pc * 8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
                if (pc == 0) { dst[128] = mall; dst[129] = lsum; }
            }
        }
        __syncthreads();   // the rings are free again
        unit += gridDim.x;
    }
}

// Combine the partials of sequences that were split over several chunks: one 128-thread block per
// (sequence, query head), thread d owns output element d.  Blocks of unsplit sequences exit at once.
// The kernel is a chain of dependent loads and nothing else, so the chain is kept at two hops: the
// sequence's (first item, chunk count) pair in one load at an address known from the launch arguments,
// then the chunks' partials, all independent, folded in one pass.
__global__ __launch_bounds__(128) void decode_merge_kernel(const int32_t* __restrict__ plan, const float* __restrict__ partial,
                                                           uint16_t* __restrict__ out, int hq_total, int batch, int max_items) {
    // the grid is exactly batch x hq_total blocks
    const int b = blockIdx.x / hq_total, hq = blockIdx.x % hq_total;
    // (first item, chunk count) in ONE 8-byte load: as two fields the compiler fetched the count, tested
    // it, and only then fetched the other
    const unsigned long long info =
        *reinterpret_cast<const unsigned long long*>(plan + 8 + (int64_t)DEC_REC_INTS * max_items + 2 * b);
    const int first = (int)(unsigned)(info & 0xffffffffull), nch = (int)(unsigned)(info >> 32);
    if (nch <= 1) return;
    const int d = threadIdx.x;
    const float* base = partial + ((int64_t)first * hq_total + hq) * DEC_PART_STRIDE;
    const int64_t step = (int64_t)hq_total * DEC_PART_STRIDE;
    float mall = -INFINITY, lsum = 0.f, o = 0.f;
    // eight chunks per round trip (sequences rarely have more at serving batch sizes): chunks past the end
    // re-read the last one and are not folded in
    for (int i0 = 0; i0 < nch; i0 += 8) {
        float mi[8], li[8], ai[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = i0 + j < nch;
            const float* p = base + (live ? i0 + j : nch - 1) * step;
            // an empty chunk (m = -inf, l = 0, acc = 0) folds in as a no-op, so the fold below has no branches
            // (with branches the compiler moved every load into its branch: one round trip per chunk)
            const float pm = p[128], pl = p[129], pa = p[d];
            mi[j] = live ? pm : -INFINITY; li[j] = live ? pl : 0.f; ai[j] = live ? pa : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float mnew = fmaxf(mall, mi[j]);
            const float mref = mnew == -INFINITY ? 0.f : mnew;
            const float so = __builtin_amdgcn_exp2f(mall - mref), sn = __builtin_amdgcn_exp2f(mi[j] - mref);
            lsum = lsum * so + li[j] * sn;
            o = o * so + ai[j] * sn;
            mall = mnew;
        }
    }
    out[((int64_t)b * hq_total + hq) * 128 + d] = f32_to_h16bits(lsum > 0.f ? o / lsum : 0.f);
}

static int dec_launch(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache, int64_t n_slots,
                      const int32_t* kv_indices, const int32_t* plan_ws, float* partial_ws, void* out, int64_t batch, int n_q_heads,
                      int n_kv_heads, float sm_scale, const DecQkv* fx, int min_q_width, const char* who, nvx_stream_t stream) {
    NVX_REQUIRE(batch > 0 && n_q_heads > 0 && n_kv_heads > 0 && n_slots > 0, NVX_EINVAL, "%s: bad size", who);
    NVX_REQUIRE(q && k_cache && v_cache && kv_indices && plan_ws && partial_ws && out, NVX_EINVAL, "%s: null pointer", who);
    NVX_REQUIRE(n_q_heads % n_kv_heads == 0, NVX_EINVAL, "%s: n_q_heads must be a multiple of n_kv_heads", who);
    NVX_REQUIRE(q_row_stride % 8 == 0 && q_row_stride >= (int64_t)min_q_width, NVX_EINVAL, "%s: bad row stride", who);
    NVX_REQUIRE(nvx_aligned16(q) && nvx_aligned16(k_cache) && nvx_aligned16(v_cache) && nvx_aligned16(out) &&
                nvx_aligned16(partial_ws) && nvx_aligned16(plan_ws), NVX_EINVAL, "%s: pointers must be 16-byte aligned", who);
    const int g = n_q_heads / n_kv_heads;
    const float scale_log2 = sm_scale * DEC_LOG2E;
    hipStream_t st = (hipStream_t)stream;
    const int64_t max_units = dec_max_items(batch, n_kv_heads) * n_kv_heads;
    const unsigned grid = (unsigned)(max_units < 2048 ? max_units : 2048);
    const DecQkv none = {};
#define NVX_DEC(GG, TT, TQ)                                                                                                      \
    do {                                                                                                                         \
        if (fx) hipLaunchKernelGGL((paged_decode_kernel<GG, TQ, true>), dim3(grid), dim3(256), 0, st, (const uint16_t*)q, q_row_stride, \
                                   (const uint16_t*)k_cache, (const uint16_t*)v_cache, n_slots, kv_indices, plan_ws, partial_ws, \
                                   (uint16_t*)out, n_kv_heads, scale_log2, *fx);                                                 \
        else hipLaunchKernelGGL((paged_decode_kernel<GG, TT, false>), dim3(grid), dim3(256), 0, st, (const uint16_t*)q, q_row_stride,  \
                                (const uint16_t*)k_cache, (const uint16_t*)v_cache, n_slots, kv_indices, plan_ws, partial_ws,    \
                                (uint16_t*)out, n_kv_heads, scale_log2, none);                                                   \
    } while (0)
    // tokens per quarter-wave per tile: plain variant / variant fed with the raw projection (which keeps
    // more values live across the loop, so it takes a smaller tile to stay within the same registers)
    static int ring_mode = -1;   // NVX_DEC_RING: 0 = rows in registers for every G, 1 = LDS ring + MFMA for every G, unset = by G
    if (ring_mode < 0) { const char* e = getenv("NVX_DEC_RING"); ring_mode = e ? (atoi(e) ? 1 : 0) : 2; }
    // by G: wide groups, and G = 4 where a launch moves few bytes (one or two kv heads per rank: the kernel's shorter
    // vector work per unit is what counts there; at 8 kv heads the two kernels are level)
    if (ring_mode == 1 || (ring_mode == 2 && (g >= 8 || (g >= 4 && n_kv_heads <= 2)))) {
        constexpr int NS = DEC_M_NS;
        const size_t smem = (size_t)4 * NS * DEC_M_STAGE_BYTES;
#define NVX_DEC_MFMA_K(GG)                                                                                                        \
        do {                                                                                                                      \
            if (fx) hipLaunchKernelGGL((paged_decode_mfma_kernel<GG, true, NS>), dim3(grid), dim3(256), smem, st, (const uint16_t*)q, \
                                       q_row_stride, (const uint16_t*)k_cache, (const uint16_t*)v_cache, n_slots, kv_indices, plan_ws, \
                                       partial_ws, (uint16_t*)out, n_kv_heads, scale_log2, *fx);                                  \
            else hipLaunchKernelGGL((paged_decode_mfma_kernel<GG, false, NS>), dim3(grid), dim3(256), smem, st, (const uint16_t*)q,   \
                                    q_row_stride, (const uint16_t*)k_cache, (const uint16_t*)v_cache, n_slots, kv_indices, plan_ws, \
                                    partial_ws, (uint16_t*)out, n_kv_heads, scale_log2, none);                                    \
        } while (0)
        switch (g) {
            case 1: NVX_DEC_MFMA_K(1); break;
            case 2: NVX_DEC_MFMA_K(2); break;
            case 4: NVX_DEC_MFMA_K(4); break;
            case 8: NVX_DEC_MFMA_K(8); break;
            default: return nvx_fail(NVX_EUNSUPPORTED, "%s: query/kv head ratio %d not in {1,2,4,8}", who, g);
        }
#undef NVX_DEC_MFMA_K
    } else
    switch (g) {
        case 1: NVX_DEC(1, 4, 4); break;
        case 2: NVX_DEC(2, 4, DEC_TQ_G2); break;   // plain: 112 VGPRs, four workgroups per CU; measured best of TPQ 2/4/8/16
        case 4: NVX_DEC(4, 4, DEC_TQ_G4); break;
        case 8: NVX_DEC(8, 2, 2); break;
        default: return nvx_fail(NVX_EUNSUPPORTED, "%s: query/kv head ratio %d not in {1,2,4,8}", who, g);
    }
#undef NVX_DEC
    int rc = nvx_check_launch(who);
    if (rc) return rc;
    hipLaunchKernelGGL(decode_merge_kernel, dim3((unsigned)(batch * n_q_heads)), dim3(128), 0, st, plan_ws, partial_ws,
                       (uint16_t*)out, n_q_heads, (int)batch, (int)dec_max_items(batch, n_kv_heads));
    return nvx_check_launch(who);
}

extern "C" int NVX_ENTRY(nvx_paged_decode)(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                                int64_t n_slots, const int32_t* kv_indices, const int32_t* plan_ws, float* partial_ws,
                                void* out, int64_t batch, int n_q_heads, int n_kv_heads, float sm_scale,
                                nvx_stream_t stream) {
    return dec_launch(q, q_row_stride, k_cache, v_cache, n_slots, kv_indices, plan_ws, partial_ws, out, batch, n_q_heads,
                      n_kv_heads, sm_scale, nullptr, n_q_heads * 128, "nvx_paged_decode", stream);
}

#ifndef NVX_F16   // the entry point that allocates nothing exists for bfloat16 only (one built-in array per library)
// ---------------------------------------------------------------------------------------------
// Entry point for callers that cannot allocate: plan, attention and merge out of one file-scope device
// array instead of caller-provided workspaces.  Each GPU the library runs on holds its own copy of the
// array (at its own address) until the process ends; every call on a GPU uses that GPU's copy, so calls that
// overlap on one GPU race.  Nothing in it has to start at zero: the plan kernel writes the header, every record
// and every sequence's (first item, count) pair before the attention kernel reads them, and the merge reads
// only partials the attention kernel of the same call wrote.
#define NVX_NOALLOC_MAX_BATCH 256
#define NVX_NOALLOC_PLAN_INTS 20480                  // >= nvx_decode_plan_ws_ints(256, 1) = 19016
#define NVX_NOALLOC_PARTIAL_FLOATS (3 * 1024 * 1024)   // 12 MiB: e.g. 32 q / 8 kv heads, 16 / 2 or 8 / 1 at batch 256
#define NVX_NOALLOC_BYTES (NVX_NOALLOC_PLAN_INTS * 4 + NVX_NOALLOC_PARTIAL_FLOATS * 4)
static __device__ __align__(256) unsigned char scratch[NVX_NOALLOC_BYTES];

#define NVX_MAX_DEVICES 64
static std::mutex noalloc_mutex;
static unsigned char* noalloc_ptr[NVX_MAX_DEVICES];   // per device: the address of ITS copy of `scratch`

// Address of the current device's copy of the array (looked up once per device, with that device current).
static int noalloc_scratch(unsigned char** ptr, const char* who) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return nvx_fail(NVX_ELAUNCH, "%s: hipGetDevice: %s", who, hipGetErrorString(e));
    NVX_REQUIRE(dev >= 0 && dev < NVX_MAX_DEVICES, NVX_EUNSUPPORTED, "%s: device ordinal %d outside [0, %d)", who, dev, NVX_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(noalloc_mutex);
    if (!noalloc_ptr[dev]) {
        // The first call on a device may come from a thread whose stream is being captured into a graph.  The look-up
        // is no stream operation, but it may load the code object, which a capture in the global or thread-local
        // mode treats as an error: this thread is switched to the relaxed mode for the look-up alone (what an
        // allocator that must run under capture does), and back.
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess || !p) return nvx_fail(NVX_ELAUNCH, "%s: hipGetSymbolAddress(scratch): %s", who, hipGetErrorString(e));
        noalloc_ptr[dev] = static_cast<unsigned char*>(p);
    }
    *ptr = noalloc_ptr[dev];
    return NVX_OK;
}

extern "C" int64_t nvx_paged_decode_noalloc_max_batch(void) { return NVX_NOALLOC_MAX_BATCH; }

extern "C" int nvx_paged_decode_noalloc(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                                        int64_t n_slots, const int32_t* kv_indices, const int32_t* kv_len,
                                        const int64_t* kv_start, void* out, int64_t batch, int n_q_heads, int n_kv_heads,
                                        float sm_scale, nvx_stream_t stream) {
    const char* who = "nvx_paged_decode_noalloc";
    NVX_REQUIRE(batch > 0 && n_q_heads > 0 && n_kv_heads > 0, NVX_EINVAL, "%s: bad size", who);
    NVX_REQUIRE(kv_len && kv_start, NVX_EINVAL, "%s: null pointer", who);
    NVX_REQUIRE(batch <= NVX_NOALLOC_MAX_BATCH, NVX_ENOSPACE, "%s: batch %lld exceeds the built-in scratch (at most %d sequences)", who,
                (long long)batch, NVX_NOALLOC_MAX_BATCH);
    const int64_t plan_ints = nvx_decode_plan_ws_ints(batch, n_kv_heads);
    const int64_t part_floats = nvx_decode_partial_ws_floats(batch, n_q_heads, n_kv_heads);
    NVX_REQUIRE(plan_ints <= NVX_NOALLOC_PLAN_INTS && part_floats <= NVX_NOALLOC_PARTIAL_FLOATS, NVX_ENOSPACE,
                "%s: batch %lld with %d query / %d kv heads needs %lld plan ints and %lld partial floats; the built-in scratch "
                "holds %d and %d", who, (long long)batch, n_q_heads, n_kv_heads, (long long)plan_ints, (long long)part_floats,
                NVX_NOALLOC_PLAN_INTS, NVX_NOALLOC_PARTIAL_FLOATS);
    unsigned char* ptr = nullptr;
    int rc = noalloc_scratch(&ptr, who);
    if (rc) return rc;
    int32_t* plan_ws = reinterpret_cast<int32_t*>(ptr);
    float* partial_ws = reinterpret_cast<float*>(ptr + (size_t)NVX_NOALLOC_PLAN_INTS * 4);
    rc = nvx_decode_plan(kv_len, kv_start, nullptr, batch, n_kv_heads, plan_ws, NVX_NOALLOC_PLAN_INTS, stream);
    if (rc) return rc;
    return dec_launch(q, q_row_stride, k_cache, v_cache, n_slots, kv_indices, plan_ws, partial_ws, out, batch, n_q_heads,
                      n_kv_heads, sm_scale, nullptr, n_q_heads * 128, who, stream);
}

#endif

extern "C" int NVX_ENTRY(nvx_paged_decode_qkv)(const void* qkv, int64_t qkv_row_stride, const int64_t* positions, const float* cos_sin,
                                    int64_t max_pos, const void* q_norm_weight, const void* k_norm_weight, float eps,
                                    void* k_cache, void* v_cache, int64_t n_slots, const int32_t* kv_indices,
                                    const int32_t* plan_ws, float* partial_ws, void* out, int64_t batch, int n_q_heads,
                                    int n_kv_heads, float sm_scale, nvx_stream_t stream) {
    NVX_REQUIRE(positions && cos_sin && q_norm_weight && k_norm_weight && max_pos > 0, NVX_EINVAL, "nvx_paged_decode_qkv: null pointer");
    NVX_REQUIRE(nvx_aligned16(cos_sin) && nvx_aligned16(q_norm_weight) && nvx_aligned16(k_norm_weight), NVX_EINVAL,
                "nvx_paged_decode_qkv: pointers must be 16-byte aligned");
    const DecQkv fx = {positions, cos_sin, max_pos, (const uint16_t*)q_norm_weight, (const uint16_t*)k_norm_weight, eps,
                       (uint16_t*)k_cache, (uint16_t*)v_cache};
    return dec_launch(qkv, qkv_row_stride, k_cache, v_cache, n_slots, kv_indices, plan_ws, partial_ws, out, batch, n_q_heads,
                      n_kv_heads, sm_scale, &fx, (n_q_heads + 2 * n_kv_heads) * 128, "nvx_paged_decode_qkv", stream);
}

}  // namespace NVX_NS

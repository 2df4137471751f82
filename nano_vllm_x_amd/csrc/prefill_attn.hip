// Paged prefill attention for gfx950: causal, variable length, page size 1, MFMA for QK^T and PV.
//
// Work unit = (sequence, block of query positions, kv head).  A workgroup holds, per wave, 32 query
// rows of one query head in registers (B operand of K.Q^T); the G query heads that share a kv head
// sit in different waves of the same workgroup, so every gathered K/V tile is used G times.
//
// Per 64-key tile:
//   gather   K and V rows (256 B each, located through the block table) global -> VGPR -> LDS,
//            issued one tile ahead so the HBM latency hides under the MFMAs of the current tile;
//            K rows are XOR-swizzled by (key & 15) for conflict-free ds_read_b128, V rows by
//            (key & 3) << 2 for conflict-free ds_read_b64_tr_b16;
//   S^T    = K . Q^T      v_mfma_f32_32x32x16_bf16, keys on the MFMA rows, queries on the lanes,
//            so a query's scores live in ONE lane pair: softmax needs no LDS and one lane swap;
//   P^T    = exp2(S^T - m) in fp32, rounded to bf16 in registers: the accumulator tile is already
//            the B operand of the next MFMA (k order permuted, matched by the V fragment reads);
//   O^T   += V^T . P^T    V^T fragments by the LDS transpose read; O^T keeps queries on the lanes,
//            so the online-softmax rescale is a per-lane scalar multiply.
#include "nvx_dev.h"

#define PF_KVBLK 64
#define PF_LOG2E 1.4426950408889634f

typedef __attribute__((ext_vector_type(4))) short pf_s16x4;
typedef __attribute__((address_space(3))) pf_s16x4 pf_lds_s16x4;

// plan_ws (int32): three prefix tables, for 32, 64 and 128 query positions per workgroup:
//   table t at ws + t*(batch+2): [0] = total blocks, [1 + b] = first block of sequence b, [1 + batch] = total
extern "C" int64_t nvx_prefill_plan_ws_ints(int64_t batch) { return 3 * (batch + 2); }

__global__ __launch_bounds__(1024) void prefill_plan_kernel(const int32_t* __restrict__ qo_indptr, int batch,
                                                            int32_t* __restrict__ ws) {
    __shared__ int scan[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    for (int t = 0; t < 3; ++t) {
        const int qpos = 32 << t;
        int32_t* tab = ws + t * (batch + 2);
        if (tid == 0) carry = 0;
        __syncthreads();
        for (int base = 0; base < batch; base += 1024) {
            const int b = base + tid;
            int nb = 0;
            if (b < batch) { const int lq = qo_indptr[b + 1] - qo_indptr[b]; nb = lq > 0 ? (lq + qpos - 1) / qpos : 0; }
            scan[tid] = nb;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {
                const int v = tid >= off ? scan[tid - off] : 0;
                __syncthreads();
                scan[tid] += v;
                __syncthreads();
            }
            if (b < batch) tab[1 + b] = carry + scan[tid] - nb;
            __syncthreads();
            if (tid == 1023) carry += scan[1023];
            __syncthreads();
        }
        if (tid == 0) { tab[0] = carry; tab[1 + batch] = carry; }
        __syncthreads();
    }
}

extern "C" int nvx_prefill_plan(const int32_t* qo_indptr, int64_t batch, int32_t* plan_ws, int64_t plan_ws_ints,
                                nvx_stream_t stream) {
    NVX_REQUIRE(batch > 0, NVX_EINVAL, "nvx_prefill_plan: bad size");
    NVX_REQUIRE(qo_indptr && plan_ws, NVX_EINVAL, "nvx_prefill_plan: null pointer");
    NVX_REQUIRE(plan_ws_ints >= nvx_prefill_plan_ws_ints(batch), NVX_ENOSPACE, "nvx_prefill_plan: workspace too small");
    hipLaunchKernelGGL(prefill_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, qo_indptr, (int)batch, plan_ws);
    return nvx_check_launch("nvx_prefill_plan");
}

// Staging registers of one 64-key tile (K and V rows of 256 B, 16 lanes per row).  Named members, not
// arrays: arrays that live across the tile loop were left in scratch memory by the compiler.
struct PfStage { nvx_u32x4 k0, k1, k2, k3, v0, v1, v2, v3; };

template <int KEYS_PER_PASS>
__device__ __forceinline__ void pf_gather_one(nvx_u32x4& kd, nvx_u32x4& vd, int p, int tile, int krow0, int lkv,
                                              const int32_t* __restrict__ seq_slots, int64_t n_slots,
                                              const uint16_t* __restrict__ kbase, const uint16_t* __restrict__ vbase,
                                              int64_t row_elems) {
    int key = tile * PF_KVBLK + krow0 + p * KEYS_PER_PASS;
    key = key < lkv ? key : lkv - 1;   // rows past the end are masked later
    int64_t slot = seq_slots[key];
    slot = (slot < 0 || slot >= n_slots) ? 0 : slot;
    kd = *reinterpret_cast<const nvx_u32x4*>(kbase + slot * row_elems);
    vd = *reinterpret_cast<const nvx_u32x4*>(vbase + slot * row_elems);
}
template <int PASSES, int KEYS_PER_PASS>
__device__ __forceinline__ void pf_gather(PfStage& st, int tile, int krow0, int lkv, const int32_t* __restrict__ seq_slots,
                                          int64_t n_slots, const uint16_t* __restrict__ kbase,
                                          const uint16_t* __restrict__ vbase, int64_t row_elems) {
    pf_gather_one<KEYS_PER_PASS>(st.k0, st.v0, 0, tile, krow0, lkv, seq_slots, n_slots, kbase, vbase, row_elems);
    pf_gather_one<KEYS_PER_PASS>(st.k1, st.v1, 1, tile, krow0, lkv, seq_slots, n_slots, kbase, vbase, row_elems);
    if constexpr (PASSES == 4) {
        pf_gather_one<KEYS_PER_PASS>(st.k2, st.v2, 2, tile, krow0, lkv, seq_slots, n_slots, kbase, vbase, row_elems);
        pf_gather_one<KEYS_PER_PASS>(st.k3, st.v3, 3, tile, krow0, lkv, seq_slots, n_slots, kbase, vbase, row_elems);
    }
}
__device__ __forceinline__ void pf_write_one(const nvx_u32x4& kd, const nvx_u32x4& vd, int key, int piece, unsigned char* k_lds,
                                             unsigned char* v_lds) {
    *reinterpret_cast<nvx_u32x4*>(k_lds + key * 256 + ((piece ^ (key & 15)) << 4)) = kd;
    *reinterpret_cast<nvx_u32x4*>(v_lds + key * 256 + ((piece ^ ((key & 3) << 2)) << 4)) = vd;
}
template <int PASSES, int KEYS_PER_PASS>
__device__ __forceinline__ void pf_stage_write(const PfStage& st, unsigned char* k_lds, unsigned char* v_lds, int krow0, int piece) {
    pf_write_one(st.k0, st.v0, krow0, piece, k_lds, v_lds);
    pf_write_one(st.k1, st.v1, krow0 + KEYS_PER_PASS, piece, k_lds, v_lds);
    if constexpr (PASSES == 4) {
        pf_write_one(st.k2, st.v2, krow0 + 2 * KEYS_PER_PASS, piece, k_lds, v_lds);
        pf_write_one(st.k3, st.v3, krow0 + 3 * KEYS_PER_PASS, piece, k_lds, v_lds);
    }
}

// ---------------------------------------------------------------------------------------------
template <int G, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void paged_prefill_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ k_cache,
    const uint16_t* __restrict__ v_cache, int64_t n_slots, const int32_t* __restrict__ qo_indptr,
    const int32_t* __restrict__ kv_indptr, const int32_t* __restrict__ kv_indices, const int32_t* __restrict__ plan,
    uint16_t* __restrict__ out, int batch, int hkv, float scale_log2) {
    constexpr int NT = NWAVES * 64;
    constexpr int POSB = NWAVES / G;          // 32-row position blocks per workgroup
    constexpr int QPOS = 32 * POSB;           // query positions per workgroup
    constexpr int TAB = QPOS == 32 ? 0 : (QPOS == 64 ? 1 : 2);
    constexpr int KEYS_PER_PASS = NT / 16;    // 16 lanes move one 256-byte row
    constexpr int PASSES = PF_KVBLK / KEYS_PER_PASS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* k_lds = smem;                      // [64 keys][256 B], pieces swizzled by key & 15
    unsigned char* v_lds = smem + PF_KVBLK * 256;     // [64 keys][256 B], pieces swizzled by (key & 3) << 2

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l32 = lane & 31;
    const int32_t* tab = plan + TAB * (batch + 2);
    const int total_blocks = tab[0];
    const int unit = blockIdx.x;
    const int gblk = unit / hkv, head = unit % hkv;
    if (gblk >= total_blocks) return;
    // sequence that owns query block gblk: last b with tab[1 + b] <= gblk
    int lo = 0, hi = batch - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[1 + mid] <= gblk) lo = mid; else hi = mid - 1;
    }
    const int b = lo;
    const int qb = gblk - tab[1 + b];
    const int q0 = qo_indptr[b], lq = qo_indptr[b + 1] - q0;
    const int kv0 = kv_indptr[b], lkv = kv_indptr[b + 1] - kv0;
    const int offset = lkv - lq;              // query i sees keys 0 .. i + offset
    const int g = wave % G, pb = wave / G;
    const int qi = qb * QPOS + pb * 32 + l32;  // this lane's query (position within the new tokens)
    const bool q_ok = qi < lq;
    const int hq_total = hkv * G;
    const int hq = head * G + g;
    int kmax = qb * QPOS + QPOS + offset;       // keys visible to the workgroup's last query
    kmax = kmax > lkv ? lkv : kmax;
    const int wave_last_key = qb * QPOS + pb * 32 + 31 + offset;  // last key any query of this wave sees
    const int wave_first_vis = qb * QPOS + pb * 32 + offset;      // last key the wave's FIRST query sees
    const int n_tiles = (kmax + PF_KVBLK - 1) / PF_KVBLK;
    const int64_t row_elems = (int64_t)hkv * 128;

    // Q^T fragments: 8 k-steps of 16 dims; lane holds dims 16*ks + 8*half .. +7 of its query row
    nvx_bf16x8 qf[8];
    {
        const uint16_t* qrow = q + (int64_t)(q0 + (q_ok ? qi : 0)) * q_stride + hq * 128 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            uint4 t = *reinterpret_cast<const uint4*>(qrow + ks * 16);
            if (!q_ok) t = make_uint4(0, 0, 0, 0);
            qf[ks] = __builtin_bit_cast(nvx_bf16x8, t);
        }
    }

    nvx_f32x16 o_acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging registers for the tile after the one being computed
    PfStage stage;
    static_assert(PASSES == 2 || PASSES == 4, "tile staging is written for 256- or 512-thread workgroups");
    const int piece = tid & 15, krow0 = tid >> 4;
    const int gofs = head * 128 + piece * 8;
    if (n_tiles > 0) pf_gather<PASSES, KEYS_PER_PASS>(stage, 0, krow0, lkv, kv_indices + kv0, n_slots, k_cache + gofs, v_cache + gofs, row_elems);
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();            // every wave is done reading the previous tile
        pf_stage_write<PASSES, KEYS_PER_PASS>(stage, k_lds, v_lds, krow0, piece);
        __syncthreads();            // tile visible
        if (tile + 1 < n_tiles) pf_gather<PASSES, KEYS_PER_PASS>(stage, tile + 1, krow0, lkv, kv_indices + kv0, n_slots, k_cache + gofs, v_cache + gofs, row_elems);

        const int key_base = tile * PF_KVBLK;
        if (key_base > wave_last_key) continue;   // wave-uniform: nothing in this tile is visible to this wave

        // ---- S^T = K . Q^T ----
        nvx_f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
            const int key = kb * 32 + l32;
            const unsigned char* krow = k_lds + key * 256;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint4 a = *reinterpret_cast<const uint4*>(krow + (((2 * ks + half) ^ (key & 15)) << 4));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nvx_bf16x8, a), qf[ks], st[kb], 0, 0, 0);
            }
        }
        // ---- mask, online softmax (this lane: one query, 32 of the tile's 64 keys) ----
        // Scores stay unscaled; the scale rides on the fused multiply-add inside the exponent.  Only
        // tiles that cross the causal diagonal (or the end of the sequence) pay for the mask.
        const int vis = q_ok ? qi + offset : -1;   // last visible key index of this lane's query
        if (key_base + PF_KVBLK - 1 > wave_first_vis) {   // wave-uniform
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key_base + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    st[kb][r] = key <= vis ? st[kb][r] : -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * scale_log2);     // scale_log2 > 0
        const float mref = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - mref);
        m_run = m_new;
        l_run *= alpha;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
        }
        nvx_bf16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(st[kb][8 * s2 + 2 * j], scale_log2, -mref));
                    const float p1 = __builtin_amdgcn_exp2f(fmaf(st[kb][8 * s2 + 2 * j + 1], scale_log2, -mref));
                    l_run += p0 + p1;
                    w[j] = pack_bf16x2(p0, p1);
                }
                pf[kb][s2] = __builtin_bit_cast(nvx_bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            }
        // ---- O^T += V^T . P^T ----
        // fragment element j of lane half h is key 16*s2 + 8*(j>>2) + 4*h + (j&3) of the 32-key block;
        // the transpose read hands lane i of a 16-lane group column i of a 4-key x 16-dim block
        const int grp = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int key_a = kb * 32 + 16 * s2 + 4 * half + tq;   // row this lane addresses, first read
                const int key_b = key_a + 8;                            // second read
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int pc = d * 4 + 2 * (grp & 1) + (tp >> 1);
                    const unsigned char* pa = v_lds + key_a * 256 + ((pc ^ ((key_a & 3) << 2)) << 4) + 8 * (tp & 1);
                    const unsigned char* pbp = v_lds + key_b * 256 + ((pc ^ ((key_b & 3) << 2)) << 4) + 8 * (tp & 1);
                    const pf_s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pf_lds_s16x4*)pa);
                    const pf_s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pf_lds_s16x4*)pbp);
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                    o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nvx_bf16x8, vf), pf[kb][s2], o_acc[d], 0, 0, 0);
                }
            }
    }

    // ---- epilogue: O = O^T / l, one query per lane, dims in registers ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (q_ok) {
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        uint16_t* orow = out + ((int64_t)(q0 + qi) * hq_total + hq) * 128;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int dim = d * 32 + 8 * r4 + 4 * half;
                *reinterpret_cast<uint2*>(orow + dim) =
                    make_uint2(pack_bf16x2(o_acc[d][4 * r4] * inv, o_acc[d][4 * r4 + 1] * inv),
                               pack_bf16x2(o_acc[d][4 * r4 + 2] * inv, o_acc[d][4 * r4 + 3] * inv));
            }
    }
}

extern "C" int nvx_paged_prefill(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
                                 int64_t n_slots, const int32_t* qo_indptr, const int32_t* kv_indptr,
                                 const int32_t* kv_indices, const int32_t* plan_ws, void* out, int64_t n_tokens,
                                 int64_t batch, int n_q_heads, int n_kv_heads, float sm_scale, nvx_stream_t stream) {
    NVX_REQUIRE(batch > 0 && n_tokens >= 0 && n_q_heads > 0 && n_kv_heads > 0 && n_slots > 0, NVX_EINVAL, "nvx_paged_prefill: bad size");
    if (n_tokens == 0) return NVX_OK;
    NVX_REQUIRE(q && k_cache && v_cache && qo_indptr && kv_indptr && kv_indices && plan_ws && out, NVX_EINVAL,
                "nvx_paged_prefill: null pointer");
    NVX_REQUIRE(n_q_heads % n_kv_heads == 0, NVX_EINVAL, "nvx_paged_prefill: n_q_heads must be a multiple of n_kv_heads");
    NVX_REQUIRE(q_row_stride % 8 == 0 && q_row_stride >= (int64_t)n_q_heads * 128, NVX_EINVAL, "nvx_paged_prefill: bad q row stride");
    NVX_REQUIRE(nvx_aligned16(q) && nvx_aligned16(k_cache) && nvx_aligned16(v_cache) && nvx_aligned16(out), NVX_EINVAL,
                "nvx_paged_prefill: pointers must be 16-byte aligned");
    const int g = n_q_heads / n_kv_heads;
    const float scale_log2 = sm_scale * PF_LOG2E;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 2 * PF_KVBLK * 256;
#define NVX_PF(GG, NW, QPOS)                                                                                          \
    do {                                                                                                               \
        const int64_t blocks = ((n_tokens + (QPOS) - 1) / (QPOS) + batch) * n_kv_heads;                                \
        hipLaunchKernelGGL((paged_prefill_kernel<GG, NW>), dim3((unsigned)blocks), dim3((NW) * 64), lds, st,           \
                           (const uint16_t*)q, q_row_stride, (const uint16_t*)k_cache, (const uint16_t*)v_cache,       \
                           n_slots, qo_indptr, kv_indptr, kv_indices, plan_ws, (uint16_t*)out, (int)batch, n_kv_heads, \
                           scale_log2);                                                                                \
    } while (0)
    switch (g) {
        case 1: NVX_PF(1, 4, 128); break;
        case 2: NVX_PF(2, 8, 128); break;
        case 4: NVX_PF(4, 4, 32); break;
        case 8: NVX_PF(8, 8, 32); break;
        default: return nvx_fail(NVX_EUNSUPPORTED, "nvx_paged_prefill: query/kv head ratio %d not in {1,2,4,8}", g);
    }
#undef NVX_PF
    return nvx_check_launch("nvx_paged_prefill");
}

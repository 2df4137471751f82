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

#include <type_traits>

namespace NVX_NS {


#define PF_KVBLK 64
#define PF_LOG2E 1.4426950408889634f

typedef __attribute__((ext_vector_type(4))) short pf_s16x4;
typedef __attribute__((address_space(3))) pf_s16x4 pf_lds_s16x4;

// plan_ws (int32): three prefix tables, for 32, 64 and 128 query positions per workgroup:
//   table t at ws + t*(batch+2): [0] = total blocks, [1 + b] = first block of sequence b, [1 + batch] = total
#ifndef NVX_F16   // independent of the 16-bit format: defined by the bfloat16 build only
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

#endif
// Slot ids of a tile (64 keys) travel block table -> LDS one tile before the rows they name, as ONE
// 256-byte copy issued by wave 0 (lane l fetches the id of key l).  No thread holds ids that are still
// in flight: a load issued from inline assembly hands the compiler a register whose value arrives
// later, and nothing stops the register allocator from copying that register before then (seen as rare
// wrong rows); a load the compiler knows of makes it wait for the row copies as well.
__device__ __forceinline__ void pf_dma4(const int32_t* g, unsigned lds_addr) {
    const unsigned uni = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(g), "s"(uni) : "memory");
}
// by wave 0: ids of tile `tile` of a sequence (slot list seq_slots, lkv keys; rows past the end repeat the
// last key and are masked later) into the 64-int LDS array at `dst`
__device__ __forceinline__ void pf_fetch_slot_ids(const int32_t* __restrict__ seq_slots, int tile, int lkv, int lane, unsigned dst) {
    pf_dma4(seq_slots + min(tile * PF_KVBLK + lane, lkv - 1), dst);
}
// every vector-memory operation of this wave has completed, the copies into LDS included (the compiler
// does not know of those)
__device__ __forceinline__ void pf_wait_copies() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
// One row pair (K and V, 256 B each, 16 lanes per row) straight from the pool into LDS, no register
// staging: a wave instruction writes 64 consecutive 16-byte slots (4 rows) starting at a wave-uniform
// LDS address, so the swizzle is applied on the global side: the lane that fills position `pos` of row r
// fetches piece pos ^ swz(r).
// The copy is issued from inline assembly: through the compiler's builtin every later LDS read is held
// back until the copy has landed (the compiler cannot tell the two buffers apart), which puts the whole
// HBM latency in front of every tile.  The waits are therefore explicit: pf_wait_copies() before the
// barrier that publishes a tile.
__device__ __forceinline__ void pf_dma16(const unsigned char* g, unsigned lds_addr) {
    // lds_addr is wave-uniform by construction; the explicit broadcast puts it in a scalar register even
    // where the compiler cannot prove that.  m0 has no other user in this kernel: the toolchain refuses m0 in a clobber list (reserved register), so tests/test_abi_exports.py disassembles the library and checks that every access to m0 is one of these moves.
    const unsigned uni = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(uni) : "memory");
}
// LDS reads by 32-bit LDS address (address arithmetic with XOR would hide the address space from the
// compiler if it were done on generic pointers)
typedef __attribute__((address_space(3))) const nvx_u32x4 pf_lds_u4;
__device__ __forceinline__ nvx_u32x4 pf_lds_read16(unsigned addr) { return *reinterpret_cast<pf_lds_u4*>((uintptr_t)addr); }
__device__ __forceinline__ pf_s16x4 pf_lds_read_tr(unsigned addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<pf_lds_s16x4*>((uintptr_t)addr));
}
__device__ __forceinline__ unsigned pf_lds_addr(const unsigned char* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
// kbase / vbase already include the head and this lane's (swizzled) piece; unsigned 32-bit slot
// arithmetic keeps an address at one compare, one select and one 64-bit multiply-add
__device__ __forceinline__ void pf_gather_one(int slot_id, unsigned n_slots, const unsigned char* __restrict__ kbase,
                                              const unsigned char* __restrict__ vbase, unsigned row_bytes,
                                              unsigned k_dst, unsigned v_dst) {
    unsigned slot = (unsigned)slot_id;   // -1 -> 0xffffffff: out of range
    slot = slot < n_slots ? slot : 0u;
    const unsigned long long off = (unsigned long long)slot * row_bytes;
    pf_dma16(kbase + off, k_dst);
    pf_dma16(vbase + off, v_dst);
}
// ids: LDS address of the tile's 64 slot ids; k_dst / v_dst: start of this WAVE's four rows of pass 0 in
// the tile buffer (wave-uniform)
typedef __attribute__((address_space(3))) const int pf_lds_i32;
template <int PASSES, int KEYS_PER_PASS>
__device__ __forceinline__ void pf_gather(unsigned ids, int krow0, unsigned n_slots, const unsigned char* __restrict__ kbase,
                                          const unsigned char* __restrict__ vbase, unsigned row_bytes,
                                          unsigned k_dst, unsigned v_dst) {
    constexpr int STEP = KEYS_PER_PASS * 256;
    int id[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) id[p] = *reinterpret_cast<pf_lds_i32*>((uintptr_t)(ids + 4 * (krow0 + p * KEYS_PER_PASS)));
#pragma unroll
    for (int p = 0; p < PASSES; ++p) pf_gather_one(id[p], n_slots, kbase, vbase, row_bytes, k_dst + p * STEP, v_dst + p * STEP);
}

#if defined(PF_STAMPS) && !defined(NVX_F16)
// Build-time instrumentation (tools/build_variant.sh ... "-DPF_STAMPS"): shader-clock stamps of waves 0 and 3 of the
// workgroups of blocks 8 and 9 (kv head 0; blocks of a sequence are taken last first, so these are query blocks 119 and
// 118 of a 8 192-token sequence) at eight points of tiles 16..47, read back with nvx_debug_pf_stamps.
// Never in the shipped library.
__device__ unsigned long long pf_stamps[2 * 2 * 32 * 8];
extern "C" int nvx_debug_pf_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pf_stamps), sizeof(pf_stamps)); }
#define PF_STAMP(k) do { if (stamp_on && tile >= 16 && tile < 48 && lane == 0) \
    pf_stamps[(((stamp_blk - 8) * 2 + (wave ? 1 : 0)) * 32 + (tile - 16)) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PF_STAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
template <int G, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 2) void paged_prefill_kernel(
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
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    // two tile buffers of [K: 64 keys x 256 B, pieces swizzled by key & 15][V: same, swizzled by (key & 3) << 2],
    // two arrays of 64 slot ids, the counters of the owner lookup
    constexpr int BUF_BYTES = 2 * PF_KVBLK * 256;
    constexpr int IDS_BYTES = 4 * PF_KVBLK;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5, l32 = lane & 31;
    const int32_t* tab = plan + TAB * (batch + 2);
    const int total_blocks = tab[0];
    const int unit = blockIdx.x;
    const int gblk = unit / hkv, head = unit % hkv;
    if (gblk >= total_blocks) return;
    // Sequence that owns query block gblk: the last b with tab[1 + b] <= gblk.  The table is
    // non-decreasing, so b + 1 is the number of entries <= gblk: every thread tests some entries and the
    // workgroup adds up.  (A binary search is log2(batch) dependent loads, several microseconds in front
    // of every workgroup's first tile; this is one.)
    // (the counters sit behind the tile buffers: the fragment addressing XORs address bits and needs the
    // buffers at a 256-byte boundary, which a static LDS array in front of them would break)
    int* owner_count = reinterpret_cast<int*>(smem + 2 * BUF_BYTES + 2 * IDS_BYTES);
    int below = 0;
    for (int i = tid; i < batch; i += NT) below += tab[1 + i] <= gblk ? 1 : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) below += __shfl_xor(below, m, 64);
    if (lane == 0) owner_count[wave] = below;
    __syncthreads();
    int b = -1;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) b += owner_count[w];
    b = __builtin_amdgcn_readfirstlane(b < 0 ? 0 : b);
    // blocks of a sequence are taken last first: the last query block sees the most keys, and the launch should end
    // on short blocks, not on the longest ones
    const int qb = tab[2 + b] - 1 - gblk;
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

    // Q^T fragments: 8 k-steps of 16 dims; lane holds dims 16*ks + 8*half .. +7 of its query row
    nvx_h16x8 qf[8];
    {
        const uint16_t* qrow = q + (int64_t)(q0 + (q_ok ? qi : 0)) * q_stride + hq * 128 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            uint4 t = *reinterpret_cast<const uint4*>(qrow + ks * 16);
            if (!q_ok) t = make_uint4(0, 0, 0, 0);
            qf[ks] = __builtin_bit_cast(nvx_h16x8, t);
        }
    }

    nvx_f32x16 o_acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    static_assert(PASSES == 2 || PASSES == 4, "tile gather is written for 256- or 512-thread workgroups");
    // Pipeline, one barrier per tile: while tile t is computed from one buffer, the rows of tile t + 1 travel
    // pool -> LDS into the other and the slot ids of tile t + 2 block table -> LDS.
    const int pos = tid & 15, krow0 = tid >> 4;
    const int32_t* seq_slots = kv_indices + kv0;
    // this lane's piece of a row: KEYS_PER_PASS is a multiple of 16, so the swizzle is the same for every pass
    const unsigned char* kg = reinterpret_cast<const unsigned char*>(k_cache) + head * 256 + ((pos ^ (krow0 & 15)) << 4);
    const unsigned char* vg = reinterpret_cast<const unsigned char*>(v_cache) + head * 256 + ((pos ^ ((krow0 & 3) << 2)) << 4);
    const unsigned row_bytes = (unsigned)hkv * 256u;
    const unsigned n_slots_u = n_slots > 0xffffffffll ? 0xffffffffu : (unsigned)n_slots;
    const unsigned wave_rows = (unsigned)__builtin_amdgcn_readfirstlane(wave * 4 * 256);
    const unsigned lds0 = pf_lds_addr(smem);
    // per-lane byte offsets of the fragment reads inside a tile buffer (see the reads below)
    const int k_lane_off = l32 * 256 + ((half ^ (l32 & 15)) << 4);
    const int v_lane_off = [&] {
        const int grp = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
        const int row = 4 * half + tq;
        const int pc0 = 2 * (grp & 1) + (tp >> 1);
        return row * 256 + ((pc0 ^ ((row & 3) << 2)) << 4) + 8 * (tp & 1);
    }();
    unsigned ring0 = lds0, ring1 = lds0 + BUF_BYTES;                 // tile being computed, tile arriving
    unsigned ids0 = lds0 + 2 * BUF_BYTES, ids1 = ids0 + IDS_BYTES;     // slot ids of the next tile, of the one after
    if (n_tiles > 0) {
        if (wave == 0) {
            pf_fetch_slot_ids(seq_slots, 0, lkv, lane, ids0);
            if (n_tiles > 1) pf_fetch_slot_ids(seq_slots, 1, lkv, lane, ids1);
        }
        pf_wait_copies();
        __syncthreads();
        // the first TWO tiles travel together: a workgroup's first round trip to the pool is the one nothing
        // can hide, so it carries as much as the buffers hold
        pf_gather<PASSES, KEYS_PER_PASS>(ids0, krow0, n_slots_u, kg, vg, row_bytes, ring0 + wave_rows, ring0 + PF_KVBLK * 256 + wave_rows);
        if (n_tiles > 1)
            pf_gather<PASSES, KEYS_PER_PASS>(ids1, krow0, n_slots_u, kg, vg, row_bytes, ring1 + wave_rows, ring1 + PF_KVBLK * 256 + wave_rows);
        pf_wait_copies();
    }
    __syncthreads();
    { const unsigned t0 = ids0; ids0 = ids1; ids1 = t0; }
    // at the top of tile t: ids0 holds the ids of tile t + 1, ids1 is free (its ids were used a tile ago)
#ifdef PF_STAMPS
    const int stamp_blk = gblk;
    const bool stamp_on = (gblk == 8 || gblk == 9) && head == 0 && (wave == 0 || wave == 3);
#endif
    for (int tile = 0; tile < n_tiles; ++tile) {
        PF_STAMP(0);
        const unsigned k_lds = ring0;   // LDS byte addresses
        const unsigned v_lds = k_lds + PF_KVBLK * 256;
        if (tile + 1 < n_tiles) {   // the other buffer held tile t - 1: every wave left it before the last barrier
            if (tile > 0)           // (tile 1 came with tile 0)
                pf_gather<PASSES, KEYS_PER_PASS>(ids0, krow0, n_slots_u, kg, vg, row_bytes, ring1 + wave_rows, ring1 + PF_KVBLK * 256 + wave_rows);
            if (tile + 2 < n_tiles && wave == 0) pf_fetch_slot_ids(seq_slots, tile + 2, lkv, lane, ids1);
        }

        PF_STAMP(1);
        const int key_base = tile * PF_KVBLK;
        // One tile for this wave: NKB = 2 when both 32-key blocks hold keys some query of the wave sees, 1 when the
        // second block lies wholly beyond the causal diagonal for all of them (half the waves of a workgroup on
        // its last tile): its score and value MFMAs, exponentials and LDS reads are then not issued at all.
        auto tile_body = [&](auto nkb_tag) __attribute__((always_inline)) {
        constexpr int NKB = decltype(nkb_tag)::value;

        // ---- S^T = K . Q^T ----
        // all K fragments are requested before the first MFMA, and the 32-key blocks alternate so that no MFMA
        // waits for the one before it
        nvx_f32x16 st[NKB];
        nvx_u32x4 kf[NKB][8];
        // fragment ks of key row l32 (+32 for the second block) sits at piece (2 ks + half) ^ (key & 15): one
        // per-tile base, the k-step applied as an XOR on the address bits, the block as an immediate offset
        const unsigned kfrag0 = k_lds + k_lane_off;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                kf[kb][ks] = pf_lds_read16((kfrag0 ^ (ks << 5)) + kb * 32 * 256);
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
                st[kb] = mfma_32x32x16_h16(__builtin_bit_cast(nvx_h16x8, kf[kb][ks]), qf[ks], st[kb]);
        // scheduling order for this region: the LDS reads, then the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 8 * NKB, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * NKB, 0);
        PF_STAMP(2);
        // ---- mask, online softmax (this lane: one query, 16 of each 32-key block) ----
        // Scores stay unscaled; the scale rides on the fused multiply-add inside the exponent.  Only
        // tiles that cross the causal diagonal (or the end of the sequence) pay for the mask.
        const int vis = q_ok ? qi + offset : -1;   // last visible key index of this lane's query
        if (key_base + 32 * NKB - 1 > wave_first_vis) {   // wave-uniform
            asm volatile("" : : : "memory");   // keeps the branch: as selects on every tile the mask costs 96 VALU ops
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key_base + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    st[kb][r] = key <= vis ? st[kb][r] : -INFINITY;
                }
        }
        // (one chain of maxima per 32-key block: the chains are dependent instructions, and two run side by side)
        float mxb[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            mxb[kb] = st[kb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mxb[kb] = fmaxf(mxb[kb], st[kb][r]);
        }
        float mx = NKB == 2 ? fmaxf(mxb[0], mxb[NKB - 1]) : mxb[0];
        mx = fmaxf(mx, xor32(mx));
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
        PF_STAMP(3);
        // ---- P^T, then O^T += V^T . P^T ----
        // V^T fragment element j of lane half h is key 16*s2 + 8*(j>>2) + 4*h + (j&3) of the 32-key block; the
        // transpose read hands lane i of a 16-lane group column i of a 4-key x 16-dim block.  The lane
        // addresses row 4 half + tq (+ 16 s2 + 32 kb, and + 8 for the second read: immediate offsets); the
        // dim block d enters the swizzled piece index as an XOR on address bits 6-7.
        // Order, fixed with scheduling barriers because the compiler otherwise sinks every LDS read to
        // just before its MFMA and waits out the LDS latency 32 times per tile: the V fragments of the
        // first two (kb, s2) groups are requested before the exponentials and land under them; each group
        // of four MFMAs is followed by the reads of the group two ahead.
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        constexpr int NGRP = 2 * NKB;   // groups of 16 keys
        const unsigned vfrag0 = v_lds + v_lane_off;
        s16x8 vf[NGRP][4];
        auto read_group = [&](int grp_i) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned pa = (vfrag0 ^ (d << 6)) + grp_i * 16 * 256;   // group = 2 kb + s2: 16 keys each
                const pf_s16x4 va = pf_lds_read_tr(pa);
                const pf_s16x4 vb = pf_lds_read_tr(pa + 8 * 256);
                vf[grp_i][d] = s16x8{va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
            }
        };
        read_group(0);
        read_group(1);
        __builtin_amdgcn_sched_barrier(0);
        nvx_h16x8 pf[NGRP];
        auto probabilities = [&](int gi) {   // P^T of 16 keys: 8 exponentials per lane
            const int kb = gi >> 1, s2 = gi & 1;
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(st[kb][8 * s2 + 2 * j], scale_log2, -mref));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(st[kb][8 * s2 + 2 * j + 1], scale_log2, -mref));
                l_run += p0 + p1;
                w[j] = pack_h16x2(p0, p1);
            }
            pf[gi] = __builtin_bit_cast(nvx_h16x8, make_uint4(w[0], w[1], w[2], w[3]));
        };
        probabilities(0);
        __builtin_amdgcn_sched_barrier(0);
        PF_STAMP(4);
        // the four MFMAs of a group run on the matrix pipe while the vector ALU makes the next group's
        // probabilities: one MFMA, then a share of the exponentials, four times over
#pragma unroll
        for (int gi = 0; gi < NGRP; ++gi) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                o_acc[d] = mfma_32x32x16_h16(__builtin_bit_cast(nvx_h16x8, vf[gi][d]), pf[gi], o_acc[d]);
            if (gi + 1 < NGRP) probabilities(gi + 1);
            if (gi + 2 < NGRP) read_group(gi + 2);
            if (gi + 1 < NGRP) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x402, 8, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        PF_STAMP(5);
        };
        if (key_base + 32 <= wave_last_key) tile_body(std::integral_constant<int, 2>{});        // wave-uniform
        else if (key_base <= wave_last_key) tile_body(std::integral_constant<int, 1>{});        // else: nothing visible
        pf_wait_copies();
        PF_STAMP(6);
        __syncthreads();   // tile + 1 and the ids of tile + 2 visible; this tile's buffer free
        PF_STAMP(7);
        { const unsigned t0 = ring0; ring0 = ring1; ring1 = t0; }
        { const unsigned t0 = ids0; ids0 = ids1; ids1 = t0; }
    }

    // ---- epilogue: O = O^T / l ----
    // A lane holds 64 of the 128 dims of ONE query row, so stores straight from registers are 64 scattered
    // 8-byte writes per instruction (a third of the kernel's time on short sequences).  Each wave instead
    // turns its 32 x 128 block around in LDS (the tile buffers are free: the loop ended with a barrier;
    // rows 272 B apart keep the 8-byte writes off each other's banks and the 16-byte reads aligned) and
    // stores whole 256-byte rows, four per instruction.
    const float l_tot = l_run + xor32(l_run);
    {
        constexpr int ROW = 272;
        const unsigned stage = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(wave * 32 * ROW);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) u32x2 lds_u2;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int dim = d * 32 + 8 * r4 + 4 * half;
                *reinterpret_cast<lds_u2*>((uintptr_t)(stage + l32 * ROW + dim * 2)) =
                    u32x2{pack_h16x2(o_acc[d][4 * r4] * inv, o_acc[d][4 * r4 + 1] * inv),
                          pack_h16x2(o_acc[d][4 * r4 + 2] * inv, o_acc[d][4 * r4 + 3] * inv)};
            }
        // a wave's LDS operations complete in order: no barrier between its writes and its reads
        const int rq = lane >> 4, pc = lane & 15;
        const int q_first = qb * QPOS + pb * 32;   // the wave's first query position
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + rq;
            const nvx_u32x4 v = pf_lds_read16(stage + row * ROW + pc * 16);
            if (q_first + row < lq)
                *reinterpret_cast<nvx_u32x4*>(out + ((int64_t)(q0 + q_first + row) * hq_total + hq) * 128 + pc * 8) = v;
        }
    }
}

extern "C" int NVX_ENTRY(nvx_paged_prefill)(const void* q, int64_t q_row_stride, const void* k_cache, const void* v_cache,
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
    // two tile buffers of K and V, two tiles of slot ids, per-wave counters; at the end the same memory
    // stages the output, 32 rows of 272 B per wave
    const size_t loop_bytes = 2 * 2 * PF_KVBLK * 256 + 2 * 4 * PF_KVBLK + 64;
#define NVX_PF(GG, NW, QPOS)                                                                                           \
    do {                                                                                                                \
        const size_t lds_bytes = loop_bytes > (size_t)(NW) * 32 * 272 ? loop_bytes : (size_t)(NW) * 32 * 272;           \
        static const hipError_t attr = hipFuncSetAttribute((const void*)paged_prefill_kernel<GG, NW>,                   \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        NVX_REQUIRE(attr == hipSuccess, NVX_ELAUNCH, "nvx_paged_prefill: cannot reserve %d KB of LDS", (int)(lds_bytes / 1024)); \
        const int64_t blocks = ((n_tokens + (QPOS) - 1) / (QPOS) + batch) * n_kv_heads;                                 \
        hipLaunchKernelGGL((paged_prefill_kernel<GG, NW>), dim3((unsigned)blocks), dim3((NW) * 64), lds_bytes, st,      \
                           (const uint16_t*)q, q_row_stride, (const uint16_t*)k_cache, (const uint16_t*)v_cache,        \
                           n_slots, qo_indptr, kv_indptr, kv_indices, plan_ws, (uint16_t*)out, (int)batch, n_kv_heads,  \
                           scale_log2);                                                                                 \
    } while (0)
    switch (g) {
        case 1: NVX_PF(1, 4, 128); break;
        // two 4-wave workgroups per CU (each in its own phase of the tile loop) measured faster than one
        // 8-wave workgroup of 128 queries (0.75 vs 0.86 ms on the bench prompts); a third tile buffer (row
        // copies two tiles ahead) changed neither
        case 2: NVX_PF(2, 4, 64); break;
        case 4: NVX_PF(4, 4, 32); break;   // 8 waves / 64 queries measured slower here too (1.77 vs 1.57 ms at 32 q heads)
        case 8: NVX_PF(8, 8, 32); break;
        default: return nvx_fail(NVX_EUNSUPPORTED, "nvx_paged_prefill: query/kv head ratio %d not in {1,2,4,8}", g);
    }
#undef NVX_PF
    return nvx_check_launch("nvx_paged_prefill");
}

}  // namespace NVX_NS

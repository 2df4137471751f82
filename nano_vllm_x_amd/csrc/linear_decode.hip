// Skinny linear layer for decode steps on gfx950: Y[M, N] = X[M, K] * W[N, K]^T with M <= 256 rows
// (one per running sequence), bf16 in, fp32 accumulate.
//
// At these sizes the layer is a stream of the weight matrix (each byte used once per step) with the
// whole activation matrix re-read from L2 by every workgroup, so the design is about bytes in flight
// and a short dependent chain, not FLOPs:
//   * a workgroup owns 16 rows of W over one slice of K (grid = N/16 x k_splits, so that even a
//     1024-row projection puts >= 256 workgroups on the chip); its 512 threads pull that slice with
//     16-byte loads issued all at once and park it in LDS in MFMA operand order (XOR-swizzled so that
//     both the row-major writes and the fragment reads are bank-conflict free);
//   * each of the 8 waves owns every eighth 16-row tile of X: B fragments come straight from global memory
//     (L2-resident activations) in double-buffered groups of 8 k-steps, A fragments from LDS,
//     v_mfma_f32_16x16x32_bf16 accumulates; no cross-wave reduction is needed;
//   * epilogues: bf16 store; SiLU(gate) * up for a fused gate/up projection (the workgroup owns 16
//     gate rows and the 16 matching up rows); fp32 partials when K is split across workgroups, which
//     the RMSNorm kernel that follows sums (nvx_rmsnorm_splitk).
#include "nvx_dev.h"

#include <stdlib.h>

namespace NVX_NS {

#define LIN_THREADS 512
#define LIN_WAVES 8

__device__ __forceinline__ nvx_f32x4 mfma16(nvx_u32x4 a, nvx_u32x4 b, nvx_f32x4 c) {
    return mfma_16x16x32_h16(a, b, c);
}

// LDS slot (16 bytes each) of the A fragment piece: k-step j, lane quad q (k offset 8q), W row r.
__device__ __forceinline__ int lin_slot(int j, int q, int r) { return j * 64 + q * 16 + (r ^ ((q << 2) | (j & 3))); }

// MODE 0: y bf16 (M, N).  MODE 1: w rows [0, N/2) are gate rows, [N/2, N) up rows; y bf16 (M, N/2) =
// bf16(bf16(silu(gate)) * up) on the bf16-rounded projections, as the unfused ops compute it.
// MODE 2: fp32 partials (k_splits, M, N).
// KC = the K slice of one workgroup (compile time, so that every loop below is straight-line code:
// with per-element bounds checks the compiler serialises each load behind a full wait).
// STAGED = false: activation fragments are loaded straight from global memory in MFMA operand order:
//   shortest dependent chain, but every load instruction is 64 separate 16-byte requests (16 rows x 4
//   k pieces), which the address unit serves at about one per clock, so the time grows with the number
//   of activation rows (~0.045 us per row).
// STAGED = true: each wave loads its rows coalesced (32 lanes x 16 B = 512 contiguous bytes per row),
//   parks them in a private 8 KB LDS area (pieces XOR-swizzled by row) and reads the fragments from
//   there: one more LDS hop per 256 k, a quarter of the address-unit work.  Used above 48 rows.
// WIDE (modes 0 and 2): the workgroup owns 32 rows of W instead of 16, so half as many workgroups re-read
// the activations from L2.  gridDim.z > 1 deals the 16-row tiles of X round-robin to that many workgroups
// per W slice (their ids differ by a multiple of 8, so they share an XCD and its L2 copy of the slice).
template <int MODE, int KC, bool STAGED, bool WIDE = false>
__global__ __launch_bounds__(LIN_THREADS) void linear_decode_kernel(
    const uint16_t* __restrict__ x, int64_t x_stride, const uint16_t* __restrict__ w, void* __restrict__ y, int m_rows, int n_rows,
    int k_total) {
    static_assert(!(WIDE && MODE == 1), "the gate/up mode already owns 32 rows of W");
    constexpr int HALVES = (MODE == 1 || WIDE) ? 2 : 1;
    constexpr int STEPS = KC / 32;               // MFMA k-steps per tile
    constexpr int GPT = STAGED ? STEPS / 8 : ((MODE == 1 && KC == 1024) ? 4 : (KC > 1024 ? KC / 512 : 2));   // groups per tile
    constexpr int GROUP = STEPS / GPT;           // k-steps per group (direct: half a tile in flight; staged: 8 = 256 k)
    constexpr int PIECES_ROW = KC / 8;           // 16-byte pieces per W row
    constexpr int PIECES_HALF = 16 * PIECES_ROW;
    constexpr int WLOADS = HALVES * PIECES_HALF / LIN_THREADS;   // pieces per thread, all in flight at once
    static_assert(GROUP % 2 == 0, "k-steps are permuted in pairs");
    static_assert(KC % 256 == 0, "K slice must be a multiple of 256");
    __shared__ nvx_u32x4 lds[HALVES * STEPS * 64];
    __shared__ nvx_u32x4 stage[STAGED ? LIN_WAVES * 16 * 32 : 1];   // per wave: 16 rows x 32 pieces of 16 B
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int quad = lane >> 4, l16 = lane & 15;
    const int n_out = MODE == 1 ? n_rows / 2 : n_rows;
    const int n0 = blockIdx.x * (WIDE ? 32 : 16);
    const int k0 = blockIdx.y * KC;
    const int half_stride = MODE == 1 ? n_out : 16;   // W row distance between the two halves
    const int msplit = gridDim.z, mz = blockIdx.z;

    // ---- this wave's work: (tile, group of 8 k-steps) items; the first item's X loads go in flight before W
    const int tiles = ((m_rows + 15) / 16 - mz + msplit - 1) / msplit;   // of this workgroup: mz, mz + msplit, ...
    const int my_tiles = tiles > wave ? (tiles - wave + LIN_WAVES - 1) / LIN_WAVES : 0;
    const int n_items = my_tiles * GPT;
    nvx_u32x4 b_even[GROUP], b_odd[STAGED ? 1 : GROUP];
    auto load_b = [&](nvx_u32x4 (&dst)[GROUP], int item) {
        const int tile = (wave + LIN_WAVES * (item / GPT)) * msplit + mz, group = item % GPT;
        if constexpr (STAGED) {
            // load u: rows 2u and 2u+1 of the tile, 32 lanes x 16 B = the 256 k of this group, contiguous
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int row = tile * 16 + 2 * u + (lane >> 5);
                row = row < m_rows ? row : m_rows - 1;
                dst[u] = *reinterpret_cast<const nvx_u32x4*>(x + (int64_t)row * x_stride + k0 + group * 256 + (lane & 31) * 8);
            }
        } else {
            int row = tile * 16 + l16;
            row = row < m_rows ? row : m_rows - 1;
            // k order within a pair of k-steps is permuted so that the four lane quads of two consecutive
            // loads cover one whole 128-byte line of the row (quad q: bytes 32q..32q+15, then 32q+16..32q+31)
            // instead of half of two lines; W is parked in LDS in the same order
            const uint16_t* p = x + (int64_t)row * x_stride + k0 + group * (GROUP * 32) + quad * 16;
#pragma unroll
            for (int i = 0; i < GROUP; ++i) dst[i] = *reinterpret_cast<const nvx_u32x4*>(p + (i >> 1) * 64 + (i & 1) * 8);
        }
    };
    if (n_items > 0) load_b(b_even, 0);

    // ---- W slice -> LDS, parked in operand order ---------------------------------------------------
    {
        nvx_u32x4 v[WLOADS];
#pragma unroll
        for (int u = 0; u < WLOADS; ++u) {
            const int p = u * LIN_THREADS + tid;
            const int h = p / PIECES_HALF, ph = p % PIECES_HALF;
            const int r = ph / PIECES_ROW, kp = ph % PIECES_ROW;
            v[u] = *reinterpret_cast<const nvx_u32x4*>(w + (int64_t)(n0 + h * half_stride + r) * k_total + k0 + kp * 8);
        }
#pragma unroll
        for (int u = 0; u < WLOADS; ++u) {
            const int p = u * LIN_THREADS + tid;
            const int h = p / PIECES_HALF, ph = p % PIECES_HALF;
            const int r = ph / PIECES_ROW, kp = ph % PIECES_ROW;
            lds[h * STEPS * 64 + lin_slot(2 * (kp >> 3) + (kp & 1), (kp >> 1) & 3, r)] = v[u];   // piece kp -> (k-step, quad)
        }
    }
    __syncthreads();

    // ---- tiles of X ------------------------------------------------------------------------------
    nvx_f32x4 acc[HALVES];
    nvx_u32x4* my_stage = stage + (STAGED ? wave * (16 * 32) : 0);
    auto compute = [&](const nvx_u32x4 (&bfrag)[GROUP], int item) {
        const int tile = (wave + LIN_WAVES * (item / GPT)) * msplit + mz, g = item % GPT;
        if (g == 0) {
#pragma unroll
            for (int h = 0; h < HALVES; ++h) acc[h] = nvx_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < GROUP; ++i) {
            nvx_u32x4 b;
            if constexpr (STAGED) {   // piece of k-step i for quad q in the pair-permuted k order, row l16
                const int piece = 8 * (i >> 1) + 2 * quad + (i & 1);
                b = my_stage[l16 * 32 + (piece ^ l16)];
            } else {
                b = bfrag[i];
            }
#pragma unroll
            for (int h = 0; h < HALVES; ++h)
                acc[h] = mfma16(lds[h * STEPS * 64 + lin_slot(g * GROUP + i, quad, l16)], b, acc[h]);
        }
        if (g != GPT - 1) return;
        // C layout: this lane holds X row (tile*16 + l16), W rows n0 + 4*quad + 0..3
        const int row = tile * 16 + l16;
        if (row >= m_rows) return;
        const int col = n0 + quad * 4;
        if (MODE == 0) {
#pragma unroll
            for (int h = 0; h < HALVES; ++h) {
                uint16_t* dst = reinterpret_cast<uint16_t*>(y) + (int64_t)row * n_rows + col + h * 16;
                *reinterpret_cast<uint2*>(dst) = make_uint2(pack_h16x2(acc[h][0], acc[h][1]), pack_h16x2(acc[h][2], acc[h][3]));
            }
        } else if (MODE == 1) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gate = round_h16(acc[0][r]), up = round_h16(acc[HALVES - 1][r]);
                const float sg = round_h16(gate / (1.0f + __expf(-gate)));
                o[r] = sg * up;
            }
            uint16_t* dst = reinterpret_cast<uint16_t*>(y) + (int64_t)row * n_out + col;
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3]));
        } else {
#pragma unroll
            for (int h = 0; h < HALVES; ++h) {
                float* dst = reinterpret_cast<float*>(y) + ((int64_t)blockIdx.y * m_rows + row) * n_rows + col + h * 16;
                *reinterpret_cast<float4*>(dst) = make_float4(acc[h][0], acc[h][1], acc[h][2], acc[h][3]);
            }
        }
    };
    if constexpr (STAGED) {
        // per item: registers -> this wave's staging area, next item's loads into the freed registers,
        // fragments back from LDS (a wave's LDS operations complete in order: no barrier needed)
        for (int item = 0; item < n_items; ++item) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = 2 * u + (lane >> 5);
                my_stage[r * 32 + ((lane & 31) ^ r)] = b_even[u];
            }
            if (item + 1 < n_items) load_b(b_even, item + 1);
            compute(b_even, item);
        }
    } else {
        // the next item's loads go in flight before the current item is multiplied
        for (int item = 0; item < n_items; item += 2) {
            if (item + 1 < n_items) load_b(b_odd, item + 1);
            compute(b_even, item);
            if (item + 1 >= n_items) break;
            if (item + 2 < n_items) load_b(b_even, item + 2);
            compute(b_odd, item + 1);
        }
    }
}

// Gate/up projection with SiLU(gate) * up for staged batches, 32 gate rows + the 32 matching up rows per
// workgroup and the 16-row tiles of X dealt to gridDim.z workgroups per W slice (the same trade as WIDE
// above: half the activation traffic at the same weight traffic and workgroup count).  64 rows of W over
// K = 1024 would need 128 KB of LDS beside the 64 KB of staging areas, so W passes through LDS in two
// halves of K and the accumulators live across them: each wave owns at most ONE tile of X
// (tiles <= 8 * gridDim.z, checked by the host).
template <int KC>
__global__ __launch_bounds__(LIN_THREADS) void linear_gateup_wide_kernel(
    const uint16_t* __restrict__ x, int64_t x_stride, const uint16_t* __restrict__ w, uint16_t* __restrict__ y, int m_rows,
    int n_rows, int k_total) {
    constexpr int KPH = 2, KP = KC / KPH;        // K phases, k per phase
    constexpr int STEPS = KP / 32, GROUPS = KP / 256;
    constexpr int PIECES_ROW = KP / 8, PIECES_Q = 16 * PIECES_ROW;
    constexpr int WLOADS = 4 * PIECES_Q / LIN_THREADS;
    static_assert(KP % 256 == 0, "a phase is a whole number of 256-k groups");
    __shared__ nvx_u32x4 lds[4 * STEPS * 64];              // quarters: gate rows 0-15, 16-31, up rows 0-15, 16-31
    __shared__ nvx_u32x4 stage[LIN_WAVES * 16 * 32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int quad = lane >> 4, l16 = lane & 15;
    const int n_out = n_rows / 2;
    const int n0 = blockIdx.x * 32;
    const int tile = wave * gridDim.z + blockIdx.z;
    const bool has = tile * 16 < m_rows;
    nvx_u32x4* my_stage = stage + wave * (16 * 32);
    nvx_u32x4 b[8];
    auto load_x = [&](int phase, int group) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int row = tile * 16 + 2 * u + (lane >> 5);
            row = row < m_rows ? row : m_rows - 1;
            b[u] = *reinterpret_cast<const nvx_u32x4*>(x + (int64_t)row * x_stride + phase * KP + group * 256 + (lane & 31) * 8);
        }
    };
    if (has) load_x(0, 0);
    nvx_f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = nvx_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int phase = 0; phase < KPH; ++phase) {
        if (phase) __syncthreads();           // every wave is done with the previous half of W
        {
            nvx_u32x4 v[WLOADS];
#pragma unroll
            for (int u = 0; u < WLOADS; ++u) {
                const int p = u * LIN_THREADS + tid;
                const int q = p / PIECES_Q, ph = p % PIECES_Q;
                const int r = ph / PIECES_ROW, kp = ph % PIECES_ROW;
                const int row = n0 + (q & 1) * 16 + (q >> 1) * n_out + r;
                v[u] = *reinterpret_cast<const nvx_u32x4*>(w + (int64_t)row * k_total + phase * KP + kp * 8);
            }
#pragma unroll
            for (int u = 0; u < WLOADS; ++u) {
                const int p = u * LIN_THREADS + tid;
                const int q = p / PIECES_Q, ph = p % PIECES_Q;
                const int r = ph / PIECES_ROW, kp = ph % PIECES_ROW;
                lds[q * STEPS * 64 + lin_slot(2 * (kp >> 3) + (kp & 1), (kp >> 1) & 3, r)] = v[u];
            }
        }
        __syncthreads();
        if (has) {
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = 2 * u + (lane >> 5);
                    my_stage[r * 32 + ((lane & 31) ^ r)] = b[u];
                }
                if (g + 1 < GROUPS) load_x(phase, g + 1);
                else if (phase + 1 < KPH) load_x(phase + 1, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int piece = 8 * (i >> 1) + 2 * quad + (i & 1);
                    const nvx_u32x4 bf = my_stage[l16 * 32 + (piece ^ l16)];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = mfma16(lds[q * STEPS * 64 + lin_slot(g * 8 + i, quad, l16)], bf, acc[q]);
                }
            }
        }
    }
    const int row = tile * 16 + l16;
    if (!has || row >= m_rows) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gate = round_h16(acc[h][r]), up = round_h16(acc[2 + h][r]);
            const float sg = round_h16(gate / (1.0f + __expf(-gate)));
            o[r] = sg * up;
        }
        uint16_t* dst = y + (int64_t)row * n_out + n0 + h * 16 + quad * 4;
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3]));
    }
}

// rows above which the activations go through the LDS staging area (NVX_LIN_STAGED_MIN overrides: tuning)
static int lin_staged_min_rows() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("NVX_LIN_STAGED_MIN"); v = e ? atoi(e) : 48; }
    return v;
}

static int lin_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// Tuning knobs.  NVX_LIN_WIDE_MIN = rows from which modes 0/2 use 32-row workgroups (0 = never; default:
// whenever the activations are staged), NVX_LIN_MSPLIT = workgroups per W slice in that case.  Measured on
// MI355X with the Qwen3-0.6B projections (tools/kernel_bench.py linear, us, 16-row -> 32-row x 2):
// 176 rows: qkv 9.4 -> 7.8, o 6.1 -> 5.2, down 7.7 -> 6.6; 64 rows: 6.1 -> 6.0, 4.2 -> 4.0, 5.1 -> 5.0.
// 32-row workgroups alone are slower (half the workgroups, each with twice the serial work), and dealing
// the X tiles of a 16-row slice to two workgroups is slower still (W is then fetched twice, X as often as
// before): the pair halves the X traffic at the same W traffic and the same number of workgroups.
static int lin_wide_min_rows() { static int v = -1; if (v < 0) v = lin_env_int("NVX_LIN_WIDE_MIN", 1); return v; }
static int lin_msplit() { static int v = -1; if (v < 0) { v = lin_env_int("NVX_LIN_MSPLIT", 2); if (v < 1 || v > 8) v = 2; } return v; }

// rows from which the gate/up mode uses its 64-row kernel: 64 rows 8.2 vs 7.8 us, 123 rows 9.1 vs 9.7,
// 176 rows 10.7 vs 12.2, 256 rows 12.3 vs 14.9 (two passes of W through LDS cost a barrier pair)
static int lin_gateup_wide_min_rows() { static int v = -1; if (v < 0) v = lin_env_int("NVX_LIN_GATEUP_WIDE_MIN", 112); return v; }

static bool lin_chunk_ok(int64_t kc) { return kc == 512 || kc == 768 || kc == 1024; }
// whole-K sizes built for mode 0 only: a tensor-parallel rank's share of K in the row-parallel projections
// (2048 / 8 and 3072 / 2 for the 0.6B geometry), whose 16-bit result goes to the all-reduce
static bool lin_whole_k_ok(int64_t k) { return k == 256 || k == 1536; }

#ifndef NVX_F16   // independent of the 16-bit format: defined by the bfloat16 build only
extern "C" int nvx_linear_decode_splits(int64_t n, int64_t k) {
    // smallest split whose K slice is one of the built sizes, preferring >= 256 workgroups; 0 = none
    int best = 0;
    for (int s = 1; s <= 16; ++s) {
        if (k % s || !lin_chunk_ok(k / s)) continue;
        best = s;
        if ((n / 16) * s >= 256) break;
    }
    return best;
}
#endif

extern "C" int NVX_ENTRY(nvx_linear_decode)(const void* x, int64_t x_row_stride, const void* w, void* y, int64_t m, int64_t n, int64_t k,
                                 int mode, int k_splits, nvx_stream_t stream) {
    NVX_REQUIRE(x && w && y, NVX_EINVAL, "nvx_linear_decode: null pointer");
    NVX_REQUIRE(m > 0 && m <= 256, NVX_EUNSUPPORTED, "nvx_linear_decode: 1..256 rows (got %lld); larger batches are a library GEMM", (long long)m);
    NVX_REQUIRE(mode >= 0 && mode <= 2, NVX_EINVAL, "nvx_linear_decode: mode must be 0, 1 or 2");
    NVX_REQUIRE(k_splits >= 1 && (mode == 2 || k_splits == 1), NVX_EINVAL, "nvx_linear_decode: only mode 2 (fp32 partials) may split K");
    const bool whole_k = mode == 0 && k_splits == 1 && lin_whole_k_ok(k);
    NVX_REQUIRE(k > 0 && k % k_splits == 0 && (lin_chunk_ok(k / k_splits) || whole_k), NVX_EUNSUPPORTED,
                "nvx_linear_decode: K / k_splits must be 512, 768 or 1024, or K 256 or 1536 in mode 0 (got %lld / %d)", (long long)k, k_splits);
    const int64_t n_out = mode == 1 ? n / 2 : n;
    NVX_REQUIRE(n > 0 && n_out % 16 == 0 && (mode != 1 || n % 2 == 0), NVX_EUNSUPPORTED, "nvx_linear_decode: output width must be a multiple of 16");
    NVX_REQUIRE(x_row_stride >= k && x_row_stride % 8 == 0, NVX_EINVAL, "nvx_linear_decode: bad x row stride");
    NVX_REQUIRE(nvx_aligned16(x) && nvx_aligned16(w) && nvx_aligned16(y), NVX_EINVAL, "nvx_linear_decode: pointers must be 16-byte aligned");
    const int kc = (int)(k / k_splits);
    hipStream_t st = (hipStream_t)stream;
    const bool staged = m > lin_staged_min_rows();
    const bool wide = mode != 1 && staged && !whole_k && lin_wide_min_rows() > 0 && m >= lin_wide_min_rows() && n_out % 32 == 0;
    const int msplit = wide && ((n_out / 32) * k_splits) % 8 == 0 ? lin_msplit() : 1;
    const dim3 grid((unsigned)(n_out / (wide ? 32 : 16)), (unsigned)k_splits, (unsigned)msplit);
#define NVX_LIN(MODE, KC, ST, WD) hipLaunchKernelGGL((linear_decode_kernel<MODE, KC, ST, WD>), grid, dim3(LIN_THREADS), 0, st, (const uint16_t*)x, \
                                                     x_row_stride, (const uint16_t*)w, y, (int)m, (int)n, (int)k)
#define NVX_LIN_KC(MODE, ST, WD) do { if (kc == 512) NVX_LIN(MODE, 512, ST, WD); else if (kc == 768) NVX_LIN(MODE, 768, ST, WD); else NVX_LIN(MODE, 1024, ST, WD); } while (0)
#define NVX_LIN_ST(MODE) do { if (staged) NVX_LIN_KC(MODE, true, false); else NVX_LIN_KC(MODE, false, false); } while (0)
    const int64_t tiles = (m + 15) / 16;
    if (whole_k) {
        if (kc == 256) { if (staged) NVX_LIN(0, 256, true, false); else NVX_LIN(0, 256, false, false); }
        else { if (staged) NVX_LIN(0, 1536, true, false); else NVX_LIN(0, 1536, false, false); }
        return nvx_check_launch("nvx_linear_decode");
    }
    if (mode == 1 && staged && lin_gateup_wide_min_rows() > 0 && m >= lin_gateup_wide_min_rows() && kc == 1024 && n_out % 32 == 0 &&
        (n_out / 32) % 8 == 0 && lin_msplit() == 2 && tiles <= 2 * LIN_WAVES) {
        hipLaunchKernelGGL((linear_gateup_wide_kernel<1024>), dim3((unsigned)(n_out / 32), 1, 2), dim3(LIN_THREADS), 0, st,
                           (const uint16_t*)x, x_row_stride, (const uint16_t*)w, (uint16_t*)y, (int)m, (int)n, (int)k);
        return nvx_check_launch("nvx_linear_decode");
    }
    if (wide) { if (mode == 0) NVX_LIN_KC(0, true, true); else NVX_LIN_KC(2, true, true); }
    else if (mode == 0) NVX_LIN_ST(0); else if (mode == 1) NVX_LIN_ST(1); else NVX_LIN_ST(2);
#undef NVX_LIN_ST
#undef NVX_LIN_KC
#undef NVX_LIN
    return nvx_check_launch("nvx_linear_decode");
}

}  // namespace NVX_NS

// Projection of a decode step with tens to hundreds of rows over a weight of tens to hundreds of MB (gfx950):
// y (m, n) = x (m, k) . W^T (n, k), m <= 256, the weight in the packed tile layout of nvx_lmhead_pack.
//
// At these sizes the layer is still a stream of the weight (each byte used once per step), but the rows no longer fit
// the LDS of a compute unit for the whole of k (nvx_linear_stream's way), and a BLAS library's kernels for such shapes
// leave half of the HBM rate unused (Qwen3-8B, 128 rows: o_proj 25 us for 33.5 MB, down_proj 50-67 us for 100 MB).
// Blocking:
//   * a workgroup (4 waves) owns 128 rows of W -- every wave two 16-row tiles -- over ONE slice of k; the grid is
//     (row groups x k-slices), about one workgroup per compute unit;
//   * W travels HBM -> operand registers only (one coalesced 1 KB per wave instruction, the A operand of one
//     v_mfma_f32_16x16x32 step); a wave keeps a ring of four 128-wide k-chunks of its two tiles in flight (32 KB) and
//     refills every register right after the products that read it;
//   * the rows pass through LDS in chunks of 128 k, in B-operand order (pieces XOR-swizzled so that the row-major
//     writes and the fragment reads are both conflict-free), double-buffered: the next chunk is fetched (L2) while the
//     current one is multiplied, one barrier per chunk.  Every fragment read feeds the products of both tiles of the wave,
//     so LDS moves 1/2 m/16 KB per KB of weight;
//   * epilogues: 16-bit store; SiLU(gate) * up (the wave's two tiles are a gate tile and the matching up tile); fp32
//     partials per k-slice, summed by nvx_rmsnorm_splitk or nvx_splitk_sum.
#include "nvx_dev.h"

namespace NVX_NS {

#define LB_NT 2        // 16-row tiles of W per wave
#define LB_CK 4        // k-steps (of 32) per chunk
#define LB_RING 4      // chunks of W in flight per wave
#define LB_MAX_ROWS 256

__device__ __forceinline__ nvx_u32x4 lb_stream16(const nvx_u32x4* p) { return __builtin_nontemporal_load(p); }

// 16-byte slot of the rows' chunk buffer: m-tile mt, k-step kk of the chunk, lane quad q (k offset 8 q), row r of the tile
__device__ __forceinline__ int lb_slot(int mt, int kk, int q, int r) { return (mt * LB_CK + kk) * 64 + q * 16 + (r ^ ((kk << 2) | q)); }

// MODE 0: y 16-bit (m, n).  MODE 1: W = [gate rows; up rows], y 16-bit (m, n / 2) = SiluAndMul of the 16-bit-rounded
// projection.  MODE 2: y fp32 (n_slices, m, n) partial products.
template <int MT, int MODE>
__global__ __launch_bounds__(256, 1) void linear_block_kernel(
    const nvx_u32x4* __restrict__ wp, const uint16_t* __restrict__ x, int64_t x_stride, int m_rows, int n_tiles, int n_cols,
    int k_steps, int n_slices, void* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
    nvx_u32x4* xs = reinterpret_cast<nvx_u32x4*>(lb_smem);      // [2 buffers][MT][LB_CK][64]
    constexpr int BUF = MT * LB_CK * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int quad = lane >> 4, l16 = lane & 15;
    const int slice = blockIdx.x % n_slices, group = blockIdx.x / n_slices;
    const int chunks_total = k_steps / LB_CK;
    const int c0 = (int)((long long)slice * chunks_total / n_slices), c1 = (int)((long long)(slice + 1) * chunks_total / n_slices);
    const int n_chunks = c1 - c0;                                 // a multiple of LB_RING (checked by the host)

    // this wave's two tiles
    int tile[LB_NT];
    if (MODE == 1) { tile[0] = group * 4 + wave; tile[1] = tile[0] + n_tiles / 2; }
    else { tile[0] = (group * 4 + wave) * 2; tile[1] = tile[0] + 1; }
    const int last_tile = n_tiles - 1;
    const bool live0 = MODE == 1 ? tile[0] < n_tiles / 2 : tile[0] < n_tiles;
    const bool live1 = MODE == 1 ? live0 : tile[1] < n_tiles;
    const nvx_u32x4* wt[LB_NT];
#pragma unroll
    for (int t = 0; t < LB_NT; ++t)     // (a tile past the end reads the last tile: loaded, never stored)
        wt[t] = wp + ((int64_t)(tile[t] < n_tiles ? tile[t] : last_tile) * k_steps + (int64_t)c0 * LB_CK) * 64 + lane;

    // ring of W: chunk j of the slice sits in ring slot j % RING (256 rows: the accumulators take half the register file,
    // and the products, not the stream, set the pace: two chunks in flight)
    constexpr int RING = MT > 12 ? 2 : LB_RING;
    nvx_u32x4 a[RING][LB_NT][LB_CK];
#pragma unroll
    for (int r = 0; r < RING; ++r)
#pragma unroll
        for (int t = 0; t < LB_NT; ++t)
#pragma unroll
            for (int kk = 0; kk < LB_CK; ++kk) a[r][t][kk] = lb_stream16(wt[t] + (r * LB_CK + kk) * 64);

    // rows of chunk c -> registers (coalesced: 16 lanes x 16 B = the 256 bytes a row holds of the chunk), -> LDS
    nvx_u32x4 xr[MT];
    auto load_x = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int idx = tid + 256 * i, m = idx >> 4, pc = idx & 15;
            nvx_u32x4 v = {0u, 0u, 0u, 0u};
            if (m < m_rows) v = *reinterpret_cast<const nvx_u32x4*>(x + (int64_t)m * x_stride + (int64_t)(c0 + chunk) * (LB_CK * 32) + pc * 8);
            xr[i] = v;
        }
    };
    auto park_x = [&](int buf) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int idx = tid + 256 * i, m = idx >> 4, pc = idx & 15;
            xs[buf * BUF + lb_slot(m >> 4, pc >> 2, pc & 3, m & 15)] = xr[i];
        }
    };
    load_x(0);
    park_x(0);
    __syncthreads();

    nvx_f32x4 acc[LB_NT][MT];
#pragma unroll
    for (int t = 0; t < LB_NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = nvx_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int cb = 0; cb < n_chunks; cb += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            const int chunk = cb + r;
            const bool more = chunk + 1 < n_chunks;
            if (more) load_x(chunk + 1);
            const nvx_u32x4* xb = xs + (chunk & 1) * BUF;
            // the refill of this ring slot: chunk + RING (past the end: the slice's last chunk again, never used)
            const int nxt = chunk + RING < n_chunks ? chunk + RING : n_chunks - 1;
#pragma unroll
            for (int kk = 0; kk < LB_CK; ++kk) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const nvx_u32x4 b = xb[lb_slot(mt, kk, quad, l16)];
#pragma unroll
                    for (int t = 0; t < LB_NT; ++t) acc[t][mt] = mfma_16x16x32_h16(a[r][t][kk], b, acc[t][mt]);
                }
#pragma unroll
                for (int t = 0; t < LB_NT; ++t) a[r][t][kk] = lb_stream16(wt[t] + (nxt * LB_CK + kk) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) park_x((chunk + 1) & 1);
            __syncthreads();
        }
    }

    // this lane: columns tile * 16 + 4 quad .. + 3 of row mt * 16 + l16
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + l16;
        if (m >= m_rows) continue;
        if (MODE == 1) {
            if (!live0) continue;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gate = round_h16(acc[0][mt][j]), up = round_h16(acc[1][mt][j]);
                const float sg = round_h16(gate / (1.0f + __expf(-gate)));
                o[j] = sg * up;
            }
            uint16_t* dst = reinterpret_cast<uint16_t*>(y) + (int64_t)m * n_cols + tile[0] * 16 + quad * 4;
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3]));
        } else {
#pragma unroll
            for (int t = 0; t < LB_NT; ++t) {
                if (!(t == 0 ? live0 : live1)) continue;
                const int col = tile[t] * 16 + quad * 4;
                if (col >= n_cols) continue;
                if (MODE == 0) {
                    uint16_t* dst = reinterpret_cast<uint16_t*>(y) + (int64_t)m * n_cols + col;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_h16x2(acc[t][mt][0], acc[t][mt][1]), pack_h16x2(acc[t][mt][2], acc[t][mt][3]));
                } else {
                    float* dst = reinterpret_cast<float*>(y) + ((int64_t)slice * m_rows + m) * n_cols + col;
                    *reinterpret_cast<nvx_f32x4*>(dst) = acc[t][mt];
                }
            }
        }
    }
}

static int lb_workgroups() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n = cus;
    }
    return n;
}

#ifndef NVX_F16   // format-independent helpers: defined by the bfloat16 build only
extern "C" int64_t nvx_linear_block_max_rows(void) { return LB_MAX_ROWS; }

// k-slices for a weight of n rows (mode 1: n / 2 gate + n / 2 up rows) over k: the most slices that keep the grid within
// the compute units and every slice a whole number of ring turns; 0 = the shape is not supported.  Modes 0 and 1 store
// final values and take one slice.
extern "C" int nvx_linear_block_slices(int64_t n, int64_t k, int mode) {
    if (n <= 0 || k <= 0 || k % (LB_CK * 32 * LB_RING) != 0 || n % 16 != 0 || (mode == 1 && n % 32 != 0)) return 0;
    if (mode != 2) return 1;
    const int64_t tiles = n / 16, groups = (tiles + 4 * LB_NT - 1) / (4 * LB_NT);
    const int64_t turns = k / (LB_CK * 32 * LB_RING);
    int best = 1;
    for (int s = 1; s <= 64; ++s)
        if (turns % s == 0 && groups * s <= lb_workgroups()) best = s;
    return best;
}
#endif

extern "C" int NVX_ENTRY(nvx_linear_block)(const void* w_packed, int64_t n, int64_t k, const void* x, int64_t x_row_stride, int64_t m,
                                           int mode, int k_slices, void* y, nvx_stream_t stream) {
    const char* who = "nvx_linear_block";
    NVX_REQUIRE(m >= 0 && n > 0 && k > 0, NVX_EINVAL, "%s: bad size", who);
    if (m == 0) return NVX_OK;
    NVX_REQUIRE(w_packed && x && y, NVX_EINVAL, "%s: null pointer", who);
    NVX_REQUIRE(mode >= 0 && mode <= 2, NVX_EINVAL, "%s: mode must be 0, 1 or 2", who);
    NVX_REQUIRE(m <= LB_MAX_ROWS, NVX_ENOSPACE, "%s: %lld rows (at most %d)", who, (long long)m, LB_MAX_ROWS);
    NVX_REQUIRE(n % 16 == 0 && (mode != 1 || n % 32 == 0), NVX_EUNSUPPORTED, "%s: n must be a multiple of 16 (32 for gate/up), got %lld", who, (long long)n);
    NVX_REQUIRE(k_slices >= 1 && (mode == 2 || k_slices == 1), NVX_EINVAL, "%s: only mode 2 (fp32 partials) may slice k", who);
    NVX_REQUIRE(k % ((int64_t)LB_CK * 32 * LB_RING * k_slices) == 0 && k <= (1 << 20), NVX_EUNSUPPORTED,
                "%s: k must be a multiple of %d x k_slices (got %lld / %d)", who, LB_CK * 32 * LB_RING, (long long)k, k_slices);
    NVX_REQUIRE(x_row_stride % 8 == 0 && x_row_stride >= k && nvx_aligned16(x) && nvx_aligned16(w_packed) && nvx_aligned16(y), NVX_EINVAL,
                "%s: rows must be 16-byte aligned", who);
    const int n_tiles = (int)(n / 16);
    const int units = mode == 1 ? n_tiles / 2 : (n_tiles + LB_NT - 1) / LB_NT;     // wave-sized pieces of W
    const int groups = (units + 3) / 4;
    const int grid = groups * k_slices;
    const int n_cols = (int)(mode == 1 ? n / 2 : n);
    hipStream_t st = (hipStream_t)stream;
#define NVX_LB(MTV, MD)                                                                                                               \
    do {                                                                                                                              \
        const size_t lds = (size_t)2 * MTV * LB_CK * 1024;                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_block_kernel<MTV, MD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((linear_block_kernel<MTV, MD>), dim3((unsigned)grid), dim3(256), lds, st, (const nvx_u32x4*)w_packed,      \
                           (const uint16_t*)x, x_row_stride, (int)m, n_tiles, n_cols, (int)(k / 32), k_slices, y);                    \
    } while (0)
#define NVX_LB_MT(MD)                                                                                                                 \
    do {                                                                                                                              \
        switch ((int)((m + 31) / 32)) {                                                                                               \
            case 1: NVX_LB(2, MD); break;                                                                                             \
            case 2: NVX_LB(4, MD); break;                                                                                             \
            case 3: NVX_LB(6, MD); break;                                                                                             \
            case 4: NVX_LB(8, MD); break;                                                                                             \
            case 5: case 6: NVX_LB(12, MD); break;                                                                                    \
            default: NVX_LB(16, MD); break;                                                                                           \
        }                                                                                                                             \
    } while (0)
    if (mode == 0) NVX_LB_MT(0); else if (mode == 1) NVX_LB_MT(1); else NVX_LB_MT(2);
#undef NVX_LB_MT
#undef NVX_LB
    return nvx_check_launch(who);
}

}  // namespace NVX_NS

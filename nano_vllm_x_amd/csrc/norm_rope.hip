// RMSNorm, rotary embedding, fused qk-norm + rotary + KV scatter, SiluAndMul.
// All four are HBM-bound streaming kernels: 16-byte loads/stores per lane, fp32 math in registers,
// wave64 shuffles for the row reductions, no LDS.
#include "nvx_dev.h"
#include "rope_dev.h"

#define NVX_RMSNORM_ROW_KERNEL_MAX_ROWS 1024

namespace NVX_NS {

// ---------------------------------------------------------------------------------------------
// RMSNorm: one wave per row, the row stays in registers between the two passes.
// ---------------------------------------------------------------------------------------------
template <int ITERS, bool HAS_RES, bool W_F32>
__global__ __launch_bounds__(256) void rmsnorm_kernel(
    const uint16_t* __restrict__ x, int64_t x_stride, const uint16_t* residual, const void* __restrict__ weight,
    uint16_t* y, uint16_t* residual_out, int64_t rows, int hidden, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* xr = x + row * x_stride;
    float v[ITERS][8];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int col = it * 512 + lane * 8;
        if (col < hidden) {
            const uint4 a = *reinterpret_cast<const uint4*>(xr + col);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
            if (HAS_RES) {
                const uint4 r = *reinterpret_cast<const uint4*>(residual + row * hidden + col);
                const uint32_t rw[4] = {r.x, r.y, r.z, r.w};
                uint32_t ow[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = h16lo_to_f32(aw[j]) + h16lo_to_f32(rw[j]);
                    const float hi = h16hi_to_f32(aw[j]) + h16hi_to_f32(rw[j]);
                    v[it][2 * j] = lo;
                    v[it][2 * j + 1] = hi;
                    ow[j] = pack_h16x2(lo, hi);
                }
                *reinterpret_cast<uint4*>(residual_out + row * hidden + col) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[it][2 * j] = h16lo_to_f32(aw[j]);
                    v[it][2 * j + 1] = h16hi_to_f32(aw[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += v[it][j] * v[it][j];
        }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int col = it * 512 + lane * 8;
        if (col < hidden) {
            float w[8];
            if (W_F32) {
                const float4 w0 = *reinterpret_cast<const float4*>((const float*)weight + col);
                const float4 w1 = *reinterpret_cast<const float4*>((const float*)weight + col + 4);
                w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w;
                w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
            } else {
                const uint4 wv = *reinterpret_cast<const uint4*>((const uint16_t*)weight + col);
                const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) { w[2 * j] = h16lo_to_f32(ww[j]); w[2 * j + 1] = h16hi_to_f32(ww[j]); }
            }
            uint32_t ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ow[j] = pack_h16x2((v[it][2 * j] * inv) * w[2 * j], (v[it][2 * j + 1] * inv) * w[2 * j + 1]);
            *reinterpret_cast<uint4*>(y + row * hidden + col) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
    }
}

// The same norm for FEW, WIDE rows (a decode batch at hidden sizes of 2048 and more): one 256-thread workgroup per row,
// thread t owns elements 8t .. 8t+7 of every 2048-wide slab, every load of the row (x, residual, weight) is issued before
// the first use, the sum of squares crosses the waves through LDS.  With one WAVE per row (above) a 4096-wide row is eight
// dependent round trips on a handful of CUs: 9.5 us at 29 rows where this takes a third of that.  Per-element arithmetic
// is the kernel's above; only the order in which the squares are summed differs.
template <int SLABS, bool HAS_RES, bool W_F32>
__global__ __launch_bounds__(256) void rmsnorm_row_kernel(
    const uint16_t* __restrict__ x, int64_t x_stride, const uint16_t* residual, const void* __restrict__ weight,
    uint16_t* y, uint16_t* residual_out, int hidden, float eps) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const uint16_t* xr = x + row * x_stride;
    uint4 a[SLABS], r[SLABS], wv[SLABS];
    float4 wf[SLABS][2];
#pragma unroll
    for (int it = 0; it < SLABS; ++it) {
        const int col = it * 2048 + tid * 8;
        if (col < hidden) {
            a[it] = *reinterpret_cast<const uint4*>(xr + col);
            if (HAS_RES) r[it] = *reinterpret_cast<const uint4*>(residual + row * hidden + col);
            if (W_F32) {
                wf[it][0] = *reinterpret_cast<const float4*>((const float*)weight + col);
                wf[it][1] = *reinterpret_cast<const float4*>((const float*)weight + col + 4);
            } else {
                wv[it] = *reinterpret_cast<const uint4*>((const uint16_t*)weight + col);
            }
        }
    }
    float v[SLABS][8];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < SLABS; ++it) {
        const int col = it * 2048 + tid * 8;
        if (col < hidden) {
            const uint32_t aw[4] = {a[it].x, a[it].y, a[it].z, a[it].w};
            if (HAS_RES) {
                const uint32_t rw[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
                uint32_t ow[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = h16lo_to_f32(aw[j]) + h16lo_to_f32(rw[j]);
                    const float hi = h16hi_to_f32(aw[j]) + h16hi_to_f32(rw[j]);
                    v[it][2 * j] = lo;
                    v[it][2 * j + 1] = hi;
                    ow[j] = pack_h16x2(lo, hi);
                }
                *reinterpret_cast<uint4*>(residual_out + row * hidden + col) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[it][2 * j] = h16lo_to_f32(aw[j]); v[it][2 * j + 1] = h16hi_to_f32(aw[j]); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += v[it][j] * v[it][j];
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
#pragma unroll
    for (int it = 0; it < SLABS; ++it) {
        const int col = it * 2048 + tid * 8;
        if (col < hidden) {
            float w[8];
            if (W_F32) {
                w[0] = wf[it][0].x; w[1] = wf[it][0].y; w[2] = wf[it][0].z; w[3] = wf[it][0].w;
                w[4] = wf[it][1].x; w[5] = wf[it][1].y; w[6] = wf[it][1].z; w[7] = wf[it][1].w;
            } else {
                const uint32_t ww[4] = {wv[it].x, wv[it].y, wv[it].z, wv[it].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) { w[2 * j] = h16lo_to_f32(ww[j]); w[2 * j + 1] = h16hi_to_f32(ww[j]); }
            }
            uint32_t ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ow[j] = pack_h16x2((v[it][2 * j] * inv) * w[2 * j], (v[it][2 * j + 1] * inv) * w[2 * j + 1]);
            *reinterpret_cast<uint4*>(y + row * hidden + col) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
    }
}

template <int SLABS>
static void launch_rmsnorm_rows(const void* x, int64_t xs, const void* res, const void* w, int wf32, void* y, void* ro,
                                int64_t rows, int hidden, float eps, hipStream_t st) {
    const dim3 grid((unsigned)rows), block(256);
#define NVX_RMSR(HR, WF) hipLaunchKernelGGL((rmsnorm_row_kernel<SLABS, HR, WF>), grid, block, 0, st, (const uint16_t*)x, xs, \
        (const uint16_t*)res, w, (uint16_t*)y, (uint16_t*)ro, hidden, eps)
    if (res) { if (wf32) NVX_RMSR(true, true); else NVX_RMSR(true, false); }
    else     { if (wf32) NVX_RMSR(false, true); else NVX_RMSR(false, false); }
#undef NVX_RMSR
}

template <int ITERS>
static void launch_rmsnorm(const void* x, int64_t xs, const void* res, const void* w, int wf32, void* y, void* ro,
                           int64_t rows, int hidden, float eps, hipStream_t st) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define NVX_RMS(HR, WF) hipLaunchKernelGGL((rmsnorm_kernel<ITERS, HR, WF>), grid, block, 0, st, (const uint16_t*)x, xs, \
        (const uint16_t*)res, w, (uint16_t*)y, (uint16_t*)ro, rows, hidden, eps)
    if (res) { if (wf32) NVX_RMS(true, true); else NVX_RMS(true, false); }
    else     { if (wf32) NVX_RMS(false, true); else NVX_RMS(false, false); }
#undef NVX_RMS
}

extern "C" int NVX_ENTRY(nvx_rmsnorm)(const void* x, int64_t x_row_stride, const void* residual, const void* weight,
                           int weight_is_f32, void* y, void* residual_out, int64_t rows, int64_t hidden,
                           float eps, nvx_stream_t stream) {
    NVX_REQUIRE(rows >= 0 && hidden > 0, NVX_EINVAL, "nvx_rmsnorm: bad size");
    if (rows == 0) return NVX_OK;
    NVX_REQUIRE(x && weight && y, NVX_EINVAL, "nvx_rmsnorm: null pointer");
    NVX_REQUIRE(!residual || residual_out, NVX_EINVAL, "nvx_rmsnorm: residual without residual_out");
    NVX_REQUIRE(hidden % 8 == 0 && hidden <= 8192 && x_row_stride % 8 == 0 && x_row_stride >= hidden,
                NVX_EUNSUPPORTED, "nvx_rmsnorm: hidden must be a multiple of 8, <= 8192 (got %lld)", (long long)hidden);
    NVX_REQUIRE(nvx_aligned16(x) && nvx_aligned16(y) && nvx_aligned16(weight) &&
                (!residual || (nvx_aligned16(residual) && nvx_aligned16(residual_out))),
                NVX_EINVAL, "nvx_rmsnorm: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int h = (int)hidden;
    // few wide rows (decode batches of the larger models): a workgroup per row
    if (h > 1024 && rows <= NVX_RMSNORM_ROW_KERNEL_MAX_ROWS) {
        if (h <= 2048) launch_rmsnorm_rows<1>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
        else if (h <= 4096) launch_rmsnorm_rows<2>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
        else if (h <= 6144) launch_rmsnorm_rows<3>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
        else launch_rmsnorm_rows<4>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
        return nvx_check_launch("nvx_rmsnorm");
    }
    if (h <= 512) launch_rmsnorm<1>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
    else if (h <= 1024) launch_rmsnorm<2>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
    else if (h <= 2048) launch_rmsnorm<4>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
    else if (h <= 4096) launch_rmsnorm<8>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
    else if (h <= 5120) launch_rmsnorm<10>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
    else launch_rmsnorm<16>(x, x_row_stride, residual, weight, weight_is_f32, y, residual_out, rows, h, eps, st);
    return nvx_check_launch("nvx_rmsnorm");
}

// RMSNorm whose input is the set of fp32 partial products a K-split linear layer left behind
// (nvx_linear_decode mode 2): x = bf16(sum over splits), i.e. what the unsplit projection would have
// stored, then the same arithmetic as rmsnorm_kernel.  Decode batches have few rows, so a whole
// 256-thread workgroup takes one row (one wave per row would leave most of the chip idle and make
// every lane walk splits x row/512 loads): thread t owns elements 4t .. 4t+3 of every 1024-wide
// slab, all loads are issued at once, the sum of squares crosses the waves through LDS.
template <int SLABS, bool HAS_RES, int NT>
__global__ __launch_bounds__(NT) void rmsnorm_splitk_kernel(
    const float* __restrict__ partials, int n_splits, const uint16_t* residual, const uint16_t* __restrict__ weight,
    uint16_t* y, uint16_t* residual_out, int rows, int hidden, float eps) {
    __shared__ float red[NT / 64];
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const int64_t split_stride = (int64_t)rows * hidden;
    float v[SLABS][4];
    uint2 wv[SLABS];
    float ss = 0.f;
    // The kernel is a chain of memory round trips and little else, so every load of a slab (up to four or
    // eight splits at a time, the residual, the weight) is issued before the first use.  With a plain loop over
    // the splits the compiler waited for each load before issuing the next: six round trips in a row.
    // Wide rows (hidden > 1024) get 1024 threads, so that a 4096-wide row is ONE slab: with 256 threads its four
    // slabs were walked one after the other, 4 x ceil(splits / 4) round trips (8 us for 8 splits where this takes 4).
#pragma unroll
    for (int it = 0; it < SLABS; ++it) {
        const int col = it * (4 * NT) + tid * 4;
        if (col < hidden) {
            const float* p = partials + (int64_t)row * hidden + col;
            uint2 r = make_uint2(0u, 0u);
            if (HAS_RES) r = *reinterpret_cast<const uint2*>(residual + (int64_t)row * hidden + col);
            wv[it] = *reinterpret_cast<const uint2*>(weight + col);
            nvx_f32x4 a = {0.f, 0.f, 0.f, 0.f};
            constexpr int UNR = NT == 256 ? 4 : 8;   // splits in flight (narrow rows come from projections cut into <= 4)
            for (int s0 = 0; s0 < n_splits; s0 += UNR) {
                nvx_f32x4 b[UNR];
#pragma unroll
                for (int j = 0; j < UNR; ++j) {   // splits past the end re-read split 0 (cached) and are not added
                    const int sidx = s0 + j < n_splits ? s0 + j : 0;
                    b[j] = *reinterpret_cast<const nvx_f32x4*>(p + sidx * split_stride);
                }
#pragma unroll
                for (int j = 0; j < UNR; ++j) {   // same order of additions as a loop over the splits
                    if (s0 + j == 0) a = b[j];
                    else if (s0 + j < n_splits) a += b[j];
                }
            }
            if (HAS_RES) {
                v[it][0] = round_h16(a[0]) + h16lo_to_f32(r.x);
                v[it][1] = round_h16(a[1]) + h16hi_to_f32(r.x);
                v[it][2] = round_h16(a[2]) + h16lo_to_f32(r.y);
                v[it][3] = round_h16(a[3]) + h16hi_to_f32(r.y);
                *reinterpret_cast<uint2*>(residual_out + (int64_t)row * hidden + col) =
                    make_uint2(pack_h16x2(v[it][0], v[it][1]), pack_h16x2(v[it][2], v[it][3]));
            } else {
                v[it][0] = round_h16(a[0]); v[it][1] = round_h16(a[1]); v[it][2] = round_h16(a[2]); v[it][3] = round_h16(a[3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) ss += v[it][j] * v[it][j];
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) ss += red[w];
    const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
#pragma unroll
    for (int it = 0; it < SLABS; ++it) {
        const int col = it * (4 * NT) + tid * 4;
        if (col < hidden) {
            *reinterpret_cast<uint2*>(y + (int64_t)row * hidden + col) =
                make_uint2(pack_h16x2((v[it][0] * inv) * h16lo_to_f32(wv[it].x), (v[it][1] * inv) * h16hi_to_f32(wv[it].x)),
                           pack_h16x2((v[it][2] * inv) * h16lo_to_f32(wv[it].y), (v[it][3] * inv) * h16hi_to_f32(wv[it].y)));
        }
    }
}

template <int SLABS, int NT>
static void launch_rmsnorm_splitk(const float* p, int s, const void* res, const void* w, void* y, void* ro, int rows, int hidden,
                                  float eps, hipStream_t st) {
    const dim3 grid((unsigned)rows), block(NT);
    if (res) hipLaunchKernelGGL((rmsnorm_splitk_kernel<SLABS, true, NT>), grid, block, 0, st, p, s, (const uint16_t*)res, (const uint16_t*)w,
                                (uint16_t*)y, (uint16_t*)ro, rows, hidden, eps);
    else hipLaunchKernelGGL((rmsnorm_splitk_kernel<SLABS, false, NT>), grid, block, 0, st, p, s, (const uint16_t*)res, (const uint16_t*)w,
                            (uint16_t*)y, (uint16_t*)ro, rows, hidden, eps);
}

extern "C" int NVX_ENTRY(nvx_rmsnorm_splitk)(const float* partials, int n_splits, const void* residual, const void* weight, void* y,
                                  void* residual_out, int64_t rows, int64_t hidden, float eps, nvx_stream_t stream) {
    NVX_REQUIRE(rows >= 0 && rows <= 65536 && hidden > 0 && n_splits >= 1 && n_splits <= 64, NVX_EINVAL, "nvx_rmsnorm_splitk: bad size");
    if (rows == 0) return NVX_OK;
    NVX_REQUIRE(partials && weight && y, NVX_EINVAL, "nvx_rmsnorm_splitk: null pointer");
    NVX_REQUIRE(!residual || residual_out, NVX_EINVAL, "nvx_rmsnorm_splitk: residual without residual_out");
    NVX_REQUIRE(hidden % 4 == 0 && hidden <= 8192, NVX_EUNSUPPORTED, "nvx_rmsnorm_splitk: hidden must be a multiple of 4, <= 8192 (got %lld)",
                (long long)hidden);
    NVX_REQUIRE(nvx_aligned16(partials) && nvx_aligned16(y) && nvx_aligned16(weight) &&
                (!residual || (nvx_aligned16(residual) && nvx_aligned16(residual_out))),
                NVX_EINVAL, "nvx_rmsnorm_splitk: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int h = (int)hidden, r = (int)rows;
    if (h <= 1024) launch_rmsnorm_splitk<1, 256>(partials, n_splits, residual, weight, y, residual_out, r, h, eps, st);
    else if (h <= 2048) launch_rmsnorm_splitk<1, 512>(partials, n_splits, residual, weight, y, residual_out, r, h, eps, st);
    else if (h <= 4096) launch_rmsnorm_splitk<1, 1024>(partials, n_splits, residual, weight, y, residual_out, r, h, eps, st);
    else launch_rmsnorm_splitk<2, 1024>(partials, n_splits, residual, weight, y, residual_out, r, h, eps, st);
    return nvx_check_launch("nvx_rmsnorm_splitk");
}

__global__ __launch_bounds__(256) void rope_kernel(
    const int64_t* __restrict__ positions, const float* __restrict__ cos_sin, int64_t max_pos,
    const uint16_t* q, int64_t q_stride, const uint16_t* k, int64_t k_stride, uint16_t* q_out, uint16_t* k_out,
    int64_t n_tokens, int hq, int hkv) {
    const int l16 = threadIdx.x & 15;
    const int heads = hq + hkv;
    const int64_t rowid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (rowid >= n_tokens * heads) return;
    const int64_t tok = rowid / heads;
    const int h = (int)(rowid % heads);
    const bool is_q = h < hq;
    const uint16_t* src = is_q ? q + tok * q_stride + h * 128 : k + tok * k_stride + (h - hq) * 128;
    uint16_t* dst = is_q ? q_out + (tok * hq + h) * 128 : k_out + (tok * hkv + (h - hq)) * 128;
    const HalfQuad x = load_half_quads(src, l16);
    const float* cs = cos_sin + clamp_pos(positions[tok], max_pos) * 128;
    const uint4 o = rotate_interleave(x, cs, l16);
    *reinterpret_cast<uint4*>(dst + 8 * l16) = o;
}

extern "C" int NVX_ENTRY(nvx_rope)(const int64_t* positions, const float* cos_sin, int64_t max_pos, const void* q,
                        int64_t q_row_stride, const void* k, int64_t k_row_stride, void* q_out, void* k_out,
                        int64_t n_tokens, int n_q_heads, int n_kv_heads, nvx_stream_t stream) {
    NVX_REQUIRE(n_tokens >= 0 && n_q_heads >= 0 && n_kv_heads >= 0 && max_pos > 0, NVX_EINVAL, "nvx_rope: bad size");
    if (n_tokens == 0 || n_q_heads + n_kv_heads == 0) return NVX_OK;
    NVX_REQUIRE(positions && cos_sin, NVX_EINVAL, "nvx_rope: null pointer");
    NVX_REQUIRE((n_q_heads == 0 || (q && q_out)) && (n_kv_heads == 0 || (k && k_out)), NVX_EINVAL, "nvx_rope: null pointer");
    NVX_REQUIRE(q_row_stride % 8 == 0 && k_row_stride % 8 == 0, NVX_EUNSUPPORTED, "nvx_rope: row strides must be multiples of 8");
    NVX_REQUIRE(nvx_aligned16(q) && nvx_aligned16(k) && nvx_aligned16(q_out) && nvx_aligned16(k_out) && nvx_aligned16(cos_sin),
                NVX_EINVAL, "nvx_rope: pointers must be 16-byte aligned");
    const int64_t rows = n_tokens * (n_q_heads + n_kv_heads);
    const unsigned blocks = (unsigned)((rows + 15) / 16);
    hipLaunchKernelGGL(rope_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, positions, cos_sin, max_pos,
                       (const uint16_t*)q, q_row_stride, (const uint16_t*)k, k_row_stride, (uint16_t*)q_out,
                       (uint16_t*)k_out, n_tokens, n_q_heads, n_kv_heads);
    return nvx_check_launch("nvx_rope");
}

// ---------------------------------------------------------------------------------------------
// Fused: per-head RMSNorm of q and k, rotary, q -> contiguous buffer, k/v -> paged cache rows.
// ---------------------------------------------------------------------------------------------
#define QRS_ROWS 2   // (token, head) rows per 16-lane group, all their loads issued before the first is used (measured: 1 row 654 us, 2 rows 564, 4 rows 608, 8 rows 701 for the bench prefill)
__global__ __launch_bounds__(256) void qknorm_rope_store_kernel(
    const uint16_t* __restrict__ qkv, int64_t qkv_stride, const int64_t* __restrict__ positions,
    const float* __restrict__ cos_sin, int64_t max_pos, const uint16_t* __restrict__ qw, const uint16_t* __restrict__ kw,
    float eps, const int64_t* __restrict__ slots, uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_cache,
    uint16_t* __restrict__ v_cache, int64_t n_slots, int64_t n_tokens, int hq, int hkv) {
    const int l16 = threadIdx.x & 15;
    const int heads = hq + 2 * hkv;
    const int64_t n_rows = n_tokens * heads;
    const int64_t row0 = ((int64_t)blockIdx.x * 16 + (threadIdx.x >> 4)) * QRS_ROWS;   // consecutive rows: mostly one token
    if (row0 >= n_rows) return;
    HalfQuad x[QRS_ROWS];
    int64_t tok[QRS_ROWS], slot[QRS_ROWS], pos[QRS_ROWS];
    int h[QRS_ROWS];
#pragma unroll
    for (int i = 0; i < QRS_ROWS; ++i) {
        const int64_t rowid = row0 + i < n_rows ? row0 + i : n_rows - 1;   // rows past the end repeat the last one and are not stored
        tok[i] = rowid / heads;
        h[i] = (int)(rowid - tok[i] * heads);
        x[i] = load_half_quads(qkv + tok[i] * qkv_stride + (int64_t)h[i] * 128, l16);
        slot[i] = slots[tok[i]];
        pos[i] = positions[tok[i]];
    }
#pragma unroll
    for (int i = 0; i < QRS_ROWS; ++i) {
        // the row's sum of squares is computed for every row kind so that the shuffles stay uniform
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += x[i].a[j] * x[i].a[j] + x[i].b[j] * x[i].b[j];
        ss = quarter_sum(ss);
        if (row0 + i >= n_rows) continue;
        const bool slot_ok = slot[i] >= 0 && slot[i] < n_slots;
        if (h[i] >= hq + hkv) {  // v: plain copy into the slot row
            if (!slot_ok) continue;
            uint16_t* dst = v_cache + (slot[i] * hkv + (h[i] - hq - hkv)) * 128;
            *reinterpret_cast<uint2*>(dst + 4 * l16) = make_uint2(pack_h16x2(x[i].a[0], x[i].a[1]), pack_h16x2(x[i].a[2], x[i].a[3]));
            *reinterpret_cast<uint2*>(dst + 64 + 4 * l16) = make_uint2(pack_h16x2(x[i].b[0], x[i].b[1]), pack_h16x2(x[i].b[2], x[i].b[3]));
            continue;
        }
        const bool is_q = h[i] < hq;
        if (!is_q && !slot_ok) continue;
        const float* cs = cos_sin + clamp_pos(pos[i], max_pos) * 128;
        const uint4 o = norm_rotate_interleave(x[i], ss, is_q ? qw : kw, eps, cs, l16);
        uint16_t* dst = is_q ? q_out + (tok[i] * hq + h[i]) * 128 : k_cache + (slot[i] * hkv + (h[i] - hq)) * 128;
        *reinterpret_cast<uint4*>(dst + 8 * l16) = o;
    }
}

extern "C" int NVX_ENTRY(nvx_qknorm_rope_store)(const void* qkv, int64_t qkv_row_stride, const int64_t* positions,
                                     const float* cos_sin, int64_t max_pos, const void* q_norm_weight,
                                     const void* k_norm_weight, float eps, const int64_t* slots, void* q_out,
                                     void* k_cache, void* v_cache, int64_t n_slots, int64_t n_tokens,
                                     int n_q_heads, int n_kv_heads, nvx_stream_t stream) {
    NVX_REQUIRE(n_tokens >= 0 && n_q_heads > 0 && n_kv_heads > 0 && max_pos > 0 && n_slots >= 0, NVX_EINVAL,
                "nvx_qknorm_rope_store: bad size");
    if (n_tokens == 0) return NVX_OK;
    NVX_REQUIRE(qkv && positions && cos_sin && q_norm_weight && k_norm_weight && slots && q_out && k_cache && v_cache,
                NVX_EINVAL, "nvx_qknorm_rope_store: null pointer");
    NVX_REQUIRE(qkv_row_stride % 8 == 0 && qkv_row_stride >= (int64_t)(n_q_heads + 2 * n_kv_heads) * 128, NVX_EINVAL,
                "nvx_qknorm_rope_store: bad qkv row stride");
    NVX_REQUIRE(nvx_aligned16(qkv) && nvx_aligned16(q_out) && nvx_aligned16(k_cache) && nvx_aligned16(v_cache) &&
                nvx_aligned16(cos_sin) && nvx_aligned16(q_norm_weight) && nvx_aligned16(k_norm_weight),
                NVX_EINVAL, "nvx_qknorm_rope_store: pointers must be 16-byte aligned");
    const int64_t rows = n_tokens * (n_q_heads + 2 * n_kv_heads);
    const unsigned blocks = (unsigned)((rows + 16 * QRS_ROWS - 1) / (16 * QRS_ROWS));
    hipLaunchKernelGGL(qknorm_rope_store_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv,
                       qkv_row_stride, positions, cos_sin, max_pos, (const uint16_t*)q_norm_weight,
                       (const uint16_t*)k_norm_weight, eps, slots, (uint16_t*)q_out, (uint16_t*)k_cache,
                       (uint16_t*)v_cache, n_slots, n_tokens, n_q_heads, n_kv_heads);
    return nvx_check_launch("nvx_qknorm_rope_store");
}

// ---------------------------------------------------------------------------------------------
// SiluAndMul
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_h16(float x) { return round_h16(x / (1.0f + __expf(-x))); }

// IDX: the type the element index is counted in.  A 64-bit division per element costs about a hundred instructions;
// every decode batch and most prefill batches fit 32 bits.
template <typename IDX>
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                       int64_t rows, int d8) {
    const IDX total = (IDX)(rows * d8);
    for (IDX id = (IDX)blockIdx.x * 256 + threadIdx.x; id < total; id += (IDX)gridDim.x * 256) {
        const int64_t r = (int64_t)(id / (IDX)d8);
        const int c = (int)(id % (IDX)d8);
        const uint4 g = *reinterpret_cast<const uint4*>(x + (r * 2 * d8 + c) * 8);
        const uint4 u = *reinterpret_cast<const uint4*>(x + (r * 2 * d8 + d8 + c) * 8);
        const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, uw[4] = {u.x, u.y, u.z, u.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_h16x2(silu_h16(h16lo_to_f32(gw[j])) * h16lo_to_f32(uw[j]),
                               silu_h16(h16hi_to_f32(gw[j])) * h16hi_to_f32(uw[j]));
        *reinterpret_cast<uint4*>(y + id * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int NVX_ENTRY(nvx_silu_mul)(const void* x, void* y, int64_t rows, int64_t d, nvx_stream_t stream) {
    NVX_REQUIRE(rows >= 0 && d > 0, NVX_EINVAL, "nvx_silu_mul: bad size");
    if (rows == 0) return NVX_OK;
    NVX_REQUIRE(x && y, NVX_EINVAL, "nvx_silu_mul: null pointer");
    NVX_REQUIRE(d % 8 == 0, NVX_EUNSUPPORTED, "nvx_silu_mul: d must be a multiple of 8");
    NVX_REQUIRE(nvx_aligned16(x) && nvx_aligned16(y), NVX_EINVAL, "nvx_silu_mul: pointers must be 16-byte aligned");
    const int64_t total = rows * (d / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (total + 256 * 256 * 16 < 0x7fffffffll)
        hipLaunchKernelGGL(silu_mul_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)x, (uint16_t*)y, rows, (int)(d / 8));
    else
        hipLaunchKernelGGL(silu_mul_kernel<int64_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)x, (uint16_t*)y, rows, (int)(d / 8));
    return nvx_check_launch("nvx_silu_mul");
}

// ---------------------------------------------------------------------------------------------
// Sum of the fp32 partial products a K-split projection left behind, (splits, rows, n) -> 16-bit (rows, n); with SILU the
// sum is a [gate | up] projection and the result is SiluAndMul of it, (rows, n / 2) -- the sum is rounded to the 16-bit
// format first, as the unsplit projection would have stored it, then the same arithmetic as silu_mul_kernel.
template <bool SILU>
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ partials, int n_splits, int rows, int n,
                                                         uint16_t* __restrict__ y) {
    const int out_cols = SILU ? n / 2 : n;
    const int per_row = out_cols / 4;
    const int64_t split_stride = (int64_t)rows * n;
    for (int id = blockIdx.x * 256 + threadIdx.x; id < rows * per_row; id += gridDim.x * 256) {
        const int r = id / per_row, c = (id % per_row) * 4;
        const float* p = partials + (int64_t)r * n + c;
        nvx_f32x4 g = *reinterpret_cast<const nvx_f32x4*>(p);
        nvx_f32x4 u = {0.f, 0.f, 0.f, 0.f};
        if (SILU) u = *reinterpret_cast<const nvx_f32x4*>(p + out_cols);
        for (int s = 1; s < n_splits; ++s) {
            g += *reinterpret_cast<const nvx_f32x4*>(p + s * split_stride);
            if (SILU) u += *reinterpret_cast<const nvx_f32x4*>(p + s * split_stride + out_cols);
        }
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = SILU ? silu_h16(round_h16(g[j])) * round_h16(u[j]) : g[j];
        *reinterpret_cast<uint2*>(y + (int64_t)r * out_cols + c) = make_uint2(pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3]));
    }
}

extern "C" int NVX_ENTRY(nvx_splitk_sum)(const float* partials, int n_splits, int64_t rows, int64_t n, int silu_mul, void* y,
                                         nvx_stream_t stream) {
    NVX_REQUIRE(rows >= 0 && n > 0 && n_splits >= 1 && n_splits <= 64, NVX_EINVAL, "nvx_splitk_sum: bad size");
    if (rows == 0) return NVX_OK;
    NVX_REQUIRE(partials && y, NVX_EINVAL, "nvx_splitk_sum: null pointer");
    NVX_REQUIRE(n % (silu_mul ? 8 : 4) == 0 && rows * n < 0x7fffffffll, NVX_EUNSUPPORTED,
                "nvx_splitk_sum: n must be a multiple of %d and rows x n fit 31 bits", silu_mul ? 8 : 4);
    NVX_REQUIRE(nvx_aligned16(partials) && (reinterpret_cast<uintptr_t>(y) & 7u) == 0, NVX_EINVAL, "nvx_splitk_sum: alignment");
    const int64_t work = rows * ((silu_mul ? n / 2 : n) / 4);
    int64_t blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (silu_mul)
        hipLaunchKernelGGL(splitk_sum_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partials, n_splits, (int)rows, (int)n, (uint16_t*)y);
    else
        hipLaunchKernelGGL(splitk_sum_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partials, n_splits, (int)rows, (int)n, (uint16_t*)y);
    return nvx_check_launch("nvx_splitk_sum");
}

}  // namespace NVX_NS

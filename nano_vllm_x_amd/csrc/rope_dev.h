// Per-head RMSNorm + rotary embedding on a 128-wide head row held by 16 lanes, shared by the stand-alone
// kernels (norm_rope.hip) and the decode-attention kernel that applies them to its own query and to
// the new key (decode_attn.hip), so that both produce the same bits.
#pragma once
#include "nvx_dev.h"

namespace NVX_NS {

// ---------------------------------------------------------------------------------------------
// Rotary helpers.  A 128-wide head row is handled by 16 lanes; lane l holds elements 4l..4l+3 of
// the first half (x1) and 64+4l..64+4l+3 of the second half (x2) and produces the interleaved
// outputs 8l..8l+7 = (o1_0, o2_0, o1_1, o2_1, ...): one 16-byte store.
// ---------------------------------------------------------------------------------------------
struct HalfQuad { float a[4]; float b[4]; };

__device__ __forceinline__ HalfQuad load_half_quads(const uint16_t* row, int l16) {
    const uint2 p = *reinterpret_cast<const uint2*>(row + 4 * l16);
    const uint2 q = *reinterpret_cast<const uint2*>(row + 64 + 4 * l16);
    HalfQuad h;
    h.a[0] = h16lo_to_f32(p.x); h.a[1] = h16hi_to_f32(p.x); h.a[2] = h16lo_to_f32(p.y); h.a[3] = h16hi_to_f32(p.y);
    h.b[0] = h16lo_to_f32(q.x); h.b[1] = h16hi_to_f32(q.x); h.b[2] = h16lo_to_f32(q.y); h.b[3] = h16hi_to_f32(q.y);
    return h;
}

// Eager bf16 arithmetic: cos/sin rounded to bf16, every product and sum rounded to bf16.
__device__ __forceinline__ uint4 rotate_interleave(const HalfQuad& x, const float* cs_row, int l16) {
    const float4 c4 = *reinterpret_cast<const float4*>(cs_row + 4 * l16);
    const float4 s4 = *reinterpret_cast<const float4*>(cs_row + 64 + 4 * l16);
    const float c[4] = {round_h16(c4.x), round_h16(c4.y), round_h16(c4.z), round_h16(c4.w)};
    const float s[4] = {round_h16(s4.x), round_h16(s4.y), round_h16(s4.z), round_h16(s4.w)};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float o1 = round_h16(x.a[j] * c[j]) - round_h16(x.b[j] * s[j]);
        const float o2 = round_h16(x.b[j] * c[j]) + round_h16(x.a[j] * s[j]);
        o[j] = pack_h16x2(o1, o2);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ int64_t clamp_pos(int64_t p, int64_t max_pos) {
    return p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
}


// RMSNorm over the 128 elements of a head (ss = their sum of squares, already reduced over the 16
// lanes), rounded to bf16 as the unfused norm hands it to the rotary op, then rotated and interleaved.
__device__ __forceinline__ uint4 norm_rotate_interleave(const HalfQuad& x, float ss, const uint16_t* __restrict__ w, float eps,
                                                        const float* cs_row, int l16) {
    const uint2 wa = *reinterpret_cast<const uint2*>(w + 4 * l16);
    const uint2 wb = *reinterpret_cast<const uint2*>(w + 64 + 4 * l16);
    const float wA[4] = {h16lo_to_f32(wa.x), h16hi_to_f32(wa.x), h16lo_to_f32(wa.y), h16hi_to_f32(wa.y)};
    const float wB[4] = {h16lo_to_f32(wb.x), h16hi_to_f32(wb.x), h16lo_to_f32(wb.y), h16hi_to_f32(wb.y)};
    const float inv = 1.0f / sqrtf(ss / 128.0f + eps);
    HalfQuad n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        n.a[j] = round_h16((x.a[j] * inv) * wA[j]);
        n.b[j] = round_h16((x.b[j] * inv) * wB[j]);
    }
    return rotate_interleave(n, cs_row, l16);
}

__device__ __forceinline__ float half_quad_sumsq(const HalfQuad& x) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) ss += x.a[j] * x.a[j] + x.b[j] * x.b[j];
    return quarter_sum(ss);
}

}  // namespace NVX_NS

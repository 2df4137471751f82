// Noise and winner keys of the Gumbel-max draw, shared by the sampler (sampler.hip) and by the LM head that draws as it
// computes its logits (lmhead_sample.hip): both must give the same token for the same logits, seed and step.
#pragma once
#include "nvx_dev.h"

namespace NVX_NS {

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }

__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(M0, ctr.x), lo0 = M0 * ctr.x;
        const uint32_t hi1 = mulhi32(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0; key.y += W1;
    }
    return ctr;
}

// Gumbel noise from 23 random bits.  (r >> 9) + 0.5 is exact in fp32 (24 significant bits), so u lies
// strictly inside (0, 1); with 24 bits the top value rounds to u = 1, -log(u) = 0 and the noise becomes
// +infinity: that token would win whatever its logit.  The inner logarithm is clamped at 2^-25 as well, in
// case the fast logarithm returns 0 just below 1 (the noise is then at most 17.3).
__device__ __forceinline__ float gumbel_from_bits(uint32_t r) {
    const float u = ((float)(r >> 9) + 0.5f) * (1.0f / 8388608.0f);
    return -__logf(fmaxf(-__logf(u), 2.98023224e-8f));
}

// order-preserving map fp32 -> uint32 (ascending)
__device__ __forceinline__ uint32_t order_key(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// (score, token) as one 64-bit key: larger score wins, then the lower token id
__device__ __forceinline__ unsigned long long pack_winner(float v, int idx) {
    return ((unsigned long long)order_key(v) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)idx);
}

}  // namespace NVX_NS

// Shared device/host helpers for the gfx950 kernels of libnvx_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/nvx.h"

#define NVX_WAVE 64
#define NVX_HEAD_DIM 128

// ---- host-side error plumbing -----------------------------------------------------------
extern thread_local char nvx_err_buf[512];
int nvx_fail(int code, const char* fmt, ...);
int nvx_check_launch(const char* what);

#define NVX_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return nvx_fail((code), __VA_ARGS__); } while (0)

static inline bool nvx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- the 16-bit floating-point format of activations, weights and KV rows --------------------------
// Every kernel source is compiled twice: once as is (bfloat16, the format of the Qwen3 checkpoints) and once with
// -DNVX_F16 (IEEE half, the reference's default dtype, nanovllmx/core/model_runner.py:14-23).  What differs between the
// two builds is all here: the conversions, the packed dot product and the MFMA instructions.  Kernels and their static
// state live in namespace NVX_NS so that the two builds of a source can be linked into one library, and an entry point
// `name` of the bfloat16 build is `name_f16` in the other (NVX_ENTRY).
typedef __attribute__((ext_vector_type(4))) float nvx_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int nvx_u32x4;   // a 16-byte move that stays in registers
typedef __attribute__((ext_vector_type(16))) float nvx_f32x16;
typedef __attribute__((ext_vector_type(4))) short nvx_s16x4;

#ifdef NVX_F16
#define NVX_NS nvx_f16
#define NVX_ENTRY(name) name##_f16
#define NVX_FORMAT_NAME "float16"
typedef _Float16 nvx_h16;
#else
#define NVX_NS nvx_bf16
#define NVX_ENTRY(name) name
#define NVX_FORMAT_NAME "bfloat16"
typedef __bf16 nvx_h16;
#endif
typedef __attribute__((ext_vector_type(2))) nvx_h16 nvx_h16x2;
typedef __attribute__((ext_vector_type(4))) nvx_h16 nvx_h16x4;
typedef __attribute__((ext_vector_type(8))) nvx_h16 nvx_h16x8;

namespace NVX_NS {

#ifdef NVX_F16
__device__ __forceinline__ float h16lo_to_f32(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(packed & 0xffffu)); }
__device__ __forceinline__ float h16hi_to_f32(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(packed >> 16)); }
__device__ __forceinline__ float h16bits_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
#else
__device__ __forceinline__ float h16lo_to_f32(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float h16hi_to_f32(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
__device__ __forceinline__ float h16bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
#endif

// Round-to-nearest-even f32 -> 16-bit pattern; a plain cast lowers to the hardware conversion (v_cvt_pk_bf16_f32 /
// v_cvt_f16_f32 on gfx950) and keeps NaNs NaN.
__device__ __forceinline__ uint16_t f32_to_h16bits(float f) {
    nvx_h16 h = (nvx_h16)f;
    return __builtin_bit_cast(uint16_t, h);
}
// two conversions, packed (round to nearest even, as the scalar conversion)
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, nvx_h16x2));
}
// value of f after a round trip through the 16-bit format (what eager arithmetic in that format keeps of every result)
__device__ __forceinline__ float round_h16(float f) { return h16bits_to_f32(f32_to_h16bits(f)); }

// c + a.x * b.x + a.y * b.y on packed pairs, fp32 accumulate (v_dot2c_f32_bf16 / v_dot2_f32_f16)
__device__ __forceinline__ float dot2_h16(uint32_t a, uint32_t b, float c) {
#ifdef NVX_F16
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(nvx_h16x2, a), __builtin_bit_cast(nvx_h16x2, b), c, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(nvx_h16x2, a), __builtin_bit_cast(nvx_h16x2, b), c, false);
#endif
}
// the MFMA shapes the kernels use, on raw 16-byte / 8-byte operand registers
__device__ __forceinline__ nvx_f32x16 mfma_32x32x16_h16(nvx_h16x8 a, nvx_h16x8 b, nvx_f32x16 c) {
#ifdef NVX_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ nvx_f32x4 mfma_16x16x32_h16(nvx_u32x4 a, nvx_u32x4 b, nvx_f32x4 c) {
#ifdef NVX_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(nvx_h16x8, a), __builtin_bit_cast(nvx_h16x8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(nvx_h16x8, a), __builtin_bit_cast(nvx_h16x8, b), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ nvx_f32x4 mfma_16x16x16_h16(nvx_s16x4 a, nvx_s16x4 b, nvx_f32x4 c) {
#ifdef NVX_F16
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(nvx_h16x4, a), __builtin_bit_cast(nvx_h16x4, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
#endif
}

// ---- wave-level reductions (wave = 64 lanes) ------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
// Sum across the 16 lanes of one quarter-wave (a DPP "row": lanes 16g .. 16g+15); every lane ends with
// the row total.  DPP moves ride on the VALU add itself (v_add_f32_dpp); __shfl_xor would compile to
// ds_bpermute_b32, a round trip through the LDS crossbar per step.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// The value of lane ^ 32 (a one-dimensional workgroup whose size is a multiple of 64: lane = threadIdx.x & 63), moved
// in the vector ALU: v_permlane32_swap_b32 exchanges the upper half of one register with the lower half of another;
// __shfl_xor(v, 32) would be a ds_bpermute_b32 round trip through LDS.  Given v in both operands, the first result
// holds the values of lanes 0-31 in both halves and the second those of lanes 32-63 -- or, if the compiler gives both
// operands one register, both results hold the halves exchanged: the selection below is right either way.
__device__ __forceinline__ float xor32(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32u) ? r[0] : r[1]);
}
// The value of lane ^ 16, the same way (v_permlane16_swap_b32 exchanges the odd 16-lane rows of one register with the
// even rows of another).
__device__ __forceinline__ float xor16(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 16u) ? r[0] : r[1]);
}
__device__ __forceinline__ float quarter_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_move<0x141>(v);   // row_half_mirror: quad 0 <-> quad 1, quad 2 <-> quad 3
    v += dpp_move<0x140>(v);   // row_mirror: lower 8 <-> upper 8
    return v;
}

}  // namespace NVX_NS

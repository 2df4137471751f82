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

// ---- bf16 <-> f32 on raw bit patterns -----------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 nvx_bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 nvx_bf16x8;
typedef __attribute__((ext_vector_type(4))) float nvx_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int nvx_u32x4;   // a 16-byte move that stays in registers
typedef __attribute__((ext_vector_type(16))) float nvx_f32x16;

__device__ __forceinline__ float bf16lo_to_f32(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
__device__ __forceinline__ float bf16bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// Round-to-nearest-even f32 -> bf16 bits; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
// and keeps NaNs NaN.
__device__ __forceinline__ uint16_t f32_to_bf16bits(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
// two conversions in one v_cvt_pk_bf16_f32 (round to nearest even, as the scalar conversion)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// value of f after a round trip through bf16 (what eager bf16 arithmetic keeps of every result)
__device__ __forceinline__ float round_bf16(float f) { return bf16bits_to_f32(f32_to_bf16bits(f)); }

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
__device__ __forceinline__ float quarter_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_move<0x141>(v);   // row_half_mirror: quad 0 <-> quad 1, quad 2 <-> quad 3
    v += dpp_move<0x140>(v);   // row_mirror: lower 8 <-> upper 8
    return v;
}

// Shared device/host helpers for libyvhip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "yv_hip.h"

#define YV_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

static inline int yv_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? YV_OK : YV_ERR_LAUNCH;
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even f32 -> bf16 through the hardware convert (NaN stays NaN)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {      // one v_cvt_pk_bf16_f32 (RNE)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

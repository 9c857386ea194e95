// Shared helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/speinet_hip.h"

void spei_set_error(const char* fmt, ...);

#define SPEI_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            spei_set_error(__VA_ARGS__);   \
            return -1;                     \
        }                                  \
    } while (0)

#define SPEI_CHECK_LAUNCH(name)                                                          \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            spei_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
            return -2;                                                                   \
        }                                                                                \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

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

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16-byte register staging type.  NOT HIP's uint4: arrays of that struct type are not scalarised by the compiler and end
// up in scratch memory (measured: 144-160 B/lane of scratch traffic in the staging loops).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

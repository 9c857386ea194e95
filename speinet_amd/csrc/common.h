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

// ---- in-register 4x4 transpose across the 4 lanes of a quad (DPP quad_perm, no LDS) --------------------------------
// Before: lane t of a quad holds a[0..3] = one column (n = lane) of 4 consecutive accumulator rows.
// After : lane t holds row t's 4 consecutive columns  (b[e] on lane t == a[t] on lane e).
// Used by the GEMM epilogues to turn "4 bytes per lane" stores into 16-byte row-segment stores.
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose4(float& a0, float& a1, float& a2, float& a3, int t) {
    const bool b0 = t & 1, b1 = t & 2;
    float s = b0 ? a0 : a1, r = dpp_quad<0xB1>(s);      // quad_perm [1,0,3,2]
    if (b0) a0 = r; else a1 = r;
    s = b0 ? a2 : a3; r = dpp_quad<0xB1>(s);
    if (b0) a2 = r; else a3 = r;
    s = b1 ? a0 : a2; r = dpp_quad<0x4E>(s);            // quad_perm [2,3,0,1]
    if (b1) a0 = r; else a2 = r;
    s = b1 ? a1 : a3; r = dpp_quad<0x4E>(s);
    if (b1) a1 = r; else a3 = r;
}

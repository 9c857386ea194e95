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

// Tuning knobs (tile-shape / ablation switches read by tools/ablate_*.py).  The shipping build has NONE: every knob is
// its compile-time default and the dispatch code folds.  `python -m speinet_amd.build --tuning` defines SPEI_TUNING and
// lets the named environment variable override the default (read once per process).
#ifdef SPEI_TUNING
#include <stdlib.h>
static inline int spei_knob(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
// In-kernel phase stamps (tools/stamp_phases.py): SPEI_STAMP_PTR = device address of a long long[workgroups][16] buffer the
// caller allocated; wave 0 of a workgroup writes s_memtime at each SPEI_STAMP(i).  Compiled out of the shipping build.
static inline long long* spei_stamp_buffer() { const char* v = getenv("SPEI_STAMP_PTR"); return v ? (long long*)strtoull(v, nullptr, 0) : nullptr; }
#define SPEI_STAMP(buf, i)                                                                            \
    do {                                                                                              \
        if ((buf) && threadIdx.x == 0) (buf)[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
// the shader clock (s_memtime: core cycles, per XCD) into slot i: with SPEI_STAMP at the same point it gives the clock a phase ran at
#define SPEI_STAMP_CLK(buf, i)                                                                        \
    do {                                                                                              \
        if ((buf) && threadIdx.x == 0) (buf)[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
// the same from lane 0 of another wave (thread t): kernels whose waves take different roles
#define SPEI_STAMP_AT(buf, i, t)                                                                      \
    do {                                                                                              \
        if ((buf) && threadIdx.x == (t)) (buf)[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
static constexpr int spei_knob(const char*, int dflt) { return dflt; }
static inline long long* spei_stamp_buffer() { return nullptr; }
#define SPEI_STAMP(buf, i) do { } while (0)
#define SPEI_STAMP_CLK(buf, i) do { } while (0)
#define SPEI_STAMP_AT(buf, i, t) do { } while (0)
#endif

// Raise a kernel's dynamic-LDS limit before its first launch with `lds` bytes ON THE CURRENT DEVICE.  The attribute is
// per device: a process that drives several GPUs (the reference's nn.DataParallel calls forward from one thread per
// device) must set it on each.  The KERNEL is the template parameter (a non-type one): one table per kernel
// instantiation, keyed by device — instantiations that share a function-pointer TYPE do not share a table.  The races
// are benign (the call is idempotent).
template <auto Kernel>
inline void ensure_dyn_lds(size_t lds) {
    static size_t have[32] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 32 || lds > have[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 32) have[dev] = lds;
    }
}

// Compute units of the current device (cached per device): grid size of the persistent kernels.
inline int spei_num_cus() {
    static int have[32] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 32) return 256;
    if (have[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        have[dev] = n;
    }
    return have[dev];
}

// Persistent kernels (one workgroup per CU, tile = blockIdx.x + k gridDim.x): the workgroups of a launch start together and do the
// same work per tile, so chip-wide they issue every burst of loads at the same instant — 256 x 100 KB at once takes HBM 5 us to deliver,
// whatever the average rate.  Each workgroup therefore starts `unit` x phase microseconds late, phase 0..3 by its place inside its XCD
// (blockIdx / 8), + 4 for the workgroups that have one tile fewer than the others (they finish early anyway): the bursts of different
// CUs then fall at different times for the whole launch.
__device__ __forceinline__ void spei_stagger_start(int ntiles, int unit) {
    const int G = gridDim.x, b = blockIdx.x;
    const int mine = (ntiles - b + G - 1) / G, most = (ntiles + G - 1) / G;
    const int phase = (mine < most ? 4 : 0) + ((b >> 3) & 3);
    for (int i = 0; i < phase * unit; ++i) __builtin_amdgcn_s_sleep(32);      // 32 x 64 cycles ~ 1 us
}

// ---- cross-lane reductions on the VALU (DPP + gfx950 permlane swaps): no LDS-pipe traffic, unlike __shfl_xor -------
// (ds_bpermute; 12 of them per wave_sum made the LayerNorm / gate-statistics kernels LDS-instruction-bound)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
struct OpSum { static __device__ __forceinline__ float f(float a, float b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); } };
// v <- op(v, value of lane ^ M) for M = 8 (row_ror:8), 16 (v_permlane16_swap), 32 (v_permlane32_swap)
template <int M, typename Op>
__device__ __forceinline__ float xor_combine(float v) {
    if constexpr (M == 8) {
        return Op::f(v, dpp_mov<0x128>(v));
    } else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return Op::f(__uint_as_float(r[0]), __uint_as_float(r[1]));
    } else {
        static_assert(M == 32, "xor_combine: M must be 8, 16 or 32");
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return Op::f(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
}
// all-reduce over the 64 lanes: xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror (the values are already uniform
// inside the quads / octets they pair), then the two swaps
template <typename Op>
__device__ __forceinline__ float wave_allreduce(float v) {
    v = Op::f(v, dpp_mov<0xB1>(v));
    v = Op::f(v, dpp_mov<0x4E>(v));
    v = Op::f(v, dpp_mov<0x141>(v));
    v = Op::f(v, dpp_mov<0x140>(v));
    v = xor_combine<16, Op>(v);
    return xor_combine<32, Op>(v);
}
__device__ __forceinline__ float wave_sum(float v) { return wave_allreduce<OpSum>(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_allreduce<OpMax>(v); }

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS operations (lgkmcnt), NOT for its global loads.
// __syncthreads() carries a workgroup-scope fence, and on gfx950 (one vmcnt for loads and stores) that drains every global load
// the wave has in flight before it reaches the barrier: nothing can be prefetched across it (stamps, round 4: a phase whose loads
// were issued one barrier early was no shorter).  Use where the waves exchange data through LDS only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the two 16-bit operand formats of the matrix pipe -----------------------------------------------------------------
// bf16 (8-bit significand) and IEEE half (11-bit significand) run v_mfma_f32_32x32x16_* at the same rate with fp32
// accumulation; kernels are templates on the element type LP, the C-ABI selects with SPEI_BF16 / SPEI_F16.  Half keeps
// three more bits of every weight and activation (PSNR delta of the 720p forward 8x smaller) at a range of
// +-65504.  Conversions do NOT saturate: an activation beyond that range becomes +-inf and the frame comes out NaN —
// loud, where a clamp would be silently wrong (the harness checks its output for non-finite values).
template <typename T>
struct lpv {
    typedef T x8 __attribute__((ext_vector_type(8)));
    typedef T x4 __attribute__((ext_vector_type(4)));
    typedef T x2 __attribute__((ext_vector_type(2)));
};
__device__ __forceinline__ f32x16 mfma16(lpv<__bf16>::x8 a, lpv<__bf16>::x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(lpv<_Float16>::x8 a, lpv<_Float16>::x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// fp32 -> LP, round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 for pairs)
template <typename LP>
__device__ __forceinline__ LP to_lp(float v) { return (LP)v; }
template <typename LP>
__device__ __forceinline__ typename lpv<LP>::x4 to_lp4(f32x4 v) {
    typename lpv<LP>::x4 h;
    h[0] = to_lp<LP>(v[0]); h[1] = to_lp<LP>(v[1]); h[2] = to_lp<LP>(v[2]); h[3] = to_lp<LP>(v[3]);
    return h;
}
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

// Fused Swin MLP branch on the gfx950 bf16 matrix pipe:
//
//     out = x + fc2( GELU( fc1( LayerNorm(x) ) ) )          (reference model/swinir.py:12-29 Mlp, :279 block tail)
//
// Unfused this is LayerNorm -> GEMM 256->512 + GELU -> GEMM 512->256 + residual: the normalised tokens and the 512-wide
// hidden activations each make a round trip through HBM (59 + 118 MB per call at 720p, 72 calls per frame).  Here a
// 512-thread workgroup owns 64 tokens end to end:
//   1. one wave per token row: fp32 load, two-pass moments by wave shuffles, normalise (the LayerNorm affine is folded
//      into fc1 by pack.py) -> bf16 row in LDS (pitch 2*256+16 B);
//   2. fc1: A fragments from that slab, B fragments streamed from HBM/L2 in MFMA fragment order (as conv_slab_bf16),
//      bias + exact-erf GELU on the accumulator -> bf16 hidden slab in LDS (pitch 2*512+16 B), never in HBM;
//   3. fc2: A fragments from the hidden slab, bias, transposing per-wave LDS tile -> + residual x -> 16-byte stores.
// HBM traffic: x read (twice, the second time from L2 for the residual) and out written: 118 MB instead of 531 MB.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int D = 256, HID = 512, MT = 64;          // model dim, hidden dim, tokens per workgroup
constexpr int PA = 2 * D + 16, PH = 2 * HID + 16;   // LDS row pitches (bytes)
constexpr int EP = 36;                              // epilogue tile pitch (floats)
constexpr int G = 2, RING = 4;

struct MlpParams {
    const float* x;
    float* out;
    const __bf16* w1;     // fragment order [HID/32][1][D/16][64][8]
    const float* b1;
    const __bf16* w2;     // fragment order [D/32][1][HID/16][64][8]
    const float* b2;
    int M;
};

// GELU with erf from Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): ~15 VALU ops instead of ~60 for erff.  Only used
// where the result is rounded to bf16 right away (2^-9 relative), i.e. in the "bf16" arithmetic mode.
__device__ __forceinline__ float gelu_fast(float v) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = poly * t * __expf(-z * z);
    const float erf_abs = 1.0f - e;
    return 0.5f * v * (1.0f + copysignf(erf_abs, v));
}

__global__ __launch_bounds__(512) void mlp_fused_kernel(const MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sa = smem;                               // [MT][PA]   normalised tokens, bf16
    unsigned char* sh = smem + MT * PA;                     // [MT][PH]   hidden activations, bf16
    float* etile = reinterpret_cast<float*>(smem + MT * PA + MT * PH) + (threadIdx.x >> 6) * (32 * EP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave;                                    // 8 column slots; every wave covers both 32-token halves
    const int fr = lane & 31, fk = lane >> 5;
    const int m0 = blockIdx.x * MT;

    // ---- 1. LayerNorm(256) per token row, one wave per row ---------------------------------------------------
    // all 8 rows of this wave are requested before the first reduction (one HBM latency, not eight)
    f32x4 rows[MT / 8];
#pragma unroll
    for (int i = 0; i < MT / 8; ++i) {
        const int m = min(m0 + wave + 8 * i, p.M - 1);
        rows[i] = reinterpret_cast<const f32x4*>(p.x + (size_t)m * D)[lane];
    }
#pragma unroll
    for (int i = 0; i < MT / 8; ++i) {
        const int r = wave + 8 * i;
        const float4 v = make_float4(rows[i][0], rows[i][1], rows[i][2], rows[i][3]);
        const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        bf16x4 h;
        h[0] = (__bf16)(dx * rstd); h[1] = (__bf16)(dy * rstd); h[2] = (__bf16)(dz * rstd); h[3] = (__bf16)(dw * rstd);
        *reinterpret_cast<bf16x4*>(sa + r * PA + lane * 8) = h;
    }

    bf16x8 bring[RING][G];
    const __bf16* bptr;
    auto load_b = [&](int slot, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < G; ++s) bring[slot][s] = *reinterpret_cast<const bf16x8*>(bptr + (size_t)(g * G + s) * 512);
    };
    // weight stream of the first fc1 pass starts before the barrier
    // Every workgroup walks the K dimension of each GEMM from a different starting group (rot): otherwise all 256 CUs
    // request the same weight fragment from the same L2 channel at the same time.
    constexpr int NG1 = (D / 16) / G;                       // 8 groups per 32-column tile of fc1
    constexpr int NG2 = (HID / 16) / G;                     // 16 groups per tile of fc2
    const int rot1 = blockIdx.x & (NG1 - 1), rot2 = (blockIdx.x * 5) & (NG2 - 1);
    bptr = p.w1 + (size_t)wn * (D / 16) * 512 + lane * 8;
#pragma unroll
    for (int d = 0; d < RING; ++d) load_b(d, (d + rot1) & (NG1 - 1));
    __syncthreads();

    // ---- 2. fc1 + GELU -> hidden slab ------------------------------------------------------------------------------
    // every B fragment is fetched by exactly one wave of the workgroup and used for both token halves
    const unsigned char* abase = sa + fr * PA + fk * 16;
    for (int pass = 0; pass < HID / 256; ++pass) {
        const int nt = pass * 8 + wn;                       // 32-column tile of the hidden dim
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < NG1; g0 += RING) {
#pragma unroll
            for (int d = 0; d < RING; ++d) {
                const int g = g0 + d, gr = (g + rot1) & (NG1 - 1);
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(abase + (gr * G + s) * 32);
                    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(abase + 32 * PA + (gr * G + s) * 32);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bring[d][s], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bring[d][s], acc[1], 0, 0, 0);
                }
                if (g + RING < NG1) load_b(d, (g + RING + rot1) & (NG1 - 1));
            }
        }
        // next pass's (or fc2's) weight stream
        const bool more1 = pass + 1 < HID / 256;
        if (more1) bptr = p.w1 + (size_t)(nt + 8) * (D / 16) * 512 + lane * 8;
        else bptr = p.w2 + (size_t)wn * (HID / 16) * 512 + lane * 8;
#pragma unroll
        for (int d = 0; d < RING; ++d) load_b(d, more1 ? ((d + rot1) & (NG1 - 1)) : ((d + rot2) & (NG2 - 1)));
        const float bias = p.b1[nt * 32 + fr];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = (r & 3) + 8 * (r >> 2) + 4 * fk;
                *reinterpret_cast<__bf16*>(sh + (i * 32 + q) * PH + (nt * 32 + fr) * 2) = (__bf16)gelu_fast(acc[i][r] + bias);
            }
    }
    __syncthreads();

    // ---- 3. fc2 + bias + residual ------------------------------------------------------------------------------------
    const unsigned char* hbase = sh + fr * PH + fk * 16;
    const int erow = lane >> 3, ecol = (lane & 7) * 4;
    {
        const int nt = wn;                                  // 8 waves x 32 columns = the 256 output channels
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < NG2; g0 += RING) {
#pragma unroll
            for (int d = 0; d < RING; ++d) {
                const int g = g0 + d, gr = (g + rot2) & (NG2 - 1);
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(hbase + (gr * G + s) * 32);
                    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(hbase + 32 * PH + (gr * G + s) * 32);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bring[d][s], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bring[d][s], acc[1], 0, 0, 0);
                }
                if (g + RING < NG2) load_b(d, (g + RING + rot2) & (NG2 - 1));
            }
        }
        const float bias = p.b2[nt * 32 + fr];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = (r & 3) + 8 * (r >> 2) + 4 * fk;
                etile[q * EP + fr] = acc[i][r] + bias;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = erow + 8 * k;
                const int m = m0 + i * 32 + q;
                if (m < p.M) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(etile + q * EP + ecol);
                    v += *reinterpret_cast<const f32x4*>(p.x + (size_t)m * D + nt * 32 + ecol);
                    *reinterpret_cast<f32x4*>(p.out + (size_t)m * D + nt * 32 + ecol) = v;
                }
            }
        }
    }
}

}  // namespace

extern "C" int spei_mlp_fused_bf16(const float* x, float* out, const void* w1_frag, const float* b1, const void* w2_frag,
                                   const float* b2, int64_t M, spei_stream_t stream) {
    SPEI_REQUIRE(x && out && w1_frag && b1 && w2_frag && b2 && M > 0, "spei_mlp_fused_bf16: bad arguments");
    SPEI_REQUIRE(M < (1ll << 31), "spei_mlp_fused_bf16: too many tokens");
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)w1_frag | (uintptr_t)w2_frag) % 16 == 0, "spei_mlp_fused_bf16: 16-byte alignment required");
    MlpParams p;
    p.x = x; p.out = out; p.w1 = (const __bf16*)w1_frag; p.b1 = b1; p.w2 = (const __bf16*)w2_frag; p.b2 = b2; p.M = (int)M;
    const size_t lds = (size_t)MT * PA + (size_t)MT * PH + (size_t)8 * 32 * EP * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL(mlp_fused_kernel, dim3(cdiv(M, MT)), dim3(512), lds, (hipStream_t)stream, p);
    SPEI_CHECK_LAUNCH("spei_mlp_fused_bf16");
    return 0;
}

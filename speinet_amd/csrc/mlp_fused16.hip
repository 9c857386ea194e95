// Fused Swin MLP branch on the gfx950 16-bit matrix pipe (bf16 or half operands):
//
//     out = x + fc2( GELU( fc1( LayerNorm(x) ) ) )          (reference model/swinir.py:12-29 Mlp, :279 block tail)
//
// Unfused this is LayerNorm -> GEMM 256->512 + GELU -> GEMM 512->256 + residual: the normalised tokens and the 512-wide
// hidden activations each make a round trip through HBM (59 + 118 MB per call at 720p, 72 calls per frame).  Here a
// 512-thread workgroup owns 128 tokens end to end.  The weights (512 KB of bf16 per workgroup) stream from L2 in MFMA
// fragment order; a CU takes them in at only ~30 B/clk, so every fragment is fetched by exactly ONE wave of the
// workgroup and feeds four MFMAs (the four 32-token row tiles):
//   1. LayerNorm(256), 16 lanes per token (4 DPP steps per reduction; the affine is folded into fc1 by pack.py) -> bf16
//      token slab in LDS (pitch 2*256+16 B);
//   2. per half of the hidden dim: fc1 TRANSPOSED (weights as the A operand, tokens on the lanes) so that a lane holds
//      4 consecutive hidden channels of one token -> bias + erf-GELU -> 8-byte writes into the hidden slab (bf16, LDS only);
//      then the fc2 partial product over that half accumulates into registers;
//   3. + bias, in-register quad transpose, + residual x, 16-byte stores.
// HBM traffic: x read (twice, the second time mostly from L2 for the residual) and out written: 118 MB instead of 531 MB.
#include "common.h"

namespace {

constexpr int D = 256, HID = 512, MT = 128;         // model dim, hidden dim, tokens per workgroup
constexpr int RT = MT / 32;                         // 32-token row tiles
constexpr int PA = 2 * D + 16;                      // LDS row pitch (bytes), token slab and hidden-half slab alike
constexpr int RING = 4;                             // weight fragments in flight per wave

template <typename LP>       // LP: __bf16 or _Float16
struct MlpParams {
    const float* x;
    float* out;
    const LP* w1;         // fragment order [HID/32][1][D/16][64][8]
    const float* b1;
    const LP* w2;         // fragment order [D/32][1][HID/16][64][8]
    const float* b2;
    long long* stamps;    // tuning build: phase stamps, else NULL
    int M;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// erf-GELU v * Phi(v) on two values (packed fp32 FMAs, no transcendental): Phi(v) - 1/2 = v Q(v^2) on |v| < 4, a degree-7
// weighted least-squares fit in v^2 constrained to Phi(4) = 1; beyond that GELU is max(v, 0) to fp32.  Max abs error of
// v Phi(v) 1.1e-4 -- below the bf16 rounding (2^-9 relative) the result gets right away, for every |v| > 0.06.  Only
// used in the "bf16" arithmetic mode; the f32 / bf16x3 modes apply erff in the GEMM epilogue.  The Abramowitz-Stegun
// 7.1.26 form used before cost one rcp + one exp + 14 VALU ops per value and made the hidden-slab write VALU-bound.
__device__ __forceinline__ f32x2 gelu2(f32x2 v) {
    const f32x2 u = v * v;
    f32x2 q = u * -1.419582270e-09f + 1.126438985e-07f;
    q = q * u + -3.898368825e-06f;
    q = q * u + 7.838465745e-05f;
    q = q * u + -1.034571474e-03f;
    q = q * u + 9.623637850e-03f;
    q = q * u + -6.612132016e-02f;
    q = q * u + 3.988274675e-01f;
    const f32x2 g = v * (v * q + 0.5f);
    f32x2 r;
    r[0] = fabsf(v[0]) < 4.0f ? g[0] : fmaxf(v[0], 0.f);
    r[1] = fabsf(v[1]) < 4.0f ? g[1] : fmaxf(v[1], 0.f);
    return r;
}

// erf-GELU as gelu2, with the |v| >= 4 tails folded into a clamp of the polynomial's argument: Phi(clamp(v)) is 1 / 0 there (the fit
// is constrained to Phi(4) = 1), two v_med3 instead of two compares, two selects and two max
__device__ __forceinline__ f32x2 gelu2c(f32x2 v) {
    f32x2 c;
    c[0] = __builtin_amdgcn_fmed3f(v[0], -4.0f, 4.0f);
    c[1] = __builtin_amdgcn_fmed3f(v[1], -4.0f, 4.0f);
    const f32x2 u = c * c;
    f32x2 q = u * -1.419582270e-09f + 1.126438985e-07f;
    q = q * u + -3.898368825e-06f;
    q = q * u + 7.838465745e-05f;
    q = q * u + -1.034571474e-03f;
    q = q * u + 9.623637850e-03f;
    q = q * u + -6.612132016e-02f;
    q = q * u + 3.988274675e-01f;
    return v * (c * q + 0.5f);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) { return v + dpp_quad<CTRL>(v); }
// all-reduce over each aligned group of 16 lanes: xor 1, xor 2 (quad_perm), then row_half_mirror / row_mirror
__device__ __forceinline__ float sum16(float v) { return dpp_add<0x140>(dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(v)))); }

// Round 4 revision of the kernel below (every wave through every phase).  What changed, all from in-kernel stamps:
//   * the fc2 accumulators START from x + b2: the residual rows are loaded (row chunks + a quad transpose) while fc1 of the first half
//     runs, straight into the accumulator registers — no residual registers (32), no wait for them before the epilogue, no adds;
//   * the 32 registers pay for an 8-deep weight ring (8 KB in flight per wave, 64 KB per CU: a CU takes ~70 GB/s from L2 at ~0.7 us,
//     which needs ~50 KB in flight; with 4 KB per wave the GEMM phases ran at the latency, not the rate);
//   * fc1 accumulators start from b1 (read from LDS), the GELU tails are a clamp of the polynomial's argument (gelu2c);
//   * barriers wait for LDS only (lds_barrier): __syncthreads() drains vmcnt, i.e. every weight fragment prefetched for the next phase.
template <typename LP>
__global__ __launch_bounds__(512) void mlp_fused_kernel(const MlpParams<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    typedef typename lpv<LP>::x4 lp4;
    constexpr int RG = 8;                                   // weight fragments in flight per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sa = smem;                               // [MT][PA]   normalised tokens, bf16
    unsigned char* sh = smem + MT * PA;                     // [MT][PA]   one half (256 channels) of the hidden activations
    float* bias1 = reinterpret_cast<float*>(smem + 2 * MT * PA);   // [512]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int m0 = blockIdx.x * MT;
    const int et = fr & 3, ecol = (fr >> 2) * 4;

    SPEI_STAMP(p.stamps, 0);
    bias1[tid] = p.b1[tid];
    // Every workgroup walks the K dimension of each GEMM from a different k-step (rot): otherwise all 256 CUs request the same weight
    // fragment from the same L2 channel at the same time.  GEMM step s contracts k-step (rot + s) & 15; the token slab and the hidden
    // slab are STORED rotated by the same amount, so step s reads byte offset 32 s of a row — an immediate of the ds_read (with the
    // rotation in the read address the 2 x 64 (row tile, step) addresses are computed once and held in registers for the whole kernel).
    const int rot = blockIdx.x & 15, rotb = rot * 32;

    // ---- 1. LayerNorm(256): 16 lanes per token, 32 tokens per pass; every load is issued before the first reduction ----
    {
        const int l16 = tid & 15, rsub = tid >> 4;
        f32x4 xr[MT / 32][4];
#pragma unroll
        for (int b = 0; b < MT / 32; ++b) {
            const int m = min(m0 + b * 32 + rsub, p.M - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[b][j] = reinterpret_cast<const f32x4*>(p.x + (size_t)m * D)[l16 + 16 * j];
        }
#pragma unroll
        for (int b = 0; b < MT / 32; ++b) {
            const int r = b * 32 + rsub;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (xr[b][j][0] + xr[b][j][1]) + (xr[b][j][2] + xr[b][j][3]);
            const float mean = sum16(s) * (1.0f / 256.0f);
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xr[b][j] -= mean;
                ss += (xr[b][j][0] * xr[b][j][0] + xr[b][j][1] * xr[b][j][1]) + (xr[b][j][2] * xr[b][j][2] + xr[b][j][3] * xr[b][j][3]);
            }
            const float rstd = 1.0f / sqrtf(sum16(ss) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lp4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = to_lp<LP>(xr[b][j][e] * rstd);
                *reinterpret_cast<lp4*>(sa + r * PA + (((l16 + 16 * j) * 8 - rotb) & 511)) = hv;
            }
        }
    }

    lp8 ring[RG];
    const LP* wptr = p.w1 + (size_t)wave * 16 * 512 + lane * 8;      // fc1, half 0: hidden tile `wave`
#pragma unroll
    for (int d = 0; d < RG; ++d) ring[d] = *reinterpret_cast<const lp8*>(wptr + ((rot + d) & 15) * 512);
    // residual rows -> (after the barrier) the fc2 accumulators; requested behind the first fc1 fragments, consumed under fc1
    f32x4 res[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = min(m0 + i * 32 + 8 * k + 4 * fk + et, p.M - 1);
            res[i][k] = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * D + wave * 32 + ecol);
        }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    SPEI_STAMP(p.stamps, 1);

    f32x16 acc2[RT];                                        // fc2 accumulators: tokens x output channels [32 wave, +32)
    const unsigned char* abase = sa + fr * PA + fk * 16;
    const unsigned char* hbase = sh + fr * PA + fk * 16;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // ---- 2a. fc1^T for hidden channels [256 half + 32 wave, +32): rows = channels, columns = tokens; acc starts from b1 ----
        f32x16 acc1[RT];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias1 + half * 256 + wave * 32 + 8 * g + 4 * fk);
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[i][4 * g + e] = bv[e];
        }
        // Software pipeline, pinned with full scheduling barriers: "LDS fragments of step s+1 and the refill of the ring slot,
        // then the MFMAs of step s".  Left alone the scheduler sinks every load to right before its use (ds_read +
        // lgkmcnt(0) in front of each MFMA, global_load + vmcnt(0) one step ahead: seen in the ISA) and the loop runs at
        // LDS / L2 latency; sched_group_barrier groups did not take in this fully unrolled region.
        lp8 tn[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) tn[i] = *reinterpret_cast<const lp8*>(abase + i * 32 * PA);
        const LP* wnext = p.w2 + (size_t)wave * 32 * 512 + (size_t)half * 16 * 512 + lane * 8;      // this half's fc2 fragments
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            lp8 tc[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) tc[i] = tn[i];
            if (s + 1 < 16) {
#pragma unroll
                for (int i = 0; i < RT; ++i) tn[i] = *reinterpret_cast<const lp8*>(abase + i * 32 * PA + (s + 1) * 32);
            }
            const lp8 w = ring[s % RG];
            // the ring runs on into the NEXT phase's stream: fc2's first fragments are in flight before fc1 ends (and through the GELU)
            ring[s % RG] = s + RG < 16 ? *reinterpret_cast<const lp8*>(wptr + ((rot + s + RG) & 15) * 512)
                                       : *reinterpret_cast<const lp8*>(wnext + ((rot + s + RG - 16) & 15) * 512);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc1[i] = mfma16(w, tc[i], acc1[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        SPEI_STAMP(p.stamps, 2 + 3 * half);
        if (half == 0) {
            // the residual has landed under fc1: fc2 accumulators = x + b2 (row chunks -> accumulator layout: the transpose is its own inverse)
            const float bias = p.b2[wave * 32 + fr];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float a[4] = {res[i][k][0], res[i][k][1], res[i][k][2], res[i][k][3]};
                    quad_transpose4(a[0], a[1], a[2], a[3], et);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[i][4 * k + e] = a[e] + bias;
                }
        } else {
            lds_barrier();                                  // fc2 of half 0 is done reading the hidden slab
        }
        // rows c = (r&3) + 8*(r>>2) + 4*fk = 8g + 4fk + e, column = token fr  ->  sh[token][32 wave + c]
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const f32x2 g01 = gelu2c(f32x2{acc1[i][4 * g], acc1[i][4 * g + 1]});
                const f32x2 g23 = gelu2c(f32x2{acc1[i][4 * g + 2], acc1[i][4 * g + 3]});
                lp4 hv;
                hv[0] = to_lp<LP>(g01[0]); hv[1] = to_lp<LP>(g01[1]); hv[2] = to_lp<LP>(g23[0]); hv[3] = to_lp<LP>(g23[1]);
                *reinterpret_cast<lp4*>(sh + (i * 32 + fr) * PA + (((wave * 32 + 8 * g + 4 * fk) * 2 - rotb) & 511)) = hv;
            }
        }
        lds_barrier();
        SPEI_STAMP(p.stamps, 3 + 3 * half);
        // ---- 2b. fc2 partial product over this half of the hidden dim ------------------------------------------------------
        wptr = wnext;
        const LP* wafter = p.w1 + (size_t)(8 + wave) * 16 * 512 + lane * 8;      // fc1, half 1
#pragma unroll
        for (int i = 0; i < RT; ++i) tn[i] = *reinterpret_cast<const lp8*>(hbase + i * 32 * PA);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            lp8 tc[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) tc[i] = tn[i];
            if (s + 1 < 16) {
#pragma unroll
                for (int i = 0; i < RT; ++i) tn[i] = *reinterpret_cast<const lp8*>(hbase + i * 32 * PA + (s + 1) * 32);
            }
            const lp8 w = ring[s % RG];
            if (s + RG < 16) ring[s % RG] = *reinterpret_cast<const lp8*>(wptr + ((rot + s + RG) & 15) * 512);
            else if (half == 0) ring[s % RG] = *reinterpret_cast<const lp8*>(wafter + ((rot + s + RG - 16) & 15) * 512);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc2[i] = mfma16(tc[i], w, acc2[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        SPEI_STAMP(p.stamps, 4 + 3 * half);
        wptr = wafter;
    }

    // ---- 3. the accumulators hold x + b2 + fc2(...): quad transpose, 16-byte stores --------------------------------------------------------
#pragma unroll
    for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = acc2[i][4 * k + e];
            quad_transpose4(a[0], a[1], a[2], a[3], et);
            const int m = m0 + i * 32 + 8 * k + 4 * fk + et;
            if (m < p.M) *reinterpret_cast<f32x4*>(p.out + (size_t)m * D + wave * 32 + ecol) = f32x4{a[0], a[1], a[2], a[3]};
        }
    }
    SPEI_STAMP(p.stamps, 8);
}

}  // namespace

template <typename LP>
static int mlp_launch(const float* x, float* out, const void* w1, const float* b1, const void* w2, const float* b2, int64_t M, hipStream_t st) {
    MlpParams<LP> p;
    p.x = x; p.out = out; p.w1 = (const LP*)w1; p.b1 = b1; p.w2 = (const LP*)w2; p.b2 = b2; p.M = (int)M;
    p.stamps = spei_stamp_buffer();
    const size_t lds = (size_t)2 * MT * PA + HID * sizeof(float);
    ensure_dyn_lds<&mlp_fused_kernel<LP>>(lds);
    hipLaunchKernelGGL(mlp_fused_kernel<LP>, dim3(cdiv(M, MT)), dim3(512), lds, st, p);
    SPEI_CHECK_LAUNCH("spei_mlp_fused16");
    return 0;
}

extern "C" int spei_mlp_fused16(int fmt, const float* x, float* out, const void* w1_frag, const float* b1, const void* w2_frag,
                                const float* b2, int64_t M, spei_stream_t stream) {
    SPEI_REQUIRE(x && out && w1_frag && b1 && w2_frag && b2 && M > 0, "spei_mlp_fused16: bad arguments");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_mlp_fused16: fmt=%d", fmt);
    SPEI_REQUIRE(M < (1ll << 31), "spei_mlp_fused16: too many tokens");
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)w1_frag | (uintptr_t)w2_frag) % 16 == 0, "spei_mlp_fused16: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (fmt == SPEI_F16) return mlp_launch<_Float16>(x, out, w1_frag, b1, w2_frag, b2, M, st);
    return mlp_launch<__bf16>(x, out, w1_frag, b1, w2_frag, b2, M, st);
}

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
#include <type_traits>

namespace {

constexpr int D = 256, HID = 512, MT = 128;         // model dim, hidden dim, tokens per workgroup
constexpr int RT = MT / 32;                         // 32-token row tiles
constexpr int PA = 2 * D + 16;                      // LDS row pitch (bytes), token slab and hidden-half slab alike

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

// gelu2c on four values at once: the two packed chains advance in lockstep, so that no instruction waits on the one before it (a
// dependent v_pk_fma_f32 costs a wait state: the compiler pads with s_nop or splits the packed operation, +60 % instructions).  The
// coefficients come as scalar register PAIRS the compiler cannot see through (gelu_consts): a literal is not a packed operand, and
// with literals a third of the packed operations are emitted as two scalar ones.
struct GeluK { f32x2 c[8]; f32x2 half; };
__device__ __forceinline__ f32x2 sgpr_pair(float v) {
    f32x2 r = {v, v};
    asm volatile("" : "+s"(r));
    return r;
}
__device__ __forceinline__ GeluK gelu_consts() {
    GeluK k;
    k.c[0] = sgpr_pair(-1.419582270e-09f); k.c[1] = sgpr_pair(1.126438985e-07f); k.c[2] = sgpr_pair(-3.898368825e-06f);
    k.c[3] = sgpr_pair(7.838465745e-05f);  k.c[4] = sgpr_pair(-1.034571474e-03f); k.c[5] = sgpr_pair(9.623637850e-03f);
    k.c[6] = sgpr_pair(-6.612132016e-02f); k.c[7] = sgpr_pair(3.988274675e-01f);
    k.half = sgpr_pair(0.5f);
    return k;
}
__device__ __forceinline__ f32x4 gelu4c(f32x4 v, const GeluK& k) {
    const f32x2 va = {v[0], v[1]}, vb = {v[2], v[3]};
    f32x2 ca, cb;
    ca[0] = __builtin_amdgcn_fmed3f(v[0], -4.0f, 4.0f); ca[1] = __builtin_amdgcn_fmed3f(v[1], -4.0f, 4.0f);
    cb[0] = __builtin_amdgcn_fmed3f(v[2], -4.0f, 4.0f); cb[1] = __builtin_amdgcn_fmed3f(v[3], -4.0f, 4.0f);
    const f32x2 ua = ca * ca, ub = cb * cb;
    f32x2 qa = __builtin_elementwise_fma(ua, k.c[0], k.c[1]), qb = __builtin_elementwise_fma(ub, k.c[0], k.c[1]);
#pragma unroll
    for (int j = 2; j < 8; ++j) {
        qa = __builtin_elementwise_fma(qa, ua, k.c[j]);
        qb = __builtin_elementwise_fma(qb, ub, k.c[j]);
    }
    qa = __builtin_elementwise_fma(ca, qa, k.half);
    qb = __builtin_elementwise_fma(cb, qb, k.half);
    const f32x2 ga = va * qa, gb = vb * qb;
    return f32x4{ga[0], ga[1], gb[0], gb[1]};
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


// ---- round 4, second half: the same branch as a PERSISTENT, software-pipelined kernel ---------------------------------------------------
// Stamps of `mlp_fused_kernel` on the two stacked maps of a frame (900 workgroups, 28.2 us each): token load + LayerNorm 8.3 us, the
// four GEMM phases 9.8 (the last one 1.84 = the matrix pipe's own time), the two GELU phases 7.2, the epilogue 2.1 — one workgroup per
// CU (137 KB of LDS), every wave in the same phase, so the matrix pipe idles through 18 of the 28 us.  Here one workgroup per CU walks
// tiles of 96 tokens (tile = blockIdx.x + k gridDim.x) and every vector phase rides INSIDE a GEMM phase of the same waves:
//     P1  fc1, hidden half 0     (tokens: slab a)    + the PREVIOUS tile's results stored, this tile's residual rows requested
//     P2  fc1, hidden half 1                         + GELU of half 0 -> slab b;  fc2 accumulators = residual + bias
//     P3  fc2 over half 0        (slab b)            + GELU of half 1 -> slab a (the normalised tokens are dead);  the NEXT tile's x rows requested
//     P4  fc2 over half 1        (slab a)            + LayerNorm of the next tile -> slab c;   a <-> c
// Three 96-row slabs (3 x 50.7 KB): 96 instead of 128 tokens per 512 KB of weight fragments is what a third slab costs (P1, the one
// phase without vector work, then runs at the L2 -> CU intake: 74 GB/s), and the weight ring (8 fragments per wave) runs on from phase
// to phase and from tile to tile, so its start-up is paid once per workgroup.  Three barriers per tile (after P2, P3 and P4), each
// between a slab's last read and its next write.  No epilogue: results and residual move as one dword per lane in the accumulators'
// own layout (lanes 0-31 of a register = 32 consecutive channels of a row = one 128-byte line) under P1.
// Measured (two stacked 720p maps, same box): 111 -> 89 us per launch, 16.2 us per tile where the four GEMM phases alone need 5.5: the
// phases with vector work are bound by the SIMD's vector issue (GELU: 7 packed FMAs + 6 more instructions per value pair, 45 % of a tile's
// vector instructions) at a shader clock that falls from 2.15 to 1.6 GHz while the matrix pipe and the vector units run together
// (SPEI_STAMP_CLK); with the MFMAs compiled out the same phases take 1.7-2.4 us instead of 4.1-5.3.
constexpr int MP = 96;                              // tokens per tile
constexpr int RP3 = MP / 32;                        // 32-token row tiles per tile
constexpr int SLABP = MP * PA;                      // one slab (bytes)

template <typename LP>
__global__ __launch_bounds__(512) void mlp_pipe_kernel(const MlpParams<LP> p, const int ntiles, const int stagger) {
    typedef typename lpv<LP>::x8 lp8;
    typedef typename lpv<LP>::x4 lp4;
    constexpr int RG = 8;                                   // weight fragments in flight per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sb = smem + SLABP;                 // slab b: GELU(fc1), hidden half 0
    float* bias1 = reinterpret_cast<float*>(smem + 3 * SLABP);   // [512]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fk = lane >> 5;
    const int l16 = tid & 15, rsub = tid >> 4;              // LayerNorm staging: 16 lanes per token, 32 tokens per pass

    SPEI_STAMP(p.stamps, 0); SPEI_STAMP_CLK(p.stamps, 8 + 0);
    spei_stagger_start(ntiles, stagger);
    bias1[tid] = p.b1[tid];
    const float bias2 = p.b2[wave * 32 + fr];
    // The weight stream through buffer loads: descriptor and fragment offset in scalar registers, one vector register (lane x 16) for
    // every load of the kernel.  With flat loads the compiler hoists the ~64 loop-invariant 64-bit fragment addresses out of the tile
    // loop and spills them (80 registers in the first build).
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<LP*>(p.w1), 0, HID * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<LP*>(p.w2), 0, HID * D * 2, 0x00020000);
    const int lane16 = lane * 16;
    // fragment index of (phase, k-step 0): fc1 hidden tile `wave` of half 0 / half 1; fc2 output tile `wave`, k-steps of half 0 / half 1
    const int fbase[4] = {wave * 16, (8 + wave) * 16, wave * 32, wave * 32 + 16};
    auto wload = [&](int ph, int kstep) -> lp8 {
        return __builtin_bit_cast(lp8, __builtin_amdgcn_raw_buffer_load_b128(ph < 2 ? rs1 : rs2, lane16, (fbase[ph] + kstep) * 1024, 0));
    };

    const GeluK gk = gelu_consts();
    int tile = blockIdx.x;
    f32x4 xr[RP3][4];                                       // a tile's x rows on their way to the LayerNorm
    float lnm[RP3], lns[RP3];
    auto x_load = [&](int t, int b, int j) {
        const int m = min(t * MP + b * 32 + rsub, p.M - 1);
        xr[b][j] = reinterpret_cast<const f32x4*>(p.x + (size_t)m * D)[l16 + 16 * j];
    };
    // LayerNorm of pass b (32 tokens) in four slices: sum | centre, squares | rstd | scale, convert, write (rotated by rb bytes)
    // (inlined in the prologue and in P4: products and sums spelled out, contraction off — a tile's LayerNorm must not depend on which of
    // the two computed it)
    auto ln_slice = [&](int b, int sub, unsigned char* dst, int rb) {
#pragma clang fp contract(off)
        if (sub == 0) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (xr[b][j][0] + xr[b][j][1]) + (xr[b][j][2] + xr[b][j][3]);
            lnm[b] = sum16(s) * (1.0f / 256.0f);
        } else if (sub == 1) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xr[b][j] -= lnm[b];
                ss += __builtin_fmaf(xr[b][j][1], xr[b][j][1], xr[b][j][0] * xr[b][j][0]) + __builtin_fmaf(xr[b][j][3], xr[b][j][3], xr[b][j][2] * xr[b][j][2]);
            }
            lns[b] = ss;
        } else if (sub == 2) {
            lns[b] = 1.0f / sqrtf(__builtin_fmaf(sum16(lns[b]), 1.0f / 256.0f, 1e-5f));
        } else {
            const int r = b * 32 + rsub;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lp4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = to_lp<LP>(xr[b][j][e] * lns[b]);
                *reinterpret_cast<lp4*>(dst + r * PA + (((l16 + 16 * j) * 8 - rb) & 511)) = hv;
            }
        }
    };

    // ---- prologue: the first tile's tokens -> slab 0; the ring's first fragments ------------------------------------------------------
    int rot = tile & 15;
#pragma unroll
    for (int b = 0; b < RP3; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) x_load(tile, b, j);
    lp8 ring[RG];
#pragma unroll
    for (int d = 0; d < RG; ++d) ring[d] = wload(0, (rot + d) & 15);
#pragma unroll
    for (int b = 0; b < RP3; ++b)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) ln_slice(b, sub, smem, rot * 32);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    SPEI_STAMP(p.stamps, 1); SPEI_STAMP_CLK(p.stamps, 8 + 1);

    // x and out through buffer descriptors whose range is the M rows: dword accesses in the accumulators' own layout (register 4 k + e of
    // row tile i <-> row 32 i + 8 k + 4 fk + e, channel 32 wave + fr), out-of-range rows read 0 / are not written
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.M * (D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.M * (D * 4), 0x00020000);
    const int voff_row = fk * 4096 + fr * 4;
    f32x16 acc2[RP3];                                       // fc2 accumulators; a tile's results are stored under the NEXT tile's P1
#pragma unroll
    for (int i = 0; i < RP3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
    int cur = 0;
    bool first = true;
    for (; tile < ntiles; tile += gridDim.x) {
        unsigned char* const sa = smem + (cur ? 2 * SLABP : 0);         // this tile's tokens, later GELU(fc1) of half 1
        unsigned char* const sc = smem + (cur ? 0 : 2 * SLABP);         // the next tile's tokens
        const int tnext = tile + gridDim.x;
        const int rotn = tnext & 15, rotb = rot * 32;
        // fragment q of the tile's weight stream (q = 16 phase + step; q >= 64: the next tile's first phase)
        auto wfrag = [&](int q) -> lp8 {
            return wload((q >> 4) & 3, ((q >= 64 ? rotn : rot) + (q & 15)) & 15);
        };
        // one GEMM phase: 16 k-steps, RP3 MFMAs per step on the fragment the ring delivers; token fragments from `src` one step ahead;
        // filler(s): the vector work that rides in this phase.  Pinned with full scheduling barriers (left alone the
        // scheduler sinks every load to right before its use).
        auto gemm = [&](auto phc, const unsigned char* src, f32x16 (&acc)[RP3], auto weights_are_a, auto&& filler) {
            constexpr int ph = decltype(phc)::value;
            const unsigned char* base = src + fr * PA + fk * 16;
            lp8 tn[RP3];
#pragma unroll
            for (int i = 0; i < RP3; ++i) tn[i] = *reinterpret_cast<const lp8*>(base + i * 32 * PA);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                lp8 tc[RP3];
#pragma unroll
                for (int i = 0; i < RP3; ++i) tc[i] = tn[i];
                if (s + 1 < 16) {
#pragma unroll
                    for (int i = 0; i < RP3; ++i) tn[i] = *reinterpret_cast<const lp8*>(base + i * 32 * PA + (s + 1) * 32);
                }
                const lp8 w = ring[(ph * 16 + s) % RG];
                ring[(ph * 16 + s) % RG] = wfrag(ph * 16 + s + RG);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < RP3; ++i) {
                    if constexpr (decltype(weights_are_a)::value) acc[i] = mfma16(w, tc[i], acc[i]);
                    else acc[i] = mfma16(tc[i], w, acc[i]);
                }
                filler(s);
                // An MFMA is a pure function of its operands: an IR pass moves the chains whose results are needed last (row tiles 1, 2 of
                // P2: GELU reads them late in P3) to the end of the phase — behind the workgroup barrier — and keeps their 64 + 128
                // operand registers alive until then (68 spilled registers).  An empty volatile asm on each accumulator pins them.
#pragma unroll
                for (int i = 0; i < RP3; ++i) asm volatile("" : "+v"(acc[i]));
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // GELU of one (row tile i, register group g) of a fc1^T accumulator set: rows c = 8 g + 4 fk + e (hidden channel 32 wave + c
        // of the half), column = token fr  ->  dst[token][32 wave + c], 8-byte writes
        auto gelu_slice = [&](const f32x16 (&acc)[RP3], int s, unsigned char* dst) {
            if (s < 4 * RP3) {
                const int i = s >> 2, g = s & 3;
                const f32x4 gv = gelu4c(f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]}, gk);
                *reinterpret_cast<lp4*>(dst + (i * 32 + fr) * PA + (((wave * 32 + 8 * g + 4 * fk) * 2 - rotb) & 511)) = to_lp4<LP>(gv);
            }
        };
        auto init_rows = [&](f32x16 (&acc)[RP3], const float* b) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b + wave * 32 + 8 * g + 4 * fk);
#pragma unroll
                for (int i = 0; i < RP3; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][4 * g + e] = bv[e];
            }
        };

        // byte offsets of (row 4 fk of the tile, channel fr) for the previous tile's stores and this tile's residual loads; before the
        // first tile the "previous" offset is negative = beyond any descriptor range (M <= 2^21 rows): those stores are dropped
        const int vthis = voff_row + tile * (MP * 1024), vprev = vthis - (int)gridDim.x * (MP * 1024);
        f32x16 acc1a[RP3], acc1b[RP3];
        f32x16 res[RP3];                                    // this tile's residual rows, in the fc2 accumulators' layout
        init_rows(acc1a, bias1);
        // P1 runs at the weight intake (3 MFMAs per fragment; 1.7 of 1.84 us are the L2 -> CU stream) and has issue slots to spare: the
        // PREVIOUS tile's 48 result registers are stored here and this tile's 48 residual values requested, one dword per lane: lanes 0-31
        // of a register hold 32 consecutive channels of one row (a 128-byte line), so no transposition is needed in either direction
        // (the 16-byte form cost 24 quad transposes = 300 vector instructions per tile) and rows past M fall to the descriptor's range check
        gemm(std::integral_constant<int, 0>{}, sa, acc1a, std::true_type{}, [&](int s) {
            if (s < 4 * RP3) {
                const int i = s >> 2, k = s & 3;
                // the row lives in the VECTOR offset (the descriptor's range check covers vector offset + immediate only; the scalar
                // offset carries the wave's 128-byte channel block, which never leaves the row)
                const int vst = vprev + (i * 32 + 8 * k) * 1024, vld = vthis + (i * 32 + 8 * k) * 1024;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc2[i][4 * k + e]), rso, vst + e * 1024, wave * 128, 0);
                    res[i][4 * k + e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsx, vld + e * 1024, wave * 128, 0));
                }
            }
        });
        if (first) { SPEI_STAMP(p.stamps, 2); SPEI_STAMP_CLK(p.stamps, 8 + 2); }
        init_rows(acc1b, bias1 + 256);
        gemm(std::integral_constant<int, 1>{}, sa, acc1b, std::true_type{}, [&](int s) {
            gelu_slice(acc1a, s, sb);
            if (s >= 4 * RP3) {                             // the last four steps carry no GELU: fc2's accumulators = residual + bias
#pragma unroll
                for (int i = 0; i < RP3; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[i][4 * (s - 4 * RP3) + e] = res[i][4 * (s - 4 * RP3) + e] + bias2;
            }
        });
        lds_barrier();                                      // slab b complete; every wave is done reading the tokens in slab a
        if (first) { SPEI_STAMP(p.stamps, 3); SPEI_STAMP_CLK(p.stamps, 8 + 3); }
        gemm(std::integral_constant<int, 2>{}, sb, acc2, std::false_type{}, [&](int s) {
            gelu_slice(acc1b, s, sa);
            // the next tile's rows: requested here, normalised under P4 (a tile past the end re-reads row M - 1); pass b's 16 registers
            // are requested once GELU has retired row tile b of half 1 (steps 4 b + 4, 4 b + 5): the phase's peak stays under 256 registers
            if (s >= 4 && (s & 3) < 2) {
                x_load(tnext, (s >> 2) - 1, 2 * (s & 1));
                x_load(tnext, (s >> 2) - 1, 2 * (s & 1) + 1);
            }
        });
        lds_barrier();                                      // slab a holds half 1; slab b is free
        if (first) { SPEI_STAMP(p.stamps, 4); SPEI_STAMP_CLK(p.stamps, 8 + 4); }
        gemm(std::integral_constant<int, 3>{}, sa, acc2, std::false_type{}, [&](int s) {
            if (s < 4 * RP3) ln_slice(s >> 2, s & 3, sc, rotn * 32);
        });
        if (first) { SPEI_STAMP(p.stamps, 5); SPEI_STAMP_CLK(p.stamps, 8 + 5); }
        lds_barrier();                                      // slab c holds the next tile; slab a is free
        if (first) { SPEI_STAMP(p.stamps, 6); SPEI_STAMP_CLK(p.stamps, 8 + 6); }
        first = false;
        cur ^= 1;
        rot = rotn;
    }
    // the last tile's results (`tile` is one stride past it)
#pragma unroll
    for (int i = 0; i < RP3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc2[i][r]), rso,
                                                  voff_row + (tile - (int)gridDim.x) * (MP * 1024) + (i * 32 + 8 * (r >> 2) + (r & 3)) * 1024, wave * 128, 0);
    SPEI_STAMP(p.stamps, 7); SPEI_STAMP_CLK(p.stamps, 8 + 7);
}

}  // namespace

template <typename LP>
static int mlp_launch(const float* x, float* out, const void* w1, const float* b1, const void* w2, const float* b2, int64_t M, hipStream_t st) {
    MlpParams<LP> p;
    p.x = x; p.out = out; p.w1 = (const LP*)w1; p.b1 = b1; p.w2 = (const LP*)w2; p.b2 = b2; p.M = (int)M;
    p.stamps = spei_stamp_buffer();
    static const int pipe = spei_knob("SPEI_MLP_PIPE", 1);               // tuning build: 0 = the one-tile-per-workgroup kernel
    if (pipe && M <= (1ll << 21)) {                                      // 32-bit byte offsets into x / out with room for a negative tile
        const int ntiles = cdiv(M, MP);
        const size_t lds = (size_t)3 * SLABP + HID * sizeof(float);
        ensure_dyn_lds<&mlp_pipe_kernel<LP>>(lds);
        static const int stagger = spei_knob("SPEI_PIPE_STAGGER", 0);
        hipLaunchKernelGGL(mlp_pipe_kernel<LP>, dim3(ntiles < spei_num_cus() ? ntiles : spei_num_cus()), dim3(512), lds, st, p, ntiles, stagger);
        SPEI_CHECK_LAUNCH("spei_mlp_fused16");
        return 0;
    }
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

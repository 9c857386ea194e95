// Token-stationary Swin kernels on the gfx950 16-bit matrix pipe (bf16 or half operands), round 3.
//
// What bounded the round-2 fused kernels (attn_fused16.hip, mlp_fused16.hip): every 256-thread half streamed
// the block's 512 KB of weights from L2 into REGISTERS once per 50 / 128 tokens, and a CU takes in only ~70 GB/s from L2; the
// phases of a workgroup (LayerNorm staging, GEMMs, GELU / softmax, epilogue) ran one after the other behind barriers, so the
// matrix pipe idled through every VALU phase (26 % busy).  Here the roles are swapped:
//   * the WEIGHTS of a block are pre-arranged at pack time (speinet_amd/pack.py) into one linear stream of 1 KiB MFMA fragments
//     in the order the kernel consumes them; a workgroup moves the stream through a 4-slot LDS ring of 32 KiB chunks with
//     LDS-DMA (global_load_lds_dwordx4: one fragment per wave-instruction, lane-linear, no VGPRs) — ONCE per 8 waves = 256
//     tokens (2 KB of L2 -> CU traffic per token instead of 8);
//   * the TOKENS stay in registers: a wave owns 32 tokens (the lanes are the MFMA columns) from the first load to the last
//     store — LayerNorm is an in-lane sum plus one cross-half swap, every GEMM is computed transposed (weights = A operand
//     from the ring, tokens = B operand from registers), and a 32x32 accumulator is handed to the next GEMM as its B operand
//     after a pairwise cvt (MI355X guide, "an accumulator tile as the next MFMA's operand"; the permuted k order that form
//     presents is baked into the weight stream).  Nothing a wave produces is read by another wave: the only synchronisation is
//     one barrier per ring chunk;
//   * waves 4-7 run half a chunk behind waves 0-3 (they defer the second GEMM of a chunk into the next iteration), so that on
//     each SIMD one wave's VALU phase (GELU) sits beside its partner's MFMA phase instead of both idling the matrix pipe.
//
//     spei_mlp_tok16:   out = x + fc2( GELU( fc1( LayerNorm(x) ) ) )          (reference model/swinir.py:12-29 Mlp, :279)
#include "common.h"
#include <type_traits>

namespace {

// timing ablations, tuning build only (SPEI_TOK_DBG=n, read at launch): 1 no GELU, 2 no ring reads (compile-time only), 4 no
// barrier, 8 no LDS-DMA issue, 16 no stagger (waves 4-7 in the same order as 0-3).  The shipping build folds them away.
#ifdef SPEI_TUNING
#define TOK_DBG (p.dbg)
#else
#define TOK_DBG 0
#endif
constexpr int D = 256, HID = 512;
constexpr int FRAG = 1024;                 // one MFMA operand fragment: 64 lanes x 16 B
constexpr int CHUNK = 32 * FRAG;           // ring slot
constexpr int RING = 4;                    // slots: two being read (waves 4-7 lag by half a chunk), two in flight

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// N consecutive fragments global -> LDS: fragment i, lane l: 16 bytes from src_lane + 1024 i to dst + 1024 i + 16 l (dst
// wave-uniform; the instruction offset applies to both addresses: one pointer pair and one M0 value for the run)
template <int N>
__device__ __forceinline__ void glds_run(const unsigned char* src_lane, unsigned char* dst) {
    static_assert(N == 2 || N == 4 || N == 8, "glds_run");
    gbl_void* g = (gbl_void*)(src_lane);
    lds_void* l = (lds_void*)(uintptr_t)(dst);
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
    if constexpr (N >= 4) {
        __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
    }
    if constexpr (N >= 8) {
        g = (gbl_void*)(src_lane + 4096);
        l = (lds_void*)(uintptr_t)(dst + 4096);
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
    }
}

// erf-GELU v * Phi(v): Phi(v) - 1/2 = v Q(v^2) on |v| <= 4 (degree-7 weighted least squares in v^2, constrained to Phi(4) = 1:
// the polynomial of mlp_fused16.hip), evaluated at the clamped argument so that both tails come out without a select:
// v >= 4 -> v * 1, v <= -4 -> v * 0.  Max abs error 1.1e-4.
__device__ __forceinline__ float gelu1(float v) {
    const float c = __builtin_amdgcn_fmed3f(v, -4.0f, 4.0f);
    const float u = c * c;
    float q = u * -1.419582270e-09f + 1.126438985e-07f;
    q = q * u + -3.898368825e-06f;
    q = q * u + 7.838465745e-05f;
    q = q * u + -1.034571474e-03f;
    q = q * u + 9.623637850e-03f;
    q = q * u + -6.612132016e-02f;
    q = q * u + 3.988274675e-01f;
    return v * (c * q + 0.5f);
}

// ---- ring -> register fragment pipeline, by hand ----------------------------------------------------------------------------
// ds_read_b128 and its s_waitcnt as inline asm: hipcc's own placement for this loop was "read 2 (or 5) fragments, s_waitcnt
// lgkmcnt(0), MFMAs", the whole LDS latency exposed each time (2.4 us per chunk where the MFMAs need 0.9; stamps, round 3).  LDS
// reads return in order, so waiting until at most N of MY later reads are outstanding retires the fragment; the wait takes the
// fragment as an in/out operand, which ties the MFMA that consumes it behind the wait.  Reads the compiler issues on its own
// (bias rows) can only make these waits longer, never shorter.
template <int N, typename F>
__device__ __forceinline__ void lds_wait(F& f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}
// 16 fragments at addr + FRAG * i through a PD-deep register pipeline; step(i, fragment) issues the MFMA(s) of fragment i.
// lds_pre issues the first PD reads (possibly ahead of a VALU phase), lds_pipe the rest, each right after the MFMA that freed
// its register.
template <int PD, typename F>
__device__ __forceinline__ void lds_pre(F (&fq)[PD], unsigned addr) {
#pragma unroll
    for (int i = 0; i < PD; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fq[i]) : "v"(addr), "n"(i * 1024));
}
template <int PD, int I, typename F, typename Step>
__device__ __forceinline__ void lds_pipe(F (&fq)[PD], unsigned addr, Step&& step) {
    if constexpr (I < 16) {
        constexpr int after = (15 - I) < (PD - 1) ? (15 - I) : (PD - 1);       // my reads issued after fragment I's
        lds_wait<after>(fq[I % PD]);
        step(std::integral_constant<int, I>{}, fq[I % PD]);
        if constexpr (I + PD < 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fq[I % PD]) : "v"(addr), "n"((I + PD) * 1024));
        lds_pipe<PD, I + 1>(fq, addr, step);
    }
}

template <int S, typename LP>
__device__ __forceinline__ typename lpv<LP>::x8 acc_frag(const f32x16& a) {      // registers 8S..8S+7 as the k-step-S operand
    typename lpv<LP>::x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = to_lp<LP>(a[8 * S + j]);
    return r;
}

template <typename LP>
struct MlpTokParams {
    const float* x;
    float* out;
    const unsigned char* wstream;   // 16 chunks x 32 fragments (pack.py mlp_stream): chunk c = fc1 tile c (16 k-steps), then fc2
    const float* b1;                // [512]                                                    (n-tile j, k-steps 2c, 2c+1) x 8
    const float* b2;                // [256]
    long long* stamps;              // tuning build: phase stamps, else NULL
    int M;
    int dbg;                        // tuning build: TOK_DBG bits
};

template <typename LP, int NW>
__global__ __launch_bounds__(64 * NW) void mlp_tok_kernel(const MlpTokParams<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // ring[RING][CHUNK] | bias1[512]  (ONE array: guide §5)
    float* bias1 = reinterpret_cast<float*>(smem + RING * CHUNK);
    constexpr int PER = 32 / NW;                                              // fragments a wave moves per chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, h = lane >> 5;
    const bool late = wave >= NW / 2 && !(TOK_DBG & 16);                                         // the half that runs half a chunk behind
    const int tok = blockIdx.x * (32 * NW) + wave * 32 + fr;
    const int tokc = min(tok, p.M - 1);
    const unsigned char* wsrc = p.wstream + lane * 16;

    auto issue = [&](int c) {                                                 // this wave's PER consecutive fragments of chunk c
        glds_run<PER>(wsrc + (size_t)(c * 32 + wave * PER) * FRAG, smem + (c & (RING - 1)) * CHUNK + wave * PER * FRAG);
    };
    SPEI_STAMP(p.stamps, 0);
    issue(0);
    issue(1);
    for (int i = tid; i < HID; i += 64 * NW) bias1[i] = p.b1[i];

    // ---- LayerNorm(256) of the lane's token: the lane holds k = 16 s + 8 h + j, its partner (lane ^ 32) the other half ----
    lp8 xh[16];
    {
        f32x4 xr[32];
        const float* xp = p.x + (size_t)tokc * D + 8 * h;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            xr[2 * s] = *reinterpret_cast<const f32x4*>(xp + 16 * s);
            xr[2 * s + 1] = *reinterpret_cast<const f32x4*>(xp + 16 * s + 4);
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) sum += (xr[i][0] + xr[i][1]) + (xr[i][2] + xr[i][3]);
        const float mean = xor_combine<32, OpSum>(sum) * (1.0f / 256.0f);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            xr[i] -= mean;
            ss += (xr[i][0] * xr[i][0] + xr[i][1] * xr[i][1]) + (xr[i][2] * xr[i][2] + xr[i][3] * xr[i][3]);
        }
        const float rstd = 1.0f / sqrtf(xor_combine<32, OpSum>(ss) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) xh[s][j] = to_lp<LP>(xr[2 * s + (j >> 2)][j & 3] * rstd);
    }

    SPEI_STAMP(p.stamps, 1);
    f32x16 acc2[8];                                   // out^T: rows = output channel 32 j + .., columns = the wave's tokens
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    lp8 hk0, hk1;                                     // GELU(fc1) of the current chunk as the B operand of fc2 (k-steps 0, 1)

    constexpr int PD = 3;                             // fragments in flight per wave (lds_pipe)
    lp8 fq[PD];
    const unsigned lds0 = (unsigned)(uintptr_t)smem + lane * 16;
    auto fc2_pre = [&](int c) { lds_pre<PD>(fq, lds0 + (c & (RING - 1)) * CHUNK + 16 * FRAG); };    // ahead of the GELU
    // s_setprio 1 around every MFMA run: issue on a SIMD is arbitrated by priority, then age, and the partner wave's GELU (a few
    // hundred back-to-back VALU instructions) otherwise takes the issue slots the MFMAs need (MI355X guide, "Two waves per SIMD")
    auto fc2 = [&](int c) {
        __builtin_amdgcn_s_setprio(1);
        lds_pipe<PD, 0>(fq, lds0 + (c & (RING - 1)) * CHUNK + 16 * FRAG, [&](auto ic, const lp8& a) {
            constexpr int i = decltype(ic)::value;
            acc2[i >> 1] = mfma16(a, (i & 1) ? hk1 : hk0, acc2[i >> 1]);
        });
        __builtin_amdgcn_s_setprio(0);
    };

    for (int c = 0; c < 16; ++c) {
        // this wave's share of chunk c has landed (chunk c + 1's may still be in flight), then everybody's has, and everybody is
        // done with chunk c - 2 (waves 4-7: with the fc2 part of chunk c - 2, read during iteration c - 1)
        if (c + 1 < 16) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(TOK_DBG & 4)) __builtin_amdgcn_s_barrier();
        if (c + 2 < 16 && !(TOK_DBG & 8)) issue(c + 2);
        if (c == 1) SPEI_STAMP(p.stamps, 2);
        if (c == 9) SPEI_STAMP(p.stamps, 3);
        if (late && c > 0) { fc2_pre(c - 1); fc2(c - 1); }
        // fc1^T: hidden channels [32 c, 32 c + 32) x tokens
        const unsigned slot = lds0 + (c & (RING - 1)) * CHUNK;
        lds_pre<PD>(fq, slot);
        f32x16 acc1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias1 + 32 * c + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1[4 * g + e] = bv[e];
        }
        __builtin_amdgcn_s_setprio(1);
        lds_pipe<PD, 0>(fq, slot, [&](auto ic, const lp8& a) { acc1 = mfma16(a, xh[decltype(ic)::value], acc1); });
        __builtin_amdgcn_s_setprio(0);
        if (!late) fc2_pre(c);                        // in flight under the GELU
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = (TOK_DBG & 1) ? acc1[r] : gelu1(acc1[r]);
        hk0 = acc_frag<0, LP>(acc1);
        hk1 = acc_frag<1, LP>(acc1);
        if (!late) fc2(c);
    }
    if (late) { fc2_pre(15); fc2(15); }
    SPEI_STAMP(p.stamps, 4);

    // ---- + bias + residual x: the lane holds 4 consecutive output channels of its token per (j, g) -> 16-byte accesses ----
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 res[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                res[jj][g] = *reinterpret_cast<const f32x4*>(p.x + (size_t)tokc * D + 32 * (4 * half + jj) + 8 * g + 4 * h);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int j = 4 * half + jj, n0 = 32 * j + 8 * g + 4 * h;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b2 + n0);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc2[j][4 * g + e] + bv[e] + res[jj][g][e];
                if (tok < p.M) *reinterpret_cast<f32x4*>(p.out + (size_t)tok * D + n0) = o;
            }
    }
    SPEI_STAMP(p.stamps, 5);
}

}  // namespace

template <typename LP>
static int mlp_tok_launch(const float* x, float* out, const void* wstream, const float* b1, const float* b2, int64_t M, hipStream_t st) {
    constexpr int NW = 8;
    MlpTokParams<LP> p;
    p.x = x; p.out = out; p.wstream = (const unsigned char*)wstream; p.b1 = b1; p.b2 = b2; p.M = (int)M;
    p.stamps = spei_stamp_buffer();
    p.dbg = spei_knob("SPEI_TOK_DBG", 0);
    const size_t lds = (size_t)RING * CHUNK + HID * sizeof(float);
    ensure_dyn_lds<&mlp_tok_kernel<LP, NW>>(lds);
    hipLaunchKernelGGL((mlp_tok_kernel<LP, NW>), dim3(cdiv(M, 32 * NW)), dim3(64 * NW), lds, st, p);
    SPEI_CHECK_LAUNCH("spei_mlp_tok16");
    return 0;
}

extern "C" int spei_mlp_tok16(int fmt, const float* x, float* out, const void* wstream, const float* b1, const float* b2, int64_t M,
                              spei_stream_t stream) {
    SPEI_REQUIRE(x && out && wstream && b1 && b2 && M > 0, "spei_mlp_tok16: bad arguments");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_mlp_tok16: fmt=%d", fmt);
    SPEI_REQUIRE(M < (1ll << 31), "spei_mlp_tok16: too many tokens");
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)wstream | (uintptr_t)b1 | (uintptr_t)b2) % 16 == 0, "spei_mlp_tok16: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (fmt == SPEI_F16) return mlp_tok_launch<_Float16>(x, out, wstream, b1, b2, M, st);
    return mlp_tok_launch<__bf16>(x, out, wstream, b1, b2, M, st);
}

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


// =====================================================================================================================================
//     spei_attn_tok16:   out = x + proj( W-MSA( q = yhat Wq,  [k, v] = LayerNorm(x) Wkv ) )      (reference model/swinir.py:238-278, :115-149)
//
// A wave owns ONE 5x5 window (25 of its 32 lanes) from the first load to the last store.  Weight stream (pack.AttnStreamW), 16 chunks
// of 32 fragments:  0-3 Wq (two heads each) | 4-11 head h: Wk_h, Wv_h | 12-15 Wproj (two 32-channel output tiles each, input
// channels in accumulator order).  Per head everything stays in registers, as in attn_fused16.hip: Q^T_h, K^T_h (weights = A operand)
// and V_h (tokens = A operand), S^T = K Q^T and O^T = V^T P^T straight from the accumulators; new here: the operands are the lane's
// own LayerNorm / y-hat fragments (no LDS slab, no staging barrier), O^T_h is handed to the projection as its B operand (no LDS
// exchange between waves), and the weights arrive through the shared ring.
template <typename LP>
struct AttnTokParams {
    const float* x;
    float* out;
    const LP* yhat;                 // [M][256]
    const unsigned char* wstream;   // 16 chunks x 32 fragments
    const float* bq;                // [256] (scale folded)
    const float* bkv;               // [512]
    const float* bproj;             // [256]
    const float* relb;              // [8][25][28]: relative position bias, key axis padded to 28 (pack.AttnStreamW)
    long long* stamps;
    int H, W, shift, nwin;
    int dbg;
};

constexpr int RELB_FLOATS = 8 * 25 * 28;

__device__ __forceinline__ void wait_vm(int n) {        // s_waitcnt vmcnt(n) for the counts this file uses (n is a constant after unrolling)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ int mask_region(int v, int n, int shift) { return v < n - 5 ? 0 : (v < n - shift ? 1 : 2); }

template <typename LP, int NW>
__global__ __launch_bounds__(64 * NW) void attn_tok_kernel(const AttnTokParams<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    static_assert(NW == 8 || NW == 4, "ring shares: 32 fragments per chunk over NW waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // ring | relb | bq | bkv | bproj
    float* relb = reinterpret_cast<float*>(smem + RING * CHUNK);
    float* sbq = relb + RELB_FLOATS;
    float* sbkv = sbq + 256;
    float* sbp = sbkv + 512;
    constexpr int PER = 32 / NW;
    constexpr int PD = 3;
    constexpr int NCH = 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, h = lane >> 5;
    const unsigned char* wsrc = p.wstream + lane * 16;
    auto issue = [&](int c) {
        glds_run<PER>(wsrc + (size_t)(c * 32 + wave * PER) * FRAG, smem + (c & (RING - 1)) * CHUNK + wave * PER * FRAG);
    };
    SPEI_STAMP(p.stamps, 0);
    issue(0);
    issue(1);
    for (int i = tid; i < RELB_FLOATS; i += 64 * NW) relb[i] = p.relb[i];
    for (int i = tid; i < 256; i += 64 * NW) { sbq[i] = p.bq[i]; sbp[i] = p.bproj[i]; }
    for (int i = tid; i < 512; i += 64 * NW) sbkv[i] = p.bkv[i];

    // ---- the lane's token: window `win`, position fr of its 25 (lanes 25..31 idle: zero operands, no stores) ----------------
    const int nwx = p.W / 5;
    const int win = blockIdx.x * NW + wave;
    const bool wok = win < p.nwin;
    const int wc = wok ? win : p.nwin - 1;
    const int wy = wc / nwx, wx = wc - wy * nwx;
    const int tq = fr < 25 ? fr : 0;
    const int ysf = wy * 5 + tq / 5, xsf = wx * 5 + tq % 5;              // shifted-frame coordinates
    int yo = ysf + p.shift, xo = xsf + p.shift;                          // roll(-shift): shifted[y] = x[(y + shift) % H]
    if (yo >= p.H) yo -= p.H;
    if (xo >= p.W) xo -= p.W;
    const size_t pix = (size_t)yo * p.W + xo;
    const bool tok = wok && fr < 25;
    // shift mask: bit r = the key of accumulator row r lies in another region than the lane's query (model/swinir.py:215-236)
    unsigned mbits = 0u;
    if (p.shift > 0) {
        const int qreg = 3 * mask_region(ysf, p.H, p.shift) + mask_region(xsf, p.W, p.shift);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = min((r & 3) + 8 * (r >> 2) + 4 * h, 24);
            const int kreg = 3 * mask_region(wy * 5 + key / 5, p.H, p.shift) + mask_region(wx * 5 + key % 5, p.W, p.shift);
            if (kreg != qreg) mbits |= 1u << r;
        }
    }

    // ---- y-hat fragments: the lane's 16 B of every k-step, straight from HBM (B operand of Q^T = Wq y^T) -------------------------
    lp8 yh[16];
    {
        const LP* yp = p.yhat + pix * D + 8 * h;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            yh[s] = *reinterpret_cast<const lp8*>(yp + 16 * s);
            if (!tok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) yh[s][j] = (LP)0.0f;
            }
        }
    }
    lp8 fq[PD];
    const unsigned lds0 = (unsigned)(uintptr_t)smem + lane * 16;
    lp8 qo[8][2];                                   // Q^T_h as the B operand of S^T, later O^T_h as the B operand of the projection
    lp8 xh[16];                                     // LayerNorm(x) of the lane's token: B operand of K^T, A operand of V
    auto bias_rows = [&](const float* b32) {        // accumulator initialised with a per-ROW bias (rows = 32 channels at b32)
        f32x16 a;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b32 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * g + e] = bv[e];
        }
        return a;
    };

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // own share of chunk c landed; everybody's; everybody done with chunk c - 2's slot.  VM operations issued after chunk c's
        // LDS-DMA that may stay in flight: chunk c + 1's, plus the stores (and their residual loads) of a projection chunk
        wait_vm(c + 1 < NCH ? PER + (c >= 13 ? 16 : 0) : (c >= 13 ? 16 : 0));
        if (!(TOK_DBG & 4)) __builtin_amdgcn_s_barrier();
        if (c + 2 < NCH) issue(c + 2);
        const unsigned slot = lds0 + (c & (RING - 1)) * CHUNK;
        if (c < 4) {
            // ---- Q^T of heads 2c, 2c + 1 ---------------------------------------------------------------------------------
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int head = 2 * c + hh;
                f32x16 acc = bias_rows(sbq + head * 32);
                lds_pre<PD>(fq, slot + hh * 16 * FRAG);
                __builtin_amdgcn_s_setprio(1);
                lds_pipe<PD, 0>(fq, slot + hh * 16 * FRAG, [&](auto ic, const lp8& a) { acc = mfma16(a, yh[decltype(ic)::value], acc); });
                __builtin_amdgcn_s_setprio(0);
                qo[head][0] = acc_frag<0, LP>(acc);
                qo[head][1] = acc_frag<1, LP>(acc);
            }
            if (c == 3) {
                // ---- LayerNorm(256) of the lane's token (the y-hat registers are free now) ------------------------------------
                SPEI_STAMP(p.stamps, 1);
                // two sweeps over the lane's half row, 4 k-steps (32 registers) at a time — the registers of a whole fp32 row on top of
                // the Q fragments do not exist: first the moments (var = E[x^2] - mean^2 in fp32), then normalise and pack; the second
                // sweep is served by L2
                const float* xp = p.x + pix * D + 8 * h;
                float sum = 0.f, sq = 0.f;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    f32x4 xr[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) xr[u] = *reinterpret_cast<const f32x4*>(xp + 64 * s4 + 16 * (u >> 1) + 4 * (u & 1));
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        sum += (xr[u][0] + xr[u][1]) + (xr[u][2] + xr[u][3]);
                        sq += (xr[u][0] * xr[u][0] + xr[u][1] * xr[u][1]) + (xr[u][2] * xr[u][2] + xr[u][3] * xr[u][3]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float mean = xor_combine<32, OpSum>(sum) * (1.0f / 256.0f);
                const float var = fmaxf(xor_combine<32, OpSum>(sq) * (1.0f / 256.0f) - mean * mean, 0.f);
                const float rstd = tok ? 1.0f / sqrtf(var + 1e-5f) : 0.f;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    f32x4 xr[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) xr[u] = *reinterpret_cast<const f32x4*>(xp + 64 * s4 + 16 * (u >> 1) + 4 * (u & 1));
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) xh[4 * s4 + (u >> 1)][4 * (u & 1) + e] = to_lp<LP>((xr[u][e] - mean) * rstd);
                    __builtin_amdgcn_sched_barrier(0);
                }
                SPEI_STAMP(p.stamps, 2);
            }
        } else if (c < 12) {
            // ---- head c - 4: K^T, V, then the window's attention, all in registers ---------------------------------------------
            const int head = c - 4;
            f32x16 kT = bias_rows(sbkv + head * 32);
            lds_pre<PD>(fq, slot);
            __builtin_amdgcn_s_setprio(1);
            lds_pipe<PD, 0>(fq, slot, [&](auto ic, const lp8& a) { kT = mfma16(a, xh[decltype(ic)::value], kT); });
            __builtin_amdgcn_s_setprio(0);
            // S^T[key][query] = sum_d K^T[d][key] Q^T[d][query]
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            st = mfma16(acc_frag<0, LP>(kT), qo[head][0], st);
            st = mfma16(acc_frag<1, LP>(kT), qo[head][1], st);
            const float* rbp = relb + (head * 25 + tq) * 28 + 4 * h;
            float mx = -INFINITY;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 rb = *reinterpret_cast<const f32x4*>(rbp + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
                    float v = -INFINITY;
                    if (key < 25) {
                        v = st[r] + rb[e];
                        if (mbits >> r & 1) v += -100.0f;
                    }
                    st[r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(st[r] - mx);
                st[r] = e;
                sum += e;
            }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] *= inv;
            const lp8 pf0 = acc_frag<0, LP>(st), pf1 = acc_frag<1, LP>(st);       // P^T packed before V is computed: K^T, S^T are dead by then
            f32x16 vv;
            {
                const float bv = sbkv[256 + head * 32 + fr];                  // V: tokens on the rows, channel d = the lane's column
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r] = bv;
            }
            lds_pre<PD>(fq, slot + 16 * FRAG);
            __builtin_amdgcn_s_setprio(1);
            lds_pipe<PD, 0>(fq, slot + 16 * FRAG, [&](auto ic, const lp8& a) { vv = mfma16(xh[decltype(ic)::value], a, vv); });
            __builtin_amdgcn_s_setprio(0);
            // O^T[d][query] = sum_key V[key][d] P^T[key][query]
            f32x16 ot;
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[r] = 0.f;
            ot = mfma16(acc_frag<0, LP>(vv), pf0, ot);
            ot = mfma16(acc_frag<1, LP>(vv), pf1, ot);
            qo[head][0] = acc_frag<0, LP>(ot);
            qo[head][1] = acc_frag<1, LP>(ot);
            if (c == 11) SPEI_STAMP(p.stamps, 3);
        } else {
            // ---- projection, output channels 64 (c - 12) .. +64: out^T = Wproj O^T, + bias + residual x, 16-byte accesses ----------
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * (c - 12) + jj;
                f32x16 acc = bias_rows(sbp + j * 32);
                lds_pre<PD>(fq, slot + jj * 16 * FRAG);
                __builtin_amdgcn_s_setprio(1);
                lds_pipe<PD, 0>(fq, slot + jj * 16 * FRAG, [&](auto ic, const lp8& a) {
                    constexpr int i = decltype(ic)::value;
                    acc = mfma16(a, qo[i >> 1][i & 1], acc);
                });
                __builtin_amdgcn_s_setprio(0);
                f32x4 res[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) res[g] = *reinterpret_cast<const f32x4*>(p.x + pix * D + 32 * j + 8 * g + 4 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[4 * g + e] + res[g][e];
                    if (tok) *reinterpret_cast<f32x4*>(p.out + pix * D + 32 * j + 8 * g + 4 * h) = o;
                }
            }
        }
    }
    SPEI_STAMP(p.stamps, 4);
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

template <typename LP>
static int attn_tok_launch(const float* x, float* out, const void* yhat, const void* wstream, const float* bq, const float* bkv,
                           const float* bproj, const float* relb, int H, int W, int shift, hipStream_t st) {
    constexpr int NW = 8;
    AttnTokParams<LP> p;
    p.x = x; p.out = out; p.yhat = (const LP*)yhat; p.wstream = (const unsigned char*)wstream; p.bq = bq; p.bkv = bkv; p.bproj = bproj;
    p.relb = relb; p.H = H; p.W = W; p.shift = shift; p.nwin = (H / 5) * (W / 5);
    p.stamps = spei_stamp_buffer();
    p.dbg = spei_knob("SPEI_TOK_DBG", 0);
    const size_t lds = (size_t)RING * CHUNK + (RELB_FLOATS + 1024) * sizeof(float);
    ensure_dyn_lds<&attn_tok_kernel<LP, NW>>(lds);
    hipLaunchKernelGGL((attn_tok_kernel<LP, NW>), dim3(cdiv(p.nwin, NW)), dim3(64 * NW), lds, st, p);
    SPEI_CHECK_LAUNCH("spei_attn_tok16");
    return 0;
}

extern "C" int spei_attn_tok16(int fmt, const float* x, float* out, const void* yhat, const void* wstream, const float* bq, const float* bkv,
                               const float* bproj, const float* relb28, int H, int W, int shift, spei_stream_t stream) {
    SPEI_REQUIRE(x && out && yhat && wstream && bq && bkv && bproj && relb28, "spei_attn_tok16: null pointer");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_attn_tok16: fmt=%d", fmt);
    SPEI_REQUIRE(H > 0 && W > 0 && H % 5 == 0 && W % 5 == 0, "spei_attn_tok16: %dx%d is not a multiple of the 5x5 window", H, W);
    SPEI_REQUIRE(shift >= 0 && shift < 5, "spei_attn_tok16: shift=%d", shift);
    SPEI_REQUIRE((int64_t)H * W < (1ll << 30), "spei_attn_tok16: map too large");
    SPEI_REQUIRE(out != x, "spei_attn_tok16: out must not alias x (a window's residual rows are re-read after other windows have stored theirs)");
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)yhat | (uintptr_t)wstream | (uintptr_t)relb28) % 16 == 0, "spei_attn_tok16: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (fmt == SPEI_F16) return attn_tok_launch<_Float16>(x, out, yhat, wstream, bq, bkv, bproj, relb28, H, W, shift, st);
    return attn_tok_launch<__bf16>(x, out, yhat, wstream, bq, bkv, bproj, relb28, H, W, shift, st);
}

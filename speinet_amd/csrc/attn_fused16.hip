// Fused attention branch of one cross-window Swin block on the gfx950 16-bit matrix pipe (bf16 or half operands):
//
//     x <- x + proj( W-MSA( q = norm1(y) Wq,  [k, v] = norm1(x) Wkv ) )
//
// (reference model/swinir.py:238-278 SwinTransformerBlock.forward up to the first residual, :115-149
// WindowAttention.forward, :215-236 calculate_mask, :32-61 window_partition/reverse, torch.roll for the shift).
// Unfused this is LayerNorm -> two GEMMs -> attention -> GEMM: q, k, v and the attention output each make a round trip
// through HBM.  Here a 256-thread workgroup owns TWO 5x5 windows (50 tokens, padded to 2 x 32 rows) and each of its four
// waves owns two of the eight heads; two workgroups share a CU (2 x 68 KB of LDS, <= 256 VGPRs), so one workgroup's HBM
// phases (token staging, residual read, stores) overlap the other's MFMA phases:
//   1. stage x (fp32 -> LayerNorm without affine -> bf16) and the pre-normalised y (bf16) of the 50 tokens into LDS, 16
//      lanes per token (row reductions are 4 DPP steps); the cyclic shift and the window partition are the token -> pixel map;
//   2. per head: Q^T_h = Wq_h y^T, K^T_h = Wk_h x^T (weights as the MFMA A operand, tokens on the lanes), V_h = x Wv_h^T
//      (tokens on the accumulator rows); weights stream from L2 in fragment order;
//   3. S^T = K Q^T and O^T = V^T P^T WITHOUT leaving registers: a 32x32 f32 accumulator X is a valid bf16 MFMA operand
//      after a cvt of registers 8s..8s+7 (as A it yields X^T.B, as B it yields A.X, with the SAME permuted k order on both
//      sides), so K^T/Q^T feed S^T directly and V / P^T feed O^T; softmax is 16 registers + one cross-half shuffle per query;
//   4. O^T -> bf16 [token][256] slab in LDS (over the y slab; the only cross-wave exchange), proj GEMM (wave = 64 output
//      channels), + bias + residual (prefetched before the GEMM), 16-byte stores.
// HBM traffic per block: x read (+ once more, mostly from L2, for the residual), y-hat read, x written.
#include "common.h"
#include <type_traits>

namespace {

constexpr int D = 256, HD = 32, NT = 25, WS = 5;
constexpr int PA = 2 * D + 16;        // LDS row pitch (bytes)
constexpr int ROWS = 64;              // 2 windows x 32 (25 tokens + 7 pad rows)

template <typename LP>       // LP: __bf16 or _Float16 (tokens, weights, probabilities as matrix-pipe operands)
struct AttnParams {
    const float* x;
    float* out;
    const LP* yhat;       // [M][256]
    const LP* wq;         // fragment order [8][1][16][64][8]
    const float* bq;
    const LP* wkv;        // fragment order [16][1][16][64][8]  (n-tiles 0..7 = K heads, 8..15 = V heads)
    const float* bkv;
    const LP* wproj;      // fragment order [8][1][16][64][8]
    const float* bproj;
    const float* relbias; // [8][25][25]
    long long* stamps;    // tuning build: phase stamps, else NULL
    int H, W, shift, nwin;
};

__device__ __forceinline__ int mask_region(int v, int n, int shift) { return v < n - WS ? 0 : (v < n - shift ? 1 : 2); }

template <int S, typename LP>
__device__ __forceinline__ typename lpv<LP>::x8 cvt8(const f32x16& a) {
    typename lpv<LP>::x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = to_lp<LP>(a[8 * S + j]);
    return r;
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) { return v + dpp_quad<CTRL>(v); }
// all-reduce over each aligned group of 16 lanes: xor 1, xor 2 (quad_perm), then row_half_mirror / row_mirror, which pair
// lanes of different quads / octets (the values are already uniform inside them)
__device__ __forceinline__ float sum16(float v) { return dpp_add<0x140>(dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(v)))); }

// HV = 1: a 256-thread workgroup owns two windows, two workgroups share a CU.  HV = 2: a 512-thread workgroup runs two such
// halves side by side (own LDS slabs, own windows, common barriers): both halves walk the weight stream in lockstep with the SAME
// rotation, so the second half's fragment loads hit the lines the first half just brought into the CU's L1.
template <typename LP, int HV>
__global__ __launch_bounds__(256 * HV, HV == 1 ? 2 : 1) void attn_fused_kernel(const AttnParams<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    typedef typename lpv<LP>::x4 lp4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int hv = HV == 1 ? 0 : (int)(threadIdx.x >> 8);
    unsigned char* smem = smem_all + (size_t)hv * (2 * ROWS * PA + 2 * ROWS * sizeof(int));
    const int wg = blockIdx.x * HV + hv;           // index of this half among all 256-thread halves
    unsigned char* xs = smem;                      // [64][PA]  LayerNorm(x), bf16
    unsigned char* ys = smem + ROWS * PA;          // [64][PA]  y-hat, bf16; later the attention output
    unsigned char* os = ys;
    int* tok_pix = reinterpret_cast<int*>(smem + 2 * ROWS * PA);   // [64] pixel index or -1
    int* tok_reg = tok_pix + ROWS;                                 // [64] shift-mask region id

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int nwx = p.W / WS;

    SPEI_STAMP(p.stamps, 0);
    if (tid < ROWS) {
        const int w = tid >> 5, t = tid & 31;
        const int win = wg * 2 + w;
        int pix = -1, reg = 0;
        if (t < NT && win < p.nwin) {
            const int wy = win / nwx, wx = win - wy * nwx;
            const int ysf = wy * WS + t / WS, xsf = wx * WS + t % WS;      // shifted-frame coordinates
            int yo = ysf + p.shift, xo = xsf + p.shift;                    // roll(-shift): shifted[y] = x[(y+shift) % H]
            if (yo >= p.H) yo -= p.H;
            if (xo >= p.W) xo -= p.W;
            pix = yo * p.W + xo;
            reg = p.shift > 0 ? 3 * mask_region(ysf, p.H, p.shift) + mask_region(xsf, p.W, p.shift) : 0;
        }
        tok_pix[tid] = pix;
        tok_reg[tid] = reg;
    }
    __syncthreads();

    // ---- 1. stage LN(x) and y-hat: 16 lanes per token, 16 tokens per pass ------------------------------------------
    {
        const int l16 = tid & 15, rsub = tid >> 4;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 xr[2][4];
            u32x4 yr[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int pix = max(tok_pix[(half * 2 + b) * 16 + rsub], 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) xr[b][j] = reinterpret_cast<const f32x4*>(p.x + (size_t)pix * D)[l16 + 16 * j];
#pragma unroll
                for (int j = 0; j < 2; ++j) yr[b][j] = reinterpret_cast<const u32x4*>(p.yhat + (size_t)pix * D)[l16 + 16 * j];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int r = (half * 2 + b) * 16 + rsub;
                const bool ok = tok_pix[r] >= 0;
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
                const float rstd = ok ? 1.0f / sqrtf(sum16(ss) * (1.0f / 256.0f) + 1e-5f) : 0.f;     // empty rows stage zeros
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lp4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = to_lp<LP>(xr[b][j][e] * rstd);
                    *reinterpret_cast<lp4*>(xs + r * PA + (l16 + 16 * j) * 8) = hv;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<u32x4*>(ys + r * PA + (l16 + 16 * j) * 16) = ok ? yr[b][j] : u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
    __syncthreads();
    SPEI_STAMP(p.stamps, 1);

    // ---- 2 + 3. the wave's two heads: Q^T, K^T, V, then attention on both windows, all in registers -------------------
    lp4 opk[2][2][4];                               // [head][window][4 d-groups]: O^T packed, written to LDS after the barrier
    // shift mask: bit r of mbits[w] = key (register row r) lies in another region than the lane's query
    unsigned mbits[2] = {0u, 0u};
    if (p.shift > 0) {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int qreg = tok_reg[w * 32 + (fr < NT ? fr : 0)];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (tok_reg[w * 32 + min(key, NT - 1)] != qreg) mbits[w] |= 1u << r;
            }
        }
    }
    const int rot = blockIdx.x & 15;                   // per-workgroup K rotation (spreads the L2 channel load)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = wave * 2 + hh;
        const int qi = fr < NT ? fr : 0;
        float rb[16];                                  // relative-position bias of (query = lane column, key = register row)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
            rb[r] = p.relbias[(h * NT + qi) * NT + min(key, NT - 1)];
        }
        f32x16 qT[2], kT[2], vv[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * fk;
            const float bqv = p.bq[h * HD + d], bkv = p.bkv[h * HD + d], bvv = p.bkv[D + h * HD + fr];
            qT[0][r] = qT[1][r] = bqv;
            kT[0][r] = kT[1][r] = bkv;
            vv[0][r] = vv[1][r] = bvv;
        }
        {
            const LP* wqp = p.wq + (size_t)h * 16 * 512 + lane * 8;
            const LP* wkp = p.wkv + (size_t)h * 16 * 512 + lane * 8;
            const LP* wvp = p.wkv + (size_t)(8 + h) * 16 * 512 + lane * 8;
            // Software pipeline pinned with full scheduling barriers: weight fragments two k-steps ahead (3-slot ring), token
            // fragments one step ahead, then the 6 MFMAs of the step.  Left alone the scheduler sinks each global_load to
            // right before its use (s_waitcnt vmcnt(0) in front of every MFMA group: seen in the ISA).
            lp8 wqf[3], wkf[3], wvf[3];
#pragma unroll
            for (int pre = 0; pre < 2; ++pre) {
                wqf[pre] = *reinterpret_cast<const lp8*>(wqp + ((rot + pre) & 15) * 512);
                wkf[pre] = *reinterpret_cast<const lp8*>(wkp + ((rot + pre) & 15) * 512);
                wvf[pre] = *reinterpret_cast<const lp8*>(wvp + ((rot + pre) & 15) * 512);
            }
            lp8 yn[2], xn[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                yn[w] = *reinterpret_cast<const lp8*>(ys + (w * 32 + fr) * PA + rot * 32 + fk * 16);
                xn[w] = *reinterpret_cast<const lp8*>(xs + (w * 32 + fr) * PA + rot * 32 + fk * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cur = i % 3;
                lp8 yc[2], xc[2];
#pragma unroll
                for (int w = 0; w < 2; ++w) { yc[w] = yn[w]; xc[w] = xn[w]; }
                if (i + 1 < 16) {
                    const int ko = ((rot + i + 1) & 15) * 32 + fk * 16;
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        yn[w] = *reinterpret_cast<const lp8*>(ys + (w * 32 + fr) * PA + ko);
                        xn[w] = *reinterpret_cast<const lp8*>(xs + (w * 32 + fr) * PA + ko);
                    }
                }
                const lp8 wqc = wqf[cur], wkc = wkf[cur], wvc = wvf[cur];
                if (i + 2 < 16) {
                    const int nxt = (i + 2) % 3, ksn = (rot + i + 2) & 15;
                    wqf[nxt] = *reinterpret_cast<const lp8*>(wqp + ksn * 512);
                    wkf[nxt] = *reinterpret_cast<const lp8*>(wkp + ksn * 512);
                    wvf[nxt] = *reinterpret_cast<const lp8*>(wvp + ksn * 512);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    qT[w] = mfma16(wqc, yc[w], qT[w]);
                    kT[w] = mfma16(wkc, xc[w], kT[w]);
                    vv[w] = mfma16(xc[w], wvc, vv[w]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        SPEI_STAMP(p.stamps, 2 + 2 * hh);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            // S^T[key][query] = sum_d K^T[d][key] Q^T[d][query]:  A = (K^T)^T from the accumulator, B = Q^T from the accumulator
            st = mfma16(cvt8<0, LP>(kT[w]), cvt8<0, LP>(qT[w]), st);
            st = mfma16(cvt8<1, LP>(kT[w]), cvt8<1, LP>(qT[w]), st);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
                float v = -INFINITY;
                if (key < NT) {
                    v = st[r] + rb[r];
                    if (mbits[w] >> r & 1) v += -100.0f;
                }
                st[r] = v;
                mx = fmaxf(mx, v);
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
            // O^T[d][query] = sum_key V[key][d] P^T[key][query]:  A = V^T from the accumulator (X^T.B form), B = P^T
            f32x16 ot;
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[r] = 0.f;
            ot = mfma16(cvt8<0, LP>(vv[w]), cvt8<0, LP>(st), ot);
            ot = mfma16(cvt8<1, LP>(vv[w]), cvt8<1, LP>(st), ot);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) opk[hh][w][g][e] = to_lp<LP>(ot[4 * g + e]);
        }
        SPEI_STAMP(p.stamps, 3 + 2 * hh);
    }
    __syncthreads();                                   // every wave is done reading the y slab
    SPEI_STAMP(p.stamps, 6);
    // rows d = (r&3) + 8*(r>>2) + 4*fk = 8g + 4fk + e, column = query token fr  ->  os[token][h*32 + d]
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<lp4*>(os + (w * 32 + fr) * PA + ((wave * 2 + hh) * HD + 8 * g + 4 * fk) * 2) = opk[hh][w][g];
    __syncthreads();
    SPEI_STAMP(p.stamps, 7);

    // ---- 4. proj: the wave produces output channels [64 wave, 64 wave + 64) for all 64 rows, + bias + residual ---------
    {
        const int et = fr & 3, ecol = (fr >> 2) * 4;
        f32x4 res[2][2][4];                            // residual x, fetched under the GEMM
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pix = max(tok_pix[i * 32 + 8 * k + 4 * fk + et], 0);
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
                    res[i][nn][k] = *reinterpret_cast<const f32x4*>(p.x + (size_t)pix * D + (wave * 2 + nn) * HD + ecol);
            }
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][nn][r] = 0.f;
        const LP* wpp = p.wproj + (size_t)(wave * 2) * 16 * 512 + lane * 8;
        const int rot4 = (blockIdx.x * 3) & 15;
        lp8 wf[3][2];
        int ks = rot4;
#pragma unroll
        for (int pre = 0; pre < 2; ++pre)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) wf[pre][nn] = *reinterpret_cast<const lp8*>(wpp + (nn * 16 + ((ks + pre) & 15)) * 512);
        lp8 an[2];
        an[0] = *reinterpret_cast<const lp8*>(os + fr * PA + ks * 32 + fk * 16);
        an[1] = *reinterpret_cast<const lp8*>(os + (32 + fr) * PA + ks * 32 + fk * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cur = i % 3, nxt = (i + 2) % 3;
            const lp8 a0 = an[0], a1 = an[1];
            if (i + 1 < 16) {
                const int ko = ((ks + 1) & 15) * 32 + fk * 16;
                an[0] = *reinterpret_cast<const lp8*>(os + fr * PA + ko);
                an[1] = *reinterpret_cast<const lp8*>(os + (32 + fr) * PA + ko);
            }
            const lp8 w0 = wf[cur][0], w1 = wf[cur][1];
            if (i + 2 < 16) {
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) wf[nxt][nn] = *reinterpret_cast<const lp8*>(wpp + (nn * 16 + ((ks + 2) & 15)) * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = mfma16(a0, w0, acc[0][0]);
            acc[1][0] = mfma16(a1, w0, acc[1][0]);
            acc[0][1] = mfma16(a0, w1, acc[0][1]);
            acc[1][1] = mfma16(a1, w1, acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            ks = (ks + 1) & 15;
        }
        SPEI_STAMP(p.stamps, 8);
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            const float bias = p.bproj[(wave * 2 + nn) * HD + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = acc[i][nn][4 * k + e] + bias;
                    quad_transpose4(a[0], a[1], a[2], a[3], et);
                    const int pix = tok_pix[i * 32 + 8 * k + 4 * fk + et];
                    if (pix >= 0)
                        *reinterpret_cast<f32x4*>(p.out + (size_t)pix * D + (wave * 2 + nn) * HD + ecol) = f32x4{a[0], a[1], a[2], a[3]} + res[i][nn][k];
                }
            }
        }
        SPEI_STAMP(p.stamps, 9);
    }
}

// ---- round 4: four windows per workgroup, one token slab, a batch of maps per launch -------------------------------------------------
// `attn_fused_kernel` streams the block's 512 KB of weight fragments once per TWO windows, and a CU takes in only ~30 B/clk from L2: its
// Q / K / V phase runs at the intake rate, not the matrix pipe's (stamps: 11.7 of a workgroup's 28.6 us; 1152 workgroups x 512 KB =
// 590 MB of weight stream per call).  Here a 256-thread workgroup owns FOUR windows, so every fragment it fetches feeds four MFMAs
// (256 B of intake per MFMA: the rate the pipe can take), and it still shares its CU with a second workgroup, because it keeps ONE
// token slab of 100 unpadded rows (52.8 KB) instead of two padded ones:
//   1. y-hat rows -> slab;  Q^T of the wave's two heads (weights as the A operand), kept as packed 16-bit MFMA operands (64 registers);
//   2. x rows -> LayerNorm -> the SAME slab;  per head: K^T pass -> S^T = K Q^T -> softmax -> P^T packed;  V pass -> O^T = V^T P^T
//      (accumulators feed the next MFMA as operands, as in attn_fused_kernel; Q, K, V passes are separate so that one 4-window
//      accumulator set (64 registers) is live at a time);
//   3. O -> the slab;  proj with accumulators that START from x + bias (the residual is loaded straight into them), 16-byte stores.
// A window's 32-row MFMA tile reads the 7 rows that follow its 25 tokens from the next window (finite data; keys >= 25 are masked,
// queries >= 25 are never stored), the last tile clamps to row 99.  Launch: gridDim.x = batch x ceil(windows / 4): the two Swin calls
// of a frame (same weights, model/speinet.py:84) run as ONE launch over stacked maps — 1152 workgroups on 512 slots instead of
// 2 x 576.
constexpr int WPG = 4;                // windows per workgroup
constexpr int TOK = WPG * NT;         // 100 slab rows

// a value the compiler must treat as unknown until this point: keeps it from computing (and keeping alive through the whole kernel)
// the per-lane addresses of every later phase's bias / relative-position loads right at the top (lessons)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

template <typename LP>
struct Attn4Params {
    const float* x;       // [batch][H*W][256]
    float* out;
    const LP* yhat;
    const LP* wq; const float* bq; const LP* wkv; const float* bkv; const LP* wproj; const float* bproj; const float* relbias;
    long long* stamps;
    int H, W, shift, nwin, groups, batch;     // nwin: windows per map; groups = ceil(nwin / 4)
};

// Eight waves, one head each (two waves per SIMD), one workgroup per CU.  A first version with four waves (two heads each, two
// workgroups per CU) needed 276 registers for the Q^T / O^T it carries across the slab's re-use and ran 168 us for two 720p maps; this
// one needs 224 with eight weight fragments in flight per wave: 158 us, and 137 us with the loads of a phase issued one phase early.
template <typename LP>
__global__ __launch_bounds__(512, 2) void attn_win4_kernel(const Attn4Params<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    typedef typename lpv<LP>::x4 lp4;
    constexpr int TPP = 32;                           // tokens per staging pass (16 lanes per token, 512 threads)
    constexpr int NPASS = (TOK + TPP - 1) / TPP;      // 4
    constexpr int RQ = 8;                             // weight fragments in flight per wave, Q / K / V passes
    constexpr int RP = 6;                             // ... projection
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* slab = smem;                                           // [100][PA]: y-hat, then LN(x), then the attention output
    int* tok_pix = reinterpret_cast<int*>(smem + TOK * PA);               // [100] pixel index within the map, or -1
    int* tok_reg = tok_pix + TOK;                                         // [100] shift-mask region id
    float* sbias = reinterpret_cast<float*>(tok_reg + TOK);               // [1024] bq | bk | bv | bproj

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int nwx = p.W / WS;
    const int bmap = blockIdx.x / p.groups, grp = blockIdx.x - bmap * p.groups;
    const size_t moff = (size_t)bmap * p.H * p.W * D;
    const float* xg = p.x + moff;
    float* og = p.out + moff;
    const LP* yg = p.yhat + moff;

    SPEI_STAMP(p.stamps, 0);
    if (tid < TOK) {
        const int wd = tid / NT, t = tid - wd * NT;
        const int win = grp * WPG + wd;
        int pix = -1, reg = 0;
        if (win < p.nwin) {
            const int wy = win / nwx, wx = win - wy * nwx;
            const int ysf = wy * WS + t / WS, xsf = wx * WS + t % WS;      // shifted-frame coordinates
            int yo = ysf + p.shift, xo = xsf + p.shift;                    // roll(-shift): shifted[y] = x[(y+shift) % H]
            if (yo >= p.H) yo -= p.H;
            if (xo >= p.W) xo -= p.W;
            pix = yo * p.W + xo;
            reg = p.shift > 0 ? 3 * mask_region(ysf, p.H, p.shift) + mask_region(xsf, p.W, p.shift) : 0;
        }
        tok_pix[tid] = pix;
        tok_reg[tid] = reg;
    }
    // biases -> LDS: an accumulator's 16 bias values then come from 4 ds_read_b128 at the top of a pass instead of 16 dependent
    // global loads (one exposed L2 round trip per pass)
    sbias[tid] = tid < D ? p.bq[tid] : p.bkv[tid - D];
    sbias[512 + tid] = tid < D ? p.bkv[D + tid] : p.bproj[tid - D];
    lds_barrier();

    const int l16 = tid & 15, rsub = tid >> 4;        // staging: 16 lanes per token, 32 tokens per pass
    // Per-workgroup K rotation (spreads the L2 channel load of the weight stream; by group within the map, so that a map's result does
    // not depend on its place in the batch): GEMM step i contracts k-step (rot + i) & 15.  The slab rows are STORED rotated by the same
    // amount, so step i reads byte offset 32 i of a row — an immediate of the ds_read.  With the rotation in the read address instead,
    // the 64 (window, step) addresses are common to all passes and the compiler keeps them in registers from the first pass to
    // the last (64 VGPRs).
    const int rot = grp & 15, rotb = rot * 32;
    // ---- 1a. ALL of the workgroup's token loads are issued here: y-hat (consumed now) and x (consumed after the Q pass: its HBM round
    // trip runs under the Q pass instead of after it; 64 registers that the Q pass can spare) -------------------------------------------
    f32x4 xr[NPASS][4];
    {
        u32x4 yr[NPASS][2];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = min(ps * TPP + rsub, TOK - 1);
            const int pix = max(tok_pix[r], 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) yr[ps][j] = reinterpret_cast<const u32x4*>(yg + (size_t)pix * D)[l16 + 16 * j];
        }
        __builtin_amdgcn_sched_barrier(0);            // y first, then x: a y load sunk behind the x loads makes its vmcnt wait drain them all
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = min(ps * TPP + rsub, TOK - 1);
            const int pix = max(tok_pix[r], 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[ps][j] = reinterpret_cast<const f32x4*>(xg + (size_t)pix * D)[l16 + 16 * j];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * TPP + rsub;
            if (r < TOK) {
                const bool ok = tok_pix[r] >= 0;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<u32x4*>(slab + r * PA + (((l16 + 16 * j) * 16 - rotb) & 511)) = ok ? yr[ps][j] : u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
    lds_barrier();
    SPEI_STAMP(p.stamps, 1);

    // slab row of (window wd, tile row fr): the 7 rows past a window's 25 tokens are the next window's first tokens; clamp at the end
    int rowoff[WPG];
#pragma unroll
    for (int wd = 0; wd < WPG; ++wd) rowoff[wd] = min(wd * NT + fr, TOK - 1) * PA + fk * 16;
    const int h = wave;                               // the wave's head

    // One pass of the wave's 4 windows over K = 256: 16 weight fragments from L2 (RQ in flight), token fragments from the slab one step
    // ahead, 4 MFMAs per step; TOKENS_ON_COLUMNS: weights as the A operand (Q^T, K^T), else tokens as A (V).  Pinned with full
    // scheduling barriers (left alone the scheduler sinks every load to right before its use).
    auto pass = [&](const LP* wp, f32x16 (&acc)[WPG], auto tokens_on_columns, auto&& behind_prologue) {
        lp8 ring[RQ];
#pragma unroll
        for (int d = 0; d < RQ; ++d) ring[d] = *reinterpret_cast<const lp8*>(wp + ((rot + d) & 15) * 512);
        behind_prologue();                            // loads that a LATER phase consumes: behind the first RQ fragments in the (in-order) vmcnt queue
        lp8 tn[WPG];
#pragma unroll
        for (int wd = 0; wd < WPG; ++wd) tn[wd] = *reinterpret_cast<const lp8*>(slab + rowoff[wd]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            lp8 tc[WPG];
#pragma unroll
            for (int wd = 0; wd < WPG; ++wd) tc[wd] = tn[wd];
            if (i + 1 < 16) {
#pragma unroll
                for (int wd = 0; wd < WPG; ++wd) tn[wd] = *reinterpret_cast<const lp8*>(slab + rowoff[wd] + (i + 1) * 32);
            }
            const lp8 wc = ring[i % RQ];
            if (i + RQ < 16) ring[i % RQ] = *reinterpret_cast<const lp8*>(wp + ((rot + i + RQ) & 15) * 512);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int wd = 0; wd < WPG; ++wd) {
                if constexpr (decltype(tokens_on_columns)::value) acc[wd] = mfma16(wc, tc[wd], acc[wd]);
                else acc[wd] = mfma16(tc[wd], wc, acc[wd]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // accumulator rows d = 8 g + 4 fk + e of a head <- bias[h * 32 + d]   (Q^T, K^T: head dim on the rows)
    auto init_rows = [&](f32x16 (&acc)[WPG], const float* b) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b + h * HD + 8 * g + 4 * fk);
#pragma unroll
            for (int wd = 0; wd < WPG; ++wd)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[wd][4 * g + e] = bv[e];
        }
    };

    // ---- 1b. Q^T of the wave's head, kept as packed MFMA operands ------------------------------------------------------------------------
    lp8 qp[WPG][2];
    {
        f32x16 acc[WPG];
        init_rows(acc, sbias);
        pass(p.wq + (size_t)h * 16 * 512 + lane * 8, acc, std::true_type{}, [] {});
#pragma unroll
        for (int wd = 0; wd < WPG; ++wd) {
            qp[wd][0] = cvt8<0, LP>(acc[wd]);
            qp[wd][1] = cvt8<1, LP>(acc[wd]);
        }
    }
    // relative-position bias of (query = lane column, key = register row): fetched here, used after the K pass
    float rb[16];
    {
        const int qi = fr < NT ? fr : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
            rb[r] = p.relbias[(h * NT + qi) * NT + min(key, NT - 1)];
        }
    }
    lds_barrier();                                  // every wave is done with the y rows
    SPEI_STAMP(p.stamps, 2);

    // ---- 2a. LayerNorm(x) (loaded at the top) -> the same slab ------------------------------------------------------------------------------
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int r = ps * TPP + rsub;
        const bool ok = r < TOK && tok_pix[min(r, TOK - 1)] >= 0;
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sm += (xr[ps][j][0] + xr[ps][j][1]) + (xr[ps][j][2] + xr[ps][j][3]);
        const float mean = sum16(sm) * (1.0f / 256.0f);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xr[ps][j] -= mean;
            ss += (xr[ps][j][0] * xr[ps][j][0] + xr[ps][j][1] * xr[ps][j][1]) + (xr[ps][j][2] * xr[ps][j][2] + xr[ps][j][3] * xr[ps][j][3]);
        }
        const float rstd = ok ? 1.0f / sqrtf(sum16(ss) * (1.0f / 256.0f) + 1e-5f) : 0.f;     // empty rows stage zeros
        if (r < TOK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lp4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = to_lp<LP>(xr[ps][j][e] * rstd);
                *reinterpret_cast<lp4*>(slab + r * PA + (((l16 + 16 * j) * 8 - rotb) & 511)) = hv;
            }
        }
    }
    // shift mask: bit r of mbits[wd] = key (register row r) lies in another region than the lane's query
    unsigned mbits[WPG];
#pragma unroll
    for (int wd = 0; wd < WPG; ++wd) {
        mbits[wd] = 0u;
        if (p.shift > 0) {
            const int qreg = tok_reg[wd * NT + (fr < NT ? fr : 0)];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (tok_reg[wd * NT + min(key, NT - 1)] != qreg) mbits[wd] |= 1u << r;
            }
        }
    }
    lds_barrier();
    SPEI_STAMP(p.stamps, 3);

    // ---- 2b. K^T -> S^T -> softmax;  V -> O^T ------------------------------------------------------------------------------------------------
    lp4 opk[WPG][4];                                  // O^T packed: [window][4 d-groups], written to the slab after the barrier
    const int et = fr & 3, ecol = (fr >> 2) * 4;
    f32x4 rv[4][4];                                   // residual rows of the projection's accumulators: requested under the V pass
    {
        lp8 pp[WPG][2];                               // P^T packed
        {
            f32x16 acc[WPG];
            init_rows(acc, sbias + D);
            pass(p.wkv + (size_t)h * 16 * 512 + lane * 8, acc, std::true_type{}, [] {});
#pragma unroll
            for (int wd = 0; wd < WPG; ++wd) {
                f32x16 st;
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = 0.f;
                // S^T[key][query] = sum_d K^T[d][key] Q^T[d][query]:  A = (K^T)^T from the accumulator, B = Q^T (packed)
                st = mfma16(cvt8<0, LP>(acc[wd]), qp[wd][0], st);
                st = mfma16(cvt8<1, LP>(acc[wd]), qp[wd][1], st);
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
                    float v = -INFINITY;
                    if (key < NT) {
                        v = st[r] + rb[r];
                        if (mbits[wd] >> r & 1) v += -100.0f;
                    }
                    st[r] = v;
                    mx = fmaxf(mx, v);
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
                pp[wd][0] = cvt8<0, LP>(st);
                pp[wd][1] = cvt8<1, LP>(st);
            }
        }
        SPEI_STAMP(p.stamps, 4);
        {
            f32x16 acc[WPG];                          // V[token][d]: tokens on the accumulator rows, head dim on the columns
            const float bv = sbias[512 + h * HD + fr];
#pragma unroll
            for (int wd = 0; wd < WPG; ++wd)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[wd][r] = bv;
            pass(p.wkv + (size_t)(8 + h) * 16 * 512 + lane * 8, acc, std::false_type{}, [&] {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = i * 32 + 8 * k + 4 * fk + et;
                        const int pix = max(tok_pix[min(r, TOK - 1)], 0);
                        rv[i][k] = *reinterpret_cast<const f32x4*>(xg + (size_t)pix * D + wave * HD + ecol);
                    }
            });
#pragma unroll
            for (int wd = 0; wd < WPG; ++wd) {
                // O^T[d][query] = sum_key V[key][d] P^T[key][query]:  A = V^T from the accumulator (X^T.B form), B = P^T (packed)
                f32x16 ot;
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[r] = 0.f;
                ot = mfma16(cvt8<0, LP>(acc[wd]), pp[wd][0], ot);
                ot = mfma16(cvt8<1, LP>(acc[wd]), pp[wd][1], ot);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) opk[wd][g][e] = to_lp<LP>(ot[4 * g + e]);
            }
        }
        SPEI_STAMP(p.stamps, 5);
    }
    // ---- 3. proj: the wave's 32 output channels of the 100 rows (4 row tiles, the last one clamped); acc starts from x + bias.  The
    // residual rows were requested under the V pass, the first projection fragments are BEFORE the barrier that ends the attention phase -----
    const int rot4 = (grp * 3) & 15;                  // the projection's own rotation; the attention output is stored rotated by it
    const LP* wpp = p.wproj + (size_t)wave * 16 * 512 + lane * 8;
    lp8 wf[RP];
#pragma unroll
    for (int d = 0; d < RP; ++d) wf[d] = *reinterpret_cast<const lp8*>(wpp + ((rot4 + d) & 15) * 512);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();                                  // every wave is done reading the x rows
    // rows d = 8 g + 4 fk + e of head h, column = query token fr (< 25)  ->  slab[wd * 25 + fr][h * 32 + d]
    if (fr < NT) {
#pragma unroll
        for (int wd = 0; wd < WPG; ++wd)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<lp4*>(slab + (wd * NT + fr) * PA + (((h * HD + 8 * g + 4 * fk) * 2 - rot4 * 32) & 511)) = opk[wd][g];
    }
    {
        f32x16 acc[4];
        const float bias = sbias[768 + wave * HD + fr];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a[4] = {rv[i][k][0], rv[i][k][1], rv[i][k][2], rv[i][k][3]};
                quad_transpose4(a[0], a[1], a[2], a[3], et);              // row chunks -> accumulator layout
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][4 * k + e] = a[e] + bias;
            }
        lds_barrier();                              // the attention output is in the slab
        SPEI_STAMP(p.stamps, 8);
        int aoff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) aoff[i] = min(i * 32 + fr, TOK - 1) * PA + fk * 16;
        lp8 an[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const lp8*>(slab + aoff[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            lp8 ac[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ac[i] = an[i];
            if (s + 1 < 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const lp8*>(slab + aoff[i] + (s + 1) * 32);
            }
            const lp8 wc = wf[s % RP];
            if (s + RP < 16) wf[s % RP] = *reinterpret_cast<const lp8*>(wpp + ((rot4 + s + RP) & 15) * 512);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma16(ac[i], wc, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        SPEI_STAMP(p.stamps, 9);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = acc[i][4 * k + e];
                quad_transpose4(a[0], a[1], a[2], a[3], et);
                const int r = i * 32 + 8 * k + 4 * fk + et;
                const int pix = r < TOK ? tok_pix[r] : -1;
                if (pix >= 0) *reinterpret_cast<f32x4*>(og + (size_t)pix * D + wave * HD + ecol) = f32x4{a[0], a[1], a[2], a[3]};
            }
        SPEI_STAMP(p.stamps, 10);
    }
}


// ---- round 4, second half: the four-window kernel as a PERSISTENT, software-pipelined kernel -----------------------------------------------
// Stamps of `attn_win4_kernel` on the two stacked maps of a frame (1152 workgroups x 29.9 us on 256 CUs = 4.5 rounds): y-hat staging
// 6.6 us, Q pass 3.6, LayerNorm 3.4, K pass + softmax 4.3, V pass 3.4, output exchange 4.0, projection 2.1, stores 2.2 — every wave of
// the CU's one workgroup in the same phase, each phase waiting for its own loads.  Here a workgroup walks tiles of four windows
// (tile = blockIdx.x + k gridDim.x) and a tile's loads, LayerNorm and stores ride inside the GEMM passes of its neighbours (the
// structure of mlp_pipe_kernel, mlp_fused16.hip):
//     Q pass   (y-hat: slab Yc)        + the NEXT tile's y-hat rows -> slab Yn
//     K pass   (LN(x): slab X)         + relative-position bias, shift mask
//     S^T + bias + mask;  V pass (slab X) + softmax;  this tile's residual rows requested INTO the projection's accumulators;  O^T -> slab Yc
//     projection (slab Yc)             + the next tile's x rows -> LayerNorm -> slab X;  the token table of the tile after next
//     results stored
// Three 100-row slabs (Yc / Yn swap roles per tile), three barriers per tile, the weight ring runs on from pass to pass and tile to tile
// (buffer loads: descriptor + fragment offset in scalar registers).  Results and residual move one dword per lane in the accumulators'
// own layout, a row's byte offset looked up in the tile's token table (entry = pixel, or -1: an offset beyond the
// descriptor's range, so absent rows read 0 and are never written).  The arithmetic — K order, accumulator starts, softmax — is
// `attn_win4_kernel`'s; the two agree to the last bit except where the compiler contracts a LayerNorm product differently.
constexpr int TABN = 128;             // token-table entries per tile (rows 100..127 are absent)

template <typename LP>
__global__ __launch_bounds__(512) void attn_pipe_kernel(const Attn4Params<LP> p, const int ntiles, const int stagger) {
    typedef typename lpv<LP>::x8 lp8;
    typedef typename lpv<LP>::x4 lp4;
    constexpr int TPP = 32;                           // tokens per staging pass (16 lanes per token, 512 threads)
    constexpr int NPASS = (TOK + TPP - 1) / TPP;      // 4
    constexpr int RG = 8;                             // weight fragments in flight per wave
    constexpr int SLAB = TOK * PA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sX = smem + SLAB;                                // LayerNorm(x) of the current tile
    int* const tabs = reinterpret_cast<int*>(smem + 3 * SLAB);           // [3][TABN] token tables: previous, current, next tile
    float* const sbias = reinterpret_cast<float*>(tabs + 3 * TABN);      // [768] bq | bk | bv
    unsigned char* const sdummy = reinterpret_cast<unsigned char*>(sbias + 768);      // one row that takes the writes of absent rows / idle lanes
    unsigned char* const tregs = sdummy + PA;                             // [2][128] shift-mask region of token t of window wd at [32 wd + t]

    int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int fr = lane & 31, fk = lane >> 5;               // (not const: laundered at the top of every tile, see the loop)
    int l16 = tid & 15, rsub = tid >> 4;
    const int nwx = p.W / WS, HW = p.H * p.W;
    const int h = wave;                               // the wave's head (Q, K, V passes) and its 32 output channels (projection)
    const int G = gridDim.x;

    SPEI_STAMP(p.stamps, 0);
    spei_stagger_start(ntiles, stagger);
    // token table of tile t: entry r = pixel of slab row r within its map, or -1; its shift-mask regions (swinir.py:215-236) as bytes
    auto fill_table = [&](int* tab, unsigned char* treg, int t) {
        {                                             // every thread writes entry tid & 127 (four identical writes: no divergent branch inside a pass)
            const int te = tid & (TABN - 1);
            int ent = -1;
            if (te < TOK && t < ntiles) {
                const int grp = t % p.groups;
                const int wd = te / NT, tt = te - wd * NT;
                const int win = grp * WPG + wd;
                if (win < p.nwin) {
                    const int wy = win / nwx, wx = win - wy * nwx;
                    const int ysf = wy * WS + tt / WS, xsf = wx * WS + tt % WS;      // shifted-frame coordinates
                    int yo = ysf + p.shift, xo = xsf + p.shift;                      // roll(-shift): shifted[y] = x[(y+shift) % H]
                    if (yo >= p.H) yo -= p.H;
                    if (xo >= p.W) xo -= p.W;
                    const int reg = p.shift > 0 ? 3 * mask_region(ysf, p.H, p.shift) + mask_region(xsf, p.W, p.shift) : 0;
                    ent = yo * p.W + xo;
                    treg[wd * 32 + tt] = (unsigned char)reg;
                }
            }
            tab[te] = ent;
        }
    };
    int tile = blockIdx.x;
    int* tcur = tabs;                                 // tables of this tile, of the next one, and the buffer the one after next is written to
    int* tnext = tabs + TABN;
    int* tspare = tabs + 2 * TABN;
    if (tid < 64) reinterpret_cast<int*>(tregs)[tid] = 0;
    lds_barrier();
    fill_table(tcur, tregs, tile);
    fill_table(tnext, tregs + TABN, tile + G);
    for (int i = tid; i < 768; i += 512) sbias[i] = i < D ? p.bq[i] : p.bkv[i - D];
    float bias_p = p.bproj[wave * HD + fr];
    lds_barrier();

    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<LP*>(p.wq), 0, D * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rskv = __builtin_amdgcn_make_buffer_rsrc(const_cast<LP*>(p.wkv), 0, 2 * D * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<LP*>(p.wproj), 0, D * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrel = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.relbias), 0, 8 * NT * NT * 4, 0x00020000);
    int lane16 = lane * 16;
    // fragment of pass ph (0 Q, 1 K, 2 V, 3 projection), k-step ks
    auto wload = [&](int ph, int ks) -> lp8 {
        const int base = ph == 2 ? (8 + h) * 16 : h * 16;
        return __builtin_bit_cast(lp8, __builtin_amdgcn_raw_buffer_load_b128(ph == 0 ? rsq : ph == 3 ? rsp : rskv, lane16, (base + ks) * 1024, 0));
    };

    // staging helpers: 16 lanes per token, 32 tokens per pass
    auto row_entry = [&](const int* tab, int ps) { return tab[min(ps * TPP + rsub, TOK - 1)]; };
    f32x4 xr[NPASS][4];
    float lnm[NPASS], lns[NPASS];
    auto x_load = [&](const float* xg, const int* tab, int ps) {
        const int pix = max(row_entry(tab, ps), 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[ps][j] = reinterpret_cast<const f32x4*>(xg + (size_t)pix * D)[l16 + 16 * j];
    };
    // (inlined at two places — prologue and projection pass: the products and sums are spelled out and contraction is off, so that a
    // tile's LayerNorm does not depend on which of the two computed it)
    auto ln_slice = [&](const int* tab, int ps, int sub, unsigned char* dst, int rb_) {
#pragma clang fp contract(off)
        if (sub == 0) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sm += (xr[ps][j][0] + xr[ps][j][1]) + (xr[ps][j][2] + xr[ps][j][3]);
            lnm[ps] = sum16(sm) * (1.0f / 256.0f);
        } else if (sub == 1) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xr[ps][j] -= lnm[ps];
                ss += __builtin_fmaf(xr[ps][j][1], xr[ps][j][1], xr[ps][j][0] * xr[ps][j][0]) +
                      __builtin_fmaf(xr[ps][j][3], xr[ps][j][3], xr[ps][j][2] * xr[ps][j][2]);
            }
            lns[ps] = ss;
        } else if (sub == 2) {
            const int r = ps * TPP + rsub;
            const bool ok = r < TOK && row_entry(tab, ps) >= 0;
            lns[ps] = ok ? 1.0f / sqrtf(__builtin_fmaf(sum16(lns[ps]), 1.0f / 256.0f, 1e-5f)) : 0.f;     // empty rows stage zeros
        } else {
            const int r = ps * TPP + rsub;
            unsigned char* const row = r < TOK ? dst + r * PA : sdummy;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lp4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = to_lp<LP>(xr[ps][j][e] * lns[ps]);
                *reinterpret_cast<lp4*>(row + (((l16 + 16 * j) * 8 - rb_) & 511)) = hv;
            }
        }
    };
    u32x4 yr[NPASS][2];
    auto y_load = [&](const LP* yg, const int* tab, int ps) {
        const int pix = max(row_entry(tab, ps), 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) yr[ps][j] = reinterpret_cast<const u32x4*>(yg + (size_t)pix * D)[l16 + 16 * j];
    };
    auto y_write = [&](const int* tab, int ps, unsigned char* dst, int rb_) {
        const int r = ps * TPP + rsub;
        unsigned char* const row = r < TOK ? dst + r * PA : sdummy;
        const bool ok = row_entry(tab, ps) >= 0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<u32x4*>(row + (((l16 + 16 * j) * 16 - rb_) & 511)) = ok ? yr[ps][j] : u32x4{0u, 0u, 0u, 0u};
    };

    // ---- prologue: the first tile's y-hat rows -> slab 0, x rows -> LayerNorm -> slab X; the ring's first fragments ---------------------
    // (integer division runs on the vector unit: without readfirstlane the map's buffer descriptors count as divergent and every
    // buffer access becomes a waterfall loop)
    int bmap = __builtin_amdgcn_readfirstlane(tile / p.groups), grp = tile - bmap * p.groups;
    int rot = grp & 15, rot4 = (grp * 3) & 15;
    {
        const size_t moff = (size_t)bmap * HW * D;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) y_load(p.yhat + moff, tcur, ps);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) x_load(p.x + moff, tcur, ps);
    }
    lp8 ring[RG];
#pragma unroll
    for (int d = 0; d < RG; ++d) ring[d] = wload(0, (rot + d) & 15);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) y_write(tcur, ps, smem, rot * 32);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) ln_slice(tcur, ps, sub, sX, rot * 32);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    SPEI_STAMP(p.stamps, 1);

    int voff_ch = fr * 4;
    const int soff_w = __builtin_amdgcn_readfirstlane(wave * 128);      // the wave's 128-byte channel block (scalar offset of every row access)
    int cur = 0;
    bool first = true;
    for (; tile < ntiles; tile += G) {
        // Everything derived from the lane's position is loop-invariant, and the compiler computes ALL of it ahead of the loop (table
        // arithmetic, slab addresses, bias splats for packed adds: ~50 registers) and spills what does not fit; every reload is a
        // scratch load in the middle of a pass, in the same in-order queue as the weight ring.  Opaque per tile, they are recomputed
        // where they are used.
        asm volatile("" : "+v"(tid), "+v"(fr), "+v"(fk), "+v"(l16), "+v"(rsub), "+v"(lane16), "+v"(voff_ch), "+v"(bias_p));
        unsigned char* const sYc = smem + (cur ? 2 * SLAB : 0);           // y-hat of this tile, later its attention output
        unsigned char* const sYn = smem + (cur ? 0 : 2 * SLAB);           // y-hat of the next tile
        const int tnx = tile + G;
        const int bq_ = __builtin_amdgcn_readfirstlane(tnx / p.groups);
        const int grpn = tnx - bq_ * p.groups, bmapn = tnx < ntiles ? bq_ : bmap;
        const int rotn = grpn & 15, rot4n = (grpn * 3) & 15;
        const size_t moff = (size_t)__builtin_amdgcn_readfirstlane(bmap) * HW * D, moffn = (size_t)__builtin_amdgcn_readfirstlane(bmapn) * HW * D;
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + moff), 0, HW * (D * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + moff, 0, HW * (D * 4), 0x00020000);
        // fragment q of the tile's weight stream (q = 16 pass + step; q >= 64: the next tile's Q pass)
        auto wfrag = [&](int q) -> lp8 {
            const int ph = (q >> 4) & 3;
            const int r = q >= 64 ? rotn : ph == 3 ? rot4 : rot;
            return wload(ph, (r + (q & 15)) & 15);
        };
        // One pass of 16 k-steps: 4 MFMAs per step on the fragment the ring delivers, token fragments from the slab one step ahead;
        // filler(s): the vector / memory work that rides in this pass.  Pinned with full scheduling barriers; the empty asm on each
        // accumulator keeps an IR pass from moving MFMAs whose results are needed late to the end of the pass (mlp_fused16.hip).
        auto pass = [&](auto phc, const unsigned char* src, const int (&roff)[4], f32x16 (&acc)[4], auto tokens_on_columns, auto&& filler) {
            constexpr int ph = decltype(phc)::value;
            lp8 tn[4];
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) tn[wd] = *reinterpret_cast<const lp8*>(src + roff[wd]);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                lp8 tc[4];
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) tc[wd] = tn[wd];
                if (i + 1 < 16) {
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) tn[wd] = *reinterpret_cast<const lp8*>(src + roff[wd] + (i + 1) * 32);
                }
                const lp8 wc = ring[(ph * 16 + i) % RG];
                ring[(ph * 16 + i) % RG] = wfrag(ph * 16 + i + RG);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    if constexpr (decltype(tokens_on_columns)::value) acc[wd] = mfma16(wc, tc[wd], acc[wd]);
                    else acc[wd] = mfma16(tc[wd], wc, acc[wd]);
                }
                filler(i);
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) asm volatile("" : "+v"(acc[wd]));
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto init_rows = [&](f32x16 (&acc)[4], const float* b) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b + h * HD + 8 * g + 4 * fk);
#pragma unroll
                for (int wd = 0; wd < 4; ++wd)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[wd][4 * g + e] = bv[e];
            }
        };
        int rowoff[4], aoff[4];
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            rowoff[wd] = min(wd * NT + fr, TOK - 1) * PA + fk * 16;       // window wd's 32-row tile (rows past its 25 tokens: the next window's)
            aoff[wd] = min(wd * 32 + fr, TOK - 1) * PA + fk * 16;         // row tile wd of the 100 rows
        }

        // ---- Q^T of the wave's head  + the previous tile's results stored + the next tile's table ------------------------------------------
        lp8 qp[4][2];
        unsigned mbits[4] = {0u, 0u, 0u, 0u};
        {
            f32x16 acc[4];
            init_rows(acc, sbias);
            pass(std::integral_constant<int, 0>{}, sYc, rowoff, acc, std::true_type{}, [&](int s) {
                // The next tile's y-hat rows: ALL loads now, as one burst.  The vector-memory queue is in order, so an access that takes
                // longer than the ring's eight steps holds up every weight fragment requested behind it: spread over a pass (one group
                // per step) each step waited for the group before it and the pass ran at (memory latency / 8) per step — 5.0 us instead
                // of 2.5.  A burst costs that wait once.
                if (s == 0) {
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) y_load(p.yhat + moffn, tnext, ps);
                }
                if (s >= 12) y_write(tnext, s - 12, sYn, rotn * 32);
            });
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                qp[wd][0] = cvt8<0, LP>(acc[wd]);
                qp[wd][1] = cvt8<1, LP>(acc[wd]);
                asm volatile("" : "+v"(qp[wd][0]), "+v"(qp[wd][1]));      // packed NOW: left alone the conversion sinks to its use after the
            }                                                             // K pass and the 64 accumulator registers stay live through it
        }
        lds_barrier();                                // every wave is done with the y-hat rows (slab Yc); the next tile's table is complete
        if (first) SPEI_STAMP(p.stamps, 2);

        // ---- K^T -> S^T -> softmax ---------------------------------------------------------------------------------------------------------------
        lp8 pp[4][2];                                 // P^T packed
        f32x16 st[4];                                 // S^T, then the probabilities, of the four windows
        {
            float rb[16];
            f32x16 acc[4];
            init_rows(acc, sbias + D);
            pass(std::integral_constant<int, 1>{}, sX, rowoff, acc, std::true_type{}, [&](int s) {
                if (s == 0) {                         // relative-position bias of (query = lane column, key = register row): 16 L1-resident loads per
                    // tile cost less than 16 registers held through every pass; one per-lane offset, the key in the immediate (keys >= 25 read the
                    // next row or, past the table's end, the descriptor's 0: they are masked below)
                    const int vb = ((h * NT + (fr < NT ? fr : 0)) * NT + 4 * fk) * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rb[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrel, vb + ((r & 3) + 8 * (r >> 2)) * 4, 0, 0));
                }
                if (s >= 12) {                        // shift mask of window s - 12: bit r of mbits = key (register row r) lies in another region
                    const int wd = s - 12;            // than the lane's query; four key regions per 32-bit read, compared bytewise
                    const unsigned char* tr = tregs + (cur ? TABN : 0) + wd * 32;
                    const unsigned qs = tr[fr < NT ? fr : 0] * 0x01010101u;
                    unsigned m = 0u;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned x = *reinterpret_cast<const unsigned*>(tr + 8 * g + 4 * fk) ^ qs;
                        const unsigned nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) >> 7 & 0x01010101u;      // byte b = (region differs)
                        m |= ((nz * 0x01020408u) >> 24) << (4 * g);
                    }
                    mbits[wd] = m;
                }
            });
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[wd][r] = 0.f;
                // S^T[key][query] = sum_d K^T[d][key] Q^T[d][query]:  A = (K^T)^T from the accumulator, B = Q^T (packed)
                st[wd] = mfma16(cvt8<0, LP>(acc[wd]), qp[wd][0], st[wd]);
                st[wd] = mfma16(cvt8<1, LP>(acc[wd]), qp[wd][1], st[wd]);
            }
            // + relative-position bias, shift mask, -inf on the 7 pad keys: here, so that bias and mask registers die before the V pass;
            // the rest of the softmax (max, exp, sum, scale, pack: 3/4 of its instructions) rides in the V pass
#pragma unroll
            for (int wd = 0; wd < 4; ++wd)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
                    float v = -INFINITY;
                    if (key < NT) {
                        v = st[wd][r] + rb[r];
                        if (mbits[wd] >> r & 1) v += -100.0f;
                    }
                    st[wd][r] = v;
                }
        }
        if (first) SPEI_STAMP(p.stamps, 3);
        // ---- V;  the residual rows requested;  O^T -----------------------------------------------------------------------------------------------
        lp4 opk[4][4];
        f32x16 accp[4];                               // the projection's accumulators
        {
            f32x16 acc[4];                            // V[token][d]: tokens on the accumulator rows, head dim on the columns
            float bv = sbias[512 + h * HD + fr];
            asm volatile("" : "+v"(bv));              // per tile: as a loop invariant the 16-register splat stays live through every pass
#pragma unroll
            for (int wd = 0; wd < 4; ++wd)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[wd][r] = bv;
            float smx = 0.f, ssum = 0.f;
            pass(std::integral_constant<int, 2>{}, sX, rowoff, acc, std::false_type{}, [&](int s) {
                const int wd = s >> 2, part = s & 3;  // softmax of window wd in four slices (steps 4 wd .. 4 wd + 3)
                if (part == 0) {
                    float mx = st[wd][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[wd][r]);
                    smx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                } else if (part == 1) {
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __expf(st[wd][r] - smx);
                        st[wd][r] = e;
                        sum += e;
                    }
                    ssum = sum;
                } else if (part == 2) {
                    const float inv = 1.0f / (ssum + __shfl_xor(ssum, 32, 64));
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[wd][r] *= inv;
                } else {
                    pp[wd][0] = cvt8<0, LP>(st[wd]);
                    pp[wd][1] = cvt8<1, LP>(st[wd]);
                    asm volatile("" : "+v"(pp[wd][0]), "+v"(pp[wd][1]));
                }
            });
            // The residual rows, straight into the projection's accumulators, behind the pass (beside the V accumulators, P^T and the ring
            // there is no room for them earlier) and in two halves around the O^T computation: a wave can have 63 vector-memory accesses
            // outstanding, and 64 loads on top of the ring's 8 would stop at the queue.  Each half is a burst (see the Q pass); O^T, its
            // exchange and the barrier cover part of their latency.
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int4 ent = *reinterpret_cast<const int4*>(tcur + i * 32 + 8 * k + 4 * fk);
                    const int en[4] = {ent.x, ent.y, ent.z, ent.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        accp[i][4 * k + e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsx, (en[e] << 10) + voff_ch, soff_w, 0));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                // O^T[d][query] = sum_key V[key][d] P^T[key][query]:  A = V^T from the accumulator (X^T.B form), B = P^T (packed)
                f32x16 ot;
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[r] = 0.f;
                ot = mfma16(cvt8<0, LP>(acc[wd]), pp[wd][0], ot);
                ot = mfma16(cvt8<1, LP>(acc[wd]), pp[wd][1], ot);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) opk[wd][g][e] = to_lp<LP>(ot[4 * g + e]);
            }
#pragma unroll
            for (int i = 2; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int4 ent = *reinterpret_cast<const int4*>(tcur + i * 32 + 8 * k + 4 * fk);
                    const int en[4] = {ent.x, ent.y, ent.z, ent.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        accp[i][4 * k + e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsx, (en[e] << 10) + voff_ch, soff_w, 0));
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        // rows d = 8 g + 4 fk + e of head h, column = query token fr (< 25)  ->  slab Yc[wd * 25 + fr][h * 32 + d]  (B1: nobody reads y-hat any more)
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            unsigned char* const row = fr < NT ? sYc + (wd * NT + fr) * PA : sdummy;      // lanes 25..31 hold the next window's queries
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<lp4*>(row + (((h * HD + 8 * g + 4 * fk) * 2 - rot4 * 32) & 511)) = opk[wd][g];
        }
        lds_barrier();                                // the attention output is in slab Yc; every wave is done reading slab X
        if (first) SPEI_STAMP(p.stamps, 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) accp[i][r] += bias_p;
        // ---- projection: the wave's 32 output channels of the 100 rows  + the next tile's x rows -> LayerNorm -> slab X ------------------------------
        pass(std::integral_constant<int, 3>{}, sYc, aoff, accp, std::false_type{}, [&](int s) {
            if (s == 0) {                             // the next tile's rows: one burst (see the Q pass), consumed in the second half of the pass
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) x_load(p.x + moffn, tnext, ps);
                fill_table(tspare, tregs + (cur ? TABN : 0), tnx + G);      // the table of the tile after next (this tile's regions are used up)
            }
            if (s >= 8) {                             // 16 LayerNorm slices
                ln_slice(tnext, (2 * s - 16) >> 2, (2 * s - 16) & 3, sX, rotn * 32);
                ln_slice(tnext, (2 * s - 15) >> 2, (2 * s - 15) & 3, sX, rotn * 32);
            }
        });
        if (first) SPEI_STAMP(p.stamps, 5);
        // results: register 4 k + e of row tile i <-> slab row 32 i + 8 k + 4 fk + e, channel 32 wave + fr; two bursts of 32 (the queue
        // holds 63), the barrier between them (the weight fragments of the next Q pass's first eight steps are already on their way)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int4 ent = *reinterpret_cast<const int4*>(tcur + i * 32 + 8 * k + 4 * fk);
                const int en[4] = {ent.x, ent.y, ent.z, ent.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[i][4 * k + e]), rso, (en[e] << 10) + voff_ch, soff_w, 0);
                __builtin_amdgcn_sched_barrier(0);    // group by group: 16 table reads in flight at once are 64 more live registers
            }
        lds_barrier();                                // slab X holds the next tile; slab Yc is free; the table after next is complete
#pragma unroll
        for (int i = 2; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int4 ent = *reinterpret_cast<const int4*>(tcur + i * 32 + 8 * k + 4 * fk);
                const int en[4] = {ent.x, ent.y, ent.z, ent.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[i][4 * k + e]), rso, (en[e] << 10) + voff_ch, soff_w, 0);
                __builtin_amdgcn_sched_barrier(0);    // group by group: 16 table reads in flight at once are 64 more live registers
            }
        if (first) SPEI_STAMP(p.stamps, 6);
        first = false;
        int* const tt = tcur; tcur = tnext; tnext = tspare; tspare = tt;
        cur ^= 1;
        grp = grpn; bmap = bmapn; rot = rotn; rot4 = rot4n;
    }
    SPEI_STAMP(p.stamps, 7);
}

}  // namespace

template <typename LP>
static int attn_launch(const float* x, float* out, const void* yhat, const void* wq, const float* bq, const void* wkv, const float* bkv,
                       const void* wproj, const float* bproj, const float* relbias, int H, int W, int shift, hipStream_t st) {
    AttnParams<LP> p;
    p.x = x; p.out = out; p.yhat = (const LP*)yhat; p.wq = (const LP*)wq; p.bq = bq; p.wkv = (const LP*)wkv;
    p.bkv = bkv; p.wproj = (const LP*)wproj; p.bproj = bproj; p.relbias = relbias;
    p.H = H; p.W = W; p.shift = shift; p.nwin = (H / WS) * (W / WS);
    p.stamps = spei_stamp_buffer();
    const size_t lds = (size_t)2 * ROWS * PA + 2 * ROWS * sizeof(int);
    static const int halves = spei_knob("SPEI_ATTN_HALVES", 1);          // tuning build: 2 = 512-thread workgroups (see the kernel)
    if (halves == 2) {
        ensure_dyn_lds<&attn_fused_kernel<LP, 2>>(2 * lds);
        hipLaunchKernelGGL((attn_fused_kernel<LP, 2>), dim3((p.nwin + 3) / 4), dim3(512), 2 * lds, st, p);
    } else {
        ensure_dyn_lds<&attn_fused_kernel<LP, 1>>(lds);
        hipLaunchKernelGGL((attn_fused_kernel<LP, 1>), dim3((p.nwin + 1) / 2), dim3(256), lds, st, p);
    }
    SPEI_CHECK_LAUNCH("spei_attn_fused16");
    return 0;
}

template <typename LP>
static int attn4_launch(const float* x, float* out, const void* yhat, const void* wq, const float* bq, const void* wkv, const float* bkv,
                        const void* wproj, const float* bproj, const float* relbias, int batch, int H, int W, int shift, hipStream_t st) {
    Attn4Params<LP> p;
    p.x = x; p.out = out; p.yhat = (const LP*)yhat; p.wq = (const LP*)wq; p.bq = bq; p.wkv = (const LP*)wkv;
    p.bkv = bkv; p.wproj = (const LP*)wproj; p.bproj = bproj; p.relbias = relbias;
    p.H = H; p.W = W; p.shift = shift; p.nwin = (H / WS) * (W / WS); p.groups = (p.nwin + WPG - 1) / WPG; p.batch = batch;
    p.stamps = spei_stamp_buffer();
    static const int pipe = spei_knob("SPEI_ATTN_PIPE", 1);              // tuning build: 0 = one tile per workgroup (attn_win4_kernel)
    if (pipe && (int64_t)H * W <= (1ll << 21)) {                         // 32-bit byte offsets into a map, 24-bit pixel indices in the table
        const int ntiles = p.groups * batch;
        const size_t lds = (size_t)3 * TOK * PA + 3 * TABN * sizeof(int) + 768 * sizeof(float) + PA + 2 * TABN;
        ensure_dyn_lds<&attn_pipe_kernel<LP>>(lds);
        static const int stagger = spei_knob("SPEI_PIPE_STAGGER", 0);
        hipLaunchKernelGGL((attn_pipe_kernel<LP>), dim3(ntiles < spei_num_cus() ? ntiles : spei_num_cus()), dim3(512), lds, st, p, ntiles, stagger);
        SPEI_CHECK_LAUNCH("spei_attn_win4_16");
        return 0;
    }
    const size_t lds = (size_t)TOK * PA + 2 * TOK * sizeof(int) + 1024 * sizeof(float);
    ensure_dyn_lds<&attn_win4_kernel<LP>>(lds);
    hipLaunchKernelGGL((attn_win4_kernel<LP>), dim3(p.groups * batch), dim3(512), lds, st, p);
    SPEI_CHECK_LAUNCH("spei_attn_win4_16");
    return 0;
}

extern "C" int spei_attn_win4_16(int fmt, const float* x, float* out, const void* yhat, const void* wq_frag, const float* bq,
                                 const void* wkv_frag, const float* bkv, const void* wproj_frag, const float* bproj,
                                 const float* relbias, int batch, int H, int W, int shift, spei_stream_t stream) {
    SPEI_REQUIRE(x && out && yhat && wq_frag && bq && wkv_frag && bkv && wproj_frag && bproj && relbias, "spei_attn_win4_16: null pointer");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_attn_win4_16: fmt=%d", fmt);
    SPEI_REQUIRE(H > 0 && W > 0 && H % WS == 0 && W % WS == 0, "spei_attn_win4_16: %dx%d is not a multiple of the 5x5 window", H, W);
    SPEI_REQUIRE(shift >= 0 && shift < WS, "spei_attn_win4_16: shift=%d", shift);
    SPEI_REQUIRE(batch >= 1 && (int64_t)batch * H * W < (1ll << 30), "spei_attn_win4_16: batch=%d of %dx%d", batch, H, W);
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)yhat | (uintptr_t)wq_frag | (uintptr_t)wkv_frag | (uintptr_t)wproj_frag) % 16 == 0,
                 "spei_attn_win4_16: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (fmt == SPEI_F16) return attn4_launch<_Float16>(x, out, yhat, wq_frag, bq, wkv_frag, bkv, wproj_frag, bproj, relbias, batch, H, W, shift, st);
    return attn4_launch<__bf16>(x, out, yhat, wq_frag, bq, wkv_frag, bkv, wproj_frag, bproj, relbias, batch, H, W, shift, st);
}

extern "C" int spei_attn_fused16(int fmt, const float* x, float* out, const void* yhat, const void* wq_frag, const float* bq,
                                 const void* wkv_frag, const float* bkv, const void* wproj_frag, const float* bproj,
                                 const float* relbias, int H, int W, int shift, spei_stream_t stream) {
    SPEI_REQUIRE(x && out && yhat && wq_frag && bq && wkv_frag && bkv && wproj_frag && bproj && relbias, "spei_attn_fused16: null pointer");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_attn_fused16: fmt=%d", fmt);
    SPEI_REQUIRE(H > 0 && W > 0 && H % WS == 0 && W % WS == 0, "spei_attn_fused16: %dx%d is not a multiple of the 5x5 window", H, W);
    SPEI_REQUIRE(shift >= 0 && shift < WS, "spei_attn_fused16: shift=%d", shift);
    SPEI_REQUIRE((int64_t)H * W < (1ll << 30), "spei_attn_fused16: map too large");
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)yhat | (uintptr_t)wq_frag | (uintptr_t)wkv_frag | (uintptr_t)wproj_frag) % 16 == 0,
                 "spei_attn_fused16: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (fmt == SPEI_F16) return attn_launch<_Float16>(x, out, yhat, wq_frag, bq, wkv_frag, bkv, wproj_frag, bproj, relbias, H, W, shift, st);
    return attn_launch<__bf16>(x, out, yhat, wq_frag, bq, wkv_frag, bkv, wproj_frag, bproj, relbias, H, W, shift, st);
}

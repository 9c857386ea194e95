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

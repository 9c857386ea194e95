// One cross-window Swin block in ONE persistent launch on the gfx950 16-bit matrix pipe (bf16 or half operands):
//
//     x1  = x  + proj( W-MSA( q = norm1(y) Wq,  [k, v] = norm1(x) Wkv ) )
//     out = x1 + fc2( GELU( fc1( norm2(x1) ) ) )
//
// (reference model/swinir.py:238-281 SwinTransformerBlock.forward, :115-149 WindowAttention.forward, :12-29 Mlp,
// :215-236 calculate_mask, :32-61 window_partition / reverse, torch.roll for the cyclic shift.)
//
// attn_fused16.hip + mlp_fused16.hip run this as two launches: x makes two HBM round trips per block, every workgroup
// re-streams 512 KB of weights from L2 for 50 .. 128 tokens, and phase stamps (tools/stamp_phases.py) show both bound by
// what a CU can take in from L2 (~70 GB/s) plus unoverlapped HBM phases, the matrix pipe 26 % busy.  Here
//   * a 512-thread workgroup (8 waves, one per CU, persistent over the launch) owns groups of THREE 5x5 windows (96 padded
//     rows): every weight fragment fetched from L2 feeds three MFMAs, and a CU streams the block's 1 MB of weights once
//     per 75 tokens instead of 512 KB per 50 (attention) + 512 KB per 128 (MLP) with a second pass over x;
//   * x is read once and written once per block: the attention branch's result x1 stays in registers (row layout: a lane
//     holds 4 consecutive channels of 12 rows) through LayerNorm2 and the MLP and is added to fc2's accumulators at the end;
//   * wave = head in the attention phase (Q^T, K^T for three windows in one weight pass, S^T / softmax in registers, then V
//     in a second pass and O^T = V^T P^T, all accumulator -> operand without LDS as in attn_fused16.hip), wave = 32 output
//     channels in proj / fc2, wave = 32 hidden channels in fc1;
//   * LDS (111 KB): token slab (norm1(x), later norm2(x1)), y-hat slab (later the attention output, later one half of the
//     hidden activations), LayerNorm2 partial sums.
// LayerNorm statistics: norm1 two-pass in registers as before; norm2 from sum / sum of squares in fp32 across the 8
// waves (256 values per row: relative error of the variance ~1e-7 (1 + mean^2/var), far below the 16-bit rounding of the
// normalised tokens).
#include "common.h"

namespace {

constexpr int D = 256, HD = 32, NT = 25, WS = 5, HID = 512;
constexpr int PA = 2 * D + 16;        // LDS row pitch (bytes) of every slab
constexpr int GW = 3;                 // windows per group
constexpr int ROWS = GW * 32;         // 96 rows per group (25 tokens + 7 pad rows per window)
constexpr int NTH = 512;
constexpr int RING = 4;               // MLP weight fragments in flight per wave

template <typename LP>
struct BlockParams {
    const float* x;
    float* out;
    const LP* yhat;       // [M][256]
    const LP* wq;         // fragment order [8][1][16][64][8]
    const float* bq;
    const LP* wkv;        // [16][1][16][64][8]  (n-tiles 0..7 = K heads, 8..15 = V heads)
    const float* bkv;
    const LP* wproj;      // [8][1][16][64][8]
    const float* bproj;
    const float* relbias; // [8][25][25]
    const LP* w1;         // [16][1][16][64][8]
    const float* b1;
    const LP* w2;         // [8][1][32][64][8]
    const float* b2;
    long long* stamps;    // tuning build: phase stamps of each workgroup's first group, else NULL
    int H, W, shift, nwin, ngroups;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

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
// all-reduce over each aligned group of 16 lanes (see attn_fused16.hip)
__device__ __forceinline__ float sum16(float v) { return dpp_add<0x140>(dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(v)))); }
// sum over the 8 lanes {4q + t, q = 0..7} of a 32-lane half that share t = lane & 3: rotate by 4 and 8 inside the rows of 16
// lanes (row_ror keeps lane & 3), then the other row of the half (xor 16)
__device__ __forceinline__ float sum_quads(float v) {
    v += dpp_quad<0x124>(v);          // row_ror:4
    v += dpp_quad<0x128>(v);          // row_ror:8
    return xor_combine<16, OpSum>(v);
}

// erf-GELU on two values, degree-7 odd polynomial of Phi (see mlp_fused16.hip: max abs error 1.1e-4)
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

// The group loop is one long unrolled body: every weight / slab address in it is invariant across groups, and left alone the
// compiler hoists ~130 of them out of the loop and spills them (330 VGPRs of scratch, seen with -Rpass-analysis).  An
// opaque per-phase copy of the lane offset keeps the address math where it is used.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <typename LP>
__global__ __launch_bounds__(NTH) void swin_block_kernel(const BlockParams<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    typedef typename lpv<LP>::x4 lp4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;                         // [96][PA]  norm1(x); later norm2(x1)
    unsigned char* ys = smem + ROWS * PA;             // [96][PA]  y-hat; later the attention output; later a hidden half
    float* part = reinterpret_cast<float*>(smem + 2 * ROWS * PA);      // [8 waves][96][2]  LayerNorm2 partial sums
    float* rstat = part + 8 * ROWS * 2;                                 // [96][2]  mean, rstd
    float* bias1 = rstat + ROWS * 2;                                    // [512]
    float* rbs = bias1 + HID;                                           // [8][25][25]  relative-position bias
    int* tok_pix = reinterpret_cast<int*>(rbs + 8 * NT * NT);           // [96] pixel index or -1
    int* tok_reg = tok_pix + ROWS;                                      // [96] shift-mask region id

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int et = fr & 3, ecol = (fr >> 2) * 4;      // row layout: lane -> (row offset et, 4 channels at ecol) of a 32 x 32 tile
    const int nwx = p.W / WS;
    const int h = wave;                               // the wave's head in the attention phase

    bias1[tid] = p.b1[tid];
    for (int i = tid; i < 8 * NT * NT; i += NTH) rbs[i] = p.relbias[i];

    for (int g = blockIdx.x; g < p.ngroups; g += gridDim.x) {
        long long* const stamps = g == (int)blockIdx.x ? p.stamps : nullptr;
        SPEI_STAMP(stamps, 0);
        // ---- token -> pixel map of the group's three windows (cyclic shift + window partition) -------------------------
        if (tid < ROWS) {
            const int w = tid >> 5, t = tid & 31;
            const int win = g * GW + w;
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

        // ---- A. stage norm1(x) and y-hat: 16 lanes per token, 32 tokens per pass, all loads of the group issued first -----
        {
            const int l16 = tid & 15, rsub = tid >> 4;
            f32x4 xr[GW][4];
            u32x4 yr[GW][2];
#pragma unroll
            for (int b = 0; b < GW; ++b) {
                const int pix = max(tok_pix[b * 32 + rsub], 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) xr[b][j] = reinterpret_cast<const f32x4*>(p.x + (size_t)pix * D)[l16 + 16 * j];
#pragma unroll
                for (int j = 0; j < 2; ++j) yr[b][j] = reinterpret_cast<const u32x4*>(p.yhat + (size_t)pix * D)[l16 + 16 * j];
            }
#pragma unroll
            for (int b = 0; b < GW; ++b) {
                const int r = b * 32 + rsub;
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
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<lp4*>(xs + r * PA + (l16 + 16 * j) * 8) = to_lp4<LP>(xr[b][j] * rstd);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<u32x4*>(ys + r * PA + (l16 + 16 * j) * 16) = ok ? yr[b][j] : u32x4{0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();
        SPEI_STAMP(stamps, 1);

        // ---- B. attention of head `wave` on the three windows, in registers ----------------------------------------------------
        lp4 opk[GW][4];                                    // O^T packed: [window][4 d-groups]
        {
            const int rot = opaque((blockIdx.x + wave) & 15);      // per-wave K rotation: spreads the L2 channel load
            const int lane8 = opaque(lane * 8);
            const int tofs = opaque(fr * PA + fk * 16);    // this lane's fragment offset inside a 32-row slab tile
            f32x16 st[GW];                                 // S^T, then P^T
            {
                f32x16 qT[GW], kT[GW];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const float bqv = p.bq[h * HD + d], bkv = p.bkv[h * HD + d];
#pragma unroll
                    for (int w = 0; w < GW; ++w) { qT[w][r] = bqv; kT[w][r] = bkv; }
                }
                const LP* wqp = p.wq + (size_t)h * 16 * 512 + lane8;
                const LP* wkp = p.wkv + (size_t)h * 16 * 512 + lane8;
                lp8 wqf[3], wkf[3];
#pragma unroll
                for (int pre = 0; pre < 2; ++pre) {
                    wqf[pre] = *reinterpret_cast<const lp8*>(wqp + ((rot + pre) & 15) * 512);
                    wkf[pre] = *reinterpret_cast<const lp8*>(wkp + ((rot + pre) & 15) * 512);
                }
                lp8 yn[GW], xn[GW];
#pragma unroll
                for (int w = 0; w < GW; ++w) {
                    yn[w] = *reinterpret_cast<const lp8*>(ys + w * 32 * PA + tofs + rot * 32);
                    xn[w] = *reinterpret_cast<const lp8*>(xs + w * 32 * PA + tofs + rot * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int cur = i % 3;
                    lp8 yc[GW], xc[GW];
#pragma unroll
                    for (int w = 0; w < GW; ++w) { yc[w] = yn[w]; xc[w] = xn[w]; }
                    if (i + 1 < 16) {
                        const int ko = ((rot + i + 1) & 15) * 32 + tofs;
#pragma unroll
                        for (int w = 0; w < GW; ++w) {
                            yn[w] = *reinterpret_cast<const lp8*>(ys + w * 32 * PA + ko);
                            xn[w] = *reinterpret_cast<const lp8*>(xs + w * 32 * PA + ko);
                        }
                    }
                    const lp8 wqc = wqf[cur], wkc = wkf[cur];
                    if (i + 2 < 16) {
                        const int nxt = (i + 2) % 3, ksn = (rot + i + 2) & 15;
                        wqf[nxt] = *reinterpret_cast<const lp8*>(wqp + ksn * 512);
                        wkf[nxt] = *reinterpret_cast<const lp8*>(wkp + ksn * 512);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < GW; ++w) {
                        qT[w] = mfma16(wqc, yc[w], qT[w]);
                        kT[w] = mfma16(wkc, xc[w], kT[w]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                SPEI_STAMP(stamps, 2);
                // S^T[key][query] = sum_d K^T[d][key] Q^T[d][query], + bias + shift mask, softmax over the keys (registers)
#pragma unroll
                for (int w = 0; w < GW; ++w) {
                    f32x16 s;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = 0.f;
                    s = mfma16(cvt8<0, LP>(kT[w]), cvt8<0, LP>(qT[w]), s);
                    s = mfma16(cvt8<1, LP>(kT[w]), cvt8<1, LP>(qT[w]), s);
                    unsigned mbits = 0u;           // bit r: key (register row r) lies in another shift region than the lane's query
                    if (p.shift > 0) {
                        const int qreg = tok_reg[w * 32 + (fr < NT ? fr : 0)];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
                            if (tok_reg[w * 32 + min(key, NT - 1)] != qreg) mbits |= 1u << r;
                        }
                    }
                    float mx = -INFINITY;
                    const float* rbq = rbs + (h * NT + (fr < NT ? fr : 0)) * NT;     // bias row of (head, query = lane column)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
                        float v = -INFINITY;
                        if (key < NT) {
                            v = s[r] + rbq[key];
                            if (mbits >> r & 1) v += -100.0f;
                        }
                        s[r] = v;
                        mx = fmaxf(mx, v);
                    }
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __expf(s[r] - mx);
                        s[r] = e;
                        sum += e;
                    }
                    sum += __shfl_xor(sum, 32, 64);
                    const float inv = 1.0f / sum;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] *= inv;
                    st[w] = s;
                }
            }
            SPEI_STAMP(stamps, 3);
            // second weight pass: V of the three windows, then O^T[d][query] = sum_key V[key][d] P^T[key][query]
            {
                f32x16 vv[GW];
                const float bvv = p.bkv[D + h * HD + fr];
#pragma unroll
                for (int w = 0; w < GW; ++w)
#pragma unroll
                    for (int r = 0; r < 16; ++r) vv[w][r] = bvv;
                const LP* wvp = p.wkv + (size_t)(8 + h) * 16 * 512 + opaque(lane8);
                lp8 wvf[3];
#pragma unroll
                for (int pre = 0; pre < 2; ++pre) wvf[pre] = *reinterpret_cast<const lp8*>(wvp + ((rot + pre) & 15) * 512);
                lp8 xn[GW];
#pragma unroll
                for (int w = 0; w < GW; ++w) xn[w] = *reinterpret_cast<const lp8*>(xs + w * 32 * PA + tofs + rot * 32);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int cur = i % 3;
                    lp8 xc[GW];
#pragma unroll
                    for (int w = 0; w < GW; ++w) xc[w] = xn[w];
                    if (i + 1 < 16) {
                        const int ko = ((rot + i + 1) & 15) * 32 + tofs;
#pragma unroll
                        for (int w = 0; w < GW; ++w) xn[w] = *reinterpret_cast<const lp8*>(xs + w * 32 * PA + ko);
                    }
                    const lp8 wvc = wvf[cur];
                    if (i + 2 < 16) wvf[(i + 2) % 3] = *reinterpret_cast<const lp8*>(wvp + ((rot + i + 2) & 15) * 512);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < GW; ++w) vv[w] = mfma16(xc[w], wvc, vv[w]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int w = 0; w < GW; ++w) {
                    f32x16 ot;
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[r] = 0.f;
                    ot = mfma16(cvt8<0, LP>(vv[w]), cvt8<0, LP>(st[w]), ot);
                    ot = mfma16(cvt8<1, LP>(vv[w]), cvt8<1, LP>(st[w]), ot);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                        for (int e = 0; e < 4; ++e) opk[w][gq][e] = to_lp<LP>(ot[4 * gq + e]);
                }
            }
        }
        SPEI_STAMP(stamps, 4);
        __syncthreads();                                   // every wave is done reading the x and y slabs
        // rows d = 8 gq + 4 fk + e of O^T, column = query token fr  ->  os[token][h*32 + d]   (os = the y slab)
        const int oofs = opaque(fr * PA + (h * HD + 4 * fk) * 2);
#pragma unroll
        for (int w = 0; w < GW; ++w)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<lp4*>(ys + w * 32 * PA + oofs + gq * 16) = opk[w][gq];

        // ---- C. proj: the wave produces channels [32 wave, +32) of the 96 rows; + bias + residual x -> x1 (row layout) -----------
        f32x4 x1r[GW][4];                                  // x1[row = i*32 + 8k + 4fk + et][channels 32 wave + ecol .. +3]
#pragma unroll
        for (int i = 0; i < GW; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pix = max(tok_pix[i * 32 + 8 * k + 4 * fk + et], 0);
                x1r[i][k] = *reinterpret_cast<const f32x4*>(p.x + (size_t)pix * D + wave * HD + ecol);      // residual, L2 hits
            }
        __syncthreads();
        SPEI_STAMP(stamps, 5);
        {
            f32x16 acc[GW];
#pragma unroll
            for (int i = 0; i < GW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const LP* wpp = p.wproj + (size_t)wave * 16 * 512 + opaque(lane * 8);
            const int rot = opaque((blockIdx.x * 3 + wave) & 15);
            lp8 wf[3];
#pragma unroll
            for (int pre = 0; pre < 2; ++pre) wf[pre] = *reinterpret_cast<const lp8*>(wpp + ((rot + pre) & 15) * 512);
            lp8 an[GW];
            const int tofs = opaque(fr * PA + fk * 16);
#pragma unroll
            for (int i = 0; i < GW; ++i) an[i] = *reinterpret_cast<const lp8*>(ys + i * 32 * PA + tofs + rot * 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                lp8 ac[GW];
#pragma unroll
                for (int i = 0; i < GW; ++i) ac[i] = an[i];
                if (s + 1 < 16) {
                    const int ko = ((rot + s + 1) & 15) * 32 + tofs;
#pragma unroll
                    for (int i = 0; i < GW; ++i) an[i] = *reinterpret_cast<const lp8*>(ys + i * 32 * PA + ko);
                }
                const lp8 wc = wf[s % 3];
                if (s + 2 < 16) wf[(s + 2) % 3] = *reinterpret_cast<const lp8*>(wpp + ((rot + s + 2) & 15) * 512);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GW; ++i) acc[i] = mfma16(ac[i], wc, acc[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
            SPEI_STAMP(stamps, 6);
            const float bias = p.bproj[wave * HD + fr];
            // accumulator layout (column = channel on the lane, 16 rows in registers) -> row layout, + residual; LayerNorm2
            // partial sums over this wave's 32 channels of every row
#pragma unroll
            for (int i = 0; i < GW; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = acc[i][4 * k + e] + bias;
                    quad_transpose4(a[0], a[1], a[2], a[3], et);
                    x1r[i][k] += f32x4{a[0], a[1], a[2], a[3]};
                    const f32x4 v = x1r[i][k];
                    const float s1 = sum_quads((v[0] + v[1]) + (v[2] + v[3]));
                    const float s2 = sum_quads((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                    if (fr < 4) {                              // lanes 4q + t with q == 0: one per row
                        const int row = i * 32 + 8 * k + 4 * fk + et;
                        *reinterpret_cast<f32x2*>(part + (wave * ROWS + row) * 2) = f32x2{s1, s2};
                    }
                }
        }
        __syncthreads();                                   // partial sums complete; every wave is done reading the O slab
        if (tid < ROWS) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(part + (wv * ROWS + tid) * 2);
                s1 += v[0];
                s2 += v[1];
            }
            const float mean = s1 * (1.0f / 256.0f);
            const float var = fmaxf(s2 * (1.0f / 256.0f) - mean * mean, 0.f);
            *reinterpret_cast<f32x2*>(rstat + tid * 2) = f32x2{mean, 1.0f / sqrtf(var + 1e-5f)};
        }
        __syncthreads();
        // norm2(x1) -> token slab (over the norm1(x) slab: dead since the barrier after B)
#pragma unroll
        for (int i = 0; i < GW; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = i * 32 + 8 * k + 4 * fk + et;
                const f32x2 ms = *reinterpret_cast<const f32x2*>(rstat + row * 2);
                *reinterpret_cast<lp4*>(xs + row * PA + (wave * HD + ecol) * 2) = to_lp4<LP>((x1r[i][k] - ms[0]) * ms[1]);
            }

        // ---- D. MLP: fc1^T (wave = 32 hidden channels per half) -> GELU -> hidden half slab -> fc2 partial (wave = 32 channels) ----
        f32x16 acc2[GW];
#pragma unroll
        for (int i = 0; i < GW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
        {
            const int rot = opaque((blockIdx.x * 5 + wave) & 15);
            const int lane8 = opaque(lane * 8);
            lp8 ring[RING];
            const LP* wptr = p.w1 + (size_t)wave * 16 * 512 + lane8;          // fc1, half 0: hidden tile `wave`
#pragma unroll
            for (int d = 0; d < RING; ++d) ring[d] = *reinterpret_cast<const lp8*>(wptr + ((rot + d) & 15) * 512);
            __syncthreads();                               // token slab complete
            SPEI_STAMP(stamps, 7);
            const int lofs = opaque(fr * PA + fk * 16);
            const unsigned char* abase = xs + lofs;
            const unsigned char* hbase = ys + lofs;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x16 acc1[GW];
#pragma unroll
                for (int i = 0; i < GW; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
                lp8 tn[GW];
#pragma unroll
                for (int i = 0; i < GW; ++i) tn[i] = *reinterpret_cast<const lp8*>(abase + i * 32 * PA + rot * 32);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    lp8 tc[GW];
#pragma unroll
                    for (int i = 0; i < GW; ++i) tc[i] = tn[i];
                    if (s + 1 < 16) {
#pragma unroll
                        for (int i = 0; i < GW; ++i) tn[i] = *reinterpret_cast<const lp8*>(abase + i * 32 * PA + ((rot + s + 1) & 15) * 32);
                    }
                    const lp8 w = ring[s % RING];
                    if (s + RING < 16) ring[s % RING] = *reinterpret_cast<const lp8*>(wptr + ((rot + s + RING) & 15) * 512);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < GW; ++i) acc1[i] = mfma16(w, tc[i], acc1[i]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                SPEI_STAMP(stamps, 8 + 3 * half);
                // fc2 weight stream of this half starts under the GELU
                wptr = p.w2 + (size_t)wave * 32 * 512 + (size_t)half * 16 * 512 + opaque(lane8);
#pragma unroll
                for (int d = 0; d < RING; ++d) ring[d] = *reinterpret_cast<const lp8*>(wptr + ((rot + d) & 15) * 512);
                if (half == 1) __syncthreads();            // fc2 of half 0 is done reading the hidden slab
                // rows c = 8 gq + 4 fk + e of acc1 = hidden channel, column = token fr  ->  sh[token][32 wave + c]
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias1 + half * 256 + wave * 32 + 8 * gq + 4 * fk);
#pragma unroll
                    for (int i = 0; i < GW; ++i) {
                        const f32x2 g01 = gelu2(f32x2{acc1[i][4 * gq] + bv[0], acc1[i][4 * gq + 1] + bv[1]});
                        const f32x2 g23 = gelu2(f32x2{acc1[i][4 * gq + 2] + bv[2], acc1[i][4 * gq + 3] + bv[3]});
                        lp4 hv;
                        hv[0] = to_lp<LP>(g01[0]); hv[1] = to_lp<LP>(g01[1]); hv[2] = to_lp<LP>(g23[0]); hv[3] = to_lp<LP>(g23[1]);
                        *reinterpret_cast<lp4*>(ys + (i * 32 + fr) * PA + (wave * 32 + 8 * gq + 4 * fk) * 2) = hv;
                    }
                }
                __syncthreads();
                SPEI_STAMP(stamps, 9 + 3 * half);
#pragma unroll
                for (int i = 0; i < GW; ++i) tn[i] = *reinterpret_cast<const lp8*>(hbase + i * 32 * PA + rot * 32);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    lp8 tc[GW];
#pragma unroll
                    for (int i = 0; i < GW; ++i) tc[i] = tn[i];
                    if (s + 1 < 16) {
#pragma unroll
                        for (int i = 0; i < GW; ++i) tn[i] = *reinterpret_cast<const lp8*>(hbase + i * 32 * PA + ((rot + s + 1) & 15) * 32);
                    }
                    const lp8 w = ring[s % RING];
                    if (s + RING < 16) ring[s % RING] = *reinterpret_cast<const lp8*>(wptr + ((rot + s + RING) & 15) * 512);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < GW; ++i) acc2[i] = mfma16(tc[i], w, acc2[i]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                SPEI_STAMP(stamps, 10 + 3 * half);
                if (half == 0) {
                    wptr = p.w1 + (size_t)(8 + wave) * 16 * 512 + opaque(lane8);      // fc1, half 1
#pragma unroll
                    for (int d = 0; d < RING; ++d) ring[d] = *reinterpret_cast<const lp8*>(wptr + ((rot + d) & 15) * 512);
                }
            }
        }

        // ---- E. out = x1 + fc2 + b2: row layout, 16-byte stores ------------------------------------------------------------------------
        {
            const float bias = p.b2[wave * HD + fr];
#pragma unroll
            for (int i = 0; i < GW; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = acc2[i][4 * k + e] + bias;
                    quad_transpose4(a[0], a[1], a[2], a[3], et);
                    const int pix = tok_pix[i * 32 + 8 * k + 4 * fk + et];
                    if (pix >= 0)
                        *reinterpret_cast<f32x4*>(p.out + (size_t)pix * D + wave * HD + ecol) = f32x4{a[0], a[1], a[2], a[3]} + x1r[i][k];
                }
        }
        SPEI_STAMP(stamps, 14);
        __syncthreads();                                   // the token table and the slabs are reused by the next group
    }
}

template <typename LP>
int block_launch(const float* x, float* out, const void* yhat, const void* wq, const float* bq, const void* wkv, const float* bkv,
                 const void* wproj, const float* bproj, const float* relbias, const void* w1, const float* b1, const void* w2,
                 const float* b2, int H, int W, int shift, hipStream_t st) {
    BlockParams<LP> p;
    p.x = x; p.out = out; p.yhat = (const LP*)yhat; p.wq = (const LP*)wq; p.bq = bq; p.wkv = (const LP*)wkv; p.bkv = bkv;
    p.wproj = (const LP*)wproj; p.bproj = bproj; p.relbias = relbias; p.w1 = (const LP*)w1; p.b1 = b1; p.w2 = (const LP*)w2; p.b2 = b2;
    p.H = H; p.W = W; p.shift = shift; p.nwin = (H / WS) * (W / WS);
    p.ngroups = (p.nwin + GW - 1) / GW;
    p.stamps = spei_stamp_buffer();
    const size_t lds = (size_t)2 * ROWS * PA + (size_t)(8 * ROWS * 2 + ROWS * 2 + HID + 8 * NT * NT) * sizeof(float) + 2 * ROWS * sizeof(int);
    ensure_dyn_lds<&swin_block_kernel<LP>>(lds);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = p.ngroups < cus ? p.ngroups : cus;            // one persistent workgroup per CU
    hipLaunchKernelGGL(swin_block_kernel<LP>, dim3(grid), dim3(NTH), lds, st, p);
    SPEI_CHECK_LAUNCH("spei_swin_block16");
    return 0;
}

}  // namespace

extern "C" int spei_swin_block16(int fmt, const float* x, float* out, const void* yhat, const void* wq_frag, const float* bq,
                                 const void* wkv_frag, const float* bkv, const void* wproj_frag, const float* bproj,
                                 const float* relbias, const void* w1_frag, const float* b1, const void* w2_frag, const float* b2,
                                 int H, int W, int shift, spei_stream_t stream) {
    SPEI_REQUIRE(x && out && yhat && wq_frag && bq && wkv_frag && bkv && wproj_frag && bproj && relbias && w1_frag && b1 && w2_frag && b2,
                 "spei_swin_block16: null pointer");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_swin_block16: fmt=%d", fmt);
    SPEI_REQUIRE(H > 0 && W > 0 && H % WS == 0 && W % WS == 0, "spei_swin_block16: %dx%d is not a multiple of the 5x5 window", H, W);
    SPEI_REQUIRE(shift >= 0 && shift < WS, "spei_swin_block16: shift=%d", shift);
    SPEI_REQUIRE((int64_t)H * W < (1ll << 30), "spei_swin_block16: map too large");
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)yhat | (uintptr_t)wq_frag | (uintptr_t)wkv_frag | (uintptr_t)wproj_frag |
                  (uintptr_t)w1_frag | (uintptr_t)w2_frag) % 16 == 0, "spei_swin_block16: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (fmt == SPEI_F16)
        return block_launch<_Float16>(x, out, yhat, wq_frag, bq, wkv_frag, bkv, wproj_frag, bproj, relbias, w1_frag, b1, w2_frag, b2, H, W, shift, st);
    return block_launch<__bf16>(x, out, yhat, wq_frag, bq, wkv_frag, bkv, wproj_frag, bproj, relbias, w1_frag, b1, w2_frag, b2, H, W, shift, st);
}

// Fused attention branch of one cross-window Swin block on the gfx950 bf16 matrix pipe:
//
//     x <- x + proj( W-MSA( q = norm1(y) Wq,  [k, v] = norm1(x) Wkv ) )
//
// (reference model/swinir.py:238-278 SwinTransformerBlock.forward up to the first residual, :115-149
// WindowAttention.forward, :215-236 calculate_mask, :32-61 window_partition/reverse, torch.roll for the shift).
// Unfused this is LayerNorm -> two GEMMs -> attention -> GEMM: q, k, v and the attention output each make a round trip
// through HBM.  Here a 512-thread workgroup owns TWO 5x5 windows (50 tokens, padded to 2 x 32 rows) and wave h owns
// head h:
//   1. stage x (fp32 -> LayerNorm without affine -> bf16) and the pre-normalised y (bf16) of the 50 tokens into LDS;
//      the cyclic shift and the window partition are the token -> pixel map;
//   2. Q^T_h = Wq_h y^T, K^T_h = Wk_h x^T (weights as the MFMA A operand, tokens on the lanes), V_h = x Wv_h^T (tokens on
//      the accumulator rows); weights stream from HBM/L2 in fragment order;
//   3. S^T = K Q^T and O^T = V^T P^T WITHOUT leaving registers: a 32x32 f32 accumulator X is a valid bf16 MFMA operand
//      after a cvt of registers 8s..8s+7 (as A it yields X^T.B, as B it yields A.X, with the SAME permuted k order on both
//      sides), so K^T/Q^T feed S^T directly and V / P^T feed O^T; softmax is 16 registers + one cross-half shuffle per query;
//   4. O^T -> bf16 [token][256] slab in LDS (the only cross-wave exchange), proj GEMM, + bias + residual, 16-byte stores.
// HBM traffic per block: x read (+ once more from L2 for the residual), y-hat read, x written.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int D = 256, HD = 32, NT = 25, WS = 5;
constexpr int PA = 2 * D + 16;        // LDS row pitch (bytes)
constexpr int ROWS = 64;              // 2 windows x 32 (25 tokens + 7 pad rows)

struct AttnParams {
    const float* x;
    float* out;
    const __bf16* yhat;   // [M][256]
    const __bf16* wq;     // fragment order [8][1][16][64][8]
    const float* bq;
    const __bf16* wkv;    // fragment order [16][1][16][64][8]  (n-tiles 0..7 = K heads, 8..15 = V heads)
    const float* bkv;
    const __bf16* wproj;  // fragment order [8][1][16][64][8]
    const float* bproj;
    const float* relbias; // [8][25][25]
    int H, W, shift, nwin;
};

__device__ __forceinline__ int mask_region(int v, int n, int shift) { return v < n - WS ? 0 : (v < n - shift ? 1 : 2); }

template <int S>
__device__ __forceinline__ bf16x8 cvt8(const f32x16& a) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * S + j];
    return r;
}

__global__ __launch_bounds__(512) void attn_fused_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;                      // [64][PA]  LayerNorm(x), bf16
    unsigned char* ys = smem + ROWS * PA;          // [64][PA]  y-hat, bf16
    unsigned char* os = smem + 2 * ROWS * PA;      // [64][PA]  attention output, bf16
    int* tok_pix = reinterpret_cast<int*>(smem + 3 * ROWS * PA);   // [64] pixel index or -1
    int* tok_reg = tok_pix + ROWS;                                 // [64] shift-mask region id

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int h = wave;
    const int nwx = p.W / WS;

    if (tid < ROWS) {
        const int w = tid >> 5, t = tid & 31;
        const int win = blockIdx.x * 2 + w;
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

    // ---- 1. stage LN(x) and y-hat: wave w handles rows w, w+8, ... ----------------------------------------------
    {
        f32x4 xr[ROWS / 8];
        u32x4 yr[ROWS / 8];      // only lanes < 32 carry y data (32 lanes x 16 B = one 512-byte row)
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int pix = max(tok_pix[wave + 8 * i], 0);
            xr[i] = reinterpret_cast<const f32x4*>(p.x + (size_t)pix * D)[lane];
            yr[i] = reinterpret_cast<const u32x4*>(p.yhat + (size_t)pix * D)[lane & 31];
        }
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int r = wave + 8 * i;
            const bool ok = tok_pix[r] >= 0;
            f32x4 v = xr[i];
            const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
            v -= mean;
            const float var = wave_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) * (1.0f / 256.0f);
            v *= 1.0f / sqrtf(var + 1e-5f);
            bf16x4 hv;
            hv[0] = (__bf16)(ok ? v[0] : 0.f); hv[1] = (__bf16)(ok ? v[1] : 0.f);
            hv[2] = (__bf16)(ok ? v[2] : 0.f); hv[3] = (__bf16)(ok ? v[3] : 0.f);
            *reinterpret_cast<bf16x4*>(xs + r * PA + lane * 8) = hv;
            if (lane < 32) *reinterpret_cast<u32x4*>(ys + r * PA + lane * 16) = ok ? yr[i] : u32x4{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();

    // ---- 2. per head: Q^T, K^T (weights = A operand, tokens on lanes) and V (tokens on rows) ------------------------
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
        const __bf16* wqp = p.wq + (size_t)h * 16 * 512 + lane * 8;
        const __bf16* wkp = p.wkv + (size_t)h * 16 * 512 + lane * 8;
        const __bf16* wvp = p.wkv + (size_t)(8 + h) * 16 * 512 + lane * 8;
        const int rot = blockIdx.x & 15;                 // per-workgroup K rotation (L2 channel spreading)
        bf16x8 wqf[2], wkf[2], wvf[2];
        int ks = rot;
        wqf[0] = *reinterpret_cast<const bf16x8*>(wqp + ks * 512);
        wkf[0] = *reinterpret_cast<const bf16x8*>(wkp + ks * 512);
        wvf[0] = *reinterpret_cast<const bf16x8*>(wvp + ks * 512);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cur = i & 1;
            const int ksn = (ks + 1) & 15;
            if (i + 1 < 16) {
                wqf[cur ^ 1] = *reinterpret_cast<const bf16x8*>(wqp + ksn * 512);
                wkf[cur ^ 1] = *reinterpret_cast<const bf16x8*>(wkp + ksn * 512);
                wvf[cur ^ 1] = *reinterpret_cast<const bf16x8*>(wvp + ksn * 512);
            }
            const int ko = ks * 32 + fk * 16;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const bf16x8 yf = *reinterpret_cast<const bf16x8*>(ys + (w * 32 + fr) * PA + ko);
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + (w * 32 + fr) * PA + ko);
                qT[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wqf[cur], yf, qT[w], 0, 0, 0);
                kT[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wkf[cur], xf, kT[w], 0, 0, 0);
                vv[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, wvf[cur], vv[w], 0, 0, 0);
            }
            ks = ksn;
        }
    }

    // ---- 3. attention of head h on both windows, all in registers ------------------------------------------------------
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        // S^T[key][query] = sum_d K^T[d][key] Q^T[d][query]:  A = (K^T)^T from the accumulator, B = Q^T from the accumulator
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt8<0>(kT[w]), cvt8<0>(qT[w]), st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt8<1>(kT[w]), cvt8<1>(qT[w]), st, 0, 0, 0);
        const int qi = fr < NT ? fr : 0;
        const int qreg = tok_reg[w * 32 + qi];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
            float v = -INFINITY;
            if (key < NT) {
                v = st[r] + p.relbias[(h * NT + qi) * NT + key];
                if (p.shift > 0 && tok_reg[w * 32 + key] != qreg) v += -100.0f;
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
        ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt8<0>(vv[w]), cvt8<0>(st), ot, 0, 0, 0);
        ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt8<1>(vv[w]), cvt8<1>(st), ot, 0, 0, 0);
        // rows d = (r&3) + 8*(r>>2) + 4*fk, column = query token fr  ->  os[token][h*32 + d]
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (__bf16)ot[4 * g + e];
            *reinterpret_cast<bf16x4*>(os + (w * 32 + fr) * PA + (h * HD + 8 * g + 4 * fk) * 2) = o4;
        }
    }
    __syncthreads();

    // ---- 4. proj: wave h produces output channels [32h, 32h+32) for all 64 rows, + bias + residual --------------------
    {
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const __bf16* wpp = p.wproj + (size_t)h * 16 * 512 + lane * 8;
        const int rot = (blockIdx.x * 3) & 15;
        bf16x8 wf[2];
        int ks = rot;
        wf[0] = *reinterpret_cast<const bf16x8*>(wpp + ks * 512);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cur = i & 1;
            const int ksn = (ks + 1) & 15;
            if (i + 1 < 16) wf[cur ^ 1] = *reinterpret_cast<const bf16x8*>(wpp + ksn * 512);
            const int ko = ks * 32 + fk * 16;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(os + fr * PA + ko);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(os + (32 + fr) * PA + ko);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[cur], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[cur], acc[1], 0, 0, 0);
            ks = ksn;
        }
        const float bias = p.bproj[h * HD + fr];
        const int et = fr & 3, ecol = (fr >> 2) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = acc[i][4 * k + e] + bias;
                quad_transpose4(a[0], a[1], a[2], a[3], et);
                const int row = i * 32 + 8 * k + 4 * fk + et;
                const int pix = tok_pix[row];
                if (pix >= 0) {
                    f32x4 v = f32x4{a[0], a[1], a[2], a[3]};
                    v += *reinterpret_cast<const f32x4*>(p.x + (size_t)pix * D + h * HD + ecol);
                    *reinterpret_cast<f32x4*>(p.out + (size_t)pix * D + h * HD + ecol) = v;
                }
            }
        }
    }
}

}  // namespace

extern "C" int spei_attn_fused_bf16(const float* x, float* out, const void* yhat, const void* wq_frag, const float* bq,
                                    const void* wkv_frag, const float* bkv, const void* wproj_frag, const float* bproj,
                                    const float* relbias, int H, int W, int shift, spei_stream_t stream) {
    SPEI_REQUIRE(x && out && yhat && wq_frag && bq && wkv_frag && bkv && wproj_frag && bproj && relbias, "spei_attn_fused_bf16: null pointer");
    SPEI_REQUIRE(H > 0 && W > 0 && H % WS == 0 && W % WS == 0, "spei_attn_fused_bf16: %dx%d is not a multiple of the 5x5 window", H, W);
    SPEI_REQUIRE(shift >= 0 && shift < WS, "spei_attn_fused_bf16: shift=%d", shift);
    SPEI_REQUIRE((int64_t)H * W < (1ll << 30), "spei_attn_fused_bf16: map too large");
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)yhat | (uintptr_t)wq_frag | (uintptr_t)wkv_frag | (uintptr_t)wproj_frag) % 16 == 0,
                 "spei_attn_fused_bf16: 16-byte alignment required");
    AttnParams p;
    p.x = x; p.out = out; p.yhat = (const __bf16*)yhat; p.wq = (const __bf16*)wq_frag; p.bq = bq; p.wkv = (const __bf16*)wkv_frag;
    p.bkv = bkv; p.wproj = (const __bf16*)wproj_frag; p.bproj = bproj; p.relbias = relbias;
    p.H = H; p.W = W; p.shift = shift; p.nwin = (H / WS) * (W / WS);
    const size_t lds = (size_t)3 * ROWS * PA + 2 * ROWS * sizeof(int);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL(attn_fused_kernel, dim3((p.nwin + 1) / 2), dim3(512), lds, (hipStream_t)stream, p);
    SPEI_CHECK_LAUNCH("spei_attn_fused_bf16");
    return 0;
}

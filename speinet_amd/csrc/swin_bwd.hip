// Backward kernels of the cross-window-attention SwinIR blocks (the training step of trainer/trainer_swint.py
// and trainer_swint_hsa_nsf.py calls loss.backward() through model/swinir.py:238-281): LayerNorm(256), erf-GELU and the 5x5
// window attention core.  fp32 throughout (the gradients are compared with the reference's fp32 autograd); every reduction has
// a fixed order, so gradients are bitwise reproducible run to run.  The linear layers' data / weight gradients are the GEMM
// family's (spei_igemm_f32 with transposed weights, spei_conv_wgrad_f32 with ksize 1).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm(256) backward (nn.LayerNorm, model/swinir.py:244-245,279,528,776): with xhat = (x - mean) * rstd, g = gamma * dy:
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),   dgamma = sum_rows dy * xhat,   dbeta = sum_rows dy.
// One wave per token row (4 channels per lane, the forward kernel's mapping); a wave walks rows blockIdx*4 + wave, + 4 gridDim,
// ... and keeps its dgamma / dbeta partials in registers; the four waves of a block are combined through LDS in wave order:
// part[block][0][256] = dgamma partial, part[block][1][256] = dbeta partial (the host adds the blocks in index order).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm256_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ dy, float* __restrict__ dx,
                                                               float* __restrict__ part, int64_t M) {
    __shared__ float red[4][2][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
    if (gamma) g = reinterpret_cast<const float4*>(gamma)[lane];
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
    for (int64_t m = wave; m < M; m += nwaves) {
        const float4 v = reinterpret_cast<const float4*>(x + m * 256)[lane];
        const float4 d = reinterpret_cast<const float4*>(dy + m * 256)[lane];
        const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
        const float cx = v.x - mean, cy = v.y - mean, cz = v.z - mean, cw = v.w - mean;
        const float var = wave_sum((cx * cx + cy * cy) + (cz * cz + cw * cw)) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float hx = cx * rstd, hy = cy * rstd, hz = cz * rstd, hw = cw * rstd;
        const float gx = g.x * d.x, gy = g.y * d.y, gz = g.z * d.z, gw = g.w * d.w;
        const float m1 = wave_sum((gx + gy) + (gz + gw)) * (1.0f / 256.0f);
        const float m2 = wave_sum((gx * hx + gy * hy) + (gz * hz + gw * hw)) * (1.0f / 256.0f);
        float4 o;
        o.x = rstd * (gx - m1 - hx * m2);
        o.y = rstd * (gy - m1 - hy * m2);
        o.z = rstd * (gz - m1 - hz * m2);
        o.w = rstd * (gw - m1 - hw * m2);
        reinterpret_cast<float4*>(dx + m * 256)[lane] = o;
        ag.x += d.x * hx; ag.y += d.y * hy; ag.z += d.z * hz; ag.w += d.w * hw;
        ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
    }
    float* r0 = &red[wv][0][lane * 4];
    r0[0] = ag.x; r0[1] = ag.y; r0[2] = ag.z; r0[3] = ag.w;
    float* r1 = &red[wv][1][lane * 4];
    r1[0] = ab.x; r1[1] = ab.y; r1[2] = ab.z; r1[3] = ab.w;
    __syncthreads();
    const int c = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        part[((size_t)blockIdx.x * 2 + k) * 256 + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
}

// ---------------------------------------------------------------------------------------------------------------------
// erf-GELU (nn.GELU, model/swinir.py:19-27) forward on a saved pre-activation, and its derivative:
//   gelu(v) = v Phi(v),   gelu'(v) = Phi(v) + v phi(v),   Phi(v) = (1 + erf(v / sqrt 2)) / 2,  phi(v) = exp(-v^2 / 2) / sqrt(2 pi)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_d(float v) {
    return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * expf(-0.5f * v * v);
}
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ pre, float* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(pre)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dy, float* __restrict__ dpre,
                                                       int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(pre)[i], d = reinterpret_cast<const float4*>(dy)[i];
    reinterpret_cast<float4*>(dpre)[i] = make_float4(d.x * gelu_d(v.x), d.y * gelu_d(v.y), d.z * gelu_d(v.z), d.w * gelu_d(v.w));
}

// ---------------------------------------------------------------------------------------------------------------------
// Window attention backward (model/swinir.py:115-149 under autograd).  Same decomposition as the forward kernel in swin.hip:
// two 256-thread workgroups per 5x5 window, one wave per head, tokens padded 25 -> 32, everything on v_mfma_f32_32x32x2_f32.
// A 32x32 accumulator holds rows in its registers and columns on its lanes, and used as the next MFMA's A operand it
// contracts over its ROWS.  Softmax statistics want the keys in the registers (the "transposed" pass, as in the forward);
// dV and dK contract over the queries, so a second, non-transposed pass recomputes P and dS with the queries in the registers,
// taking the row maxima / normalisers / dot products of the first pass from LDS:
//   pass T (lane = query q, registers = keys k):
//     S^T = K Q^T + bias + mask,  P = softmax_k,  dP^T = V dO^T,  r_q = sum_k P dP,  dS^T = P (dP - r_q)
//     dBias[q][k] = dS  (per window: the host sums the windows),   dQ = dS K   (A = dS^T registers)
//   pass N (lane = key k, registers = queries q):
//     S = Q K^T + bias + mask,  P = exp(S - max_q) / sum_q,  dP = dO V^T,  dS = P (dP - r_q)
//     dV = P^T dO (A = P registers),   dK = dS^T Q (A = dS registers)
// q arrives pre-scaled (the caller folds head_dim^-0.5 into the q projection, as the inference path does), so dq is the
// gradient with respect to the scaled q.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WS = 5, NT = 25, HD = 32, LDT = HD + 1;

__device__ __forceinline__ int mask_region(int v, int n, int shift) { return v < n - WS ? 0 : (v < n - shift ? 1 : 2); }

__global__ __launch_bounds__(256) void window_attention_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                                   const float* __restrict__ relbias, const float* __restrict__ dout,
                                                                   float* __restrict__ dq, float* __restrict__ dkv,
                                                                   float* __restrict__ dbias_part, int H, int W, int shift) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    {   // blockIdx.y = sample of a batch of equally sized maps stored one after the other
        const size_t z = (size_t)blockIdx.y * H * W;
        q += z * 256; kv += z * 512; dout += z * 256; dq += z * 256; dkv += z * 512;
        dbias_part += (size_t)blockIdx.y * (H / WS) * (W / WS) * 8 * NT * NT;
    }
    typedef float (*Tile)[32][LDT];
    Tile sQ = reinterpret_cast<Tile>(smem);
    Tile sK = reinterpret_cast<Tile>(smem + 4 * 32 * LDT);
    Tile sV = reinterpret_cast<Tile>(smem + 8 * 32 * LDT);
    Tile sD = reinterpret_cast<Tile>(smem + 12 * 32 * LDT);
    float* stat = smem + 16 * 32 * LDT;                       // [4 heads][3][32]: row max, 1 / row sum, r_q
    int* tok_pix = reinterpret_cast<int*>(stat + 4 * 3 * 32);
    int* tok_reg = tok_pix + 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nwx = W / WS;
    const int win = blockIdx.x >> 1;
    const int wy = win / nwx, wx = win - wy * nwx;
    if (threadIdx.x < 32) {
        int pix = 0, reg = 0;
        if (threadIdx.x < NT) {
            const int ys = wy * WS + threadIdx.x / WS, xs = wx * WS + threadIdx.x % WS;
            int yo = ys + shift, xo = xs + shift;
            if (yo >= H) yo -= H;
            if (xo >= W) xo -= W;
            pix = yo * W + xo;
            reg = shift > 0 ? 3 * mask_region(ys, H, shift) + mask_region(xs, W, shift) : 0;
        }
        tok_pix[threadIdx.x] = pix;
        tok_reg[threadIdx.x] = reg;
    }
    __syncthreads();
    const int h = (blockIdx.x & 1) * 4 + wave;
    const int hl = wave;
    for (int i = lane; i < 32 * 8; i += 64) {
        const int r = i >> 3, c4 = (i & 7) * 4;
        float4 vq = make_float4(0.f, 0.f, 0.f, 0.f), vk = vq, vv = vq, vd = vq;
        if (r < NT) {
            const size_t p = (size_t)tok_pix[r];
            vq = *reinterpret_cast<const float4*>(q + p * 256 + h * HD + c4);
            vk = *reinterpret_cast<const float4*>(kv + p * 512 + h * HD + c4);
            vv = *reinterpret_cast<const float4*>(kv + p * 512 + 256 + h * HD + c4);
            vd = *reinterpret_cast<const float4*>(dout + p * 256 + h * HD + c4);
        }
        float* d = &sQ[hl][r][c4]; d[0] = vq.x; d[1] = vq.y; d[2] = vq.z; d[3] = vq.w;
        d = &sK[hl][r][c4];        d[0] = vk.x; d[1] = vk.y; d[2] = vk.z; d[3] = vk.w;
        d = &sV[hl][r][c4];        d[0] = vv.x; d[1] = vv.y; d[2] = vv.z; d[3] = vv.w;
        d = &sD[hl][r][c4];        d[0] = vd.x; d[1] = vd.y; d[2] = vd.z; d[3] = vd.w;
    }
    __syncthreads();
    const int fr = lane & 31, fk = lane >> 5;
    float* smx = stat + (hl * 3 + 0) * 32;
    float* sinv = stat + (hl * 3 + 1) * 32;
    float* srq = stat + (hl * 3 + 2) * 32;

    // ---- pass T: lane = query fr, register r = key (r&3) + 8 (r>>2) + 4 fk ---------------------------------------------------
    {
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < HD; kk += 2) {
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[hl][fr][kk + fk], sQ[hl][fr][kk + fk], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[hl][fr][kk + fk], sD[hl][fr][kk + fk], dp, 0, 0, 0);
        }
        const int qi = fr < NT ? fr : 0;
        const int qreg = tok_reg[qi];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
            float v = -INFINITY;
            if (key < NT) {
                v = st[r] + relbias[(h * NT + qi) * NT + key];
                if (shift > 0 && tok_reg[key] != qreg) v += -100.0f;
            }
            st[r] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = expf(st[r] - mx);
            st[r] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        float rq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] *= inv;
            rq += st[r] * dp[r];
        }
        rq += __shfl_xor(rq, 32, 64);
        if (fk == 0) { smx[fr] = mx; sinv[fr] = inv; srq[fr] = rq; }
        f32x16 dqa;
#pragma unroll
        for (int r = 0; r < 16; ++r) dqa[r] = 0.f;
        float* bp = dbias_part + ((size_t)win * 8 + h) * NT * NT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
            const float ds = (fr < NT && key < NT) ? st[r] * (dp[r] - rq) : 0.f;
            if (fr < NT && key < NT) bp[fr * NT + key] = ds;
            dqa = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, sK[hl][key][fr], dqa, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = (r & 3) + 8 * (r >> 2) + 4 * fk;
            if (qq < NT) dq[(size_t)tok_pix[qq] * 256 + h * HD + fr] = dqa[r];
        }
    }
    __syncthreads();                                        // the statistics of pass T are in LDS (one wave per head, but cheap and safe)

    // ---- pass N: lane = key fr, register r = query (r&3) + 8 (r>>2) + 4 fk ---------------------------------------------------
    {
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < HD; kk += 2) {
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[hl][fr][kk + fk], sK[hl][fr][kk + fk], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sD[hl][fr][kk + fk], sV[hl][fr][kk + fk], dp, 0, 0, 0);
        }
        const int key = fr;
        const int kreg = tok_reg[key < NT ? key : 0];
        f32x16 dva, dka;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dva[r] = 0.f; dka[r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = (r & 3) + 8 * (r >> 2) + 4 * fk;
            float pv = 0.f, ds = 0.f;
            if (qq < NT && key < NT) {
                float v = st[r] + relbias[(h * NT + qq) * NT + key];
                if (shift > 0 && tok_reg[qq] != kreg) v += -100.0f;
                pv = expf(v - smx[qq]) * sinv[qq];
                ds = pv * (dp[r] - srq[qq]);
            }
            dva = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, sD[hl][qq][fr], dva, 0, 0, 0);
            dka = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, sQ[hl][qq][fr], dka, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = (r & 3) + 8 * (r >> 2) + 4 * fk;          // output row = key
            if (kk < NT) {
                dkv[(size_t)tok_pix[kk] * 512 + h * HD + fr] = dka[r];
                dkv[(size_t)tok_pix[kk] * 512 + 256 + h * HD + fr] = dva[r];
            }
        }
    }
}

}  // namespace

extern "C" int64_t spei_ln_bwd_blocks(int64_t M) {
    const int64_t b = (M + 3) / 4;
    return b < 512 ? (b > 0 ? b : 1) : 512;
}

extern "C" int spei_layernorm256_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* part, int64_t M,
                                     spei_stream_t stream) {
    SPEI_REQUIRE(x && dy && dx && part && M > 0, "spei_layernorm256_bwd: bad arguments");
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma) % 16 == 0, "spei_layernorm256_bwd: 16-byte alignment required");
    hipLaunchKernelGGL(layernorm256_bwd_kernel, dim3((unsigned)spei_ln_bwd_blocks(M)), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, dx, part, M);
    SPEI_CHECK_LAUNCH("spei_layernorm256_bwd");
    return 0;
}

extern "C" int spei_gelu_fwd(const float* pre, float* out, int64_t n, spei_stream_t stream) {
    SPEI_REQUIRE(pre && out && n > 0 && n % 4 == 0, "spei_gelu_fwd: bad arguments");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, out, n / 4);
    SPEI_CHECK_LAUNCH("spei_gelu_fwd");
    return 0;
}

extern "C" int spei_gelu_bwd(const float* pre, const float* dy, float* dpre, int64_t n, spei_stream_t stream) {
    SPEI_REQUIRE(pre && dy && dpre && n > 0 && n % 4 == 0, "spei_gelu_bwd: bad arguments");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, dy, dpre, n / 4);
    SPEI_CHECK_LAUNCH("spei_gelu_bwd");
    return 0;
}

extern "C" int spei_window_attention_bwd(const float* q, const float* kv, const float* relbias, const float* dout, float* dq, float* dkv,
                                         float* dbias_part, int H, int W, int shift, int batch, spei_stream_t stream) {
    SPEI_REQUIRE(batch >= 1 && batch <= 65535, "spei_window_attention_bwd: batch=%d", batch);
    SPEI_REQUIRE(q && kv && relbias && dout && dq && dkv && dbias_part, "spei_window_attention_bwd: null pointer");
    SPEI_REQUIRE(H > 0 && W > 0 && H % WS == 0 && W % WS == 0, "spei_window_attention_bwd: %dx%d is not a multiple of the 5x5 window", H, W);
    SPEI_REQUIRE(shift >= 0 && shift < WS, "spei_window_attention_bwd: shift=%d", shift);
    const size_t lds = (size_t)(16 * 32 * LDT + 4 * 3 * 32 + 64) * sizeof(float);
    ensure_dyn_lds<&window_attention_bwd_kernel>(lds);
    hipLaunchKernelGGL(window_attention_bwd_kernel, dim3(2 * (H / WS) * (W / WS), batch), dim3(256), lds, (hipStream_t)stream, q, kv, relbias, dout,
                       dq, dkv, dbias_part, H, W, shift);
    SPEI_CHECK_LAUNCH("spei_window_attention_bwd");
    return 0;
}

// out[m][n] = x[m][n] * rowscale[m]: the DropPath factor on a branch gradient before its weight / bias gradient sums
// (model/swinir.py:278-279: x = shortcut + drop_path(branch); one factor per sample, expanded to rows by the caller).  N % 4 == 0.
namespace {
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ rowscale, float* __restrict__ out,
                                                         int64_t M, int n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * n4) return;
    const float s = rowscale[i / n4];
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
}
}  // namespace

extern "C" int spei_scale_rows(const float* x, const float* rowscale, float* out, int64_t M, int N, spei_stream_t stream) {
    SPEI_REQUIRE(x && rowscale && out && M > 0 && N > 0 && N % 4 == 0, "spei_scale_rows: bad arguments");
    const int64_t n = M * (N / 4);
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rowscale, out, M, N / 4);
    SPEI_CHECK_LAUNCH("spei_scale_rows");
    return 0;
}

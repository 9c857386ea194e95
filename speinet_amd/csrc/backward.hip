// Backward kernels, first slice of the training step (reference trainer/trainer_swint_hsa_nsf.py:18-51 calls
// loss.backward() on the same graph): everything the encoder / decoder stacks of recons_net need — convolution weight and
// bias gradients, the ResBlock's gated residual sum backward and the statistics its gates are built from.  fp32 throughout
// (gradients are compared with the reference's fp32 autograd); every reduction is two-stage with a fixed combination order, so
// gradients are bitwise reproducible run to run.  The data gradient of a convolution needs no kernel of its own: it is the
// transposed convolution spei_igemm_f32(mode = SPEI_CONV_TRANSPOSED) of dY with the weights' last two axes swapped.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// conv weight gradient:  dW[t][n][k] = sum_m dY[m][n] * X[src(m, t)][k]     (m = output pixel, zero outside the input map)
// One 32 x 32 tile of one tap per workgroup and pixel chunk on v_mfma_f32_32x32x2_f32: A = dY^T (row n, reduction index =
// pixel), B = X (reduction index = pixel, column k).  Both operands are read straight from global memory: a lane reads one
// float of a pixel row, 32 consecutive lanes one 128-byte row segment.
// ---------------------------------------------------------------------------------------------------------------------
struct WgradParams {
    const float* x;
    const float* dy;
    float* part;          // [nchunks][T][N][K]
    float* bpart;         // [nchunks][N] column sums of dy (the bias gradient), or NULL: taken by the (tap 0, k-tile 0) workgroups, which
                          // stream every dy element of their channel tile through registers anyway
    int ldx, ldy, K, N, Hin, Win, Hout, Wout, ks, stride, pad, chunk_px, ntn, ntk;
    int batch;            // equally sized maps stored one after the other: the pixel index runs over all of them
};

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    int tile = blockIdx.x;
    const int kt = tile % p.ntk; tile /= p.ntk;
    const int nt = tile % p.ntn;
    const int t = tile / p.ntn;
    const int ty = t / p.ks, tx = t - ty * p.ks;
    const int n = nt * 32 + fr, k = kt * 32 + fr;
    const int M = p.Hout * p.Wout * p.batch;                // output pixels of all samples; oy runs over the stacked maps
    const int c0 = blockIdx.y * p.chunk_px;
    const int per_wave = p.chunk_px / 4;                    // chunk_px is a multiple of 8
    const int m0 = c0 + wave * per_wave, m1 = min(M, m0 + per_wave);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float asum = 0.f;
    int m = m0 + fk;
    int oy = m / p.Wout, ox = m - oy * p.Wout;
    int smp = oy / p.Hout;                                  // sample of this output row
    oy -= smp * p.Hout;
    for (int mm = m0; mm < m1; mm += 2) {
        float a = 0.f, b = 0.f;
        if (m < m1) {
            if (n < p.N) a = p.dy[(size_t)m * p.ldy + n];
            const int iy = oy * p.stride - p.pad + ty, ix = ox * p.stride - p.pad + tx;
            if (k < p.K && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win)
                b = p.x[(((size_t)smp * p.Hin + iy) * p.Win + ix) * p.ldx + k];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        asum += a;
        m += 2;
        ox += 2;
        while (ox >= p.Wout) {
            ox -= p.Wout;
            if (++oy == p.Hout) { oy = 0; ++smp; }
        }
    }
    __shared__ float bred[4][64];
    bred[wave][lane] = asum;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (p.bpart && t == 0 && kt == 0 && tid < 32 && nt * 32 + tid < p.N) {
        float b = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) b += bred[wv][tid] + bred[wv][tid + 32];       // pixel parity 0 then 1 of each wave, waves in order
        p.bpart[(size_t)blockIdx.y * p.N + nt * 32 + tid] = b;
    }
    if (wave == 0) {
        const int T = p.ks * p.ks;
        float* dst = p.part + (((size_t)blockIdx.y * T + t) * p.N) * p.K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
            const int row = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;       // n
            if (row < p.N && k < p.K) dst[(size_t)row * p.K + k] = v;
        }
    }
}

// The same weight gradient as split products on the 16-bit matrix pipe (train_precision = "bf16x3"): per 16 output pixels of one output
// row a lane gathers its 8 pixels of dY column n and of X column k straight from global memory (32 consecutive lanes read one
// 128-byte row segment, as in the fp32 kernel), splits each value a = ah + al (ah = bf16(a), al = bf16(a - ah)) and issues
// al*bh + ah*bl + ah*bh on v_mfma_f32_32x32x16_bf16 — three MFMAs (96 matrix-pipe cycles) where the fp32 form needs eight
// v_mfma_f32_32x32x2_f32 (512).  Work unit: a 16-pixel segment of one output row (zero lanes past the row's end), segments dealt to
// the chunks / waves in order; partial layout and the two-stage sum are the fp32 kernel's (fixed order: bitwise reproducible).
struct Wgrad16Params {
    const float* x;
    const float* dy;
    float* part;
    float* bpart;
    int ldx, ldy, K, N, Hin, Win, Hout, Wout, ks, stride, pad, ntn, ntk;
    int batch, nseg_row, nseg, chunk_seg;    // segments per output row, in all, per chunk (a multiple of 4)
};

__global__ __launch_bounds__(256) void conv_wgrad16_kernel(const Wgrad16Params p) {
    typedef lpv<__bf16>::x8 bf8;
    __shared__ float red[4][16][64];
    __shared__ float bred[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    int tile = blockIdx.x;
    const int kt = tile % p.ntk; tile /= p.ntk;
    const int nt = tile % p.ntn;
    const int t = tile / p.ntn;
    const int ty = t / p.ks, tx = t - ty * p.ks;
    const int n = nt * 32 + fr, k = kt * 32 + fr;
    const bool nok = n < p.N, kok = k < p.K;
    const int per_wave = p.chunk_seg / 4;
    const int s0 = blockIdx.y * p.chunk_seg + wave * per_wave, s1 = min(p.nseg, s0 + per_wave);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float asum = 0.f;
    float a[8], b[8], an[8], bn[8];
    auto load = [&](int sg, float* av, float* bv) __attribute__((always_inline)) {
        const int row = sg / p.nseg_row, sx = sg - row * p.nseg_row;          // row = sample * Hout + oy
        const int smp = row / p.Hout, oy = row - smp * p.Hout;
        const int iy = oy * p.stride - p.pad + ty;
        const bool rowok = iy >= 0 && iy < p.Hin;
        const int ox0 = sx * 16 + 8 * fk;
        const float* dyp = p.dy + ((size_t)row * p.Wout + ox0) * p.ldy + n;
        const float* xp = p.x + (((size_t)smp * p.Hin + (rowok ? iy : 0)) * p.Win) * p.ldx + k;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ox = ox0 + j;
            const int ix = ox * p.stride - p.pad + tx;
            const bool in = ox < p.Wout;
            av[j] = (in && nok) ? dyp[(size_t)j * p.ldy] : 0.f;
            bv[j] = (in && kok && rowok && ix >= 0 && ix < p.Win) ? xp[(size_t)ix * p.ldx] : 0.f;
        }
    };
    if (s0 < s1) load(s0, an, bn);
    for (int sg = s0; sg < s1; ++sg) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = an[j]; b[j] = bn[j]; }
        if (sg + 1 < s1) load(sg + 1, an, bn);                                   // next segment's gathers under this one's products
        bf8 ah, al, bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ah[j] = (__bf16)a[j];
            al[j] = (__bf16)(a[j] - (float)ah[j]);
            bh[j] = (__bf16)b[j];
            bl[j] = (__bf16)(b[j] - (float)bh[j]);
            asum += a[j];
        }
        acc = mfma16(al, bh, acc);
        acc = mfma16(ah, bl, acc);
        acc = mfma16(ah, bh, acc);
    }
    bred[wave][lane] = asum;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (p.bpart && t == 0 && kt == 0 && tid < 32 && nt * 32 + tid < p.N) {
        float bsum = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) bsum += bred[wv][tid] + bred[wv][tid + 32];
        p.bpart[(size_t)blockIdx.y * p.N + nt * 32 + tid] = bsum;
    }
    if (wave == 0) {
        const int T = p.ks * p.ks;
        float* dst = p.part + (((size_t)blockIdx.y * T + t) * p.N) * p.K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
            const int row = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;       // n
            if (row < p.N && k < p.K) dst[(size_t)row * p.K + k] = v;
        }
    }
}

// Stride-1 layers: a workgroup takes a whole ROW of taps (ty; tx = 0 .. KS-1) of its tile and chunk.  The dY fragment is gathered and
// split once per segment for the KS taps, and the X values of the KS shifted windows are one run of 8 + KS - 1 pixels per lane,
// gathered and split once: 3 KS MFMAs per ~(20 + 3 KS) gathers / conversions per lane instead of 3 per ~16 — the per-tap form above
// was bound by those, not by the matrix pipe (conv_wgrad16: 193 us per layer at batch 20).
template <int KS>
__global__ __launch_bounds__(256) void conv_wgrad16_row_kernel(const Wgrad16Params p) {
    typedef lpv<__bf16>::x8 bf8;
    constexpr int NX = 8 + KS - 1;
    __shared__ float red[4][16][64];
    __shared__ float bred[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    int tile = blockIdx.x;
    const int kt = tile % p.ntk; tile /= p.ntk;
    const int nt = tile % p.ntn;
    const int ty = tile / p.ntn;
    const int n = nt * 32 + fr, k = kt * 32 + fr;
    const bool nok = n < p.N, kok = k < p.K;
    const int per_wave = p.chunk_seg / 4;
    const int s0 = blockIdx.y * p.chunk_seg + wave * per_wave, s1 = min(p.nseg, s0 + per_wave);
    f32x16 acc[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float asum = 0.f;
    float a[8], an[8], xx[NX], xn[NX];
    auto load = [&](int sg, float* av, float* xv) __attribute__((always_inline)) {
        const int row = sg / p.nseg_row, sx = sg - row * p.nseg_row;          // row = sample * Hout + oy   (Hin == Hout, Win == Wout)
        const int smp = row / p.Hout, oy = row - smp * p.Hout;
        const int iy = oy - p.pad + ty;
        const bool rowok = iy >= 0 && iy < p.Hin;
        const int ox0 = sx * 16 + 8 * fk;
        const float* dyp = p.dy + ((size_t)row * p.Wout + ox0) * p.ldy + n;
        const float* xp = p.x + (((size_t)smp * p.Hin + (rowok ? iy : 0)) * p.Win) * p.ldx + k;
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = (ox0 + j < p.Wout && nok) ? dyp[(size_t)j * p.ldy] : 0.f;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int ix = ox0 - p.pad + j;
            xv[j] = (kok && rowok && ix >= 0 && ix < p.Win) ? xp[(size_t)ix * p.ldx] : 0.f;
        }
    };
    if (s0 < s1) load(s0, an, xn);
    for (int sg = s0; sg < s1; ++sg) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = an[j];
#pragma unroll
        for (int j = 0; j < NX; ++j) xx[j] = xn[j];
        if (sg + 1 < s1) load(sg + 1, an, xn);
        bf8 ah, al;
        __bf16 xh[NX], xl[NX];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ah[j] = (__bf16)a[j];
            al[j] = (__bf16)(a[j] - (float)ah[j]);
            asum += a[j];
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            xh[j] = (__bf16)xx[j];
            xl[j] = (__bf16)(xx[j] - (float)xh[j]);
        }
        // pixels of the segment past the row's end carry dY = 0: their products vanish whatever X holds
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            bf8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { bh[j] = xh[j + t]; bl[j] = xl[j + t]; }
            acc[t] = mfma16(al, bh, acc[t]);
            acc[t] = mfma16(ah, bl, acc[t]);
            acc[t] = mfma16(ah, bh, acc[t]);
        }
    }
    bred[wave][lane] = asum;
    __syncthreads();
    if (p.bpart && ty == 0 && kt == 0 && tid < 32 && nt * 32 + tid < p.N) {
        float bsum = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) bsum += bred[wv][tid] + bred[wv][tid + 32];
        p.bpart[(size_t)blockIdx.y * p.N + nt * 32 + tid] = bsum;
    }
    const int T = p.ks * p.ks;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[t][r];
        __syncthreads();
        if (wave == 0) {
            float* dst = p.part + (((size_t)blockIdx.y * T + ty * p.ks + t) * p.N) * p.K;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
                const int row = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;       // n
                if (row < p.N && k < p.K) dst[(size_t)row * p.K + k] = v;
            }
        }
    }
}

// 1x1 layers (the Swin linears): no taps to share the gathers, so a workgroup takes a 64 x 64 tile (2 x 2 MFMA tiles): the dY values
// of two channel tiles and the X values of two feed twelve MFMAs per 32 gathers per lane instead of three per 16.
__global__ __launch_bounds__(256) void conv_wgrad16_1x1_kernel(const Wgrad16Params p) {
    typedef lpv<__bf16>::x8 bf8;
    __shared__ float red[4][16][64];
    __shared__ float bred[4][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int ntk2 = (p.ntk + 1) / 2;
    const int kt2 = blockIdx.x % ntk2, nt2 = blockIdx.x / ntk2;
    const int per_wave = p.chunk_seg / 4;
    const int s0 = blockIdx.y * p.chunk_seg + wave * per_wave, s1 = min(p.nseg, s0 + per_wave);
    int nn[2], kk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { nn[i] = (nt2 * 2 + i) * 32 + fr; kk[i] = (kt2 * 2 + i) * 32 + fr; }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float asum[2] = {0.f, 0.f};
    float a[2][8], x[2][8], an[2][8], xn[2][8];
    auto load = [&](int sg, float (*av)[8], float (*xv)[8]) __attribute__((always_inline)) {
        const int row = sg / p.nseg_row, sx = sg - row * p.nseg_row;          // 1x1, stride 1: input pixel == output pixel
        const int ox0 = sx * 16 + 8 * fk;
        const size_t m = (size_t)row * p.Wout + ox0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = ox0 + j < p.Wout;
                av[i][j] = (in && nn[i] < p.N) ? p.dy[(m + j) * p.ldy + nn[i]] : 0.f;
                xv[i][j] = (in && kk[i] < p.K) ? p.x[(m + j) * p.ldx + kk[i]] : 0.f;
            }
    };
    if (s0 < s1) load(s0, an, xn);
    for (int sg = s0; sg < s1; ++sg) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[i][j] = an[i][j]; x[i][j] = xn[i][j]; }
        if (sg + 1 < s1) load(sg + 1, an, xn);
        bf8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ah[i][j] = (__bf16)a[i][j];
                al[i][j] = (__bf16)(a[i][j] - (float)ah[i][j]);
                bh[i][j] = (__bf16)x[i][j];
                bl[i][j] = (__bf16)(x[i][j] - (float)bh[i][j]);
                asum[i] += a[i][j];
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = mfma16(al[i], bh[j], acc[i][j]);
                acc[i][j] = mfma16(ah[i], bl[j], acc[i][j]);
                acc[i][j] = mfma16(ah[i], bh[j], acc[i][j]);
            }
    }
    bred[wave][0][lane] = asum[0];
    bred[wave][1][lane] = asum[1];
    __syncthreads();
    if (p.bpart && kt2 == 0 && tid < 64) {
        const int i = tid >> 5, c = tid & 31;
        if ((nt2 * 2 + i) * 32 + c < p.N) {
            float bsum = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) bsum += bred[wv][i][c] + bred[wv][i][c + 32];
            p.bpart[(size_t)blockIdx.y * p.N + (nt2 * 2 + i) * 32 + c] = bsum;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[i][j][r];
            __syncthreads();
            if (wave == 0) {
                float* dst = p.part + ((size_t)blockIdx.y * p.N) * p.K;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
                    const int row = (nt2 * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;       // n
                    if (row < p.N && kk[j] < p.K) dst[(size_t)row * p.K + kk[j]] = v;
                }
            }
        }
}

// out[i] = sum over `nparts` partials, fixed order; a second array (the bias partials) rides in the same launch: its elements follow
// the first array's in the index space (one launch per weight gradient instead of two: 1 200 fewer launches per training step)
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t count, int nparts,
                                                          const float* __restrict__ part2, float* __restrict__ out2, int count2) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) {
        i -= count;
        if (i >= count2) return;
        part = part2; out = out2; count = count2;
    }
    float s = 0.f;
    for (int c = 0; c < nparts; ++c) s += part[(size_t)c * count + i];
    out[i] = s;
}

// ---------------------------------------------------------------------------------------------------------------------
// ReLU backward on a fused conv + ReLU output:  dz = dy where y > 0 else 0
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dz, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 a = reinterpret_cast<const float4*>(y)[i], d = reinterpret_cast<const float4*>(dy)[i];
    reinterpret_cast<float4*>(dz)[i] = make_float4(a.x > 0.f ? d.x : 0.f, a.y > 0.f ? d.y : 0.f, a.z > 0.f ? d.z : 0.f, a.w > 0.f ? d.w : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// ResBlock gate statistics for training (model/block.py:71-73 ZPool, :8-24 SE pooling), and the sums its backward needs.
//   stats:   rowmax / rowmean [H][C] over x, colmax / colmean [W][C] over y, mean [C] of a = x1              (PROD = false)
//   bwd:     rowsum [H][C], colsum [W][C], total [C] of a * b  (a = dOut, b = x1: the gradients of g1, g2 and s)  (PROD = true)
// Stage 1: a block owns an R x 64-pixel... tile of TR rows x TC columns, threads = channels x columns; stage 2 combines the tile
// partials in a fixed order.  C <= 128, C % 4 == 0.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GT = 16;        // tile rows = tile columns (16: a 200x200 training crop still gives 169 workgroups)

template <bool PROD>
__global__ __launch_bounds__(256) void plane_stats_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W, int C,
                                                          float* __restrict__ rowp_max, float* __restrict__ rowp_sum,
                                                          float* __restrict__ colp_max, float* __restrict__ colp_sum, int ntx, int nty) {
    {   // blockIdx.y = sample of a batch of equally sized maps: inputs and partial buffers move by one sample's extent
        const size_t z = blockIdx.y;
        a += z * H * W * C;
        if (PROD) b += z * H * W * C;
        const size_t pstride = 2 * ((size_t)ntx * H * C + (size_t)nty * W * C);
        if (rowp_max) rowp_max += z * pstride;
        rowp_sum += z * pstride;
        if (colp_max) colp_max += z * pstride;
        colp_sum += z * pstride;
    }
    // thread -> (channel c, column group): with C channels, 256 / C columns are processed side by side
    const int tid = threadIdx.x;
    const int c = tid % C, cg = tid / C, ncg = 256 / C;
    const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
    const int x0 = tx * GT, y0 = ty * GT;
    __shared__ float smax[256], ssum[256];
    // column partials: each thread owns columns x0 + cg, x0 + cg + ncg, ... ; row partials need a cross-thread combine per row
    for (int xx = cg; xx < GT; xx += ncg) {
        const int x = x0 + xx;
        float cm = -INFINITY, cs = 0.f;
        if (x < W)
            for (int yy = 0; yy < GT && y0 + yy < H; ++yy) {
                const size_t o = ((size_t)(y0 + yy) * W + x) * C + c;
                const float v = PROD ? a[o] * b[o] : a[o];
                cm = fmaxf(cm, v);
                cs += v;
            }
        if (x < W) {
            if (!PROD) colp_max[((size_t)ty * W + x) * C + c] = cm;
            colp_sum[((size_t)ty * W + x) * C + c] = cs;
        }
    }
    for (int yy = 0; yy < GT; ++yy) {
        const int y = y0 + yy;
        float rm = -INFINITY, rs = 0.f;
        if (y < H)
            for (int xx = cg; xx < GT && x0 + xx < W; xx += ncg) {
                const size_t o = ((size_t)y * W + x0 + xx) * C + c;
                const float v = PROD ? a[o] * b[o] : a[o];
                rm = fmaxf(rm, v);
                rs += v;
            }
        smax[tid] = rm;
        ssum[tid] = rs;
        __syncthreads();
        if (cg == 0 && y < H) {
            float m = smax[c], s = ssum[c];
            for (int j = 1; j < ncg; ++j) { m = fmaxf(m, smax[j * C + c]); s += ssum[j * C + c]; }
            if (!PROD) rowp_max[((size_t)tx * H + y) * C + c] = m;
            rowp_sum[((size_t)tx * H + y) * C + c] = s;
        }
        __syncthreads();
    }
}

// stage 2: lines [L][C] from `nt` partials [nt][L][C]; scale applied to the sum (1/W for a mean, 1 for a plain sum)
__global__ __launch_bounds__(256) void plane_combine_kernel(const float* __restrict__ pmax, const float* __restrict__ psum, int nt, int64_t LC,
                                                            float scale, float* __restrict__ omax, float* __restrict__ osum, int64_t pstride) {
    {   // blockIdx.y = sample: partials move by pstride floats, outputs by LC
        const int64_t z = blockIdx.y;
        if (pmax) pmax += z * pstride;
        psum += z * pstride;
        if (omax) omax += z * LC;
        osum += z * LC;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= LC) return;
    float m = -INFINITY, s = 0.f;
    for (int k = 0; k < nt; ++k) {
        if (pmax) m = fmaxf(m, pmax[(size_t)k * LC + i]);
        s += psum[(size_t)k * LC + i];
    }
    if (omax) omax[i] = m;
    osum[i] = s * scale;
}

// total[c] = scale * sum over lines of line_sum[l][c]   (one block, fixed order)
__global__ __launch_bounds__(256) void lines_total_kernel(const float* __restrict__ lines, int L, int C, float scale, float* __restrict__ total) {
    lines += (size_t)blockIdx.x * L * C;          // blockIdx.x = sample
    total += (size_t)blockIdx.x * C;
    __shared__ float red[256];
    const int tid = threadIdx.x, c = tid % C, g = tid / C, ng = 256 / C;
    float s = 0.f;
    for (int l = g; l < L; l += ng) s += lines[(size_t)l * C + c];
    red[tid] = s;
    __syncthreads();
    if (g == 0) {
        float a = 0.f;
        for (int j = 0; j < ng; ++j) a += red[j * C + c];
        total[c] = a * scale;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gated residual sum backward:  out = x + x1 * (s[c] + g1[y][c] + g2[x][c])
//   dx1 = dOut * (s + g1 + g2)
//       + d_rowmean[y][c] / W + d_colmean[x][c] / H + d_mean[c] / (H W)                       (mean pools)
//       + d_rowmax[y][c] [x1 == rowmax[y][c]] + d_colmax[x][c] [x1 == colmax[x][c]]           (max pools: the arg-max element)
// (dx = dOut is the caller's tensor itself.)
// ---------------------------------------------------------------------------------------------------------------------
struct ApplyBwdParams {
    const float *dout, *x1, *s, *g1, *g2, *rowmax, *colmax, *d_rowmax, *d_rowmean, *d_colmax, *d_colmean, *d_mean;
    float* dx1;
    int H, W, C;
};

__global__ __launch_bounds__(256) void resblock_apply_bwd_kernel(const ApplyBwdParams p) {
    const int cg = p.C / 4;
    const int64_t total = (int64_t)p.H * p.W * cg;
    const float iw = 1.0f / (float)p.W, ih = 1.0f / (float)p.H, ihw = 1.0f / ((float)p.H * (float)p.W);
    // batched launch (blockIdx.y = map): dense maps one after the other, per-map statistics and gate maps likewise
    const size_t mb = blockIdx.y;
    const size_t mo = mb * p.H * p.W * p.C, mr = mb * p.H * p.C, mc = mb * p.W * p.C, mm = mb * p.C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cg) * 4;
        const int64_t pix = i / cg;
        const int x = (int)(pix % p.W), y = (int)(pix / p.W);
        const size_t o = mo + pix * p.C + c, ro = mr + (size_t)y * p.C + c, co = mc + (size_t)x * p.C + c;
        const f32x4 d = *reinterpret_cast<const f32x4*>(p.dout + o), v = *reinterpret_cast<const f32x4*>(p.x1 + o);
        const f32x4 gate = *reinterpret_cast<const f32x4*>(p.s + mm + c) + (*reinterpret_cast<const f32x4*>(p.g1 + ro) + *reinterpret_cast<const f32x4*>(p.g2 + co));
        f32x4 r = d * gate + *reinterpret_cast<const f32x4*>(p.d_rowmean + ro) * iw + *reinterpret_cast<const f32x4*>(p.d_colmean + co) * ih +
                  *reinterpret_cast<const f32x4*>(p.d_mean + mm + c) * ihw;
        const f32x4 rm = *reinterpret_cast<const f32x4*>(p.rowmax + ro), cm = *reinterpret_cast<const f32x4*>(p.colmax + co);
        const f32x4 drm = *reinterpret_cast<const f32x4*>(p.d_rowmax + ro), dcm = *reinterpret_cast<const f32x4*>(p.d_colmax + co);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (v[e] == rm[e]) r[e] += drm[e];
            if (v[e] == cm[e]) r[e] += dcm[e];
        }
        *reinterpret_cast<f32x4*>(p.dx1 + o) = r;
    }
}

}  // namespace

// partial slices a weight gradient may be split into (the workspace holds that many copies of it): 64 for the per-tap kernels; the
// tap-row kernel has 1 / ksize of the workgroups per slice and takes up to 256 slices for ~2k workgroups per launch
static int wgrad_slices(int N, int K, int ksize) {
    if (ksize == 1) {
        // 1x1 layers (the Swin linears): 2 x 2 tiles per workgroup; their partial volume (slices x N x K floats, written and read back by
        // the partial-sum launch) is what the extra slices cost, so only as many as give ~512 workgroups
        const int per = cdiv(cdiv(N, 32), 2) * cdiv(cdiv(K, 32), 2);
        const int want = cdiv(512, per);
        return want < 8 ? 8 : (want > 64 ? 64 : want);
    }
    const int per_slice = ksize * cdiv(N, 32) * cdiv(K, 32);
    int want = cdiv(2048, per_slice);
    want = want < 16 ? 16 : (want > 256 ? 256 : want);
    return want > 64 ? want : 64;
}

// slices the workspace holds: the fp32 kernel always cuts the pixels into up to 64 chunks
static int wgrad_ws_slices(int N, int K, int ksize) { const int s = wgrad_slices(N, K, ksize); return s > 64 ? s : 64; }

extern "C" int64_t spei_wgrad_ws_floats(int Hout, int Wout, int N, int K, int ksize) {
    const int64_t sl = wgrad_ws_slices(N, K, ksize);
    return sl * ksize * ksize * N * K + sl * 256;
}

static int conv_wgrad_run(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* ws, int Hin, int Win,
                          int Hout, int Wout, int N, int K, int ksize, int stride, int pad, int batch, spei_stream_t stream, bool split = false) {
    SPEI_REQUIRE(x && dy && dw && ws, "spei_conv_wgrad_f32: null pointer");
    SPEI_REQUIRE(batch >= 1 && (int64_t)batch * Hout * Wout < (1ll << 30), "spei_conv_wgrad_f32: batch=%d", batch);
    SPEI_REQUIRE(N > 0 && K > 0 && N <= 256 && ldx >= K && ldy >= N, "spei_conv_wgrad_f32: N=%d K=%d ldx=%d ldy=%d", N, K, ldx, ldy);
    SPEI_REQUIRE(ksize == 1 || ksize == 3 || ksize == 5, "spei_conv_wgrad_f32: ksize=%d", ksize);
    SPEI_REQUIRE(stride == 1 || stride == 2, "spei_conv_wgrad_f32: stride=%d", stride);
    SPEI_REQUIRE(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && (int64_t)Hout * Wout < (1ll << 30), "spei_conv_wgrad_f32: bad map size");
    SPEI_REQUIRE(Hout == (Hin + 2 * pad - ksize) / stride + 1 && Wout == (Win + 2 * pad - ksize) / stride + 1,
                 "spei_conv_wgrad_f32: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", Hout, Wout, Hin, Win, ksize, stride, pad);
    hipStream_t st = (hipStream_t)stream;
    const int M = Hout * Wout * batch, T = ksize * ksize;
    if (split) {
        Wgrad16Params q;
        q.x = x; q.dy = dy; q.part = ws; q.ldx = ldx; q.ldy = ldy; q.K = K; q.N = N; q.batch = batch;
        q.Hin = Hin; q.Win = Win; q.Hout = Hout; q.Wout = Wout; q.ks = ksize; q.stride = stride; q.pad = pad;
        q.nseg_row = cdiv(Wout, 16);
        q.nseg = batch * Hout * q.nseg_row;
        const bool rows = stride == 1;            // a workgroup per tap ROW: 1 / ksize of the workgroups per chunk, so more chunks
        q.ntn = cdiv(N, 32); q.ntk = cdiv(K, 32);
        const int want = rows ? wgrad_slices(N, K, ksize) : 64;
        int chunk = cdiv(q.nseg, want);
        chunk = ((chunk + 3) / 4) * 4;
        if (chunk < 4) chunk = 4;
        q.chunk_seg = chunk;
        const int nchunks = cdiv(q.nseg, chunk);
        float* bpart = ws + (size_t)wgrad_ws_slices(N, K, ksize) * T * N * K;
        q.bpart = dbias ? bpart : nullptr;
        if (!rows) hipLaunchKernelGGL(conv_wgrad16_kernel, dim3(T * q.ntn * q.ntk, nchunks), dim3(256), 0, st, q);
        else if (ksize == 5) hipLaunchKernelGGL(conv_wgrad16_row_kernel<5>, dim3(ksize * q.ntn * q.ntk, nchunks), dim3(256), 0, st, q);
        else if (ksize == 3) hipLaunchKernelGGL(conv_wgrad16_row_kernel<3>, dim3(ksize * q.ntn * q.ntk, nchunks), dim3(256), 0, st, q);
        else hipLaunchKernelGGL(conv_wgrad16_1x1_kernel, dim3(((q.ntn + 1) / 2) * ((q.ntk + 1) / 2), nchunks), dim3(256), 0, st, q);
        const int64_t count = (int64_t)T * N * K;
        hipLaunchKernelGGL(partial_sum_kernel, dim3(cdiv(count + (dbias ? N : 0), 256)), dim3(256), 0, st, ws, dw, count, nchunks,
                           dbias ? bpart : nullptr, dbias, dbias ? N : 0);
        SPEI_CHECK_LAUNCH("spei_conv_wgrad_bf16x3");
        return 0;
    }
    WgradParams p;
    p.batch = batch;
    p.x = x; p.dy = dy; p.part = ws; p.ldx = ldx; p.ldy = ldy; p.K = K; p.N = N;
    p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout; p.ks = ksize; p.stride = stride; p.pad = pad;
    int chunk = cdiv(M, 64);
    chunk = ((chunk + 7) / 8) * 8;
    if (chunk < 64) chunk = 64;
    p.chunk_px = chunk;
    const int nchunks = cdiv(M, chunk);
    p.ntn = cdiv(N, 32); p.ntk = cdiv(K, 32);
    float* bpart = ws + (size_t)wgrad_ws_slices(N, K, ksize) * T * N * K;
    p.bpart = dbias ? bpart : nullptr;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(T * p.ntn * p.ntk, nchunks), dim3(256), 0, st, p);
    const int64_t count = (int64_t)T * N * K;
    hipLaunchKernelGGL(partial_sum_kernel, dim3(cdiv(count + (dbias ? N : 0), 256)), dim3(256), 0, st, ws, dw, count, nchunks,
                       dbias ? bpart : nullptr, dbias, dbias ? N : 0);
    SPEI_CHECK_LAUNCH("spei_conv_wgrad_f32");
    return 0;
}

extern "C" int spei_conv_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* ws, int Hin, int Win,
                                   int Hout, int Wout, int N, int K, int ksize, int stride, int pad, spei_stream_t stream) {
    return conv_wgrad_run(x, ldx, dy, ldy, dw, dbias, ws, Hin, Win, Hout, Wout, N, K, ksize, stride, pad, 1, stream);
}

extern "C" int spei_conv_wgrad_f32_batched(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* ws, int Hin,
                                           int Win, int Hout, int Wout, int N, int K, int ksize, int stride, int pad, int batch,
                                           spei_stream_t stream) {
    return conv_wgrad_run(x, ldx, dy, ldy, dw, dbias, ws, Hin, Win, Hout, Wout, N, K, ksize, stride, pad, batch, stream);
}

extern "C" int spei_conv_wgrad_bf16x3_batched(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* ws, int Hin,
                                              int Win, int Hout, int Wout, int N, int K, int ksize, int stride, int pad, int batch,
                                              spei_stream_t stream) {
    return conv_wgrad_run(x, ldx, dy, ldy, dw, dbias, ws, Hin, Win, Hout, Wout, N, K, ksize, stride, pad, batch, stream, true);
}

extern "C" int spei_relu_bwd(const float* y, const float* dy, float* dz, int64_t n, spei_stream_t stream) {
    SPEI_REQUIRE(y && dy && dz && n > 0 && n % 4 == 0, "spei_relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, y, dy, dz, n / 4);
    SPEI_CHECK_LAUNCH("spei_relu_bwd");
    return 0;
}

extern "C" int64_t spei_plane_ws_floats(int H, int W, int C) {
    const int64_t ntx = (W + GT - 1) / GT, nty = (H + GT - 1) / GT;
    return 2 * (ntx * H * C + nty * W * C);
}

// prod == 0: rowmax, rowmean [H][C], colmax, colmean [W][C], mean [C] of a.   prod == 1: rowmax = colmax = NULL; rowmean / colmean /
// mean receive the plain SUMS of a * b over x, over y and over the map.
static int plane_stats_run(const float* a, const float* b, int prod, int H, int W, int C, float* rowmax, float* rowmean, float* colmax,
                           float* colmean, float* mean, float* ws, int batch, spei_stream_t stream) {
    SPEI_REQUIRE(a && rowmean && colmean && mean && ws && (!prod || b) && (prod || (rowmax && colmax)), "spei_plane_stats: null pointer");
    SPEI_REQUIRE((C == 32 || C == 64 || C == 128) && H > 0 && W > 0, "spei_plane_stats: C=%d (32/64/128 built)", C);
    SPEI_REQUIRE(batch >= 1 && batch <= 65535, "spei_plane_stats: batch=%d", batch);
    hipStream_t st = (hipStream_t)stream;
    const int ntx = (W + GT - 1) / GT, nty = (H + GT - 1) / GT;
    const int64_t pstride = 2 * ((int64_t)ntx * H * C + (int64_t)nty * W * C);          // partial floats per sample
    float* rpm = ws;
    float* rps = rpm + (size_t)ntx * H * C;
    float* cpm = rps + (size_t)ntx * H * C;
    float* cps = cpm + (size_t)nty * W * C;
    const dim3 g1(ntx * nty, batch);
    if (prod) hipLaunchKernelGGL(plane_stats_kernel<true>, g1, dim3(256), 0, st, a, b, H, W, C, rpm, rps, cpm, cps, ntx, nty);
    else hipLaunchKernelGGL(plane_stats_kernel<false>, g1, dim3(256), 0, st, a, b, H, W, C, rpm, rps, cpm, cps, ntx, nty);
    const int64_t hc = (int64_t)H * C, wc = (int64_t)W * C;
    // rows: plain sums first (the channel total is the sum of the row sums), then scaled to means when asked for
    hipLaunchKernelGGL(plane_combine_kernel, dim3(cdiv(hc, 256), batch), dim3(256), 0, st, prod ? nullptr : rpm, rps, ntx, hc, 1.0f,
                       prod ? nullptr : rowmax, rowmean, pstride);
    hipLaunchKernelGGL(lines_total_kernel, dim3(batch), dim3(256), 0, st, rowmean, H, C, prod ? 1.0f : 1.0f / ((float)H * (float)W), mean);
    if (!prod) hipLaunchKernelGGL(plane_combine_kernel, dim3(cdiv(hc, 256), batch), dim3(256), 0, st, nullptr, rps, ntx, hc, 1.0f / (float)W, nullptr,
                                  rowmean, pstride);
    hipLaunchKernelGGL(plane_combine_kernel, dim3(cdiv(wc, 256), batch), dim3(256), 0, st, prod ? nullptr : cpm, cps, nty, wc,
                       prod ? 1.0f : 1.0f / (float)H, prod ? nullptr : colmax, colmean, pstride);
    SPEI_CHECK_LAUNCH("spei_plane_stats");
    return 0;
}

extern "C" int spei_plane_stats(const float* a, const float* b, int prod, int H, int W, int C, float* rowmax, float* rowmean, float* colmax,
                                float* colmean, float* mean, float* ws, spei_stream_t stream) {
    return plane_stats_run(a, b, prod, H, W, C, rowmax, rowmean, colmax, colmean, mean, ws, 1, stream);
}

extern "C" int spei_plane_stats_batched(const float* a, const float* b, int prod, int H, int W, int C, float* rowmax, float* rowmean,
                                        float* colmax, float* colmean, float* mean, float* ws, int batch, spei_stream_t stream) {
    return plane_stats_run(a, b, prod, H, W, C, rowmax, rowmean, colmax, colmean, mean, ws, batch, stream);
}

static int resblock_apply_bwd_run(const float* dout, const float* x1, const float* s, const float* g1, const float* g2, const float* rowmax,
                                  const float* colmax, const float* d_rowmax, const float* d_rowmean, const float* d_colmax,
                                  const float* d_colmean, const float* d_mean, float* dx1, int batch, int H, int W, int C, spei_stream_t stream) {
    SPEI_REQUIRE(dout && x1 && s && g1 && g2 && rowmax && colmax && d_rowmax && d_rowmean && d_colmax && d_colmean && d_mean && dx1,
                 "spei_resblock_apply_bwd: null pointer");
    SPEI_REQUIRE(C % 4 == 0 && H > 0 && W > 0 && batch >= 1 && batch <= 65535, "spei_resblock_apply_bwd: bad shape");
    ApplyBwdParams p;
    p.dout = dout; p.x1 = x1; p.s = s; p.g1 = g1; p.g2 = g2; p.rowmax = rowmax; p.colmax = colmax; p.d_rowmax = d_rowmax;
    p.d_rowmean = d_rowmean; p.d_colmax = d_colmax; p.d_colmean = d_colmean; p.d_mean = d_mean; p.dx1 = dx1; p.H = H; p.W = W; p.C = C;
    const int64_t total = (int64_t)H * W * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(resblock_apply_bwd_kernel, dim3(blocks, batch), dim3(256), 0, (hipStream_t)stream, p);
    SPEI_CHECK_LAUNCH("spei_resblock_apply_bwd");
    return 0;
}

extern "C" int spei_resblock_apply_bwd(const float* dout, const float* x1, const float* s, const float* g1, const float* g2, const float* rowmax,
                                       const float* colmax, const float* d_rowmax, const float* d_rowmean, const float* d_colmax,
                                       const float* d_colmean, const float* d_mean, float* dx1, int H, int W, int C, spei_stream_t stream) {
    return resblock_apply_bwd_run(dout, x1, s, g1, g2, rowmax, colmax, d_rowmax, d_rowmean, d_colmax, d_colmean, d_mean, dx1, 1, H, W, C, stream);
}

extern "C" int spei_resblock_apply_bwd_batched(const float* dout, const float* x1, const float* s, const float* g1, const float* g2,
                                               const float* rowmax, const float* colmax, const float* d_rowmax, const float* d_rowmean,
                                               const float* d_colmax, const float* d_colmean, const float* d_mean, float* dx1, int batch, int H,
                                               int W, int C, spei_stream_t stream) {
    return resblock_apply_bwd_run(dout, x1, s, g1, g2, rowmax, colmax, d_rowmax, d_rowmean, d_colmax, d_colmean, d_mean, dx1, batch, H, W, C, stream);
}

// ---- training: a weight in the reference's layout -> the split (hi, lo) bf16 pair in MFMA fragment order, ONE launch -----------------
// The weights change every optimizer step, so every step re-packs each of the ~260 GEMM weights twice (forward, data gradient).  As
// torch ops that is permute + contiguous, two bf16 casts, a subtraction and two fragment-order copies per weight and use: ~3 700
// launches per step.  Here one thread writes one 16-byte fragment piece of hi and of lo.
//   mode 0 (forward):        GEMM weight [t][n][k] = w[n][k][t]                 (Conv2d [N][K][ks][ks] / Linear [N][K], ks = 1)
//   mode 1 (data gradient):  GEMM weight [t][n'][k'] = w[k'][n'][T - 1 - t]     (n' over the layer's INPUT channels K, k' over its
//                            outputs N: the stride-1 data gradient is the convolution with the taps reversed and the channel axes swapped)
// Fragment order (pack._frag): [n-tile][tap][k-step of 16][lane = h * 32 + n % 32][8], k = 16 ks + 8 h + j.
namespace {
__global__ __launch_bounds__(256) void pack_split16_kernel(const float* __restrict__ w, int N, int K, int T, int mode,
                                                           __bf16* __restrict__ fhi, __bf16* __restrict__ flo) {
    const int NN = mode ? K : N, KK = mode ? N : K;                 // rows / contraction length of the GEMM weight
    const int64_t pieces = (int64_t)T * NN * KK / 8;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pieces) return;
    const int lane = (int)(i & 63);
    int64_t r = i >> 6;
    const int ksteps = KK / 16;
    const int ks = (int)(r % ksteps); r /= ksteps;
    const int t = (int)(r % T);
    const int nt = (int)(r / T);
    const int n = nt * 32 + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    bf8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const float v = mode ? w[((size_t)k * K + n) * T + (T - 1 - t)] : w[((size_t)n * K + k) * T + t];
        const __bf16 h = (__bf16)v;
        hi[j] = h;
        lo[j] = (__bf16)(v - (float)h);
    }
    *reinterpret_cast<bf8*>(fhi + i * 8) = hi;
    *reinterpret_cast<bf8*>(flo + i * 8) = lo;
}
}  // namespace

extern "C" int spei_pack_split16(const float* w, int N, int K, int ksize, int mode, void* frag_hi, void* frag_lo, spei_stream_t stream) {
    SPEI_REQUIRE(w && frag_hi && frag_lo, "spei_pack_split16: null pointer");
    SPEI_REQUIRE(N > 0 && K > 0 && N % 32 == 0 && K % 32 == 0 && ksize >= 1 && (mode == 0 || mode == 1), "spei_pack_split16: N=%d K=%d ksize=%d mode=%d", N, K, ksize, mode);
    SPEI_REQUIRE(((uintptr_t)frag_hi | (uintptr_t)frag_lo) % 16 == 0, "spei_pack_split16: 16-byte alignment required");
    const int T = ksize * ksize;
    const int64_t pieces = (int64_t)T * N * K / 8;
    hipLaunchKernelGGL(pack_split16_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, N, K, T, mode,
                       (__bf16*)frag_hi, (__bf16*)frag_lo);
    SPEI_CHECK_LAUNCH("spei_pack_split16");
    return 0;
}

// Row a11 — LD sharpness detector at inference time: six focus measures per frame
// (reference inference_SPEINet.py:54-189: sobel :54, laplacian :68, mask :79, focus_measure_mis3 :118, _gra7 :134,
// _lap1 :144, _wave1 :152, _sta3 :161, _dct3 :169, generate_vars :177-189).  HBM-bound stencils + reductions; every
// reduction goes block partials -> one fixed-order final sum per frame (bitwise reproducible).
//
//   LAP1 = mean_win sum_win lap8(g)^2                  lap8 = [[1,1,1],[1,-8,1],[1,1,1]], zero pad
//   MIS3 = mean_win sum_win sum_{8 nbrs} |g(p)-g(q)|   zero pad
//   WAV1 = sum |LH|+|HL|+|HH| of a level-1 db6 DWT, zero extension        (parity unpinned, see oracle/detector_oracle.py)
//   GRA7 = mean_win sum_win (s - box_k(s))^2, s = |sobel(g)|, box_k = k x k mean, zero pad, /k^2
//   STA3 = mean_win sum_win (g - box_k(g))^2
//   DCT3 = mean_win (sum_win mask4x4(g))^2             mask4x4 valid convolution
// windows are the non-overlapping k x k tiles of lp_pool2d (floor), "mean_win" averages over them.
#include "common.h"

namespace {

constexpr int TB = 16;     // 16 x 16 pixel tile per 256-thread block

__device__ __forceinline__ float gat(const float* g, int H, int W, int y, int x) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? g[(size_t)y * W + x] : 0.0f;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(256) void det_gray_kernel(const float* __restrict__ rgb, float* __restrict__ gray, int64_t hw, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / hw, p = i - n * hw;
        const float* f = rgb + n * 3 * hw + p;
        gray[i] = (0.2989f * f[0] + 0.587f * f[hw] + 0.114f * f[2 * hw]) / 255.0f;
    }
}

// pass 1: laplacian^2, 8-neighbour contrast (partial sums over the window-covered region) + sobel magnitude map
__global__ __launch_bounds__(256) void det_point_kernel(const float* __restrict__ gray, float* __restrict__ sob, float* __restrict__ part,
                                                        int H, int W, int k, int pb) {
    __shared__ float red[4];
    const int n = blockIdx.z;
    const float* g = gray + (size_t)n * H * W;
    const int x = blockIdx.x * TB + (threadIdx.x & 15), y = blockIdx.y * TB + (threadIdx.x >> 4);
    const int ch = (H / k) * k, cw = (W / k) * k;
    float l2 = 0.f, mis = 0.f;
    if (y < H && x < W) {
        float v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) v[a][b] = gat(g, H, W, y + a - 1, x + b - 1);
        const float c = v[1][1];
        const float lap = ((v[0][0] + v[0][1]) + (v[0][2] + v[1][0])) + ((v[1][2] + v[2][0]) + (v[2][1] + v[2][2])) - 8.0f * c;
        float m = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                if (a != 1 || b != 1) m += fabsf(c - v[a][b]);
        const float gx = (v[0][0] - v[0][2]) + 2.0f * (v[1][0] - v[1][2]) + (v[2][0] - v[2][2]);
        const float gy = (v[0][0] + 2.0f * v[0][1] + v[0][2]) - (v[2][0] + 2.0f * v[2][1] + v[2][2]);
        sob[(size_t)n * H * W + (size_t)y * W + x] = sqrtf(gx * gx + gy * gy);
        if (y < ch && x < cw) { l2 = lap * lap; mis = m; }
    }
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    const float s0 = block_sum(l2, red), s1 = block_sum(mis, red);
    if (threadIdx.x == 0) {
        part[((size_t)n * 6 + 0) * pb + blk] = s0;
        part[((size_t)n * 6 + 1) * pb + blk] = s1;
    }
}

// pass 2: (v - box_k(v))^2 for v = gray (STA3) and v = sobel magnitude (GRA7)
__global__ __launch_bounds__(256) void det_boxdev_kernel(const float* __restrict__ gray, const float* __restrict__ sob, float* __restrict__ part,
                                                         int H, int W, int k, int pb) {
    __shared__ float red[4];
    const int n = blockIdx.z;
    const float* g = gray + (size_t)n * H * W;
    const float* s = sob + (size_t)n * H * W;
    const int x = blockIdx.x * TB + (threadIdx.x & 15), y = blockIdx.y * TB + (threadIdx.x >> 4);
    const int ch = (H / k) * k, cw = (W / k) * k, h = k / 2;
    float dg = 0.f, ds = 0.f;
    if (y < ch && x < cw) {
        float ag = 0.f, as = 0.f;
        for (int a = -h; a <= h; ++a) {
            const int yy = y + a;
            if (yy < 0 || yy >= H) continue;
            for (int b = -h; b <= h; ++b) {
                const int xx = x + b;
                if (xx < 0 || xx >= W) continue;
                ag += g[(size_t)yy * W + xx];
                as += s[(size_t)yy * W + xx];
            }
        }
        const float inv = 1.0f / (float)(k * k);
        const float eg = g[(size_t)y * W + x] - ag * inv, es = s[(size_t)y * W + x] - as * inv;
        dg = eg * eg;
        ds = es * es;
    }
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    const float s0 = block_sum(ds, red), s1 = block_sum(dg, red);
    if (threadIdx.x == 0) {
        part[((size_t)n * 6 + 3) * pb + blk] = s0;     // GRA7
        part[((size_t)n * 6 + 4) * pb + blk] = s1;     // STA3
    }
}

// pass 3: DCT3 — one thread per k x k window of the valid 4x4-mask response
__global__ __launch_bounds__(256) void det_dct_kernel(const float* __restrict__ gray, float* __restrict__ part, int H, int W, int k, int pb) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const float* g = gray + (size_t)n * H * W;
    const int nwx = (W - 3) / k, nwy = (H - 3) / k;
    const int wi = blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (wi < nwx * nwy) {
        const int wy = wi / nwx, wx = wi - wy * nwx;
        float acc = 0.f;
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) {
                const float* q = g + (size_t)(wy * k + a) * W + wx * k + b;
                const float top = (q[0] + q[1] - q[2] - q[3]) + (q[W] + q[W + 1] - q[W + 2] - q[W + 3]);
                const float bot = (q[2 * W + 2] + q[2 * W + 3] - q[2 * W] - q[2 * W + 1]) + (q[3 * W + 2] + q[3 * W + 3] - q[3 * W] - q[3 * W + 1]);
                acc += top + bot;
            }
        v = acc * acc;
    }
    const float s0 = block_sum(v, red);
    if (threadIdx.x == 0) part[((size_t)n * 6 + 5) * pb + blockIdx.x] = s0;
}

// pass 4: WAV1 — one thread per level-1 coefficient position; db6, zero extension, conv + keep odd samples
__constant__ float DB6_LO[12] = {-0.00107730108499558f, 0.004777257511010651f, 0.0005538422009938016f, -0.031582039318031156f,
                                 0.02752286553001629f, 0.09750160558707936f, -0.12976686756709563f, -0.22626469396516913f,
                                 0.3152503517092432f, 0.7511339080215775f, 0.4946238903983854f, 0.11154074335008017f};
__global__ __launch_bounds__(256) void det_wav_kernel(const float* __restrict__ gray, float* __restrict__ part, int H, int W, int pb) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const float* g = gray + (size_t)n * H * W;
    const int ch = (H + 11) / 2, cw = (W + 11) / 2;
    const int ci = blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (ci < ch * cw) {
        const int i = ci / cw, j = ci - i * cw;
        float lh = 0.f, hl = 0.f, hh = 0.f;
        for (int a = 0; a < 12; ++a) {
            const int y = 2 * i + 1 - a;
            if (y < 0 || y >= H) continue;
            float rl = 0.f, rh = 0.f;
            for (int b = 0; b < 12; ++b) {
                const int x = 2 * j + 1 - b;
                if (x < 0 || x >= W) continue;
                const float px = g[(size_t)y * W + x];
                const float hb = ((b & 1) ? 1.0f : -1.0f) * DB6_LO[11 - b];      // dec_hi[b] = (-1)^(b+1) dec_lo[11-b]
                rl = fmaf(DB6_LO[b], px, rl);
                rh = fmaf(hb, px, rh);
            }
            const float ha = ((a & 1) ? 1.0f : -1.0f) * DB6_LO[11 - a];
            lh = fmaf(ha, rl, lh);
            hl = fmaf(DB6_LO[a], rh, hl);
            hh = fmaf(ha, rh, hh);
        }
        v = fabsf(lh) + fabsf(hl) + fabsf(hh);
    }
    const float s0 = block_sum(v, red);
    if (threadIdx.x == 0) part[((size_t)n * 6 + 2) * pb + blockIdx.x] = s0;
}

// final: fixed-order sums of the block partials, one wave per (frame, feature)
__global__ __launch_bounds__(64) void det_final_kernel(const float* __restrict__ part, float* __restrict__ out, int pb, int nb_tile, int nb_dct,
                                                       int nb_wav, float nwin, float nwin_dct) {
    const int n = blockIdx.x, f = blockIdx.y;
    const int nb = (f == 5) ? nb_dct : (f == 2) ? nb_wav : nb_tile;
    const float* p = part + ((size_t)n * 6 + f) * pb;
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) s += p[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[n * 6 + f] = (f == 2) ? s : (f == 5) ? s / nwin_dct : s / nwin;
}

struct DetDims {
    int tiles_x, tiles_y, nb_tile, nb_dct, nb_wav, pb;
};
DetDims dims(int H, int W, int k) {
    DetDims d;
    d.tiles_x = cdiv(W, TB);
    d.tiles_y = cdiv(H, TB);
    d.nb_tile = d.tiles_x * d.tiles_y;
    d.nb_dct = cdiv((int64_t)((W - 3) / k) * ((H - 3) / k), 256);
    d.nb_wav = cdiv((int64_t)((H + 11) / 2) * ((W + 11) / 2), 256);
    d.pb = d.nb_tile > d.nb_wav ? d.nb_tile : d.nb_wav;
    if (d.nb_dct > d.pb) d.pb = d.nb_dct;
    return d;
}

}  // namespace

extern "C" int spei_det_gray(const float* rgb, float* gray, int N, int H, int W, spei_stream_t stream) {
    SPEI_REQUIRE(rgb && gray && N > 0 && H > 0 && W > 0, "spei_det_gray: bad arguments");
    const int64_t hw = (int64_t)H * W, total = hw * N;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(det_gray_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rgb, gray, hw, total);
    SPEI_CHECK_LAUNCH("spei_det_gray");
    return 0;
}

extern "C" int64_t spei_det_ws_floats(int N, int H, int W, int k) {
    if (N <= 0 || H < 4 || W < 4 || k < 1) return 0;
    const DetDims d = dims(H, W, k);
    return (int64_t)N * H * W + (int64_t)N * 6 * d.pb;
}

extern "C" int spei_det_features(const float* gray, float* out, float* ws, int N, int H, int W, int k, spei_stream_t stream) {
    SPEI_REQUIRE(gray && out && ws && N > 0, "spei_det_features: bad arguments");
    SPEI_REQUIRE(k >= 1 && (k & 1) && H >= k + 3 && W >= k + 3, "spei_det_features: k=%d must be odd and fit the %dx%d frame", k, H, W);
    const DetDims d = dims(H, W, k);
    float* sob = ws;
    float* part = ws + (size_t)N * H * W;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(det_point_kernel, dim3(d.tiles_x, d.tiles_y, N), dim3(256), 0, st, gray, sob, part, H, W, k, d.pb);
    hipLaunchKernelGGL(det_boxdev_kernel, dim3(d.tiles_x, d.tiles_y, N), dim3(256), 0, st, gray, sob, part, H, W, k, d.pb);
    hipLaunchKernelGGL(det_dct_kernel, dim3(d.nb_dct, N), dim3(256), 0, st, gray, part, H, W, k, d.pb);
    hipLaunchKernelGGL(det_wav_kernel, dim3(d.nb_wav, N), dim3(256), 0, st, gray, part, H, W, d.pb);
    const float nwin = (float)((H / k) * (W / k)), nwin_dct = (float)(((H - 3) / k) * ((W - 3) / k));
    hipLaunchKernelGGL(det_final_kernel, dim3(N, 6), dim3(64), 0, st, part, out, d.pb, d.nb_tile, d.nb_dct, d.nb_wav, nwin, nwin_dct);
    SPEI_CHECK_LAUNCH("spei_det_features");
    return 0;
}

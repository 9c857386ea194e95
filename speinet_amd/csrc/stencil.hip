// HBM-bound stencil / resample / elementwise kernels of the SPEINet forward pass (gfx950).
#include <stdarg.h>
#include <string.h>
#include "common.h"

// ------------------------------------------------------------------------------------------------
// error plumbing + identification
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void spei_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* spei_last_error(void) { return g_err; }
extern "C" int spei_version(void) { return 100; }
extern "C" const char* spei_arch(void) { return "gfx950"; }

// ------------------------------------------------------------------------------------------------
// K15  any(x != 0)   (model/speinet.py:70-73)
// ------------------------------------------------------------------------------------------------
__global__ void any_nonzero_kernel(const float* __restrict__ x, int64_t n, int32_t* flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    bool nz = false;
    for (; i < n; i += step) nz |= (x[i] != 0.0f);   // NaN != 0 is true, like torch (x == 0) being false
    if (__any(nz) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
extern "C" int spei_any_nonzero(const float* x, int64_t n, int32_t* flag, spei_stream_t stream) {
    SPEI_REQUIRE(x && flag && n > 0, "spei_any_nonzero: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(flag, 0, sizeof(int32_t), s);
    const int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(any_nonzero_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, s, x, n, flag);
    SPEI_CHECK_LAUNCH("spei_any_nonzero");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K1  Richardson-Lucy style prior, one iteration per launch  (model/rcl.py:22-51)
//   b = box5(d)/25 (zero pad);  c = I / b, NaN -> 0, c < 0 -> 0;  d' = c * (d + lam * lap4(d))
// 32x32 output tile per 256-thread block, halo 2 staged in LDS; planes are independent (grid.z).
// ------------------------------------------------------------------------------------------------
constexpr int RL_T = 32;
__global__ __launch_bounds__(256) void rl_iter_kernel(const float* __restrict__ img, const float* __restrict__ d,
                                                      float* __restrict__ out, int H, int W, float lam) {
    __shared__ float t[RL_T + 4][RL_T + 4 + 1];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * RL_T, y0 = blockIdx.y * RL_T;
    for (int i = threadIdx.x; i < (RL_T + 4) * (RL_T + 4); i += 256) {
        const int ly = i / (RL_T + 4), lx = i - ly * (RL_T + 4);
        const int gy = y0 + ly - 2, gx = x0 + lx - 2;
        t[ly][lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? d[plane + (size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31;
    for (int ly = threadIdx.x >> 5; ly < RL_T; ly += 8) {
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        float b = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) b = fmaf(t[ly + dy][lx + dx], 0.04f, b);
        const float c0 = t[ly + 2][lx + 2];
        // conv2d with [[0,-1,0],[-1,4,-1],[0,-1,0]] in kernel order
        float lap = -t[ly + 1][lx + 2];
        lap -= t[ly + 2][lx + 1];
        lap = fmaf(4.0f, c0, lap);
        lap -= t[ly + 2][lx + 3];
        lap -= t[ly + 3][lx + 2];
        const float I = img[plane + (size_t)gy * W + gx];
        float c = I / b;
        if (c != c) c = 0.0f;
        if (c < 0.0f) c = 0.0f;
        out[plane + (size_t)gy * W + gx] = c * (c0 + lam * lap);
    }
}
extern "C" int spei_rl_prior(const float* img, float* out, float* scratch, int C, int H, int W, int iters, float lam,
                             spei_stream_t stream) {
    SPEI_REQUIRE(img && out && scratch && C > 0 && H > 0 && W > 0 && iters >= 1, "spei_rl_prior: bad arguments");
    SPEI_REQUIRE(out != img && scratch != img && out != scratch, "spei_rl_prior: buffers must be distinct");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(cdiv(W, RL_T), cdiv(H, RL_T), C);
    // ping-pong so that the last iteration lands in `out`
    const float* src = img;
    for (int it = 0; it < iters; ++it) {
        float* dst = ((iters - 1 - it) % 2 == 0) ? out : scratch;
        hipLaunchKernelGGL(rl_iter_kernel, grid, dim3(256), 0, s, img, src, dst, H, W, lam);
        src = dst;
    }
    SPEI_CHECK_LAUNCH("spei_rl_prior");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// first conv: 5x5 pad 2, 3 NCHW planes -> NHWC [H][W][32], bias + ReLU  (model/recons_video_ori.py:28-32)
// An implicit GEMM on the f32 matrix pipe (v_mfma_f32_32x32x2_f32, exact fp32 products and accumulation): M = pixels,
// N = 32, K = 75 (tap, channel) padded to 76.  8 x 32 pixel tile per 256-thread workgroup, a wave owns two tile rows;
// the input halo (3 planes) sits in LDS, the A operand of k-step kk is ONE ds_read_b32 per lane at a per-lane
// precomputed (channel, dy, dx) offset, the 38 B operands (the whole 75 x 32 weight matrix) stay in registers.
// As 75 x 32 VALU FMAs per pixel this layer was bound by the vector ALU at 96 us per 720p frame (7 per frame).
// ------------------------------------------------------------------------------------------------
constexpr int CI_TH = 8, CI_TW = 32, CI_P = CI_TW + 4 + 1;     // tile, LDS row pitch (floats)
__global__ __launch_bounds__(256) void conv5_in_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int H, int W, int tiles_x, int ntiles) {
    __shared__ float tin[3 * (CI_TH + 4) * CI_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    // per-lane constants: k = 2 kk + fk -> (tap, ci); packed weights are [tap][co][ci].  The workgroup is persistent (it
    // walks tiles blockIdx.x, + gridDim.x, ...) so that this 38-load gather is paid once, not per tile.
    int koff[38];
    float bw[38];
#pragma unroll
    for (int kk = 0; kk < 38; ++kk) {
        const int k = 2 * kk + fk;
        const int t = k / 3, ci = k - 3 * t;
        const int dy = t / 5, dx = t - 5 * dy;
        koff[kk] = k < 75 ? (ci * (CI_TH + 4) + dy) * CI_P + dx : 0;
        bw[kk] = k < 75 ? w[(t * 32 + fr) * 3 + ci] : 0.0f;
    }
    const float bv = bias[fr];
    const float* a0 = tin + (wave * 2) * CI_P + fr;
    const int et = fr & 3, ecol = (fr >> 2) * 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int x0 = tx * CI_TW, y0 = ty * CI_TH;
        for (int i = tid; i < 3 * (CI_TH + 4) * (CI_TW + 4); i += 256) {
            const int c = i / ((CI_TH + 4) * (CI_TW + 4));
            const int r = i - c * (CI_TH + 4) * (CI_TW + 4);
            const int ly = r / (CI_TW + 4), lx = r - ly * (CI_TW + 4);
            const int gy = y0 + ly - 2, gx = x0 + lx - 2;
            tin[(c * (CI_TH + 4) + ly) * CI_P + lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[((size_t)c * H + gy) * W + gx] : 0.0f;
        }
        __syncthreads();
        // all 76 A operands are requested from LDS before the first MFMA
        float av[2][38];
#pragma unroll
        for (int kk = 0; kk < 38; ++kk) {
            av[0][kk] = a0[koff[kk]];
            av[1][kk] = a0[koff[kk] + CI_P];
        }
        __syncthreads();                                  // the tile may be overwritten by the next iteration's staging
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 38; ++kk) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][kk], bw[kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][kk], bw[kk], acc[1], 0, 0, 0);
        }
        // accumulator: column (channel) fr, rows (pixels) (r&3) + 8*(r>>2) + 4*fk; 4x4 quad transpose -> 16-byte stores
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gy = y0 + wave * 2 + i;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = fmaxf(acc[i][4 * k + e] + bv, 0.f);
                quad_transpose4(a[0], a[1], a[2], a[3], et);
                const int gx = x0 + 8 * k + 4 * fk + et;
                if (gy < H && gx < W) *reinterpret_cast<f32x4*>(out + ((size_t)gy * W + gx) * 32 + ecol) = f32x4{a[0], a[1], a[2], a[3]};
            }
        }
    }
}
extern "C" int spei_conv5_in(const float* img_chw, const float* w, const float* bias, float* out_hwc, int H, int W,
                             int Cout, spei_stream_t stream) {
    SPEI_REQUIRE(img_chw && w && bias && out_hwc && H > 0 && W > 0, "spei_conv5_in: bad arguments");
    SPEI_REQUIRE(Cout == 32, "spei_conv5_in: Cout=%d (only n_feat=32 is built)", Cout);
    const int tiles_x = cdiv(W, CI_TW), ntiles = tiles_x * cdiv(H, CI_TH);
    hipLaunchKernelGGL(conv5_in_kernel, dim3(ntiles < 1024 ? ntiles : 1024), dim3(256), 0, (hipStream_t)stream,
                       img_chw, w, bias, out_hwc, H, W, tiles_x, ntiles);
    SPEI_CHECK_LAUNCH("spei_conv5_in");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// last conv: 5x5 pad 2, NHWC [H][W][32] -> 3 NCHW planes, bias, no activation (model/recons_video_ori.py:75-77)
// 8x32 pixel tile, thread = one pixel x 3 outputs; input halo in LDS with a 36-float pixel pitch
// (conflict-free ds_read_b128 across lanes), weights [3][25][32] broadcast from LDS.
// ------------------------------------------------------------------------------------------------
constexpr int CO_TH = 8, CO_TW = 32, CO_P = 36;
__global__ __launch_bounds__(256) void conv5_out_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* tin = sm;                                         // [(CO_TH+4)*(CO_TW+4)][CO_P]
    float* wl = sm + (CO_TH + 4) * (CO_TW + 4) * CO_P;       // [3][25][32]
    const int x0 = blockIdx.x * CO_TW, y0 = blockIdx.y * CO_TH;
    for (int i = threadIdx.x; i < (CO_TH + 4) * (CO_TW + 4) * 8; i += 256) {
        const int pix = i >> 3, q = i & 7;
        const int ly = pix / (CO_TW + 4), lx = pix - ly * (CO_TW + 4);
        const int gy = y0 + ly - 2, gx = x0 + lx - 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const float4*>(in + ((size_t)gy * W + gx) * ldi + q * 4);
        *reinterpret_cast<float4*>(tin + pix * CO_P + q * 4) = v;
    }
    // packed [tap][co][ci] -> [co][tap][ci]
    for (int i = threadIdx.x; i < 25 * 3 * 32; i += 256) {
        const int t = i / 96, r = i - t * 96, co = r / 32, ci = r - co * 32;
        wl[(co * 25 + t) * 32 + ci] = w[i];
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int gy = y0 + ly, gx = x0 + lx;
    float a0 = bias[0], a1 = bias[1], a2 = bias[2];
    for (int t = 0; t < 25; ++t) {
        const int dy = t / 5, dx = t - dy * 5;
        const float4* pin = reinterpret_cast<const float4*>(tin + ((ly + dy) * (CO_TW + 4) + lx + dx) * CO_P);
        const float4* w0 = reinterpret_cast<const float4*>(wl + (0 * 25 + t) * 32);
        const float4* w1 = reinterpret_cast<const float4*>(wl + (1 * 25 + t) * 32);
        const float4* w2 = reinterpret_cast<const float4*>(wl + (2 * 25 + t) * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = pin[q];
            const float4 u0 = w0[q], u1 = w1[q], u2 = w2[q];
            a0 = fmaf(v.x, u0.x, a0); a0 = fmaf(v.y, u0.y, a0); a0 = fmaf(v.z, u0.z, a0); a0 = fmaf(v.w, u0.w, a0);
            a1 = fmaf(v.x, u1.x, a1); a1 = fmaf(v.y, u1.y, a1); a1 = fmaf(v.z, u1.z, a1); a1 = fmaf(v.w, u1.w, a1);
            a2 = fmaf(v.x, u2.x, a2); a2 = fmaf(v.y, u2.y, a2); a2 = fmaf(v.z, u2.z, a2); a2 = fmaf(v.w, u2.w, a2);
        }
    }
    if (gy < H && gx < W) {
        const size_t o = (size_t)gy * W + gx, hw = (size_t)H * W;
        out[o] = a0; out[hw + o] = a1; out[2 * hw + o] = a2;
    }
}
extern "C" int spei_conv5_out(const float* in_hwc, int ldi, const float* w, const float* bias, float* out_chw, int H,
                              int W, int Cin, spei_stream_t stream) {
    SPEI_REQUIRE(in_hwc && w && bias && out_chw && H > 0 && W > 0, "spei_conv5_out: bad arguments");
    SPEI_REQUIRE(Cin == 32 && ldi % 4 == 0 && ldi >= 32, "spei_conv5_out: Cin=%d ldi=%d (only n_feat=32 is built)", Cin, ldi);
    const size_t lds = ((size_t)(CO_TH + 4) * (CO_TW + 4) * CO_P + 3 * 25 * 32) * sizeof(float);
    ensure_dyn_lds<&conv5_out_kernel>(lds);
    hipLaunchKernelGGL(conv5_out_kernel, dim3(cdiv(W, CO_TW), cdiv(H, CO_TH)), dim3(256), lds, (hipStream_t)stream,
                       in_hwc, ldi, w, bias, out_chw, H, W);
    SPEI_CHECK_LAUNCH("spei_conv5_out");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// K13  bicubic upsample x2 / x4, align_corners=False, A=-0.75, clamped taps
// (aten UpSampleBicubic2d semantics: src = (dst+0.5)/s - 0.5, t = src - floor(src))
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    float x = t + 1.0f;
    c[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;
    c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t;
    c[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t;
    c[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}
template <int VEC>
__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo,
                                                      int H, int W, int C, int s, int relu) {
    const int cg = C / VEC;
    const int64_t total = (int64_t)H * s * W * s * cg;
    const float scale = 1.0f / (float)s;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cg) * VEC;
        const int64_t pix = i / cg;
        const int ox = (int)(pix % (W * s)), oy = (int)(pix / (W * s));
        const float ry = scale * ((float)oy + 0.5f) - 0.5f, rx = scale * ((float)ox + 0.5f) - 0.5f;
        const float fy = floorf(ry), fx = floorf(rx);
        float cy[4], cx[4];
        cubic_coeffs(ry - fy, cy);
        cubic_coeffs(rx - fx, cx);
        const int iy = (int)fy, ix = (int)fx;
        float acc[VEC];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), H - 1);
            float row[VEC];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xx = min(max(ix - 1 + b, 0), W - 1);
                const float* p = in + ((size_t)yy * W + xx) * ldi + c;
                if (VEC == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < VEC; ++e) row[e] = (b == 0) ? vv[e] * cx[0] : fmaf(vv[e], cx[b], row[e]);
                } else {
                    row[0] = (b == 0) ? p[0] * cx[0] : fmaf(p[0], cx[b], row[0]);
                }
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = (a == 0) ? row[e] * cy[0] : fmaf(row[e], cy[a], acc[e]);
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fmaxf(acc[e], 0.f);
        }
        float* o = out + ((size_t)oy * (W * s) + ox) * ldo + c;
        if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else o[0] = acc[0];
    }
}
// x2, 4 channels per thread, one 2 x 2 block of outputs per thread: the four outputs of input pixel (y, x) read the 5 x 5 input
// neighbourhood (y - 2 .. y + 2, x - 2 .. x + 2) between them — 25 loads for four outputs instead of 64 (the one-output-per-thread
// kernel above is bound by its load instructions: 87 us for 181 MB).  Same separable order as above (four horizontal taps per row,
// then four vertical taps), the two coefficient sets (t = 0.75 for even, 0.25 for odd outputs) are exact in fp32.
__global__ __launch_bounds__(256) void bicubic2x_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo,
                                                        int H, int W, int C, int relu) {
    const int cg = C / 4;
    const int64_t total = (int64_t)H * W * cg;
    float ce[4], co[4];
    cubic_coeffs(0.75f, ce);          // even outputs: src = y - 0.25 -> floor y - 1, t = 0.75: taps y - 2 .. y + 1
    cubic_coeffs(0.25f, co);          // odd outputs:  src = y + 0.25 -> floor y,     t = 0.25: taps y - 1 .. y + 2
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cg) * 4;
        const int64_t pix = i / cg;
        const int x = (int)(pix % W), y = (int)(pix / W);
        int xs[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) xs[b] = min(max(x - 2 + b, 0), W - 1);
        f32x4 he[5], ho[5];           // per input row: the horizontal sums for the even and the odd output column
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const int yy = min(max(y - 2 + a, 0), H - 1);
            const float* rowp = in + (size_t)yy * W * ldi + c;
            f32x4 v[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) v[b] = *reinterpret_cast<const f32x4*>(rowp + (size_t)xs[b] * ldi);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float re = v[0][e] * ce[0], ro = v[1][e] * co[0];
#pragma unroll
                for (int b = 1; b < 4; ++b) { re = fmaf(v[b][e], ce[b], re); ro = fmaf(v[b + 1][e], co[b], ro); }
                he[a][e] = re;
                ho[a][e] = ro;
            }
        }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const f32x4* h = px ? ho : he;
                const float* cy = py ? co : ce;
                f32x4 acc;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float r = h[py][e] * cy[0];
#pragma unroll
                    for (int a = 1; a < 4; ++a) r = fmaf(h[py + a][e], cy[a], r);
                    acc[e] = relu ? fmaxf(r, 0.f) : r;
                }
                *reinterpret_cast<f32x4*>(out + ((size_t)(2 * y + py) * (2 * W) + 2 * x + px) * ldo + c) = acc;
            }
    }
}

extern "C" int spei_upsample_bicubic(const float* in, int ldi, float* out, int ldo, int H, int W, int C, int s, int act,
                                     spei_stream_t stream) {
    SPEI_REQUIRE(in && out && H > 0 && W > 0 && C > 0 && (s == 2 || s == 4), "spei_upsample_bicubic: bad arguments");
    SPEI_REQUIRE(act == SPEI_ACT_NONE || act == SPEI_ACT_RELU, "spei_upsample_bicubic: act=%d", act);
    SPEI_REQUIRE(ldi >= C && ldo >= C, "spei_upsample_bicubic: bad row strides");
    const bool v4 = (C % 4 == 0) && (ldi % 4 == 0) && (ldo % 4 == 0);
    if (v4 && s == 2 && ((uintptr_t)in | (uintptr_t)out) % 16 == 0) {
        const int64_t tot2 = (int64_t)H * W * (C / 4);
        const int blocks2 = (int)((tot2 + 255) / 256 < 8192 ? (tot2 + 255) / 256 : 8192);
        hipLaunchKernelGGL(bicubic2x_kernel, dim3(blocks2), dim3(256), 0, (hipStream_t)stream, in, ldi, out, ldo, H, W, C, act == SPEI_ACT_RELU);
        SPEI_CHECK_LAUNCH("spei_upsample_bicubic");
        return 0;
    }
    const int64_t total = (int64_t)H * s * W * s * (v4 ? C / 4 : C);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (v4) hipLaunchKernelGGL(bicubic_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, ldi, out, ldo, H, W, C, s, act == SPEI_ACT_RELU);
    else    hipLaunchKernelGGL(bicubic_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, ldi, out, ldo, H, W, C, s, act == SPEI_ACT_RELU);
    SPEI_CHECK_LAUNCH("spei_upsample_bicubic");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// x.transpose(2,3).flip(2): out[a][b][:] = in[b][W-1-a][:]   (model/SearchTransfer.py:60)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rot90_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int H, int W, int C) {
    const int cg = C / 4;
    const int64_t total = (int64_t)H * W * cg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cg) * 4;
        const int64_t pix = i / cg;
        const int b = (int)(pix % H), a = (int)(pix / H);        // out is [W][H]
        *reinterpret_cast<float4*>(out + ((size_t)a * H + b) * C + c) =
            *reinterpret_cast<const float4*>(in + ((size_t)b * W + (W - 1 - a)) * ldi + c);
    }
}
extern "C" int spei_rot90(const float* in, int ldi, float* out, int H, int W, int C, spei_stream_t stream) {
    SPEI_REQUIRE(in && out && H > 0 && W > 0 && C % 4 == 0 && ldi % 4 == 0, "spei_rot90: bad arguments");
    const int64_t total = (int64_t)H * W * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(rot90_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, ldi, out, H, W, C);
    SPEI_CHECK_LAUNCH("spei_rot90");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// out = a + b
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(o)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
    if (blockIdx.x == 0) for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) o[i] = a[i] + b[i];
}
extern "C" int spei_add(const float* a, const float* b, float* out, int64_t n, spei_stream_t stream) {
    SPEI_REQUIRE(a && b && out && n > 0, "spei_add: bad arguments");
    SPEI_REQUIRE(((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16 == 0, "spei_add: 16-byte alignment required");
    const int64_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, (hipStream_t)stream, a, b, out, n4, n);
    SPEI_CHECK_LAUNCH("spei_add");
    return 0;
}

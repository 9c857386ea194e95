// Backward of SearchTransfer / SelfTransfer and of the decoder's glue (`loss.backward()` through
// model/SearchTransfer.py:24-79 and model/speinet.py:92-120 in trainer/trainer_swint_hsa_nsf.py:34-40).  fp32; every sum runs in
// a fixed order (gather form: one thread owns one output element; the reference-side sums walk per-position query lists that
// the caller builds with a stable sort of the arg-max) — no atomics, bitwise reproducible.
//
// With a_q = unfold3x3(lr)[q] (1152 numbers), b_k = unfold3x3(ref)[k], k* = arg[q] and S[q] = <a_q, b_k*> / (|a_q| |b_k*|):
//     dS/da_q = (b_k*/|b_k*| - S[q] a_q/|a_q|) / |a_q|,     dS/db_k* = (a_q/|a_q| - S[q] b_k*/|b_k*|) / |b_k*|
// (the max over k passes the gradient to the arg-max entry only; F.normalize's eps clamp is inactive for non-zero patches).
// unfold's adjoint is fold: a map pixel collects from the up to nine patches that contain it.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// d lr[p][c] = sum over taps d with q = p - d inside the map of
//     dS[q] inv_lr[q] ( ref[arg[q] + d][c] inv_ref[arg[q]]  -  S[q] lr[p][c] inv_lr[q] )          (ref outside its map = 0)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corr_s_bwd_lr_kernel(const float* __restrict__ lr, const float* __restrict__ ref,
                                                            const float* __restrict__ inv_lr, const float* __restrict__ inv_ref,
                                                            const float* __restrict__ S, const int* __restrict__ arg,
                                                            const float* __restrict__ dS, float* __restrict__ dlr, int H, int W, int Hr, int Wr,
                                                            int C) {
    const int cg = C / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W * cg) return;
    const int c = (int)(i % cg) * 4;
    const int p = (int)(i / cg);
    const int py = p / W, px = p - py * W;
    const f32x4 a = *reinterpret_cast<const f32x4*>(lr + (size_t)p * C + c);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int qy = py - dy, qx = px - dx;
            if (qy < 0 || qy >= H || qx < 0 || qx >= W) continue;
            const int q = qy * W + qx;
            const float g = dS[q] * inv_lr[q];
            const int k = arg[q];
            const int ky = k / Wr + dy, kx = k % Wr + dx;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (ky >= 0 && ky < Hr && kx >= 0 && kx < Wr) b = *reinterpret_cast<const f32x4*>(ref + ((size_t)ky * Wr + kx) * C + c);
            acc += g * (b * inv_ref[k] - a * (S[q] * inv_lr[q]));
        }
    *reinterpret_cast<f32x4*>(dlr + (size_t)p * C + c) = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Reference side.  order[]: the queries sorted by arg (stable), start[k] .. start[k+1]: the queries whose arg-max is k.
// A reference pixel R of the map at scale s (s = 1, 2, 4: patches of 3s x 3s pixels, stride s, padding s) lies in the patches
// k of the 3x3 level-3 neighbourhood of floor(R / s); every query q of such a patch took, for its output pixel
// P = R + s (q - k), the value ref[R] / 9:
//     d ref[R][c] = 1/9 sum_{k} sum_{q in list(k)} dT[R + s (q - k)][c]                              (P outside the output = none)
// and at s = 1, with d = R - k the tap of R inside patch k, the correlation adds
//     inv_ref[k] sum_{q in list(k)} dS[q] ( lr[q + d][c] inv_lr[q] - S[q] ref[R][c] inv_ref[k] )     (lr outside its map = 0).
// dT == NULL or dS == NULL switch a term off.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void search_bwd_ref_kernel(const float* __restrict__ ref, const float* __restrict__ dT,
                                                             const float* __restrict__ lr, const float* __restrict__ inv_lr,
                                                             const float* __restrict__ inv_ref, const float* __restrict__ S,
                                                             const float* __restrict__ dS, const int* __restrict__ order,
                                                             const int* __restrict__ start, float* __restrict__ dref, int H3, int W3, int Hr3,
                                                             int Wr3, int C, int s) {
    const int cg = C / 4;
    const int Hs = Hr3 * s, Ws = Wr3 * s;              // reference map at this scale
    const int Ho = H3 * s, Wo = W3 * s;                // output (T) map at this scale
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Hs * Ws * cg) return;
    const int c = (int)(i % cg) * 4;
    const int R = (int)(i / cg);
    const int Ry = R / Ws, Rx = R - Ry * Ws;
    const int by = Ry / s, bx = Rx / s;
    f32x4 accT = {0.f, 0.f, 0.f, 0.f}, accS = {0.f, 0.f, 0.f, 0.f};
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (dS) b = *reinterpret_cast<const f32x4*>(ref + (size_t)R * C + c);
    for (int ny = -1; ny <= 1; ++ny)
        for (int nx = -1; nx <= 1; ++nx) {
            const int ky = by + ny, kx = bx + nx;
            if (ky < 0 || ky >= Hr3 || kx < 0 || kx >= Wr3) continue;
            const int k = ky * Wr3 + kx;
            const float ik = dS ? inv_ref[k] : 0.f;
            f32x4 sumS = {0.f, 0.f, 0.f, 0.f};
            for (int j = start[k]; j < start[k + 1]; ++j) {
                const int q = order[j];
                const int qy = q / W3, qx = q - qy * W3;
                const int Py = Ry + s * (qy - ky), Px = Rx + s * (qx - kx);
                const bool inb = Py >= 0 && Py < Ho && Px >= 0 && Px < Wo;
                if (dT && inb) accT += *reinterpret_cast<const f32x4*>(dT + ((size_t)Py * Wo + Px) * C + c);
                if (dS) {          // s == 1: (Py, Px) = q + d is the lr pixel under the same tap (zero padding outside the map)
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    if (inb) a = *reinterpret_cast<const f32x4*>(lr + ((size_t)Py * Wo + Px) * C + c);
                    sumS += dS[q] * (a * inv_lr[q] - b * (S[q] * ik));
                }
            }
            accS += sumS * ik;
        }
    *reinterpret_cast<f32x4*>(dref + (size_t)R * C + c) = accT * (1.0f / 9.0f) + accS;
}

// ---------------------------------------------------------------------------------------------------------------------
// Adjoint of F.interpolate(mode='bicubic', align_corners=False, scale s), A = -0.75 (spei_upsample_bicubic): the coefficient
// with which output index o reads input index i along one axis, border taps clamped onto the edge pixels.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cubic1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }          // |x| <= 1
__device__ __forceinline__ float cubic2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }   // 1 < |x| < 2
__device__ __forceinline__ float bicubic_coeff(int o, int i, int n, float inv_s) {
    const float src = ((float)o + 0.5f) * inv_s - 0.5f;
    const float fl = floorf(src);
    const int i0 = (int)fl;
    const float t = src - fl;
    const float w[4] = {cubic2(t + 1.f), cubic1(t), cubic1(1.f - t), cubic2(2.f - t)};
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int idx = i0 - 1 + j;
        idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
        if (idx == i) r += w[j];
    }
    return r;
}

// thread = (input pixel, channel): dx[iy][ix][c] = sum_oy sum_ox cy(oy, iy) cx(ox, ix) dy[oy][ox][c] over the outputs whose taps
// can reach the pixel (all of them on the clamped side of an edge pixel)
__global__ __launch_bounds__(256) void bicubic_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C, int s) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W * C) return;
    const int c = (int)(i % C);
    const int p = (int)(i / C);
    const int iy = p / W, ix = p - iy * W;
    const int Ho = H * s, Wo = W * s;
    const float inv_s = 1.0f / (float)s;
    int y0 = s * (iy - 2), y1 = s * (iy + 3), x0 = s * (ix - 2), x1 = s * (ix + 3);
    if (iy == 0) y0 = 0;
    if (iy == H - 1) y1 = Ho;
    if (ix == 0) x0 = 0;
    if (ix == W - 1) x1 = Wo;
    y0 = max(y0, 0); y1 = min(y1, Ho); x0 = max(x0, 0); x1 = min(x1, Wo);
    float acc = 0.f;
    for (int oy = y0; oy < y1; ++oy) {
        const float cy = bicubic_coeff(oy, iy, H, inv_s);
        if (cy == 0.f) continue;
        float row = 0.f;
        for (int ox = x0; ox < x1; ++ox) {
            const float cx = bicubic_coeff(ox, ix, W, inv_s);
            if (cx != 0.f) row += cx * dy[((size_t)oy * Wo + ox) * C + c];
        }
        acc += cy * row;
    }
    dx[i] = acc;
}

// out[m] = sum_n a[m][n] * b[m][n]: one wave per row (the gradient of a per-row scale, model/speinet.py:93,95,104: `* weight_S`)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t M,
                                                     int N) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += a[m * N + n] * b[m * N + n];
    s = wave_sum(s);
    if (lane == 0) out[m] = s;
}

}  // namespace

extern "C" int spei_corr_s_bwd_lr(const float* lr, const float* ref, const float* inv_lr, const float* inv_ref, const float* S,
                                  const int32_t* arg, const float* dS, float* dlr, int H, int W, int Hr, int Wr, int C, spei_stream_t stream) {
    SPEI_REQUIRE(lr && ref && inv_lr && inv_ref && S && arg && dS && dlr, "spei_corr_s_bwd_lr: null pointer");
    SPEI_REQUIRE(H > 0 && W > 0 && Hr > 0 && Wr > 0 && C > 0 && C % 4 == 0, "spei_corr_s_bwd_lr: bad sizes");
    const int64_t n = (int64_t)H * W * (C / 4);
    hipLaunchKernelGGL(corr_s_bwd_lr_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lr, ref, inv_lr, inv_ref, S,
                       arg, dS, dlr, H, W, Hr, Wr, C);
    SPEI_CHECK_LAUNCH("spei_corr_s_bwd_lr");
    return 0;
}

extern "C" int spei_search_bwd_ref(const float* ref, const float* dT, const float* lr, const float* inv_lr, const float* inv_ref,
                                   const float* S, const float* dS, const int32_t* order, const int32_t* start, float* dref, int H3, int W3,
                                   int Hr3, int Wr3, int C, int s, spei_stream_t stream) {
    SPEI_REQUIRE(order && start && dref && (dT || dS), "spei_search_bwd_ref: null pointer");
    SPEI_REQUIRE(!dS || (s == 1 && ref && lr && inv_lr && inv_ref && S), "spei_search_bwd_ref: the correlation term needs s == 1 and all maps");
    SPEI_REQUIRE((s == 1 || s == 2 || s == 4) && H3 > 0 && W3 > 0 && Hr3 > 0 && Wr3 > 0 && C > 0 && C % 4 == 0, "spei_search_bwd_ref: bad sizes");
    const int64_t n = (int64_t)Hr3 * s * Wr3 * s * (C / 4);
    hipLaunchKernelGGL(search_bwd_ref_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ref, dT, lr, inv_lr, inv_ref,
                       S, dS, order, start, dref, H3, W3, Hr3, Wr3, C, s);
    SPEI_CHECK_LAUNCH("spei_search_bwd_ref");
    return 0;
}

extern "C" int spei_upsample_bicubic_bwd(const float* dy, float* dx, int H, int W, int C, int s, spei_stream_t stream) {
    SPEI_REQUIRE(dy && dx && H > 0 && W > 0 && C > 0 && (s == 2 || s == 4), "spei_upsample_bicubic_bwd: bad arguments");
    const int64_t n = (int64_t)H * W * C;
    hipLaunchKernelGGL(bicubic_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, H, W, C, s);
    SPEI_CHECK_LAUNCH("spei_upsample_bicubic_bwd");
    return 0;
}

extern "C" int spei_rowdot(const float* a, const float* b, float* out, int64_t M, int N, spei_stream_t stream) {
    SPEI_REQUIRE(a && b && out && M > 0 && N > 0, "spei_rowdot: bad arguments");
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, M, N);
    SPEI_CHECK_LAUNCH("spei_rowdot");
    return 0;
}

// K10-K12 — SearchTransfer / SelfTransfer: patch norms, fused correlation + arg-max, gather + fold
// (reference model/SearchTransfer.py:24-51, 59-79).
//
// The reference unfolds both maps into 1152-dim (3x3x128) patch vectors, L2-normalises them, multiplies
// [Nr x 1152] x [1152 x Nl] and takes max/argmax over the reference index: R is 57 600^2 floats = 13.3 GB at
// 720p.  Here the product is an implicit GEMM over 9 taps x 128 channels on the f32 matrix pipe with the
// normalisation applied to the accumulator and a running (max, argmax) per query column kept in registers;
// R never exists.  Ties resolve to the lowest reference index, like torch.max on the CPU.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// inv[p] = 1 / max(sqrt(sum_{3x3 taps, c} f^2), 1e-12)      (F.normalize eps, SearchTransfer.py:30-31)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_invnorm_kernel(const float* __restrict__ f, int ldf, float* __restrict__ inv,
                                                            int H, int W, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= (int64_t)H * W) return;
    const int y = (int)(p / W), x = (int)(p % W);
    float acc = 0.f;
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const float* row = f + ((size_t)yy * W + xx) * ldf;
            for (int c = lane; c < C; c += 64) acc = fmaf(row[c], row[c], acc);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) inv[p] = 1.0f / fmaxf(sqrtf(acc), 1e-12f);
}

// ---------------------------------------------------------------------------------------------------
// fused correlation + arg-max
// ---------------------------------------------------------------------------------------------------
constexpr int CBM = 128, CBN = 128, CBK = 32, CLD = CBK + 1, CMAXSPLIT = 8;

struct CorrParams {
    const float* lr;
    const float* ref;
    const float* inv_lr;
    const float* inv_ref;
    float* pval;     // [splits][Nl]
    int32_t* pidx;   // [splits][Nl]
    int ldl, ldr, Hl, Wl, Hr, Wr, C, Nl, Nr, njt, jt_per_split;
};

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__global__ __launch_bounds__(256) void corr_argmax_kernel(const CorrParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][CBM][CLD]   reference patches (rows j)
    float* Bs = smem + 2 * CBM * CLD;         // [2][CBN][CLD]   query patches (rows i)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fk = lane >> 5;
    const int i0 = blockIdx.x * CBN;
    const int jt0 = blockIdx.y * p.jt_per_split;
    const int jt1 = min(p.njt, jt0 + p.jt_per_split);
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    const int kchunks = p.C / CBK;
    const int per_tile = 9 * kchunks;

    int b_y[4], b_x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = i0 + lrow + 32 * j;
        if (i < p.Nl) { b_y[j] = i / p.Wl; b_x[j] = i - b_y[j] * p.Wl; }
        else { b_y[j] = -0x10000000; b_x[j] = 0; }
    }
    int a_y[4], a_x[4];
    auto set_jtile = [&](int jt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jj = jt * CBM + lrow + 32 * j;
            if (jj < p.Nr) { a_y[j] = jj / p.Wr; a_x[j] = jj - a_y[j] * p.Wr; }
            else { a_y[j] = -0x10000000; a_x[j] = 0; }
        }
    };
    float4 ra[4], rb[4];
    auto load_tile = [&](int it) {   // it in [0, per_tile): tap-major, then k chunk
        const int t = it / kchunks, kc = it - t * kchunks;
        const int ty = t / 3 - 1, tx = t - (t / 3) * 3 - 1;
        const int kofs = kc * CBK + lcol;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = a_y[j] + ty, xx = a_x[j] + tx;
            ra[j] = ((yy >= 0) & (yy < p.Hr) & (xx >= 0) & (xx < p.Wr))
                        ? *reinterpret_cast<const float4*>(p.ref + ((size_t)yy * p.Wr + xx) * p.ldr + kofs)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
            const int y2 = b_y[j] + ty, x2 = b_x[j] + tx;
            rb[j] = ((y2 >= 0) & (y2 < p.Hl) & (x2 >= 0) & (x2 < p.Wl))
                        ? *reinterpret_cast<const float4*>(p.lr + ((size_t)y2 * p.Wl + x2) * p.ldl + kofs)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * CBM * CLD;
        float* b = Bs + buf * CBN * CLD;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* d = a + (lrow + 32 * j) * CLD + lcol;
            d[0] = ra[j].x; d[1] = ra[j].y; d[2] = ra[j].z; d[3] = ra[j].w;
            d = b + (lrow + 32 * j) * CLD + lcol;
            d[0] = rb[j].x; d[1] = rb[j].y; d[2] = rb[j].z; d[3] = rb[j].w;
        }
    };

    float bestv[2] = {-INFINITY, -INFINITY};
    int besti[2] = {0x7fffffff, 0x7fffffff};
    float il[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int i = i0 + (wn * 2 + tn) * 32 + fr;
        il[tn] = i < p.Nl ? p.inv_lr[i] : 0.f;
    }

    f32x16 acc[2][2];
    int buf = 0;
    if (jt0 < jt1) {
        set_jtile(jt0);
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int jt = jt0; jt < jt1; ++jt) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int it = 0; it < per_tile; ++it) {
            const bool last = (it + 1 == per_tile);
            const bool more = !last || (jt + 1 < jt1);
            if (more) {
                if (last) set_jtile(jt + 1);
                load_tile(last ? 0 : it + 1);
            }
            const float* a = As + buf * CBM * CLD + (wm * 64 + fr) * CLD + fk;
            const float* b = Bs + buf * CBN * CLD + (wn * 64 + fr) * CLD + fk;
#pragma unroll
            for (int kk = 0; kk < CBK; kk += 2) {
                const float a0 = a[kk], a1 = a[32 * CLD + kk];
                const float b0 = b[kk], b1 = b[32 * CLD + kk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (more) store_tile(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        // running max over this tile's 128 reference rows; per lane the rows come in increasing j
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * CBM + (wm * 2 + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (j < p.Nr) {
                    const float ir = p.inv_ref[j];
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const float v = acc[tm][tn][r] * ir * il[tn];
                        if (v > bestv[tn]) { bestv[tn] = v; besti[tn] = j; }
                    }
                }
            }
        }
    }
    // combine the two lane halves, then the two waves that share these columns
    __syncthreads();
    float* rv = smem;                                   // [2 wm][128]
    int* ri = reinterpret_cast<int*>(smem + 2 * CBN);   // [2 wm][128]
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const float ov = __shfl_xor(bestv[tn], 32, 64);
        const int oi = __shfl_xor(besti[tn], 32, 64);
        if (better(ov, oi, bestv[tn], besti[tn])) { bestv[tn] = ov; besti[tn] = oi; }
        if (fk == 0) {
            const int col = (wn * 2 + tn) * 32 + fr;
            rv[wm * CBN + col] = bestv[tn];
            ri[wm * CBN + col] = besti[tn];
        }
    }
    __syncthreads();
    if (tid < CBN) {
        const int i = i0 + tid;
        if (i < p.Nl) {
            float v = rv[tid];
            int ix = ri[tid];
            if (better(rv[CBN + tid], ri[CBN + tid], v, ix)) { v = rv[CBN + tid]; ix = ri[CBN + tid]; }
            p.pval[(size_t)blockIdx.y * p.Nl + i] = v;
            p.pidx[(size_t)blockIdx.y * p.Nl + i] = ix;
        }
    }
}

__global__ __launch_bounds__(256) void corr_final_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
                                                         int splits, int Nl, float* __restrict__ S, int32_t* __restrict__ arg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nl) return;
    float v = pval[i];
    int ix = pidx[i];
    for (int s = 1; s < splits; ++s) {
        const float ov = pval[(size_t)s * Nl + i];
        const int oi = pidx[(size_t)s * Nl + i];
        if (better(ov, oi, v, ix)) { v = ov; ix = oi; }
    }
    S[i] = v;
    arg[i] = ix == 0x7fffffff ? 0 : ix;     // every candidate NaN: torch.max would propagate NaN; keep index valid
}

// ---------------------------------------------------------------------------------------------------
// exact re-score of the two best candidates of the bf16 pass (spei_corr_slab_top2_bf16): one wave per query,
// fp32 features, products and sums in fp64 (1152 terms each) — the winner and S no longer depend on bf16 rounding
// or on any summation order.  S = dot * inv_ref[j] * inv_lr[i] as in corr_argmax_kernel; ties -> lowest index.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corr_rescore_kernel(const float* __restrict__ lr, int ldl, const float* __restrict__ ref, int ldr,
                                                           const float* __restrict__ inv_lr, const float* __restrict__ inv_ref,
                                                           int Hl, int Wl, int Hr, int Wr, int C, float* __restrict__ S,
                                                           int32_t* __restrict__ arg, const int32_t* __restrict__ arg2) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= (int64_t)Hl * Wl) return;
    const int qy = (int)(q / Wl), qx = (int)(q - (int64_t)qy * Wl);
    int cand[2] = {arg[q], arg2[q]};
    const int nc = cand[1] >= 0 ? 2 : 1;
    double d[2] = {0.0, 0.0};
    for (int t = 0; t < 9; ++t) {
        const int ty = t / 3 - 1, tx = t - (t / 3) * 3 - 1;
        const int y = qy + ty, x = qx + tx;
        if (y < 0 || y >= Hl || x < 0 || x >= Wl) continue;          // zero-padded unfold: the term vanishes
        const float* a = lr + ((size_t)y * Wl + x) * ldl;
        for (int k = 0; k < nc; ++k) {
            const int cy = cand[k] / Wr + ty, cx = cand[k] % Wr + tx;
            if (cy < 0 || cy >= Hr || cx < 0 || cx >= Wr) continue;
            const float* b = ref + ((size_t)cy * Wr + cx) * ldr;
            double acc = 0.0;
            for (int c = lane * 2; c < C; c += 128) {
                const float2 av = *reinterpret_cast<const float2*>(a + c);
                const float2 bv = *reinterpret_cast<const float2*>(b + c);
                acc += (double)av.x * (double)bv.x + (double)av.y * (double)bv.y;
            }
            d[k] += acc;
        }
    }
    for (int k = 0; k < 2; ++k)
        for (int m = 32; m >= 1; m >>= 1) d[k] += __shfl_xor(d[k], m, 64);
    if (lane == 0) {
        const double s0 = d[0] * (double)inv_ref[cand[0]];
        int win = cand[0];
        double sw = s0;
        if (nc == 2) {
            const double s1 = d[1] * (double)inv_ref[cand[1]];
            if (s1 > s0 || (s1 == s0 && cand[1] < cand[0])) { win = cand[1]; sw = s1; }
        }
        S[q] = (float)(sw * (double)inv_lr[q]);
        arg[q] = win;
    }
}

// ---------------------------------------------------------------------------------------------------
// gather + fold: out[y][x] = (1/9) sum over the 3x3 patches q covering (y,x) of ref[patch(arg[q]) at the same
// in-patch offset]; patch 3s x 3s, stride s, pad s  (unfold -> bis -> fold, SearchTransfer.py:36-46)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_fold_kernel(const float* __restrict__ ref, int ldr, const int32_t* __restrict__ arg,
                                                          float* __restrict__ out, int ldo, int H3, int W3, int Hr3, int Wr3,
                                                          int C, int s) {
    const int cg = C / 4;
    const int Ho = H3 * s, Wo = W3 * s, Hs = Hr3 * s, Wsrc = Wr3 * s;
    const int64_t total = (int64_t)Ho * Wo * cg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cg) * 4;
        const int64_t pix = i / cg;
        const int x = (int)(pix % Wo), y = (int)(pix / Wo);
        const int by = y / s, bx = x / s;
        // fold accumulates in increasing in-patch offset (py, px) => decreasing patch index.  The nine (arg -> address -> row) chains are
        // independent: all nine indices first, then all nine rows (clamped addresses, absent taps contribute +0: the sum and its
        // order are those of the branchy loop, which ran the chains one after the other: 67 us at 720p)
        int a9[9];
        bool ok9[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int qy = by + 1 - t / 3, qx = bx + 1 - t % 3;
            ok9[t] = qy >= 0 && qy < H3 && qx >= 0 && qx < W3;
            a9[t] = arg[min(max(qy, 0), H3 - 1) * W3 + min(max(qx, 0), W3 - 1)];
        }
        float4 v9[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int qy = by + 1 - t / 3, qx = bx + 1 - t % 3;
            const int py = y - qy * s + s, px = x - qx * s + s;
            const int ay = a9[t] / Wr3, ax = a9[t] - ay * Wr3;
            const int sy = ay * s - s + py, sx = ax * s - s + px;
            ok9[t] = ok9[t] && sy >= 0 && sy < Hs && sx >= 0 && sx < Wsrc;
            v9[t] = *reinterpret_cast<const float4*>(ref + ((size_t)min(max(sy, 0), Hs - 1) * Wsrc + min(max(sx, 0), Wsrc - 1)) * ldr + c);
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (ok9[t]) { acc.x += v9[t].x; acc.y += v9[t].y; acc.z += v9[t].z; acc.w += v9[t].w; }
        *reinterpret_cast<float4*>(out + pix * ldo + c) = make_float4(acc.x / 9.0f, acc.y / 9.0f, acc.z / 9.0f, acc.w / 9.0f);
    }
}

}  // namespace

extern "C" int spei_patch_invnorm(const float* f, int ldf, float* inv, int H, int W, int C, spei_stream_t stream) {
    SPEI_REQUIRE(f && inv && H > 0 && W > 0 && C > 0 && ldf >= C, "spei_patch_invnorm: bad arguments");
    hipLaunchKernelGGL(patch_invnorm_kernel, dim3(cdiv((int64_t)H * W, 4)), dim3(256), 0, (hipStream_t)stream, f, ldf, inv, H, W, C);
    SPEI_CHECK_LAUNCH("spei_patch_invnorm");
    return 0;
}

extern "C" int64_t spei_corr_ws_floats(int64_t n_lr) { return 4 * (int64_t)CMAXSPLIT * n_lr; }   // top-2 form: pairs

extern "C" int spei_corr_argmax(const float* lr, int ldl, const float* ref, int ldr, const float* inv_lr,
                                const float* inv_ref, int Hl, int Wl, int Hr, int Wr, int C, float* S, int32_t* arg,
                                float* ws, spei_stream_t stream) {
    SPEI_REQUIRE(lr && ref && inv_lr && inv_ref && S && arg && ws, "spei_corr_argmax: null pointer");
    SPEI_REQUIRE(C > 0 && C % 32 == 0 && ldl % 4 == 0 && ldr % 4 == 0 && ldl >= C && ldr >= C, "spei_corr_argmax: C=%d ldl=%d ldr=%d", C, ldl, ldr);
    SPEI_REQUIRE(Hl > 0 && Wl > 0 && Hr > 0 && Wr > 0, "spei_corr_argmax: empty map");
    SPEI_REQUIRE((int64_t)Hl * Wl < (1ll << 30) && (int64_t)Hr * Wr < (1ll << 30), "spei_corr_argmax: map too large");
    SPEI_REQUIRE(((uintptr_t)lr | (uintptr_t)ref) % 16 == 0, "spei_corr_argmax: 16-byte alignment required");
    CorrParams p;
    p.lr = lr; p.ref = ref; p.inv_lr = inv_lr; p.inv_ref = inv_ref;
    p.ldl = ldl; p.ldr = ldr; p.Hl = Hl; p.Wl = Wl; p.Hr = Hr; p.Wr = Wr; p.C = C;
    p.Nl = Hl * Wl; p.Nr = Hr * Wr;
    p.njt = cdiv(p.Nr, CBM);
    const int itiles = cdiv(p.Nl, CBN);
    int splits = 1024 / itiles;
    if (splits < 1) splits = 1;
    if (splits > CMAXSPLIT) splits = CMAXSPLIT;
    if (splits > p.njt) splits = p.njt;
    p.jt_per_split = cdiv(p.njt, splits);
    splits = cdiv(p.njt, p.jt_per_split);
    p.pval = ws;
    p.pidx = reinterpret_cast<int32_t*>(ws + (size_t)CMAXSPLIT * p.Nl);
    const size_t lds = (size_t)2 * (CBM + CBN) * CLD * sizeof(float);
    ensure_dyn_lds<&corr_argmax_kernel>(lds);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(corr_argmax_kernel, dim3(itiles, splits), dim3(256), lds, st, p);
    hipLaunchKernelGGL(corr_final_kernel, dim3(cdiv(p.Nl, 256)), dim3(256), 0, st, p.pval, p.pidx, splits, p.Nl, S, arg);
    SPEI_CHECK_LAUNCH("spei_corr_argmax");
    return 0;
}

extern "C" int spei_corr_rescore(const float* lr, int ldl, const float* ref, int ldr, const float* inv_lr, const float* inv_ref,
                                 int Hl, int Wl, int Hr, int Wr, int C, float* S, int32_t* arg, const float* S2, const int32_t* arg2,
                                 spei_stream_t stream) {
    (void)S2;     // the runner-up's bf16 score is not needed: both candidates are re-scored exactly
    SPEI_REQUIRE(lr && ref && inv_lr && inv_ref && S && arg && arg2, "spei_corr_rescore: null pointer");
    SPEI_REQUIRE(C > 0 && C % 2 == 0 && ldl % 2 == 0 && ldr % 2 == 0 && ldl >= C && ldr >= C, "spei_corr_rescore: C=%d ldl=%d ldr=%d", C, ldl, ldr);
    SPEI_REQUIRE(Hl > 0 && Wl > 0 && Hr > 0 && Wr > 0, "spei_corr_rescore: empty map");
    SPEI_REQUIRE(((uintptr_t)lr | (uintptr_t)ref) % 8 == 0, "spei_corr_rescore: 8-byte alignment required");
    hipLaunchKernelGGL(corr_rescore_kernel, dim3(cdiv((int64_t)Hl * Wl, 4)), dim3(256), 0, (hipStream_t)stream, lr, ldl, ref, ldr, inv_lr,
                       inv_ref, Hl, Wl, Hr, Wr, C, S, arg, arg2);
    SPEI_CHECK_LAUNCH("spei_corr_rescore");
    return 0;
}

extern "C" int spei_gather_fold(const float* ref, int ldr, const int32_t* arg, float* out, int ldo, int H3, int W3,
                                int Hr3, int Wr3, int C, int s, spei_stream_t stream) {
    SPEI_REQUIRE(ref && arg && out, "spei_gather_fold: null pointer");
    SPEI_REQUIRE((s == 1 || s == 2 || s == 4) && C % 4 == 0 && ldr % 4 == 0 && ldo % 4 == 0 && ldr >= C && ldo >= C,
                 "spei_gather_fold: s=%d C=%d ldr=%d ldo=%d", s, C, ldr, ldo);
    SPEI_REQUIRE(H3 > 0 && W3 > 0 && Hr3 > 0 && Wr3 > 0, "spei_gather_fold: empty map");
    const int64_t total = (int64_t)H3 * s * W3 * s * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(gather_fold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ref, ldr, arg, out, ldo, H3, W3, Hr3, Wr3, C, s);
    SPEI_CHECK_LAUNCH("spei_gather_fold");
    return 0;
}

// K11 on the bf16 matrix pipe: fused correlation + arg-max (reference model/SearchTransfer.py:33-34, 68-69).
//
// Both feature maps are re-read hundreds of times by the N3 x N3 correlation, so they are converted ONCE to
// bf16 (hi, and optionally the bf16 residual lo = bf16(x - hi)) by spei_split_bf16 and the GEMM streams 16-byte
// bf16 rows.  Modes:  lo == NULL  -> single bf16 product (v_mfma_f32_32x32x16_bf16);
//                     lo != NULL  -> bf16x3 split product (al*bh + ah*bl + ah*bh), f32-grade scores so the
//                                    arg-max matches the f32 path except on sub-1e-5 ties.
// Everything else (normalisation in the accumulator, running max per query column, lowest-index tie-break,
// split over reference tiles + final reduce) is the same as search.hip.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename LP>
__global__ __launch_bounds__(256) void split16_kernel(const float* __restrict__ x, int ld, LP* __restrict__ hi,
                                                      LP* __restrict__ lo, int64_t M, int C) {
    typedef typename lpv<LP>::x4 lp4;
    const int cg = C / 4;
    const int64_t total = M * cg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cg) * 4;
        const int64_t m = i / cg;
        const float4 v = *reinterpret_cast<const float4*>(x + m * ld + c);
        lp4 h;
        h[0] = (LP)v.x; h[1] = (LP)v.y; h[2] = (LP)v.z; h[3] = (LP)v.w;
        *reinterpret_cast<lp4*>(hi + m * C + c) = h;
        if (lo) {
            lp4 l;
            l[0] = (LP)(v.x - (float)h[0]); l[1] = (LP)(v.y - (float)h[1]);
            l[2] = (LP)(v.z - (float)h[2]); l[3] = (LP)(v.w - (float)h[3]);
            *reinterpret_cast<lp4*>(lo + m * C + c) = l;
        }
    }
}

constexpr int CBM = 128, CBN = 128, CMAXSPLIT = 8;

struct CorrParams {
    const __bf16* lrh;
    const __bf16* lrl;
    const __bf16* refh;
    const __bf16* refl;
    const float* inv_lr;
    const float* inv_ref;
    float* pval;
    int32_t* pidx;
    int Hl, Wl, Hr, Wr, C, Nl, Nr, njt, jt_per_split;
};

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

template <int BK, bool SPLIT>
__global__ __launch_bounds__(256) void corr_bf16_kernel(const CorrParams p) {
    constexpr int PITCH = 2 * BK + 16;
    constexpr int TPR = BK / 8;               // threads per row (16 B each)
    constexpr int RPP = 256 / TPR;            // rows per pass
    constexpr int NP = 128 / RPP;             // passes per operand tile
    constexpr int NPART = SPLIT ? 2 : 1;
    constexpr int T_BYTES = 128 * PITCH;
    constexpr int BUF_BYTES = 2 * NPART * T_BYTES;   // [A hi][A lo?][B hi][B lo?]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fk = lane >> 5;
    const int i0 = blockIdx.x * CBN;
    const int jt0 = blockIdx.y * p.jt_per_split;
    const int jt1 = min(p.njt, jt0 + p.jt_per_split);
    const int lrow = tid / TPR, lcol = (tid % TPR) * 8;
    const int kchunks = p.C / BK;
    const int per_tile = 9 * kchunks;

    int b_y[NP], b_x[NP], a_y[NP], a_x[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int i = i0 + lrow + RPP * j;
        if (i < p.Nl) { b_y[j] = i / p.Wl; b_x[j] = i - b_y[j] * p.Wl; }
        else { b_y[j] = -0x10000000; b_x[j] = 0; }
    }
    auto set_jtile = [&](int jt) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int jj = jt * CBM + lrow + RPP * j;
            if (jj < p.Nr) { a_y[j] = jj / p.Wr; a_x[j] = jj - a_y[j] * p.Wr; }
            else { a_y[j] = -0x10000000; a_x[j] = 0; }
        }
    };
    u32x4 rah[NP], rbh[NP], ral[SPLIT ? NP : 1], rbl[SPLIT ? NP : 1];
    const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
    auto load_tile = [&](int it) {
        const int t = it / kchunks, kc = it - t * kchunks;
        const int ty = t / 3 - 1, tx = t - (t / 3) * 3 - 1;
        const int kofs = kc * BK + lcol;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int yy = a_y[j] + ty, xx = a_x[j] + tx;
            const bool oka = (yy >= 0) & (yy < p.Hr) & (xx >= 0) & (xx < p.Wr);
            const size_t oa = ((size_t)yy * p.Wr + xx) * p.C + kofs;
            rah[j] = oka ? *reinterpret_cast<const u32x4*>(p.refh + oa) : zero;
            if (SPLIT) ral[j] = oka ? *reinterpret_cast<const u32x4*>(p.refl + oa) : zero;
            const int y2 = b_y[j] + ty, x2 = b_x[j] + tx;
            const bool okb = (y2 >= 0) & (y2 < p.Hl) & (x2 >= 0) & (x2 < p.Wl);
            const size_t ob = ((size_t)y2 * p.Wl + x2) * p.C + kofs;
            rbh[j] = okb ? *reinterpret_cast<const u32x4*>(p.lrh + ob) : zero;
            if (SPLIT) rbl[j] = okb ? *reinterpret_cast<const u32x4*>(p.lrl + ob) : zero;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* base = smem + buf * BUF_BYTES;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int o = (lrow + RPP * j) * PITCH + lcol * 2;
            *reinterpret_cast<u32x4*>(base + o) = rah[j];
            if (SPLIT) *reinterpret_cast<u32x4*>(base + T_BYTES + o) = ral[j];
            *reinterpret_cast<u32x4*>(base + NPART * T_BYTES + o) = rbh[j];
            if (SPLIT) *reinterpret_cast<u32x4*>(base + (NPART + 1) * T_BYTES + o) = rbl[j];
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
            const unsigned char* base = smem + buf * BUF_BYTES;
            const unsigned char* pa = base + (wm * 64 + fr) * PITCH + fk * 16;
            const unsigned char* pb = base + NPART * T_BYTES + (wn * 64 + fr) * PITCH + fk * 16;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 av[2], bv[2], avl[2], bvl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    av[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * PITCH + ks * 32);
                    bv[i] = *reinterpret_cast<const bf16x8*>(pb + i * 32 * PITCH + ks * 32);
                    if (SPLIT) {
                        avl[i] = *reinterpret_cast<const bf16x8*>(pa + T_BYTES + i * 32 * PITCH + ks * 32);
                        bvl[i] = *reinterpret_cast<const bf16x8*>(pb + T_BYTES + i * 32 * PITCH + ks * 32);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (SPLIT) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avl[i], bv[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bvl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                    }
            }
            if (more) store_tile(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
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
    __syncthreads();
    float* rv = reinterpret_cast<float*>(smem);
    int* ri = reinterpret_cast<int*>(smem) + 2 * CBN;
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

__global__ __launch_bounds__(256) void corr_final_bf16_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
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
    arg[i] = ix == 0x7fffffff ? 0 : ix;
}

template <int BK, bool SPLIT>
int launch_corr(const CorrParams& p, int itiles, int splits, hipStream_t st) {
    constexpr int PITCH = 2 * BK + 16;
    const size_t lds = (size_t)2 * 2 * (SPLIT ? 2 : 1) * 128 * PITCH;
    ensure_dyn_lds<&corr_bf16_kernel<BK, SPLIT>>(lds);
    hipLaunchKernelGGL((corr_bf16_kernel<BK, SPLIT>), dim3(itiles, splits), dim3(256), lds, st, p);
    return 0;
}

}  // namespace

extern "C" int spei_split16(int fmt, const float* x, int ld, void* hi, void* lo, int64_t M, int C, spei_stream_t stream) {
    SPEI_REQUIRE(x && hi && M > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "spei_split16: bad arguments");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_split16: fmt=%d", fmt);
    const int64_t total = M * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (fmt == SPEI_F16)
        hipLaunchKernelGGL(split16_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, (_Float16*)hi, (_Float16*)lo, M, C);
    else
        hipLaunchKernelGGL(split16_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, (__bf16*)hi, (__bf16*)lo, M, C);
    SPEI_CHECK_LAUNCH("spei_split16");
    return 0;
}

extern "C" int spei_corr_argmax_bf16(const void* lr_hi, const void* lr_lo, const void* ref_hi, const void* ref_lo,
                                     const float* inv_lr, const float* inv_ref, int Hl, int Wl, int Hr, int Wr, int C,
                                     float* S, int32_t* arg, float* ws, spei_stream_t stream) {
    SPEI_REQUIRE(lr_hi && ref_hi && inv_lr && inv_ref && S && arg && ws, "spei_corr_argmax_bf16: null pointer");
    SPEI_REQUIRE((lr_lo == nullptr) == (ref_lo == nullptr), "spei_corr_argmax_bf16: lo parts must both be given or both be NULL");
    SPEI_REQUIRE(C > 0 && C % 64 == 0, "spei_corr_argmax_bf16: C=%d must be a multiple of 64", C);
    SPEI_REQUIRE(Hl > 0 && Wl > 0 && Hr > 0 && Wr > 0, "spei_corr_argmax_bf16: empty map");
    SPEI_REQUIRE((int64_t)Hl * Wl < (1ll << 30) && (int64_t)Hr * Wr < (1ll << 30), "spei_corr_argmax_bf16: map too large");
    SPEI_REQUIRE(((uintptr_t)lr_hi | (uintptr_t)ref_hi | (uintptr_t)lr_lo | (uintptr_t)ref_lo) % 16 == 0, "spei_corr_argmax_bf16: 16-byte alignment required");
    CorrParams p;
    p.lrh = (const __bf16*)lr_hi; p.lrl = (const __bf16*)lr_lo; p.refh = (const __bf16*)ref_hi; p.refl = (const __bf16*)ref_lo;
    p.inv_lr = inv_lr; p.inv_ref = inv_ref;
    p.Hl = Hl; p.Wl = Wl; p.Hr = Hr; p.Wr = Wr; p.C = C;
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
    hipStream_t st = (hipStream_t)stream;
    if (lr_lo) launch_corr<32, true>(p, itiles, splits, st);
    else launch_corr<64, false>(p, itiles, splits, st);
    hipLaunchKernelGGL(corr_final_bf16_kernel, dim3(cdiv(p.Nl, 256)), dim3(256), 0, st, p.pval, p.pidx, splits, p.Nl, S, arg);
    SPEI_CHECK_LAUNCH("spei_corr_argmax_bf16");
    return 0;
}

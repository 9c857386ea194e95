// Implicit-GEMM convolution / linear on the gfx950 bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
// Same contract as igemm_f32.hip (same call sites of the reference) with two arithmetic modes:
//
//   bf16    one MFMA per product:  a ~ bf16(a), w ~ bf16(w)                         (16x the f32 MFMA rate)
//   bf16x3  split operands a = ah + al, w = wh + wl (each half bf16) and accumulate
//           al*wh + ah*wl + ah*wh in fp32: ~2^-17 relative per product, i.e. f32-grade
//           results at 3/16 of the f32 MFMA cost.
//
// Activations stay fp32 in HBM (residual streams keep full precision); they are rounded to bf16 while being
// staged into LDS.  Weights are pre-split on the host ([tap][Cout][Cin] bf16 hi / lo).  LDS rows are padded by
// 16 B (row pitch = 2*BK + 16 bytes, an odd multiple of 16 B) so every ds_read_b128 fragment read (lane -> row)
// is bank-conflict free on the 64-bank b128 path.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct IgemmParams {
    const float* a0;
    const float* a1;
    const __bf16* wh;
    const __bf16* wl;
    const float* bias;
    float* out;
    const float* res;
    const float* rowscale;
    int lda0, lda1, k0, k1;
    int ldo, ldr;
    int M, N, K;
    int Hin, Win, Hout, Wout;
    int ks, stride, pad, mode, act;
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

__device__ __forceinline__ bf16x4 cvt4(const float4 v) {
    bf16x4 r;
    r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    return r;
}
__device__ __forceinline__ float4 resid4(const float4 v, const bf16x4 h) {
    return make_float4(v.x - (float)h[0], v.y - (float)h[1], v.z - (float)h[2], v.w - (float)h[3]);
}

template <int BM, int BN, int WM, int WN, int BK, bool SPLIT>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const IgemmParams p) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int PITCH = 2 * BK + 16;               // bytes per LDS row
    constexpr int TPR_A = BK / 4;                    // threads per A row (float4 each)
    constexpr int RPP_A = 256 / TPR_A;               // A rows per pass
    constexpr int AP = BM / RPP_A;                   // A passes
    constexpr int TPR_B = BK / 8;                    // threads per B row (8 bf16 = 16 B each)
    constexpr int RPP_B = 256 / TPR_B;
    constexpr int BP = (BN + RPP_B - 1) / RPP_B;
    constexpr int NPART = SPLIT ? 2 : 1;
    constexpr int A_BYTES = BM * PITCH, B_BYTES = BN * PITCH;
    constexpr int BUF_BYTES = NPART * (A_BYTES + B_BYTES);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // buffer layout: [A hi][A lo?][B hi][B lo?]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const int arow = tid / TPR_A, acol = (tid % TPR_A) * 4;   // floats
    const int brow = tid / TPR_B, bcol = (tid % TPR_B) * 8;   // bf16 elements
    int a_oy[AP], a_ox[AP];
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        const int m = m0 + arow + RPP_A * j;
        if (m < p.M) {
            a_oy[j] = m / p.Wout;
            a_ox[j] = m - a_oy[j] * p.Wout;
        } else {
            a_oy[j] = -0x10000000;
            a_ox[j] = 0;
        }
    }
    const int kchunks = p.K / BK;
    const int niter = p.ks * p.ks * kchunks;

    float4 ra[AP];
    u32x4 rbh[BP], rbl[SPLIT ? BP : 1];
    auto load_tile = [&](int it) {
        const int t = it / kchunks;
        const int kc = it - t * kchunks;
        const int ty = t / p.ks, tx = t - ty * p.ks;
        const int kofs = kc * BK;
        const float* src;
        int ld, kk;
        if (kofs < p.k0) { src = p.a0; ld = p.lda0; kk = kofs; }
        else             { src = p.a1; ld = p.lda1; kk = kofs - p.k0; }
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            int iy, ix;
            bool ok;
            if (p.mode == SPEI_CONV) {
                iy = a_oy[j] * p.stride - p.pad + ty;
                ix = a_ox[j] * p.stride - p.pad + tx;
                ok = (iy >= 0) & (iy < p.Hin) & (ix >= 0) & (ix < p.Win);
            } else {
                const int ny = a_oy[j] + p.pad - ty, nx = a_ox[j] + p.pad - tx;
                iy = ny / p.stride;
                ix = nx / p.stride;
                ok = (ny >= 0) & (nx >= 0) & (iy * p.stride == ny) & (ix * p.stride == nx) & (iy < p.Hin) & (ix < p.Win);
            }
            if (ok) ra[j] = *reinterpret_cast<const float4*>(src + ((size_t)iy * p.Win + ix) * ld + kk + acol);
            else    ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            if (brow + RPP_B * j < BN) {
                const size_t o = ((size_t)t * p.N + n0 + brow + RPP_B * j) * p.K + kofs + bcol;
                rbh[j] = *reinterpret_cast<const u32x4*>(p.wh + o);
                if (SPLIT) rbl[j] = *reinterpret_cast<const u32x4*>(p.wl + o);
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* base = smem + buf * BUF_BYTES;
        unsigned char* ah = base;
        unsigned char* al = base + A_BYTES;
        unsigned char* bh = base + NPART * A_BYTES;
        unsigned char* bl = bh + B_BYTES;
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const int o = (arow + RPP_A * j) * PITCH + acol * 2;
            const bf16x4 h = cvt4(ra[j]);
            *reinterpret_cast<bf16x4*>(ah + o) = h;
            if (SPLIT) *reinterpret_cast<bf16x4*>(al + o) = cvt4(resid4(ra[j], h));
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            if (brow + RPP_B * j < BN) {
                const int o = (brow + RPP_B * j) * PITCH + bcol * 2;
                *reinterpret_cast<u32x4*>(bh + o) = rbh[j];
                if (SPLIT) *reinterpret_cast<u32x4*>(bl + o) = rbl[j];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int fr = lane & 31, fk = lane >> 5;
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        if (it + 1 < niter) load_tile(it + 1);
        const unsigned char* base = smem + buf * BUF_BYTES;
        const unsigned char* ah = base + (wm * TM * 32 + fr) * PITCH + fk * 16;
        const unsigned char* bh = base + NPART * A_BYTES + (wn * TN * 32 + fr) * PITCH + fk * 16;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 av[TM], bv[TN], avl[SPLIT ? TM : 1], bvl[SPLIT ? TN : 1];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                av[i] = *reinterpret_cast<const bf16x8*>(ah + i * 32 * PITCH + ks * 32);
                if (SPLIT) avl[i] = *reinterpret_cast<const bf16x8*>(ah + A_BYTES + i * 32 * PITCH + ks * 32);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bv[j] = *reinterpret_cast<const bf16x8*>(bh + j * 32 * PITCH + ks * 32);
                if (SPLIT) bvl[j] = *reinterpret_cast<const bf16x8*>(bh + B_BYTES + j * 32 * PITCH + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (SPLIT) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avl[i], bv[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bvl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
        }
        if (it + 1 < niter) store_tile(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + fr;
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (m < p.M) {
                    float v = acc[i][j][r] + bias;
                    if (p.act == SPEI_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.act == SPEI_ACT_GELU) v = gelu_erf(v);
                    if (p.rowscale) v *= p.rowscale[m];
                    if (p.res) v += p.res[(size_t)m * p.ldr + n];
                    p.out[(size_t)m * p.ldo + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK, bool SPLIT>
int launch(const IgemmParams& p, hipStream_t s) {
    constexpr int PITCH = 2 * BK + 16;
    const size_t lds = (size_t)2 * (SPLIT ? 2 : 1) * (BM + BN) * PITCH;
    ensure_dyn_lds<&igemm_bf16_kernel<BM, BN, WM, WN, BK, SPLIT>>(lds);
    dim3 grid(cdiv(p.M, BM), p.N / BN);
    hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN, BK, SPLIT>), grid, dim3(256), lds, s, p);
    SPEI_CHECK_LAUNCH("spei_igemm_bf16");
    return 0;
}

template <bool SPLIT>
int dispatch(const IgemmParams& p, hipStream_t s) {
    const bool k64 = (p.k0 % 64 == 0) && (p.k1 % 64 == 0) && !SPLIT;
    if (p.N % 128 == 0) return k64 ? launch<128, 128, 2, 2, 64, SPLIT>(p, s) : launch<128, 128, 2, 2, 32, SPLIT>(p, s);
    if (p.N % 64 == 0) return k64 ? launch<128, 64, 4, 1, 64, SPLIT>(p, s) : launch<128, 64, 4, 1, 32, SPLIT>(p, s);
    return launch<128, 32, 4, 1, 32, SPLIT>(p, s);
}

}  // namespace

extern "C" int spei_igemm_bf16(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w_hi,
                               const void* w_lo, const float* bias, float* out, int ldo, const float* residual, int ldr,
                               const float* rowscale, int Hin, int Win, int Hout, int Wout, int N, int ksize, int stride,
                               int pad, int mode, int act, spei_stream_t stream) {
    SPEI_REQUIRE(a0 && w_hi && out, "spei_igemm_bf16: null pointer");
    SPEI_REQUIRE(k0 > 0 && k0 % 32 == 0 && k1 >= 0 && k1 % 32 == 0, "spei_igemm_bf16: k0=%d k1=%d must be multiples of 32", k0, k1);
    SPEI_REQUIRE(k1 == 0 || a1, "spei_igemm_bf16: a1 missing");
    SPEI_REQUIRE(N > 0 && N % 32 == 0, "spei_igemm_bf16: N=%d must be a multiple of 32", N);
    SPEI_REQUIRE(lda0 % 4 == 0 && (k1 == 0 || lda1 % 4 == 0) && ldo >= N, "spei_igemm_bf16: bad row strides");
    SPEI_REQUIRE(lda0 >= k0 && (k1 == 0 || lda1 >= k1), "spei_igemm_bf16: lda < k");
    SPEI_REQUIRE(ksize == 1 || ksize == 3 || ksize == 5, "spei_igemm_bf16: ksize=%d", ksize);
    SPEI_REQUIRE(stride == 1 || stride == 2, "spei_igemm_bf16: stride=%d", stride);
    SPEI_REQUIRE(mode == SPEI_CONV || mode == SPEI_CONV_TRANSPOSED, "spei_igemm_bf16: mode=%d", mode);
    SPEI_REQUIRE(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "spei_igemm_bf16: empty map");
    SPEI_REQUIRE((int64_t)Hout * Wout < (1ll << 30) && (int64_t)Hin * Win < (1ll << 30), "spei_igemm_bf16: map too large");
    SPEI_REQUIRE(((uintptr_t)a0 % 16 == 0) && ((uintptr_t)w_hi % 16 == 0) && (!a1 || (uintptr_t)a1 % 16 == 0) &&
                 (!w_lo || (uintptr_t)w_lo % 16 == 0), "spei_igemm_bf16: operands must be 16-byte aligned");
    if (mode == SPEI_CONV) {
        SPEI_REQUIRE(Hout == (Hin + 2 * pad - ksize) / stride + 1 && Wout == (Win + 2 * pad - ksize) / stride + 1,
                     "spei_igemm_bf16: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", Hout, Wout, Hin, Win, ksize, stride, pad);
    } else {
        SPEI_REQUIRE(Hout == Hin * stride && Wout == Win * stride && pad == ksize / 2,
                     "spei_igemm_bf16: transposed conv expects out = in*stride, pad = k/2");
    }
    IgemmParams p;
    p.a0 = a0; p.a1 = a1; p.wh = (const __bf16*)w_hi; p.wl = (const __bf16*)w_lo; p.bias = bias; p.out = out;
    p.res = residual; p.rowscale = rowscale;
    p.lda0 = lda0; p.lda1 = lda1; p.k0 = k0; p.k1 = k1; p.ldo = ldo; p.ldr = ldr;
    p.M = Hout * Wout; p.N = N; p.K = k0 + k1;
    p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout;
    p.ks = ksize; p.stride = stride; p.pad = pad; p.mode = mode; p.act = act;
    return w_lo ? dispatch<true>(p, (hipStream_t)stream) : dispatch<false>(p, (hipStream_t)stream);
}

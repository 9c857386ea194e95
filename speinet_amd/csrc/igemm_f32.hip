// Implicit-GEMM convolution / linear on the gfx950 f32 matrix pipe (v_mfma_f32_32x32x2_f32: exact f32,
// bitwise a k-ordered fmaf chain).  One kernel serves nn.Linear, 1x1 / 3x3 / 5x5 nn.Conv2d (stride 1/2) and
// ConvTranspose2d(3, s2) of the reference (model/block.py:26-47, model/swinir.py:18-29,105-108,467,667,716,742,
// model/speinet.py:55-66, model/recons_video_ori.py:44-71).
//
//   out[m][n] = epi( sum_t sum_k A[src(m,t)][k] * W[t][n][k] )
//
// A rows are NHWC pixel rows (K contiguous), gathered per tap t with zero fill outside the image; W is packed
// [tap][Cout][Cin] so both operands are "rows with K contiguous".  Tile BM x BN per 256-thread workgroup,
// BK = 32; register-staged double buffering (global -> VGPR while the MFMAs of the current tile run, then
// VGPR -> LDS); LDS rows padded to 33 floats so the MFMA operand reads (lane -> row, fixed k) and the
// transposing ds_write_b32 stores are bank-conflict free.
#include "common.h"

namespace {

struct IgemmParams {
    const float* a0;
    const float* a1;
    const float* w;
    const float* bias;
    float* out;
    const float* res;
    const float* rowscale;
    int lda0, lda1, k0, k1;
    int ldo, ldr;
    int M, N, K;
    int Hin, Win, Hout, Wout;
    int ks, stride, pad, mode, act;
    long long sa0, sa1, so, sr, ss;   // per-sample strides in floats (blockIdx.z = sample of a batch of equally sized maps)
};

constexpr int BK = 32;
constexpr int LDSK = BK + 1;

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const IgemmParams pin) {
    IgemmParams p = pin;
    {
        const long long z = blockIdx.z;           // sample: every map pointer moves by one sample's extent
        p.a0 += z * pin.sa0;
        if (p.a1) p.a1 += z * pin.sa1;
        p.out += z * pin.so;
        if (p.res) p.res += z * pin.sr;
        if (p.rowscale) p.rowscale += z * pin.ss;
    }
    constexpr int TM = BM / WM / 32;   // 32x32 MFMA tiles per wave along M
    constexpr int TN = BN / WN / 32;
    constexpr int AP = BM / 32;        // float4 loads per thread for the A tile
    constexpr int BP = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                           // [2][BM][LDSK]
    float* Bs = smem + 2 * BM * LDSK;           // [2][BN][LDSK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread load coordinates -------------------------------------------------------------
    const int lrow = tid >> 3;          // 0..31
    const int lcol = (tid & 7) * 4;     // 0,4,..,28
    int a_oy[AP], a_ox[AP];
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        const int m = m0 + lrow + 32 * j;
        if (m < p.M) {
            a_oy[j] = m / p.Wout;
            a_ox[j] = m - a_oy[j] * p.Wout;
        } else {
            a_oy[j] = -0x10000000;      // forces every tap out of range
            a_ox[j] = 0;
        }
    }
    const int kchunks = p.K / BK;
    const int T = p.ks * p.ks;
    const int niter = T * kchunks;

    float4 ra[AP], rb[BP];
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
            } else {   // transposed: oy = iy*stride - pad + ty
                const int ny = a_oy[j] + p.pad - ty, nx = a_ox[j] + p.pad - tx;
                iy = ny / p.stride;
                ix = nx / p.stride;
                ok = (ny >= 0) & (nx >= 0) & (iy * p.stride == ny) & (ix * p.stride == nx) & (iy < p.Hin) & (ix < p.Win);
            }
            if (ok) ra[j] = *reinterpret_cast<const float4*>(src + ((size_t)iy * p.Win + ix) * ld + kk + lcol);
            else    ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int n = n0 + lrow + 32 * j;
            rb[j] = *reinterpret_cast<const float4*>(p.w + ((size_t)t * p.N + n) * p.K + kofs + lcol);
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * BM * LDSK;
        float* b = Bs + buf * BN * LDSK;
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            float* d = a + (lrow + 32 * j) * LDSK + lcol;
            d[0] = ra[j].x; d[1] = ra[j].y; d[2] = ra[j].z; d[3] = ra[j].w;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            float* d = b + (lrow + 32 * j) * LDSK + lcol;
            d[0] = rb[j].x; d[1] = rb[j].y; d[2] = rb[j].z; d[3] = rb[j].w;
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
        const float* a = As + buf * BM * LDSK + (wm * TM * 32 + fr) * LDSK + fk;
        const float* b = Bs + buf * BN * LDSK + (wn * TN * 32 + fr) * LDSK + fk;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a[i * 32 * LDSK + kk];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b[j * 32 * LDSK + kk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < niter) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ------------------------
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

template <int BM, int BN, int WM, int WN>
int launch(const IgemmParams& p, int batch, hipStream_t s) {
    const size_t lds = (size_t)2 * (BM + BN) * LDSK * sizeof(float);
    ensure_dyn_lds<&igemm_f32_kernel<BM, BN, WM, WN>>(lds);
    dim3 grid(cdiv(p.M, BM), p.N / BN, batch);
    hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN>), grid, dim3(256), lds, s, p);
    SPEI_CHECK_LAUNCH("spei_igemm_f32");
    return 0;
}

}  // namespace

static int igemm_f32_run(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const float* w,
                         const float* bias, float* out, int ldo, const float* residual, int ldr,
                         const float* rowscale, int Hin, int Win, int Hout, int Wout, int N, int ksize,
                         int stride, int pad, int mode, int act, int batch, spei_stream_t stream) {
    SPEI_REQUIRE(batch >= 1 && batch <= 65535, "spei_igemm_f32: batch=%d", batch);
    SPEI_REQUIRE(a0 && w && out, "spei_igemm_f32: null pointer");
    SPEI_REQUIRE(k0 > 0 && k0 % 32 == 0 && k1 >= 0 && k1 % 32 == 0, "spei_igemm_f32: k0=%d k1=%d must be multiples of 32", k0, k1);
    SPEI_REQUIRE(k1 == 0 || a1, "spei_igemm_f32: a1 missing");
    SPEI_REQUIRE(N > 0 && N % 32 == 0, "spei_igemm_f32: N=%d must be a multiple of 32", N);
    SPEI_REQUIRE(lda0 % 4 == 0 && (k1 == 0 || lda1 % 4 == 0) && ldo >= N, "spei_igemm_f32: bad row strides");
    SPEI_REQUIRE(lda0 >= k0 && (k1 == 0 || lda1 >= k1), "spei_igemm_f32: lda < k");
    SPEI_REQUIRE(ksize == 1 || ksize == 3 || ksize == 5, "spei_igemm_f32: ksize=%d", ksize);
    SPEI_REQUIRE(stride == 1 || stride == 2, "spei_igemm_f32: stride=%d", stride);
    SPEI_REQUIRE(mode == SPEI_CONV || mode == SPEI_CONV_TRANSPOSED, "spei_igemm_f32: mode=%d", mode);
    SPEI_REQUIRE(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "spei_igemm_f32: empty map");
    SPEI_REQUIRE((int64_t)Hout * Wout < (1ll << 30) && (int64_t)Hin * Win < (1ll << 30), "spei_igemm_f32: map too large");
    SPEI_REQUIRE(((uintptr_t)a0 % 16 == 0) && ((uintptr_t)w % 16 == 0) && (!a1 || (uintptr_t)a1 % 16 == 0),
                 "spei_igemm_f32: operands must be 16-byte aligned");
    if (mode == SPEI_CONV) {
        SPEI_REQUIRE(Hout == (Hin + 2 * pad - ksize) / stride + 1 && Wout == (Win + 2 * pad - ksize) / stride + 1,
                     "spei_igemm_f32: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", Hout, Wout, Hin, Win, ksize, stride, pad);
    } else {
        SPEI_REQUIRE(Hout == Hin * stride && Wout == Win * stride && pad == ksize / 2,
                     "spei_igemm_f32: transposed conv expects out = in*stride, pad = k/2");
    }
    IgemmParams p;
    p.a0 = a0; p.a1 = a1; p.w = w; p.bias = bias; p.out = out; p.res = residual; p.rowscale = rowscale;
    p.lda0 = lda0; p.lda1 = lda1; p.k0 = k0; p.k1 = k1; p.ldo = ldo; p.ldr = ldr;
    p.M = Hout * Wout; p.N = N; p.K = k0 + k1;
    p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout;
    p.ks = ksize; p.stride = stride; p.pad = pad; p.mode = mode; p.act = act;
    p.sa0 = (long long)Hin * Win * lda0; p.sa1 = (long long)Hin * Win * lda1;
    p.so = (long long)Hout * Wout * ldo; p.sr = (long long)Hout * Wout * ldr; p.ss = (long long)Hout * Wout;
    hipStream_t s = (hipStream_t)stream;
    if (N % 128 == 0) return launch<128, 128, 2, 2>(p, batch, s);
    if (N % 64 == 0) return launch<128, 64, 4, 1>(p, batch, s);
    return launch<128, 32, 4, 1>(p, batch, s);
}

extern "C" int spei_igemm_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const float* w,
                              const float* bias, float* out, int ldo, const float* residual, int ldr,
                              const float* rowscale, int Hin, int Win, int Hout, int Wout, int N, int ksize,
                              int stride, int pad, int mode, int act, spei_stream_t stream) {
    return igemm_f32_run(a0, lda0, k0, a1, lda1, k1, w, bias, out, ldo, residual, ldr, rowscale, Hin, Win, Hout, Wout, N, ksize, stride, pad,
                         mode, act, 1, stream);
}

extern "C" int spei_igemm_f32_batched(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const float* w,
                                      const float* bias, float* out, int ldo, const float* residual, int ldr,
                                      const float* rowscale, int Hin, int Win, int Hout, int Wout, int N, int ksize,
                                      int stride, int pad, int mode, int act, int batch, spei_stream_t stream) {
    return igemm_f32_run(a0, lda0, k0, a1, lda1, k1, w, bias, out, ldo, residual, ldr, rowscale, Hin, Win, Hout, Wout, N, ksize, stride, pad,
                         mode, act, batch, stream);
}

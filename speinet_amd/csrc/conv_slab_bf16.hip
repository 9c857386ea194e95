// Slab-resident convolution / linear GEMM on the gfx950 bf16 matrix pipe (v_mfma_f32_32x32x16_bf16).
//
// igemm_bf16.hip re-stages the A operand from global memory for every (tap, K-chunk) and pays a load ->
// ds_write -> barrier round trip per 16..64 MFMAs: at bf16 MFMA rates that loop is latency-bound.  Here each
// 256-thread workgroup
//   1. stages ONCE the input pixels its output tile needs — TH x 32 output pixels plus the (k-1) halo, ALL input
//      channels — into LDS as bf16 (fp32 in HBM -> cvt -> LDS; optional second slab with the bf16 residual for the
//      split "bf16x3" mode), pixel pitch 2*K+16 bytes (odd multiple of 16 B => conflict-free ds_read_b128);
//   2. runs a BARRIER-FREE main loop: per (tap, 32-wide K group) every wave reads its A fragments from the slab at
//      a tap-shifted address and takes its B fragments straight from global memory, where the weights were
//      pre-packed on the host in MFMA fragment order ([n-tile][tap][k-step][lane][8] bf16: one fully coalesced
//      1 KiB load per fragment), prefetched four groups ahead in a register ring;
//   3. applies the same epilogue as the other GEMM kernels (bias, ReLU/GELU, per-pixel scale, residual).
// HBM sees every input pixel once per output tile (+halo) instead of once per tap.
// Call sites replaced: as igemm_f32.hip (convolutions with stride 1/2 and linears; transposed convs stay there).
#include <stdlib.h>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct SlabParams {
    const float* a0;
    const float* a1;
    const __bf16* wh;    // fragment-ordered
    const __bf16* wl;
    const float* bias;
    float* out;
    const float* res;
    const float* rowscale;
    int lda0, lda1, k0, k1;
    int ldo, ldr;
    int N, K;
    int Hin, Win, Hout, Wout;
    int ks, stride, pad, act;
    int TH, TW, IH, IW;      // output tile, input slab (pixels)
    int tiles_x;
    int slab_bytes;          // bytes of one slab (hi); lo follows when SPLIT
    int tw_shift;            // log2(TW): 5 (2-D maps, 32-pixel tile rows) or 0 (token lists)
    int c4_shift;            // log2(K/4) or -1 when K/4 is not a power of two
    int iw_magic;            // ceil(2^20 / IW): pix / IW == (pix * iw_magic) >> 20 for pix < 2048
    int n_chunks;            // 32*WN*TN-column chunks looped inside the workgroup (gridDim.y == 1)
    int dbg;                 // ablation switches for tools/ablate_slab.py (0 in production): 1 no staging loads,
                             // 2 no MFMA loop, 4 no epilogue stores
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

constexpr int G = 2;        // k-steps (of 16) per group
constexpr int RING = 4;     // groups of B fragments in flight

template <int WM, int WN, int TM, int TN, bool SPLIT>
__global__ __launch_bounds__(256) void conv_slab_kernel(const SlabParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int pitch = 2 * p.K + 16;
    unsigned char* slab = smem;
    int* goff = reinterpret_cast<int*>(smem + (SPLIT ? 2 : 1) * p.slab_bytes);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fk = lane >> 5;
    const int tile_y = blockIdx.x / p.tiles_x, tile_x = blockIdx.x - tile_y * p.tiles_x;
    const int oy0 = tile_y * p.TH, ox0 = tile_x * p.TW;
    const int ks16 = p.K / 16;
    const int kg_per_tap = ks16 / G;
    const int ngroups = p.ks * p.ks * kg_per_tap;

    // ---- group -> slab byte offset table ------------------------------------------------------------
    for (int g = tid; g < ngroups; g += 256) {
        const int t = g / kg_per_tap, kg = g - t * kg_per_tap;
        const int ty = t / p.ks, tx = t - ty * p.ks;
        goff[g] = (ty * p.IW + tx) * pitch + kg * (G * 32);
    }

    // ---- stage the input slab: fp32 -> bf16 (hi [, lo]) ----------------------------------------------
    {
        const int c4n = p.K / 4;
        const int total = p.IH * p.IW * c4n;
        const int gy0 = oy0 * p.stride - p.pad, gx0 = ox0 * p.stride - p.pad;
        constexpr int U = 8;
        for (int base = tid; base < total; base += 256 * U) {
            float4 v[U];
            int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 256;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                off[u] = -1;
                if (idx < total) {
                    const int pix = p.c4_shift >= 0 ? (idx >> p.c4_shift) : idx / c4n;
                    const int c = (idx - pix * c4n) * 4;
                    const int iy = (int)(((unsigned)pix * (unsigned)p.iw_magic) >> 20), ix = pix - iy * p.IW;
                    const int gy = gy0 + iy, gx = gx0 + ix;
                    off[u] = pix * pitch + c * 2;
                    if (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win && !(p.dbg & 1)) {
                        const size_t gp = (size_t)gy * p.Win + gx;
                        v[u] = (c < p.k0) ? *reinterpret_cast<const float4*>(p.a0 + gp * p.lda0 + c)
                                          : *reinterpret_cast<const float4*>(p.a1 + gp * p.lda1 + (c - p.k0));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (off[u] >= 0) {
                    bf16x4 h;
                    h[0] = (__bf16)v[u].x; h[1] = (__bf16)v[u].y; h[2] = (__bf16)v[u].z; h[3] = (__bf16)v[u].w;
                    *reinterpret_cast<bf16x4*>(slab + off[u]) = h;
                    if (SPLIT) {
                        bf16x4 l;
                        l[0] = (__bf16)(v[u].x - (float)h[0]); l[1] = (__bf16)(v[u].y - (float)h[1]);
                        l[2] = (__bf16)(v[u].z - (float)h[2]); l[3] = (__bf16)(v[u].w - (float)h[3]);
                        *reinterpret_cast<bf16x4*>(slab + p.slab_bytes + off[u]) = l;
                    }
                }
            }
        }
    }

    // ---- per-lane A bases, B fragment stream ------------------------------------------------------------
    int abase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pt = (wm * TM + i) * 32 + fr;        // pixel index inside the tile
        const int py = pt >> p.tw_shift, px = pt - (py << p.tw_shift);
        abase[i] = ((py * p.stride) * p.IW + px * p.stride) * pitch + fk * 16;
    }
    const int T = p.ks * p.ks;
    const size_t frag_per_nt = (size_t)T * ks16 * 64 * 8;        // bf16 elements per 32-column n-tile
    const __bf16* bptr[TN];
    const __bf16* bptr_lo[TN];
    auto set_chunk = [&](int nc) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nt = ((blockIdx.y * p.n_chunks + nc) * WN + wn) * TN + j;
            bptr[j] = p.wh + nt * frag_per_nt + lane * 8;
            bptr_lo[j] = SPLIT ? p.wl + nt * frag_per_nt + lane * 8 : nullptr;
        }
    };
    set_chunk(0);

    bf16x8 bring[RING][G][TN];
    bf16x8 bring_lo[SPLIT ? RING : 1][G][TN];
    auto load_b = [&](int slot, int g) {
#pragma unroll
        for (int s = 0; s < G; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const size_t o = (size_t)(g * G + s) * 512;
                bring[slot][s][j] = *reinterpret_cast<const bf16x8*>(bptr[j] + o);
                if (SPLIT) bring_lo[slot][s][j] = *reinterpret_cast<const bf16x8*>(bptr_lo[j] + o);
            }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int d = 0; d < RING; ++d)
        if (d < ngroups) load_b(d, d);

    __syncthreads();     // slab + offset table visible; the only barrier of the kernel

  for (int nc = 0; nc < p.n_chunks; ++nc) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int g0 = 0; g0 < ((p.dbg & 2) ? 0 : ngroups); g0 += RING) {
#pragma unroll
        for (int d = 0; d < RING; ++d) {
            const int g = g0 + d;
            if (g < ngroups) {
                const int go = goff[g];
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    bf16x8 av[TM], avl[SPLIT ? TM : 1];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        av[i] = *reinterpret_cast<const bf16x8*>(slab + abase[i] + go + s * 32);
                        if (SPLIT) avl[i] = *reinterpret_cast<const bf16x8*>(slab + p.slab_bytes + abase[i] + go + s * 32);
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (SPLIT) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avl[i], bring[d][s][j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bring_lo[SPLIT ? d : 0][s][j], acc[i][j], 0, 0, 0);
                            }
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bring[d][s][j], acc[i][j], 0, 0, 0);
                        }
                }
                if (g + RING < ngroups) load_b(d, g + RING);
            }
        }
    }
    // start the next chunk's weight stream before the epilogue's stores
    const int nbase = ((blockIdx.y * p.n_chunks + nc) * WN + wn) * TN;
    if (nc + 1 < p.n_chunks) {
        set_chunk(nc + 1);
#pragma unroll
        for (int d = 0; d < RING; ++d)
            if (d < ngroups) load_b(d, d);
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
    // Row / address math is invariant across the chunk loop: launder one input so the compiler does not hoist
    // ~200 registers of addresses out of the loop and starve the main loop (measured: 255 VGPRs, serialised MFMAs).
    int fk_e = fk;
    asm volatile("" : "+v"(fk_e));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = (nbase + j) * 32 + fr;
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pt = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk_e;
                const int py = pt >> p.tw_shift, px = pt - (py << p.tw_shift);
                const int oy = oy0 + py, ox = ox0 + px;
                if (oy < p.Hout && ox < p.Wout && !(p.dbg & 4)) {
                    const size_t m = (size_t)oy * p.Wout + ox;
                    float v = acc[i][j][r] + bias;
                    if (p.act == SPEI_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.act == SPEI_ACT_GELU) v = gelu_erf(v);
                    if (p.rowscale) v *= p.rowscale[m];
                    if (p.res) v += p.res[m * p.ldr + n];
                    p.out[m * p.ldo + n] = v;
                }
            }
        }
    }
  }   // n chunks
}

template <int WM, int WN, int TM, int TN, bool SPLIT>
int launch(const SlabParams& p, size_t lds, hipStream_t s) {
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_slab_kernel<WM, WN, TM, TN, SPLIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    SlabParams q = p;
    q.n_chunks = p.N / (WN * TN * 32);
    static const int dbg = getenv("SPEI_SLAB_DBG") ? atoi(getenv("SPEI_SLAB_DBG")) : 0;
    q.dbg = dbg;
    dim3 grid(p.tiles_x * cdiv(p.Hout, p.TH), 1);
    hipLaunchKernelGGL((conv_slab_kernel<WM, WN, TM, TN, SPLIT>), grid, dim3(256), lds, s, q);
    SPEI_CHECK_LAUNCH("spei_conv_slab_bf16");
    return 0;
}

template <bool SPLIT>
int dispatch(SlabParams& p, hipStream_t s) {
    const int pitch = 2 * p.K + 16;
    const int nparts = SPLIT ? 2 : 1;
    const int T = p.ks * p.ks;
    const int ngroups = T * (p.K / 32);
    const bool linear = (p.Wout == 1 && p.ks == 1);
    // tile rows x 32 pixels (2-D maps) or rows x 1 (token lists); pick the largest tile whose slab(s) fit ~96 KB
    auto setup = [&](int mtile) {
        if (linear) { p.TH = mtile; p.TW = 1; }
        else { p.TH = mtile / 32; p.TW = 32; }
        p.IH = (p.TH - 1) * p.stride + p.ks;
        p.IW = (p.TW - 1) * p.stride + p.ks;
        p.slab_bytes = ((p.IH * p.IW * pitch + 15) / 16) * 16;
        p.tiles_x = cdiv(p.Wout, p.TW);
        p.tw_shift = linear ? 0 : 5;
        const int c4n = p.K / 4;
        p.c4_shift = -1;
        for (int sft = 0; sft < 12; ++sft) if ((1 << sft) == c4n) p.c4_shift = sft;
        p.iw_magic = ((1 << 20) + p.IW - 1) / p.IW;
        return (size_t)nparts * p.slab_bytes + (size_t)ngroups * sizeof(int);
    };
    const size_t budget = 96 * 1024;
    if (p.N % 128 == 0) {
        size_t lds = setup(128);
        const int64_t tiles128 = (int64_t)p.tiles_x * cdiv(p.Hout, p.TH);
        static const int min_tiles = getenv("SPEI_SLAB_MIN_TILES128") ? atoi(getenv("SPEI_SLAB_MIN_TILES128")) : 1024;
        if (lds <= budget && tiles128 >= min_tiles) return launch<1, 4, 4, 1, SPLIT>(p, lds, s);
        lds = setup(64);
        if (lds <= 160 * 1024 - 512) return launch<1, 4, 2, 1, SPLIT>(p, lds, s);
        spei_set_error("spei_conv_slab_bf16: slab of %zu bytes does not fit LDS", lds);
        return -1;
    }
    if (p.N % 64 == 0) {
        size_t lds = setup(128);
        if (lds <= 160 * 1024 - 512) return launch<2, 2, 2, 1, SPLIT>(p, lds, s);
        spei_set_error("spei_conv_slab_bf16: slab of %zu bytes does not fit LDS", lds);
        return -1;
    }
    size_t lds = setup(256);
    if (lds > budget) lds = setup(128);
    if (p.TH * p.TW == 256) return launch<4, 1, 2, 1, SPLIT>(p, lds, s);
    if (lds <= 160 * 1024 - 512) return launch<4, 1, 1, 1, SPLIT>(p, lds, s);
    spei_set_error("spei_conv_slab_bf16: slab of %zu bytes does not fit LDS", lds);
    return -1;
}

}  // namespace

extern "C" int spei_conv_slab_bf16(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1,
                                   const void* wfrag_hi, const void* wfrag_lo, const float* bias, float* out, int ldo,
                                   const float* residual, int ldr, const float* rowscale, int Hin, int Win, int Hout,
                                   int Wout, int N, int ksize, int stride, int pad, int act, spei_stream_t stream) {
    SPEI_REQUIRE(a0 && wfrag_hi && out, "spei_conv_slab_bf16: null pointer");
    SPEI_REQUIRE(k0 > 0 && k0 % 32 == 0 && k1 >= 0 && k1 % 32 == 0, "spei_conv_slab_bf16: k0=%d k1=%d must be multiples of 32", k0, k1);
    SPEI_REQUIRE(k1 == 0 || a1, "spei_conv_slab_bf16: a1 missing");
    SPEI_REQUIRE(N > 0 && N % 32 == 0, "spei_conv_slab_bf16: N=%d must be a multiple of 32", N);
    SPEI_REQUIRE(lda0 % 4 == 0 && (k1 == 0 || lda1 % 4 == 0) && ldo >= N && lda0 >= k0 && (k1 == 0 || lda1 >= k1),
                 "spei_conv_slab_bf16: bad row strides");
    SPEI_REQUIRE(ksize == 1 || ksize == 3 || ksize == 5, "spei_conv_slab_bf16: ksize=%d", ksize);
    SPEI_REQUIRE(stride == 1 || stride == 2, "spei_conv_slab_bf16: stride=%d", stride);
    SPEI_REQUIRE(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "spei_conv_slab_bf16: empty map");
    SPEI_REQUIRE((int64_t)Hout * Wout < (1ll << 30) && (int64_t)Hin * Win < (1ll << 30), "spei_conv_slab_bf16: map too large");
    SPEI_REQUIRE(Hout == (Hin + 2 * pad - ksize) / stride + 1 && Wout == (Win + 2 * pad - ksize) / stride + 1,
                 "spei_conv_slab_bf16: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", Hout, Wout, Hin, Win, ksize, stride, pad);
    SPEI_REQUIRE(((uintptr_t)a0 % 16 == 0) && ((uintptr_t)wfrag_hi % 16 == 0) && (!a1 || (uintptr_t)a1 % 16 == 0) &&
                 (!wfrag_lo || (uintptr_t)wfrag_lo % 16 == 0), "spei_conv_slab_bf16: operands must be 16-byte aligned");
    SlabParams p;
    p.a0 = a0; p.a1 = a1; p.wh = (const __bf16*)wfrag_hi; p.wl = (const __bf16*)wfrag_lo; p.bias = bias; p.out = out;
    p.res = residual; p.rowscale = rowscale;
    p.lda0 = lda0; p.lda1 = lda1; p.k0 = k0; p.k1 = k1; p.ldo = ldo; p.ldr = ldr;
    p.N = N; p.K = k0 + k1;
    p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout;
    p.ks = ksize; p.stride = stride; p.pad = pad; p.act = act;
    return wfrag_lo ? dispatch<true>(p, (hipStream_t)stream) : dispatch<false>(p, (hipStream_t)stream);
}

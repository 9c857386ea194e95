// Weight-stationary 5x5 convolution for the 32-channel level of the encoder / decoder stacks (reference model/block.py:26-47,127-131:
// the two 5x5 convs of a ResBlock at full resolution; model/recons_video_ori.py:26-43,72-77 inBlock / outBlock) on the gfx950 16-bit
// matrix pipe.
//
// Why a second conv kernel: in conv_slab_kernel every wave streams its own copy of the layer's weight fragments from L2, and at 32
// output channels a fragment (1 KiB) feeds only TM = 2 MFMAs: 512 B of L2 -> CU traffic per MFMA, 64 B/clk at the matrix pipe's rate,
// where a CU takes in ~30 B/clk.  The 32-channel layers therefore run at the weight intake, 37-46 % MFMA busy, although a
// 32 -> 32 channel 5x5 layer has only 51 KB of weights.  Here the weights do not stream at all: a wave loads the layer's 50 fragments
// ONCE into registers (200 of the 512 a wave may hold when it is alone on its SIMD; MFMA A operands), and the workgroup — persistent,
// one per CU — walks over 16 x 32 pixel tiles of all the launch's maps:
//     acc[out channel][pixel] += W[tap, k-step] (registers) x X[pixel + tap, k-step] (B operand, one ds_read_b128 from the tile's slab)
// The only L2 / HBM traffic of the main loop is the NEXT tile's slab (20 x 36 pixels x 32 channels): its loads are issued before the 200
// MFMAs of the current tile and written to the second slab buffer after them.  LDS reads: 1 KiB per MFMA and wave = half the LDS rate.
// fp32 accumulation in tap order (dy, dx), then k-steps: not the summation order of conv_slab_kernel (K rotation per workgroup), the
// same operand rounding.
#include "common.h"

namespace {

constexpr int C = 32, KS = 5, PAD = 2, NTAP = KS * KS;
constexpr int TH = 16, TW = 32;                       // output tile
constexpr int IH = TH + KS - 1, IW = TW + KS - 1;     // 20 x 36 slab pixels
constexpr int PITCH = 2 * C + 16;                     // 80 bytes per slab pixel: conflict-free ds_read_b128 of 16 consecutive pixels
constexpr int SLAB = IH * IW * PITCH;                 // 57 600 bytes
constexpr int NTHR = 256, RT = 4;                     // 4 waves, 4 tile rows each

struct WsParams {
    const void* a;        // [batch][H*W][32] fp32 or LP
    const void* wfrag;    // fragment order [1][25][2][64][8] LP
    const float* bias;    // [32] or null
    void* out;            // [batch][H*W][32] LP or fp32
    int H, W, batch, act;
    int tiles_x, tiles_y, total;
    long long* stamps;    // tuning build: phase stamps of every workgroup's SECOND tile (steady state), else NULL
};

// MFMA with the A operand (the resident weight fragment) read straight from the accumulation-register half of the wave's 512 registers:
// left to the compiler the 200 weight registers live there too, but every use is preceded by v_accvgpr_read copies into vector registers
// (2.3 per MFMA, each a dependency the MFMA waits for: 61 cycles per MFMA instead of 32, in-kernel stamps)
__device__ __forceinline__ void mfma_aw(f32x16& acc, const lpv<_Float16>::x8& w, const lpv<_Float16>::x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
}
__device__ __forceinline__ void mfma_aw(f32x16& acc, const lpv<__bf16>::x8& w, const lpv<__bf16>::x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
}

// 16-byte chunks of a slab: pixel-major; TA = float: 8 chunks of 4 channels per pixel, TA = LP: 4 chunks of 8 channels
template <typename TA> struct Chunk { static constexpr int CPP = sizeof(TA) == 4 ? 8 : 4; };

template <typename LP, typename TA, typename TO>
__global__ __launch_bounds__(NTHR, 1) void conv32_ws_kernel(const WsParams p) {
    typedef typename lpv<LP>::x8 lp8;
    typedef typename lpv<LP>::x4 lp4;
    constexpr int CPP = Chunk<TA>::CPP;
    constexpr int NCH = IH * IW * CPP;                                  // chunks per slab
    constexpr int NIT = (NCH + NTHR - 1) / NTHR;                        // staging loads per thread: 23 (fp32) / 12 (16-bit)
    // fp32 input: the next slab travels in TWO halves (12 + 11 chunks per thread), the second requested in the middle of the MFMA loop when
    // the first has been written to LDS: 92 staging registers beside the 200 weight registers and the accumulators spill
    constexpr int NPH = sizeof(TA) == 4 ? 2 : 1;
    constexpr int NITH = (NIT + NPH - 1) / NPH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;

    // ---- the layer's weights: 50 fragments per wave, resident for the whole launch ------------------------------------------------------
    lp8 wreg[NTAP][2];
    {
        const LP* wp = static_cast<const LP*>(p.wfrag) + lane * 8;
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) wreg[t][s] = *reinterpret_cast<const lp8*>(wp + (t * 2 + s) * 512);
    }
    float biasv[16];                                                    // accumulator row r <-> output channel 8 (r >> 2) + 4 fk + (r & 3)
#pragma unroll
    for (int r = 0; r < 16; ++r) biasv[r] = p.bias ? p.bias[8 * (r >> 2) + 4 * fk + (r & 3)] : 0.f;

    typedef u32x4 stage_t;                                              // one 16-byte chunk as it comes from global memory
    stage_t sreg[NITH];
    const size_t map_elems = (size_t)p.H * p.W * C;
    const int per_map = p.tiles_x * p.tiles_y;

    auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
        b = t / per_map;
        const int r = t - b * per_map;
        const int ty = r / p.tiles_x;
        oy0 = ty * TH;
        ox0 = (r - ty * p.tiles_x) * TW;
    };
    // Chunk c = it * 256 + tid of a slab: slab pixel p0 + it * PPI (PPI = 256 / CPP), 16-byte piece `sub` of it — the piece and the LDS
    // offset are per-thread constants plus an immediate, the pixel's (iy, ix) advance by a fixed step with one wrap test: no divisions
    // in the tile loop (a wave is alone on its SIMD here: every vector instruction outside the MFMA loop is time the matrix pipe idles)
    constexpr int PPI = NTHR / CPP, E = 16 / (int)sizeof(TA);
    const int p0 = tid / CPP, sub = tid - p0 * CPP;
    const int iy0 = p0 / IW, ix0 = p0 - iy0 * IW;
    const int lds0 = p0 * PITCH + sub * (sizeof(TA) == 4 ? 8 : 16);
    auto advance = [](int& iy, int& ix) {
        ix += PPI % IW;
        iy += PPI / IW;
        if (ix >= IW) { ix -= IW; ++iy; }
    };
    // issue the global loads of tile t's slab, phase ph (clamped coordinates: every load is valid; pixels outside the map become zeros
    // at the write)
    auto load_slab = [&](int t, int ph) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        const TA* base = static_cast<const TA*>(p.a) + (size_t)b * map_elems + sub * E;
        int iy = iy0, ix = ix0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it >= ph * NITH && it < (ph + 1) * NITH) {
                const int iyc = min(iy, IH - 1);                           // (the last iteration runs past the slab for some threads)
                const int gy = min(max(oy0 - PAD + iyc, 0), p.H - 1), gx = min(max(ox0 - PAD + ix, 0), p.W - 1);
                sreg[it - ph * NITH] = *reinterpret_cast<const stage_t*>(base + (gy * p.W + gx) * C);
            }
            advance(iy, ix);
        }
    };
    // ... and write them (converted to the 16-bit operand format) into slab buffer `buf`
    // (chunks [lo, hi) of the phase: the 16-bit variant spreads them over the second half of the MFMA loop)
    auto store_slab = [&](int t, int buf, int ph, int lo, int hi) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        unsigned char* sl = smem + buf * SLAB + lds0;
        int iy = iy0, ix = ix0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it >= ph * NITH + lo && it < ph * NITH + hi && it < (ph + 1) * NITH && iy < IH) {
                const int j = it - ph * NITH;
                const bool ok = ((unsigned)(oy0 - PAD + iy) < (unsigned)p.H) & ((unsigned)(ox0 - PAD + ix) < (unsigned)p.W);
                if constexpr (sizeof(TA) == 4) {
                    const f32x4 v = __builtin_bit_cast(f32x4, sreg[j]);
                    lp4 h = to_lp4<LP>(v);
                    if (!ok) h = lp4{(LP)0.f, (LP)0.f, (LP)0.f, (LP)0.f};
                    *reinterpret_cast<lp4*>(sl + it * PPI * PITCH) = h;
                } else {
                    *reinterpret_cast<stage_t*>(sl + it * PPI * PITCH) = ok ? sreg[j] : stage_t{0u, 0u, 0u, 0u};
                }
            }
            advance(iy, ix);
        }
    };

    int t = blockIdx.x;
    if (t >= p.total) return;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        load_slab(t, ph);
        store_slab(t, 0, ph, 0, NITH);
    }
    lds_barrier();

    // per-lane slab offsets of the wave's four tile rows (pixel column fr, channel half fk): the tap and k-step are immediates
    int boff[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) boff[i] = ((wave * RT + i) * IW + fr) * PITCH + fk * 16;

    int buf = 0;
    for (; t < p.total; t += gridDim.x, buf ^= 1) {
        const int tn = t + gridDim.x;
        const bool more = tn < p.total;
        const int tl = more ? tn : t;                                   // (unconditional loads: a conditional one keeps the old registers alive)
        const bool st_ = t == (int)blockIdx.x + (int)gridDim.x;
        if (st_) SPEI_STAMP(p.stamps, 0);
        load_slab(tl, 0);
        if (st_) SPEI_STAMP(p.stamps, 1);
        __builtin_amdgcn_sched_barrier(0);

        const unsigned char* sl = smem + buf * SLAB;
        f32x16 acc[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = biasv[r];
        // B fragments (pixels) two steps ahead of the MFMAs that consume them: a three-slot ring of 4 fragments
        lp8 bq[3][RT];
        auto step_off = [](int step) { const int tap = step >> 1; return ((tap / KS) * IW + (tap % KS)) * PITCH + (step & 1) * 32; };
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < RT; ++i) bq[d][i] = *reinterpret_cast<const lp8*>(sl + boff[i] + step_off(d));
#pragma unroll
        for (int step = 0; step < NTAP * 2; ++step) {
            const int tap = step >> 1, s2 = step & 1;
            if (step + 2 < NTAP * 2) {
#pragma unroll
                for (int i = 0; i < RT; ++i) bq[(step + 2) % 3][i] = *reinterpret_cast<const lp8*>(sl + boff[i] + step_off(step + 2));
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) mfma_aw(acc[i], wreg[tap][s2], bq[step % 3][i]);
            if (NPH == 2 && step == NTAP) {                             // half way: first half of the next slab -> LDS, request the second
                store_slab(tl, buf ^ 1, 0, 0, NITH);
                load_slab(tl, 1);
            }
            // 16-bit input: the next slab's 12 chunks go to LDS one per two steps in the second half of the loop (requested ~3 us ago),
            // under the MFMAs instead of after them
            if (NPH == 1 && step >= NTAP && ((step - NTAP) & 1) == 0 && (step - NTAP) / 2 < NITH)
                store_slab(tl, buf ^ 1, 0, (step - NTAP) / 2, (step - NTAP) / 2 + 1);
        }

        if (st_) SPEI_STAMP(p.stamps, 2);
        // ---- epilogue: channel quadruples of one pixel per lane -----------------------------------------------------------------------
        {
            int b, oy0, ox0;
            tile_origin(t, b, oy0, ox0);
            TO* ob = static_cast<TO*>(p.out) + (size_t)b * map_elems;
            const int ox = ox0 + fr;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int oy = oy0 + wave * RT + i;
                if (oy < p.H && ox < p.W) {
                    TO* o = ob + ((size_t)oy * p.W + ox) * C + 4 * fk;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                        if (p.act == SPEI_ACT_RELU) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                        if constexpr (sizeof(TO) == 4) *reinterpret_cast<f32x4*>(o + 8 * g) = v;
                        else *reinterpret_cast<lp4*>(o + 8 * g) = to_lp4<LP>(v);
                    }
                }
            }
        }
        if (st_) SPEI_STAMP(p.stamps, 3);
        if (NPH == 2) store_slab(tl, buf ^ 1, 1, 0, NITH);                       // (the last tile re-stages itself into the idle buffer: never read)
        else store_slab(tl, buf ^ 1, 0, (NTAP + 1) / 2, NITH);          // whatever the loop's 13 slots did not cover (nothing: 12 chunks)
        if (st_) SPEI_STAMP(p.stamps, 4);
        lds_barrier();
        if (st_) SPEI_STAMP(p.stamps, 5);                                                  // the next slab is complete; everybody is done with this one
    }
}

template <typename LP>
int launch_ws(const WsParams& p, int a16, int o16, hipStream_t st) {
    const size_t lds = 2 * (size_t)SLAB;
    const int cus = spei_num_cus();
    const dim3 grid(p.total < cus ? p.total : cus);
    if (a16 && o16) {
        ensure_dyn_lds<&conv32_ws_kernel<LP, LP, LP>>(lds);
        hipLaunchKernelGGL((conv32_ws_kernel<LP, LP, LP>), grid, dim3(NTHR), lds, st, p);
    } else if (!a16 && o16) {
        ensure_dyn_lds<&conv32_ws_kernel<LP, float, LP>>(lds);
        hipLaunchKernelGGL((conv32_ws_kernel<LP, float, LP>), grid, dim3(NTHR), lds, st, p);
    } else if (a16 && !o16) {
        ensure_dyn_lds<&conv32_ws_kernel<LP, LP, float>>(lds);
        hipLaunchKernelGGL((conv32_ws_kernel<LP, LP, float>), grid, dim3(NTHR), lds, st, p);
    } else {
        ensure_dyn_lds<&conv32_ws_kernel<LP, float, float>>(lds);
        hipLaunchKernelGGL((conv32_ws_kernel<LP, float, float>), grid, dim3(NTHR), lds, st, p);
    }
    return 0;
}

}  // namespace

extern "C" int spei_conv32_ws16(int fmt, const void* a, int a_fmt, const void* wfrag, const float* bias, void* out, int out_fmt, int batch,
                                int H, int W, int act, spei_stream_t stream) {
    SPEI_REQUIRE(a && wfrag && out, "spei_conv32_ws16: null pointer");
    SPEI_REQUIRE((fmt == SPEI_BF16 || fmt == SPEI_F16) && (a_fmt == SPEI_F32 || a_fmt == fmt) && (out_fmt == SPEI_F32 || out_fmt == fmt),
                 "spei_conv32_ws16: fmt=%d a_fmt=%d out_fmt=%d", fmt, a_fmt, out_fmt);
    SPEI_REQUIRE(batch >= 1 && H > 0 && W > 0 && (int64_t)batch * cdiv(H, TH) * cdiv(W, TW) < (1ll << 30) && (int64_t)H * W < (1ll << 26),
                 "spei_conv32_ws16: batch=%d of %dx%d", batch, H, W);
    SPEI_REQUIRE(act == SPEI_ACT_NONE || act == SPEI_ACT_RELU, "spei_conv32_ws16: act=%d", act);
    SPEI_REQUIRE(((uintptr_t)a | (uintptr_t)wfrag | (uintptr_t)out) % 16 == 0 && a != out, "spei_conv32_ws16: 16-byte alignment, out must not be the input");
    WsParams p;
    p.a = a; p.wfrag = wfrag; p.bias = bias; p.out = out; p.H = H; p.W = W; p.batch = batch; p.act = act;
    p.tiles_x = cdiv(W, TW); p.tiles_y = cdiv(H, TH); p.total = batch * p.tiles_x * p.tiles_y;
    p.stamps = spei_stamp_buffer();
    hipStream_t st = (hipStream_t)stream;
    const int a16 = a_fmt != SPEI_F32, o16 = out_fmt != SPEI_F32;
    const int rc = fmt == SPEI_F16 ? launch_ws<_Float16>(p, a16, o16, st) : launch_ws<__bf16>(p, a16, o16, st);
    if (rc) return rc;
    SPEI_CHECK_LAUNCH("spei_conv32_ws16");
    return 0;
}

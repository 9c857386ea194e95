// Slab-resident convolution / linear GEMM on the gfx950 16-bit matrix pipe (v_mfma_f32_32x32x16_bf16 / _f16).
//
// igemm_bf16.hip re-stages the A operand from global memory for every (tap, K-chunk) and pays a load ->
// ds_write -> barrier round trip per 16..64 MFMAs: at bf16 MFMA rates that loop is latency-bound.  Here each
// 256-thread workgroup
//   1. stages ONCE the input pixels its output tile needs — TH x 32 output pixels plus the (k-1) halo, ALL input
//      channels — into LDS as bf16 (fp32 in HBM -> cvt -> LDS, or bf16 in HBM -> LDS unchanged; optional second
//      slab with the bf16 residual for the split "bf16x3" mode), pixel pitch 2*K+16 bytes (odd multiple of 16 B =>
//      conflict-free ds_read_b128);
//   2. runs a BARRIER-FREE main loop: per (tap, 32-wide K group) every wave reads its A fragments from the slab at
//      a tap-shifted address and takes its B fragments straight from global memory, where the weights were
//      pre-packed on the host in MFMA fragment order ([n-tile][tap][k-step][lane][8] bf16: one fully coalesced
//      1 KiB load per fragment), prefetched four groups ahead in a register ring; all output-channel chunks are
//      looped inside the workgroup so the slab is staged once;
//   3. applies the same epilogue as the other GEMM kernels (bias, ReLU/GELU, per-pixel scale, residual) and writes
//      fp32 (residual streams) or bf16 (tensors that only feed the next GEMM / the attention kernel).
// HBM sees every input pixel once per output tile (+halo) instead of once per tap.
// Call sites replaced: as igemm_f32.hip (convolutions with stride 1/2 and linears; transposed convs stay there).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

struct SlabParams {
    const void* a0;
    const void* a1;
    const void* wh;      // fragment-ordered, 16-bit elements of the kernel's LP type
    const void* wl;
    const float* bias;
    void* out;
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
    int cp_shift;            // log2(16-byte chunks per pixel) or -1 when that count is not a power of two <= 256
    int iw_magic;            // ceil(2^20 / IW): pix / IW == (pix * iw_magic) >> 20 for pix < 2048
    int goff_bytes;          // bytes reserved for the group offset table (multiple of 16)
    int n_chunks;            // 32*WN*TN-column chunks looped inside the workgroup
    // Tap list form (ntap > 0, used for the parity classes of a stride-2 transposed conv): the taps are the (dy, dx) >= 0
    // offsets below instead of the ks x ks square, and output pixel (y, x) of the tile grid is written to pixel
    // (y * o_mul + o_row_add, x * o_mul + o_col_add) of a map that is Wfull pixels wide.
    int ntap, tap_dy[4], tap_dx[4];
    int o_mul, o_row_add, o_col_add, Wfull;
    int64_t planes;          // > 0: write output channels 0..2 as three fp32 NCHW planes of this many pixels (last conv)
    int ln;                  // 1: LayerNorm(256) without affine is applied to every input row while it is staged
                             //    (fp32 input, K == 256: one wave-instruction loads exactly one token row)
    int stagger, stagger_slots;   // first-round start delay (units of 64 cycles) per co-resident workgroup slot
    // Fused "apply" staging (template flag FA; 2-D stride-1 convs on fp32 maps): the input map is NOT a0 itself but
    //     x'[p][c] = a0[p][c] + x1[p][c] * (s[c] + g1[y][c] + g2[x][c])          (the tail of the previous ResBlock, model/block.py:136-140)
    // computed while the slab is staged; the pixels of the workgroup's own output tile are also written to fa_out (fp32, row stride K):
    // the residual stream of the next block.  x1: LP, row stride K; s [K], g1 [Hin][K], g2 [Win][K].
    const void* fa_x1;
    const float *fa_s, *fa_g1, *fa_g2;
    float* fa_out;
    // Batched launch (gridDim.y = number of maps, e.g. the 7 encoder passes of a frame through the same layer): map b of the input / the
    // output starts a0_bs / out_bs BYTES after map b - 1.  Same tiles, same arithmetic as one launch per map (bit-identical); what
    // changes is how many workgroups a launch has (450 -> 3150 at H/4: several resident rounds instead of half of one) and how many
    // launches a frame needs.  The residual map (res_bs) is batched like the output; a1, rowscale and the fused apply staging are not.
    long long a0_bs, out_bs, res_bs;
    int batch;
    long long* stamps;       // tuning build: phase stamps (tools/stamp_phases.py conv), else NULL
    int dbg;                 // ablation switches for tools/ablate_slab.py (0 in production): 1 no staging loads, 8 weight stream from one hot group,
                             // 4 no epilogue stores
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

constexpr int G = 2;        // k-steps (of 16) per group
// groups of B fragments in flight, per tile configuration (registers: RING * G * TN * 4 per lane)
#ifndef SPEI_RING_N32
#define SPEI_RING_N32 4
#endif
#ifndef SPEI_RING_N64
#define SPEI_RING_N64 4
#endif
#ifndef SPEI_RING_N128
#define SPEI_RING_N128 4
#endif
template <int WM, int WN>
constexpr int ring_depth() { return WN == 4 ? SPEI_RING_N128 : (WN == 2 ? SPEI_RING_N64 : SPEI_RING_N32); }

// one 16-byte global chunk -> LDS: 4 fp32 -> 4 LP (8 B) [+ residual], or 8 LP unchanged (16 B)
template <typename TA, bool SPLIT, typename LP>
struct Stage {                                         // TA == LP
    static constexpr int CH = 8;
    typedef u32x4 reg_t;
    static __device__ __forceinline__ reg_t zero() { return reg_t{0u, 0u, 0u, 0u}; }
    static __device__ __forceinline__ void put(unsigned char* slab, int, int off, const reg_t v) {
        *reinterpret_cast<reg_t*>(slab + off) = v;
    }
};
template <bool SPLIT, typename LP>
struct Stage<float, SPLIT, LP> {
    static constexpr int CH = 4;                       // channels per chunk
    typedef f32x4 reg_t;
    typedef typename lpv<LP>::x4 lp4;
    static __device__ __forceinline__ reg_t zero() { return reg_t{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void put(unsigned char* slab, int slab_bytes, int off, const reg_t v) {
        const lp4 h = to_lp4<LP>(v);
        *reinterpret_cast<lp4*>(slab + off) = h;
        if (SPLIT) {
            lp4 l;
            l[0] = (LP)(v[0] - (float)h[0]); l[1] = (LP)(v[1] - (float)h[1]);
            l[2] = (LP)(v[2] - (float)h[2]); l[3] = (LP)(v[3] - (float)h[3]);
            *reinterpret_cast<lp4*>(slab + slab_bytes + off) = l;
        }
    }
};

// LP: 16-bit operand type (__bf16 or _Float16); TA / TO: float or LP
template <int WM, int WN, int TM, int TN, bool SPLIT, typename TA, typename TO, typename LP, bool FA = false>
__global__ __launch_bounds__(64 * WM * WN, (WM == 4 && WN == 1 && TM <= 2 && !SPLIT) ? (FA ? 3 : 4) : 1) void conv_slab_kernel(const SlabParams p) {
    static_assert(!FA || (sizeof(TA) == 4 && !SPLIT), "the fused apply staging reads an fp32 map");
    constexpr int NT = 64 * WM * WN;                   // threads: 4 or 8 waves
    constexpr int RING = ring_depth<WM, WN>();
    typedef Stage<TA, SPLIT, LP> ST;
    typedef typename lpv<LP>::x8 lp8;
    typedef typename lpv<LP>::x4 lp4;
    typedef typename ST::reg_t sreg_t;
    constexpr int CH = ST::CH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int pitch = 2 * p.K + 16;
    unsigned char* slab = smem;
    int* goff = reinterpret_cast<int*>(smem + (SPLIT ? 2 : 1) * p.slab_bytes);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fk = lane >> 5;
    if (p.stagger > 0) {
        // Workgroups that start together stay in lockstep (all stage, then all compute, then all store: HBM idles while the
        // matrix pipe runs and vice versa).  Delay the k-th co-resident workgroup of a CU by k * stagger in the first round;
        // later workgroups start whenever one finishes and inherit the offset.
        const int slot = blockIdx.x / 256;
        if (slot < p.stagger_slots)
            for (int i = 0; i < slot * p.stagger; i += 64) __builtin_amdgcn_s_sleep(64);
    }
    const int tile_y = blockIdx.x / p.tiles_x, tile_x = blockIdx.x - tile_y * p.tiles_x;
    const int oy0 = tile_y * p.TH, ox0 = tile_x * p.TW;
    const int ks16 = p.K / 16;
    const int kg_per_tap = ks16 / G;
    const int T = p.ntap ? p.ntap : p.ks * p.ks;
    const int ngroups = T * kg_per_tap;
    // every workgroup walks the (tap, K group) sequence from a different start: otherwise all CUs stream the same weight
    // fragment from the same L2 channel at the same time
    const int rot = (blockIdx.x * 7) % ngroups;
    auto rg = [&](int g) __attribute__((always_inline)) { const int r = g + rot; return r >= ngroups ? r - ngroups : r; };

    // ---- group -> slab byte offset table ------------------------------------------------------------
    for (int g = tid; g < ngroups; g += NT) {
        const int t = g / kg_per_tap, kg = g - t * kg_per_tap;
        const int ty = p.ntap ? p.tap_dy[t] : t / p.ks, tx = p.ntap ? p.tap_dx[t] : t - (t / p.ks) * p.ks;
        goff[g] = (ty * p.IW + tx) * pitch + kg * (G * 32);
    }

    // ---- stage the input slab ----------------------------------------------------------------------------------
    SPEI_STAMP(p.stamps, 0);
    {
        const TA* a0 = reinterpret_cast<const TA*>(static_cast<const unsigned char*>(p.a0) + (size_t)blockIdx.y * p.a0_bs);
        const TA* a1 = static_cast<const TA*>(p.a1);
        const int cpn = p.K / CH;                         // 16-byte chunks per pixel
        const int npix = p.IH * p.IW;
        const int gy0 = oy0 * p.stride - p.pad, gx0 = ox0 * p.stride - p.pad;
        constexpr int U = 8;
        if (p.cp_shift >= 0) {
            // fixed channel chunk per thread, pixels walked with a constant step: no divisions in the loop
            const int c = (tid & (cpn - 1)) * CH;
            const int step = NT >> p.cp_shift;
            const int step_y = (int)(((unsigned)step * (unsigned)p.iw_magic) >> 20), step_x = step - step_y * p.IW;
            int pix = tid >> p.cp_shift;
            int iy = (int)(((unsigned)pix * (unsigned)p.iw_magic) >> 20), ix = pix - iy * p.IW;
            const TA* src = (c < p.k0) ? a0 + c : a1 + (c - p.k0);
            const int ld = (c < p.k0) ? p.lda0 : p.lda1;
            const int lofs = c * 2;
            typedef typename lpv<LP>::x4 lp4s;
            f32x4 fa_sv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (FA) fa_sv = *reinterpret_cast<const f32x4*>(p.fa_s + c);
            while (pix < npix) {
                // branch-free loads: clamp the coordinates to a valid pixel, select zero afterwards
                sreg_t v[U];
                int off[U];
                bool ok[U];
                // fused apply: x1 rides with the map loads; the gate rows / columns (L2-resident, just written by the gate kernel) go
                // through a 4-deep register pipeline — all eight in flight at once cost 64 registers and an occupancy step
                constexpr int GP = 4;
                lp4s fx1[FA ? U : 1];
                f32x4 fg1[FA ? GP : 1], fg2[FA ? GP : 1];
                unsigned fpix[FA ? U : 1], fyx[FA ? U : 1];
                bool fin[FA ? U : 1];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int gy = gy0 + iy, gx = gx0 + ix;
                    off[u] = pix < npix ? pix * pitch + lofs : -1;
                    ok[u] = (gy >= 0) & (gy < p.Hin) & (gx >= 0) & (gx < p.Win) & !(p.dbg & 1);
                    const int cy = min(max(gy, 0), p.Hin - 1), cx = min(max(gx, 0), p.Win - 1);
                    v[u] = *reinterpret_cast<const sreg_t*>(src + ((size_t)cy * p.Win + cx) * ld);
                    if constexpr (FA) {        // everything the fused apply needs, issued with the map loads (one round trip)
                        fpix[u] = (unsigned)(cy * p.Win + cx);
                        fyx[u] = ((unsigned)cy << 16) | (unsigned)cx;
                        fx1[u] = *reinterpret_cast<const lp4s*>(static_cast<const LP*>(p.fa_x1) + (size_t)fpix[u] * p.K + c);
                        if (u < GP) {
                            fg1[u] = *reinterpret_cast<const f32x4*>(p.fa_g1 + (size_t)cy * p.K + c);
                            fg2[u] = *reinterpret_cast<const f32x4*>(p.fa_g2 + (size_t)cx * p.K + c);
                        }
                        fin[u] = ok[u] & (pix < npix) & (iy >= p.pad) & (iy < p.pad + p.TH) & (ix >= p.pad) & (ix < p.pad + p.TW);
                    }
                    pix += step;
                    ix += step_x;
                    iy += step_y;
                    if (ix >= p.IW) { ix -= p.IW; ++iy; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    sreg_t val = ok[u] ? v[u] : ST::zero();
                    if constexpr (FA) {
                        if (ok[u]) {
                            // x3 = se(x1) + (cw(x1) + hc(x1));  return x3 + x   (block.py:136-140): resblock_apply_kernel's expression
                            f32x4 b;
#pragma unroll
                            for (int e = 0; e < 4; ++e) b[e] = (float)fx1[u][e];
                            val = (b * fa_sv + (b * fg1[u % GP] + b * fg2[u % GP])) + val;
                            if (fin[u]) *reinterpret_cast<f32x4*>(p.fa_out + (size_t)fpix[u] * p.K + c) = val;
                        }
                        if (u + GP < U) {          // refill the slot just consumed
                            fg1[u % GP] = *reinterpret_cast<const f32x4*>(p.fa_g1 + (size_t)(fyx[u + GP] >> 16) * p.K + c);
                            fg2[u % GP] = *reinterpret_cast<const f32x4*>(p.fa_g2 + (size_t)(fyx[u + GP] & 0xffffu) * p.K + c);
                        }
                    }
                    if constexpr (sizeof(TA) == 4) {
                        if (p.ln) {      // the 64 lanes of this wave hold the 256 channels of one row
                            const float mean = wave_sum((val[0] + val[1]) + (val[2] + val[3])) * (1.0f / 256.0f);
                            val -= mean;
                            const float var = wave_sum((val[0] * val[0] + val[1] * val[1]) + (val[2] * val[2] + val[3] * val[3])) * (1.0f / 256.0f);
                            val *= 1.0f / sqrtf(var + 1e-5f);
                        }
                    }
                    if (off[u] >= 0) ST::put(slab, p.slab_bytes, off[u], val);
                }
            }
        } else {
            const int total = npix * cpn;
            for (int idx = tid; idx < total; idx += NT) {
                const int pix = idx / cpn, c = (idx - pix * cpn) * CH;
                const int iy = pix / p.IW, ix = pix - iy * p.IW;
                const int gy = gy0 + iy, gx = gx0 + ix;
                sreg_t v = ST::zero();
                if (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win && !(p.dbg & 1)) {
                    const size_t gp = (size_t)gy * p.Win + gx;
                    v = (c < p.k0) ? *reinterpret_cast<const sreg_t*>(a0 + gp * p.lda0 + c)
                                   : *reinterpret_cast<const sreg_t*>(a1 + gp * p.lda1 + (c - p.k0));
                }
                ST::put(slab, p.slab_bytes, pix * pitch + c * 2, v);
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
    const size_t frag_per_nt = (size_t)T * ks16 * 64 * 8;        // bf16 elements per 32-column n-tile
    const LP* bptr[TN];
    const LP* bptr_lo[TN];
    auto set_chunk = [&](int nc) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nt = (nc * WN + wn) * TN + j;
            bptr[j] = static_cast<const LP*>(p.wh) + nt * frag_per_nt + lane * 8;
            bptr_lo[j] = SPLIT ? static_cast<const LP*>(p.wl) + nt * frag_per_nt + lane * 8 : nullptr;
        }
    };
    set_chunk(0);

    lp8 bring[RING][G][TN];
    lp8 bring_lo[SPLIT ? RING : 1][G][TN];
    auto load_b = [&](int slot, int g) {
#pragma unroll
        for (int s = 0; s < G; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const size_t o = (p.dbg & 8) ? (size_t)s * 512 : (size_t)(g * G + s) * 512;     // dbg 8: every group re-reads group 0 (L1-hot)
                bring[slot][s][j] = *reinterpret_cast<const lp8*>(bptr[j] + o);
                if (SPLIT) bring_lo[slot][s][j] = *reinterpret_cast<const lp8*>(bptr_lo[j] + o);
            }
    };

    f32x16 acc[TM][TN];
    // The ring is always filled / refilled UNCONDITIONALLY (indices clamped to the last group: a few redundant loads at the
    // tail).  With conditional loads the compiler cannot count outstanding loads across the branches and falls back to
    // s_waitcnt vmcnt(0..1) in front of every MFMA group (prefetch distance 1 instead of RING), and the accumulators
    // bounce between AGPRs and VGPRs at every merge point (64 v_accvgpr moves per RING groups; seen in the ISA).
#pragma unroll
    for (int d = 0; d < RING; ++d) load_b(d, rg(min(d, ngroups - 1)));
    const int nfull = (ngroups / RING) * RING;

    SPEI_STAMP(p.stamps, 1);
    __syncthreads();     // slab + offset table visible; the only barrier of the kernel
    SPEI_STAMP(p.stamps, 2);

    // output row bookkeeping: in both tilings (TH x 32 pixels of a map, TH x 1 tokens) consecutive tile pixels are
    // consecutive output rows, so row(m-tile i, q) = mbase[i] + q with q = (r&3) + 8*(r>>2) + 4*fk in [0,32)
    int mbase[TM], qlim[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pt0 = (wm * TM + i) * 32;
        if (p.tw_shift == 5) {
            const int oy = oy0 + (pt0 >> 5);
            mbase[i] = (oy * p.o_mul + p.o_row_add) * p.Wfull + ox0 * p.o_mul + p.o_col_add;
            qlim[i] = oy < p.Hout ? p.Wout - ox0 : 0;
        } else {
            mbase[i] = oy0 + pt0;
            qlim[i] = p.Hout - oy0 - pt0;
        }
    }

    for (int nc = 0; nc < p.n_chunks; ++nc) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        float biasv[TN];                     // fetched ahead of the main loop: the epilogue must not start with a load latency
#pragma unroll
        for (int j = 0; j < TN; ++j) biasv[j] = p.bias ? p.bias[((nc * WN + wn) * TN + j) * 32 + fr] : 0.f;

        // A fragments are software-pipelined one k-step ahead (two register sets) and sched_group_barrier pins
        // "LDS reads of step s+1, then MFMAs of step s": left alone the compiler waits on each ds_read right before its MFMA.
        constexpr int NRD = TM * (SPLIT ? 2 : 1), NMF = TM * TN * (SPLIT ? 3 : 1);
        lp8 an[TM], anl[SPLIT ? TM : 1];
        auto load_a = [&](int off) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                an[i] = *reinterpret_cast<const lp8*>(slab + abase[i] + off);
                if (SPLIT) anl[i] = *reinterpret_cast<const lp8*>(slab + p.slab_bytes + abase[i] + off);
            }
        };
        int go = goff[rg(0)];
        load_a(go);
        {
            // slab offset of the next group: looked up TWO groups ahead.  One group ahead the ds_read_b32 -> address -> A
            // fragment reads chain sat in front of every other MFMA group with an s_waitcnt lgkmcnt(0) (seen in the ISA).
            int go_n1 = goff[rg(min(1, ngroups - 1))];
            auto group = [&](int d, int g) __attribute__((always_inline)) {
                const int go_next = go_n1;
                go_n1 = goff[rg(min(g + 2, ngroups - 1))];
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    lp8 av[TM], avl[SPLIT ? TM : 1];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        av[i] = an[i];
                        if (SPLIT) avl[i] = anl[i];
                    }
#if !defined(SPEI_SLAB_EXP) || !(SPEI_SLAB_EXP & 1)          // tuning experiment 1: no A-fragment re-reads (same MFMAs)
                    load_a(s + 1 < G ? go + (s + 1) * 32 : go_next);     // (the very last prefetch is a harmless re-read)
#endif
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (SPLIT) {
                                acc[i][j] = mfma16(avl[i], bring[d][s][j], acc[i][j]);
                                acc[i][j] = mfma16(av[i], bring_lo[SPLIT ? d : 0][s][j], acc[i][j]);
                            }
                            acc[i][j] = mfma16(av[i], bring[d][s][j], acc[i][j]);
                        }
                    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
                }
                go = go_next;
            };
            for (int g0 = 0; g0 < nfull; g0 += RING) {               // branch-free body
#pragma unroll
                for (int d = 0; d < RING; ++d) {
                    group(d, g0 + d);
#if !defined(SPEI_SLAB_EXP) || !(SPEI_SLAB_EXP & 2)          // tuning experiment 2: no weight-ring refills (same MFMAs)
                    load_b(d, rg(min(g0 + d + RING, ngroups - 1)));
#endif
                }
            }
            // remainder (ngroups % RING groups): their fragments sit in ring slots 0.. from the last refill / the prologue
#pragma unroll
            for (int d = 0; d < RING - 1; ++d)
                if (nfull + d < ngroups) group(d, nfull + d);
        }
        if (nc == 0) SPEI_STAMP(p.stamps, 3);
        // start the next chunk's weight stream before the epilogue's stores
        const int nbase = (nc * WN + wn) * TN;
        if (nc + 1 < p.n_chunks) {
            set_chunk(nc + 1);
#pragma unroll
            for (int d = 0; d < RING; ++d) load_b(d, rg(min(d, ngroups - 1)));
        }

        // ---- epilogue ----------------------------------------------------------------------------------------
        // Row / address math is invariant across the chunk loop: launder one input so the compiler does not hoist
        // ~200 registers of addresses out of the loop and starve the main loop (measured: 255 VGPRs, serialised MFMAs).
        // The accumulator holds one column (n) per lane and 16 rows in registers: stored directly that is 4 bytes per
        // lane per instruction.  A 4x4 transpose inside each lane quad (DPP, no LDS) gives every lane 4 consecutive
        // channels of one output row: residual loads and stores are 16 B per lane, full 128-byte row segments,
        // 4 instead of 16 store instructions per 32 x 32 tile.
        int fk_e = fk;
        asm volatile("" : "+v"(fk_e));
        TO* outp = reinterpret_cast<TO*>(static_cast<unsigned char*>(p.out) + (size_t)blockIdx.y * p.out_bs);
        const float* resp = p.res ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.res) + (size_t)blockIdx.y * p.res_bs) : nullptr;
        int fr_e = fr;
        asm volatile("" : "+v"(fr_e));
        const int et = fr_e & 3, ecol = (fr_e >> 2) * 4;
        // The activation and the optional epilogue inputs are uniform over the launch: one branch here instead of ~100 scalar
        // branches per wave inside the loops (the plain ReLU / identity epilogues of the ResBlock convs are the hot ones).
        auto epilogue = [&](auto act_c, auto plain_c) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_c)::value;          // -1: decided per value at run time
            constexpr bool PLAIN = decltype(plain_c)::value;     // no rowscale, no residual, no plane output
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int ncol0 = (nbase + j) * 32;
                    const float bias = biasv[j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float a[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[i][j][4 * k + e] + bias;
                            if constexpr (ACT == SPEI_ACT_RELU) v = fmaxf(v, 0.f);
                            else if constexpr (ACT == SPEI_ACT_GELU) v = gelu_erf(v);
                            else if constexpr (ACT < 0) {
                                if (p.act == SPEI_ACT_RELU) v = fmaxf(v, 0.f);
                                else if (p.act == SPEI_ACT_GELU) v = gelu_erf(v);
                            }
                            a[e] = v;
                        }
                        quad_transpose4(a[0], a[1], a[2], a[3], et);
                        const int q = 8 * k + 4 * fk_e + et;         // accumulator row (r&3) + 8*(r>>2) + 4*fk with r = 4k + et
                        if (q < qlim[i] && !(p.dbg & 4)) {
                            const unsigned m = (unsigned)(mbase[i] + q * p.o_mul);
                            f32x4 v = f32x4{a[0], a[1], a[2], a[3]};
                            if constexpr (!PLAIN) {
                                if (p.rowscale) { const float rs = p.rowscale[m]; v *= rs; }
                                if (p.res) v += *reinterpret_cast<const f32x4*>(resp + m * (unsigned)p.ldr + ncol0 + ecol);
                                if (p.planes) {                       // 32 -> 3 channel tail: only the lanes holding channels 0..3 store
                                    if (ncol0 + ecol == 0) {
                                        float* po = reinterpret_cast<float*>(outp) + m;
                                        po[0] = v[0]; po[p.planes] = v[1]; po[2 * p.planes] = v[2];
                                    }
                                    continue;
                                }
                            }
                            if constexpr (sizeof(TO) == 4) {
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outp) + m * (unsigned)p.ldo + ncol0 + ecol) = v;
                            } else {
                                *reinterpret_cast<lp4*>(reinterpret_cast<LP*>(outp) + m * (unsigned)p.ldo + ncol0 + ecol) = to_lp4<LP>(v);
                            }
                        }
                    }
                }
            }
        };
        typedef std::integral_constant<int, SPEI_ACT_RELU> act_relu;
        typedef std::integral_constant<int, SPEI_ACT_NONE> act_none;
        typedef std::integral_constant<int, -1> act_any;
        if (!p.rowscale && !p.res && !p.planes && p.act == SPEI_ACT_RELU) epilogue(act_relu{}, std::true_type{});
        else if (!p.rowscale && !p.res && !p.planes && p.act == SPEI_ACT_NONE) epilogue(act_none{}, std::true_type{});
        else epilogue(act_any{}, std::false_type{});
    }   // n chunks
    SPEI_STAMP(p.stamps, 4);
}

template <int WM, int WN, int TM, int TN, bool SPLIT, typename TA, typename TO, typename LP, bool FA = false>
int launch(const SlabParams& p, size_t lds, hipStream_t s) {
    ensure_dyn_lds<&conv_slab_kernel<WM, WN, TM, TN, SPLIT, TA, TO, LP, FA>>(lds);
    SlabParams q = p;
    q.n_chunks = p.N / (WN * TN * 32);
    static const int dbg = spei_knob("SPEI_SLAB_DBG", 0);
    q.dbg = dbg;
    q.stamps = spei_stamp_buffer();
    static const int stagger = spei_knob("SPEI_SLAB_STAGGER", 0), slots = spei_knob("SPEI_SLAB_SLOTS", 3);
    q.stagger = stagger; q.stagger_slots = slots;
    dim3 grid(p.tiles_x * cdiv(p.Hout, p.TH), p.batch > 0 ? p.batch : 1);
    hipLaunchKernelGGL((conv_slab_kernel<WM, WN, TM, TN, SPLIT, TA, TO, LP, FA>), grid, dim3(64 * WM * WN), lds, s, q);
    SPEI_CHECK_LAUNCH("spei_conv_slab16");
    return 0;
}

template <bool SPLIT, typename TA, typename TO, typename LP, bool FA = false>
int dispatch(SlabParams& p, hipStream_t s) {
    const int pitch = 2 * p.K + 16;
    const int nparts = SPLIT ? 2 : 1;
    const int T = p.ntap ? p.ntap : p.ks * p.ks;
    const int ngroups = T * (p.K / 32);
    const bool linear = (p.Wout == 1 && p.ks == 1 && !p.ntap);
    int ext_y = p.ks, ext_x = p.ks;                           // slab extent beyond the (strided) tile
    if (p.ntap) {
        ext_y = ext_x = 1;
        for (int t = 0; t < p.ntap; ++t) {
            ext_y = p.tap_dy[t] + 1 > ext_y ? p.tap_dy[t] + 1 : ext_y;
            ext_x = p.tap_dx[t] + 1 > ext_x ? p.tap_dx[t] + 1 : ext_x;
        }
    }
    constexpr int CH = sizeof(TA) == 4 ? 4 : 8;
    // tile rows x 32 pixels (2-D maps) or rows x 1 (token lists); pick the largest tile whose slab(s) fit ~96 KB
    auto setup = [&](int mtile) {
        if (linear) { p.TH = mtile; p.TW = 1; }
        else { p.TH = mtile / 32; p.TW = 32; }
        p.IH = (p.TH - 1) * p.stride + ext_y;
        p.IW = (p.TW - 1) * p.stride + ext_x;
        p.slab_bytes = ((p.IH * p.IW * pitch + 15) / 16) * 16;
        p.tiles_x = cdiv(p.Wout, p.TW);
        p.tw_shift = linear ? 0 : 5;
        const int cpn = p.K / CH;
        p.cp_shift = -1;
        for (int sft = 0; sft <= 8; ++sft) if ((1 << sft) == cpn) p.cp_shift = sft;
        p.iw_magic = ((1 << 20) + p.IW - 1) / p.IW;
        p.goff_bytes = ((ngroups * (int)sizeof(int) + 15) / 16) * 16;
        return (size_t)nparts * p.slab_bytes + (size_t)p.goff_bytes;
    };
    static const int budget_kb = spei_knob("SPEI_SLAB_BUDGET", 96);
    const size_t budget = (size_t)budget_kb * 1024;
    const size_t hard = 160 * 1024 - 512;
    size_t lds;
    // Experimental tile shapes (tuning build only: SPEI_SLAB_CFG_N32 / _N64 / _N128 pick one by number; 0 = shipping choice).
    // TM x TN output tiles per wave decide how many MFMAs each operand fragment feeds: an A fragment (LDS) feeds TN, a B
    // fragment (1 KiB from L2) feeds TM.
#ifdef SPEI_TUNING
    if (!SPLIT) {
        static const int c32 = spei_knob("SPEI_SLAB_CFG_N32", 0), c64 = spei_knob("SPEI_SLAB_CFG_N64", 0), c128 = spei_knob("SPEI_SLAB_CFG_N128", 0);
        const bool ok2d = !linear;
        if (p.N == 32 && c32 && ok2d) {
            if (c32 == 1) { lds = setup(256); if (lds <= hard && p.IH * p.IW < 2048) return launch<2, 1, 4, 1, SPLIT, TA, TO, LP, FA>(p, lds, s); }
            if (c32 == 2) { lds = setup(512); if (lds <= hard && p.IH * p.IW < 2048) return launch<4, 1, 4, 1, SPLIT, TA, TO, LP, FA>(p, lds, s); }
            if (c32 == 3) { lds = setup(128); if (lds <= hard && p.IH * p.IW < 2048) return launch<1, 1, 4, 1, SPLIT, TA, TO, LP, FA>(p, lds, s); }
        }
        if (p.N % 64 == 0 && p.N % 128 != 0 && c64 && ok2d) {
            if (c64 == 1) { lds = setup(128); if (lds <= hard && p.IH * p.IW < 2048) return launch<1, 2, 4, 1, SPLIT, TA, TO, LP, FA>(p, lds, s); }
            if (c64 == 2) { lds = setup(256); if (lds <= hard && p.IH * p.IW < 2048) return launch<2, 1, 4, 2, SPLIT, TA, TO, LP, FA>(p, lds, s); }
            if (c64 == 3) { lds = setup(256); if (lds <= hard && p.IH * p.IW < 2048) return launch<2, 2, 4, 1, SPLIT, TA, TO, LP, FA>(p, lds, s); }
            if (c64 == 4) { lds = setup(128); if (lds <= hard && p.IH * p.IW < 2048) return launch<1, 1, 4, 2, SPLIT, TA, TO, LP, FA>(p, lds, s); }
        }
        if (p.N % 128 == 0 && c128 && ok2d) {
            if (c128 == 1) { lds = setup(128); if (lds <= hard && p.IH * p.IW < 2048) return launch<1, 2, 4, 2, SPLIT, TA, TO, LP, FA>(p, lds, s); }
            if (c128 == 2) { lds = setup(256); if (lds <= hard && p.IH * p.IW < 2048) return launch<2, 2, 4, 2, SPLIT, TA, TO, LP, FA>(p, lds, s); }
            if (c128 == 3) { lds = setup(128); if (lds <= hard && p.IH * p.IW < 2048) return launch<1, 4, 4, 1, SPLIT, TA, TO, LP, FA>(p, lds, s); }
            if (c128 == 4) { lds = setup(256); if (lds <= hard && p.IH * p.IW < 2048) return launch<2, 4, 4, 1, SPLIT, TA, TO, LP, FA>(p, lds, s); }
        }
    }
#endif
    if (p.N % 128 == 0) {
        lds = setup(128);
        if (lds <= budget && p.IH * p.IW < 2048) return launch<1, 4, 4, 1, SPLIT, TA, TO, LP, FA>(p, lds, s);
        lds = setup(64);
        if (lds <= hard && p.IH * p.IW < 2048) return launch<1, 4, 2, 1, SPLIT, TA, TO, LP, FA>(p, lds, s);
    } else if (p.N % 64 == 0) {
        lds = setup(128);
        if (lds <= hard && p.IH * p.IW < 2048) return launch<2, 2, 2, 1, SPLIT, TA, TO, LP, FA>(p, lds, s);
    } else {
        lds = setup(256);
        if (lds <= budget && p.IH * p.IW < 2048) return launch<4, 1, 2, 1, SPLIT, TA, TO, LP, FA>(p, lds, s);
        lds = setup(128);
        if (lds <= hard && p.IH * p.IW < 2048) return launch<4, 1, 1, 1, SPLIT, TA, TO, LP, FA>(p, lds, s);
    }
    spei_set_error("spei_conv_slab16: slab of %zu bytes (K=%d, k=%d, stride %d) does not fit LDS", lds, p.K, p.ks, p.stride);
    return -1;
}

}  // namespace

// a_fmt / out_fmt: SPEI_F32 or the call's 16-bit format `fmt` (SPEI_BF16 / SPEI_F16)
template <typename LP>
static int dispatch_io(SlabParams& p, bool split, bool a16, bool o16, hipStream_t st) {
    if (split) return dispatch<true, float, float, LP>(p, st);
    if (p.fa_x1) return o16 ? dispatch<false, float, LP, LP, true>(p, st) : dispatch<false, float, float, LP, true>(p, st);
    if (a16) return o16 ? dispatch<false, LP, LP, LP>(p, st) : dispatch<false, LP, float, LP>(p, st);
    return o16 ? dispatch<false, float, LP, LP>(p, st) : dispatch<false, float, float, LP>(p, st);
}
static int dispatch_fmt(SlabParams& p, int fmt, bool split, bool a16, bool o16, hipStream_t st) {
    if (fmt == SPEI_F16) {
        if (split) { spei_set_error("the split (bf16x3) form is built for bf16 only"); return -1; }
        return dispatch_io<_Float16>(p, false, a16, o16, st);
    }
    return dispatch_io<__bf16>(p, split, a16, o16, st);
}
#define SPEI_REQUIRE_FMT(who, fmt, a_fmt, out_fmt)                                                                      \
    SPEI_REQUIRE(((fmt) == SPEI_BF16 || (fmt) == SPEI_F16) && ((a_fmt) == SPEI_F32 || (a_fmt) == (fmt)) &&              \
                 ((out_fmt) == SPEI_F32 || (out_fmt) == (fmt)), who ": fmt=%d a_fmt=%d out_fmt=%d", fmt, a_fmt, out_fmt)

static int conv_slab16_run(int fmt, const void* a0, int lda0, int k0, const void* a1, int lda1, int k1, int a_fmt,
                           const void* wfrag_hi, const void* wfrag_lo, const float* bias, void* out, int ldo,
                           int out_fmt, const float* residual, int ldr, const float* rowscale, int Hin, int Win,
                           int Hout, int Wout, int N, int ksize, int stride, int pad, int act, int ln_input,
                           const void* fa_x1, const float* fa_s, const float* fa_g1, const float* fa_g2, float* fa_out,
                           spei_stream_t stream, int batch = 1) {
    SPEI_REQUIRE(a0 && wfrag_hi && out, "spei_conv_slab16: null pointer");
    SPEI_REQUIRE(batch >= 1 && batch <= 65535, "spei_conv_slab16_batched: batch=%d", batch);
    SPEI_REQUIRE(batch == 1 || (!a1 && !rowscale && !fa_x1 && !ln_input && lda0 == k0 && ldo == N && (!residual || ldr == N) && Wout > 1),
                 "spei_conv_slab16_batched: dense maps, one input, no row scale / fused staging");
    if (fa_x1) {
        SPEI_REQUIRE(fa_s && fa_g1 && fa_g2 && fa_out, "spei_conv_slab16_fa: null pointer");
        SPEI_REQUIRE(a_fmt == SPEI_F32 && !wfrag_lo && !a1 && k1 == 0 && lda0 == k0 && stride == 1 && ksize > 1 && !ln_input && Wout > 1,
                     "spei_conv_slab16_fa: needs a dense fp32 map, one input, stride 1, a 2-D kernel, single-product arithmetic");
        SPEI_REQUIRE(k0 == 32 || k0 == 64 || k0 == 128 || k0 == 256, "spei_conv_slab16_fa: K=%d (32 / 64 / 128 / 256)", k0);
        SPEI_REQUIRE(((uintptr_t)fa_x1 | (uintptr_t)fa_s | (uintptr_t)fa_g1 | (uintptr_t)fa_g2 | (uintptr_t)fa_out) % 16 == 0 && fa_out != a0,
                     "spei_conv_slab16_fa: 16-byte alignment, and fa_out must not be the input map (neighbouring tiles read its halo)");
    }
    SPEI_REQUIRE_FMT("spei_conv_slab16", fmt, a_fmt, out_fmt);
    const bool a16 = a_fmt != SPEI_F32, o16 = out_fmt != SPEI_F32;
    SPEI_REQUIRE(k0 > 0 && k0 % 32 == 0 && k1 >= 0 && k1 % 32 == 0, "spei_conv_slab16: k0=%d k1=%d must be multiples of 32", k0, k1);
    SPEI_REQUIRE(k1 == 0 || a1, "spei_conv_slab16: a1 missing");
    SPEI_REQUIRE(N > 0 && N % 32 == 0, "spei_conv_slab16: N=%d must be a multiple of 32", N);
    const int ael = a16 ? 8 : 4;
    SPEI_REQUIRE(lda0 % ael == 0 && (k1 == 0 || lda1 % ael == 0) && ldo >= N && lda0 >= k0 && (k1 == 0 || lda1 >= k1),
                 "spei_conv_slab16: bad row strides");
    SPEI_REQUIRE(ksize == 1 || ksize == 3 || ksize == 5, "spei_conv_slab16: ksize=%d", ksize);
    SPEI_REQUIRE(stride == 1 || stride == 2, "spei_conv_slab16: stride=%d", stride);
    SPEI_REQUIRE(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "spei_conv_slab16: empty map");
    SPEI_REQUIRE((int64_t)Hout * Wout * (int64_t)(ldo > ldr ? ldo : ldr) < (1ll << 32) && (int64_t)Hin * Win < (1ll << 30),
                 "spei_conv_slab16: map too large for 32-bit element offsets");
    SPEI_REQUIRE(Hout == (Hin + 2 * pad - ksize) / stride + 1 && Wout == (Win + 2 * pad - ksize) / stride + 1,
                 "spei_conv_slab16: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", Hout, Wout, Hin, Win, ksize, stride, pad);
    SPEI_REQUIRE(((uintptr_t)a0 % 16 == 0) && ((uintptr_t)wfrag_hi % 16 == 0) && (!a1 || (uintptr_t)a1 % 16 == 0) &&
                 (!wfrag_lo || (uintptr_t)wfrag_lo % 16 == 0), "spei_conv_slab16: operands must be 16-byte aligned");
    SPEI_REQUIRE(!(wfrag_lo && (a16 || o16 || fmt != SPEI_BF16)), "spei_conv_slab16: the split (bf16x3) mode is bf16 with fp32 activations");
    SPEI_REQUIRE(ldo % 4 == 0 && (!residual || ldr % 4 == 0) && ((uintptr_t)out % 16 == 0) && (!residual || (uintptr_t)residual % 16 == 0),
                 "spei_conv_slab16: out / residual must be 16-byte aligned with row strides that are multiples of 4");
    SlabParams p = {};
    p.a0 = a0; p.a1 = a1; p.wh = wfrag_hi; p.wl = wfrag_lo; p.bias = bias; p.out = out;
    p.res = residual; p.rowscale = rowscale;
    p.lda0 = lda0; p.lda1 = lda1; p.k0 = k0; p.k1 = k1; p.ldo = ldo; p.ldr = ldr;
    p.N = N; p.K = k0 + k1;
    p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout;
    p.ks = ksize; p.stride = stride; p.pad = pad; p.act = act;
    SPEI_REQUIRE(!ln_input || (!a16 && k0 == 256 && k1 == 0 && ksize == 1), "spei_conv_slab16: ln_input needs a 256-wide fp32 linear");
    p.ln = ln_input;
    p.ntap = 0; p.o_mul = 1; p.o_row_add = 0; p.o_col_add = 0; p.Wfull = Wout; p.planes = 0;
    p.fa_x1 = fa_x1; p.fa_s = fa_s; p.fa_g1 = fa_g1; p.fa_g2 = fa_g2; p.fa_out = fa_out;
    p.batch = batch;
    p.a0_bs = (long long)Hin * Win * lda0 * (a16 ? 2 : 4);
    p.out_bs = (long long)Hout * Wout * ldo * (o16 ? 2 : 4);
    p.res_bs = (long long)Hout * Wout * ldr * 4;
    return dispatch_fmt(p, fmt, wfrag_lo != nullptr, a16, o16, (hipStream_t)stream);
}

extern "C" int spei_conv_slab16(int fmt, const void* a0, int lda0, int k0, const void* a1, int lda1, int k1, int a_fmt,
                                const void* wfrag_hi, const void* wfrag_lo, const float* bias, void* out, int ldo,
                                int out_fmt, const float* residual, int ldr, const float* rowscale, int Hin, int Win,
                                int Hout, int Wout, int N, int ksize, int stride, int pad, int act, int ln_input,
                                spei_stream_t stream) {
    return conv_slab16_run(fmt, a0, lda0, k0, a1, lda1, k1, a_fmt, wfrag_hi, wfrag_lo, bias, out, ldo, out_fmt, residual, ldr, rowscale, Hin,
                           Win, Hout, Wout, N, ksize, stride, pad, act, ln_input, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int spei_conv_slab16_batched(int fmt, const void* a0, int k0, int a_fmt, const void* wfrag, const void* wfrag_lo, const float* bias,
                                        void* out, int out_fmt, const float* residual, int batch, int Hin, int Win, int Hout, int Wout, int N,
                                        int ksize, int stride, int pad, int act, spei_stream_t stream) {
    return conv_slab16_run(fmt, a0, k0, k0, nullptr, 0, 0, a_fmt, wfrag, wfrag_lo, bias, out, N, out_fmt, residual, residual ? N : 0, nullptr, Hin,
                           Win, Hout, Wout, N, ksize, stride, pad, act, 0, nullptr, nullptr, nullptr, nullptr, nullptr, stream, batch);
}

extern "C" int spei_conv_slab16_fa(int fmt, const float* x, int K, const void* x1, const float* s, const float* g1, const float* g2,
                                   float* x_out, const void* wfrag, const float* bias, void* out, int ldo, int out_fmt, int H, int W, int N,
                                   int ksize, int act, spei_stream_t stream) {
    return conv_slab16_run(fmt, x, K, K, nullptr, 0, 0, SPEI_F32, wfrag, nullptr, bias, out, ldo, out_fmt, nullptr, 0, nullptr, H, W, H, W, N,
                           ksize, 1, ksize / 2, act, 0, x1, s, g1, g2, x_out, stream);
}

// the four output-parity classes of a 3x3 stride-2 transposed conv as tap-list launches of the slab kernel; wl: the bf16 low halves of the
// class weights (split = bf16x3 arithmetic), else NULL
static int convt2_launch(int fmt, const void* a0, int lda0, int k0, bool a16, const void* const (&wf)[2][2], const void* const (*wl)[2],
                         const float* bias, void* out, int ldo, bool o16, int Hin, int Win, int N, int act, hipStream_t stream) {
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            SlabParams p = {};
            p.a0 = a0; p.a1 = nullptr; p.wh = wf[py][px]; p.wl = wl ? wl[py][px] : nullptr; p.bias = bias; p.out = out;
            p.res = nullptr; p.rowscale = nullptr;
            p.lda0 = lda0; p.lda1 = 0; p.k0 = k0; p.k1 = 0; p.ldo = ldo; p.ldr = 0;
            p.N = N; p.K = k0;
            p.Hin = Hin; p.Win = Win; p.Hout = Hin; p.Wout = Win;          // the tile grid is the input grid
            p.ks = 1; p.stride = 1; p.pad = 0; p.act = act; p.ln = 0;
            // out[2y+py][2x+px] = sum over (ky, kx) with in[y+dy][x+dx]: parity 0 -> k = 1 (d = 0); parity 1 -> k = 0 (d = 1), k = 2 (d = 0)
            p.ntap = 0;
            for (int iy = 0; iy < (py ? 2 : 1); ++iy)
                for (int ix = 0; ix < (px ? 2 : 1); ++ix) {
                    p.tap_dy[p.ntap] = py ? 1 - iy : 0;
                    p.tap_dx[p.ntap] = px ? 1 - ix : 0;
                    ++p.ntap;
                }
            p.o_mul = 2; p.o_row_add = py; p.o_col_add = px; p.Wfull = 2 * Win; p.planes = 0;
            const int rc = dispatch_fmt(p, fmt, wl != nullptr, a16, o16, stream);
            if (rc) return rc;
        }
    return 0;
}

extern "C" int spei_convt2_slab16x3(const float* a0, int lda0, int k0, const void* const* whi4, const void* const* wlo4, const float* bias,
                                    float* out, int ldo, int Hin, int Win, int N, int act, spei_stream_t stream) {
    SPEI_REQUIRE(a0 && whi4 && wlo4 && out, "spei_convt2_slab16x3: null pointer");
    for (int i = 0; i < 4; ++i)
        SPEI_REQUIRE(whi4[i] && wlo4[i] && ((uintptr_t)whi4[i] | (uintptr_t)wlo4[i]) % 16 == 0, "spei_convt2_slab16x3: class %d weights missing or unaligned", i);
    SPEI_REQUIRE(k0 > 0 && k0 % 32 == 0 && N > 0 && N % 32 == 0, "spei_convt2_slab16x3: K=%d N=%d must be multiples of 32", k0, N);
    SPEI_REQUIRE(lda0 % 4 == 0 && lda0 >= k0 && ldo >= N && ldo % 4 == 0, "spei_convt2_slab16x3: bad row strides");
    SPEI_REQUIRE(Hin > 0 && Win > 0 && (int64_t)Hin * Win * 4 * ldo < (1ll << 32), "spei_convt2_slab16x3: bad map size");
    SPEI_REQUIRE(((uintptr_t)a0 | (uintptr_t)out) % 16 == 0, "spei_convt2_slab16x3: operands must be 16-byte aligned");
    const void* const wf[2][2] = {{whi4[0], whi4[1]}, {whi4[2], whi4[3]}};
    const void* const wl[2][2] = {{wlo4[0], wlo4[1]}, {wlo4[2], wlo4[3]}};
    return convt2_launch(SPEI_BF16, a0, lda0, k0, false, wf, wl, bias, out, ldo, false, Hin, Win, N, act, (hipStream_t)stream);
}

extern "C" int spei_convt2_slab16(int fmt, const void* a0, int lda0, int k0, int a_fmt, const void* wfrag00, const void* wfrag01,
                                  const void* wfrag10, const void* wfrag11, const float* bias, void* out, int ldo, int out_fmt,
                                  int Hin, int Win, int N, int act, spei_stream_t stream) {
    SPEI_REQUIRE(a0 && wfrag00 && wfrag01 && wfrag10 && wfrag11 && out, "spei_convt2_slab16: null pointer");
    SPEI_REQUIRE_FMT("spei_convt2_slab16", fmt, a_fmt, out_fmt);
    const bool a16 = a_fmt != SPEI_F32, o16 = out_fmt != SPEI_F32;
    SPEI_REQUIRE(k0 > 0 && k0 % 32 == 0 && N > 0 && N % 32 == 0, "spei_convt2_slab16: K=%d N=%d must be multiples of 32", k0, N);
    SPEI_REQUIRE(lda0 % (a16 ? 8 : 4) == 0 && lda0 >= k0 && ldo >= N && ldo % 4 == 0, "spei_convt2_slab16: bad row strides");
    SPEI_REQUIRE(Hin > 0 && Win > 0 && (int64_t)Hin * Win * 4 * ldo < (1ll << 32), "spei_convt2_slab16: bad map size");
    SPEI_REQUIRE(((uintptr_t)a0 | (uintptr_t)out | (uintptr_t)wfrag00 | (uintptr_t)wfrag01 | (uintptr_t)wfrag10 | (uintptr_t)wfrag11) % 16 == 0,
                 "spei_convt2_slab16: operands must be 16-byte aligned");
    const void* const wf[2][2] = {{wfrag00, wfrag01}, {wfrag10, wfrag11}};
    return convt2_launch(fmt, a0, lda0, k0, a16, wf, nullptr, bias, out, ldo, o16, Hin, Win, N, act, (hipStream_t)stream);
}

extern "C" int spei_conv5_out_slab16(int fmt, const void* in, int ldi, int in_fmt, const void* wfrag, const float* bias32, float* out_chw,
                                     int H, int W, spei_stream_t stream) {
    SPEI_REQUIRE(in && wfrag && bias32 && out_chw && H > 0 && W > 0, "spei_conv5_out_slab16: bad arguments");
    SPEI_REQUIRE_FMT("spei_conv5_out_slab16", fmt, in_fmt, SPEI_F32);
    const bool a16 = in_fmt != SPEI_F32;
    SPEI_REQUIRE(ldi >= 32 && ldi % (a16 ? 8 : 4) == 0 && ((uintptr_t)in | (uintptr_t)wfrag | (uintptr_t)out_chw) % 16 == 0,
                 "spei_conv5_out_slab16: alignment / stride");
    SPEI_REQUIRE((int64_t)H * W < (1ll << 30), "spei_conv5_out_slab16: map too large");
    SlabParams p = {};
    p.a0 = in; p.a1 = nullptr; p.wh = wfrag; p.wl = nullptr; p.bias = bias32; p.out = out_chw;
    p.res = nullptr; p.rowscale = nullptr;
    p.lda0 = ldi; p.lda1 = 0; p.k0 = 32; p.k1 = 0; p.ldo = 1; p.ldr = 0;
    p.N = 32; p.K = 32;
    p.Hin = H; p.Win = W; p.Hout = H; p.Wout = W;
    p.ks = 5; p.stride = 1; p.pad = 2; p.act = SPEI_ACT_NONE; p.ln = 0;
    p.ntap = 0; p.o_mul = 1; p.o_row_add = 0; p.o_col_add = 0; p.Wfull = W; p.planes = (int64_t)H * W;
    return dispatch_fmt(p, fmt, false, a16, false, (hipStream_t)stream);
}

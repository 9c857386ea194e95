// K11, slab-resident: fused 3x3-patch correlation + arg-max on the bf16 matrix pipe
// (reference model/SearchTransfer.py:26-34, 61-69).
//
//   S[i] = max_j <P_ref[j], P_lr[i]> * inv_ref[j] * inv_lr[i]        arg[i] = lowest maximising j
//
// search_bf16.hip re-stages 128 x 64 operand tiles per (tap, K chunk): 18 global->LDS round trips and barriers per
// reference tile, i.e. every feature pixel crosses L2 -> LDS nine times.  Here the 3x3 unfold is pure LDS
// addressing:
//   * the query block (NI rows x 32 lr positions) and its 1-pixel halo sit in LDS for the whole kernel, all 128
//     channels (pixel pitch 2*128+16 B);
//   * the reference map is streamed as (2 WJ) x 32 position blocks + halo, KC channels at a time, double buffered
//     (pixel pitch 2*KC+16 B): one global->LDS stage per 9 taps x KC/16 k-steps = 72..144 MFMAs per wave;
//   * WJ x 2 waves: WJ along the reference positions (64 each), 2 along the query positions.  The bf16 configuration
//     runs WJ = 4 (512 threads, 256 reference positions per block, 155 KB of LDS): two waves per SIMD, so one wave's
//     arg-max fold / stage stores / barrier wait overlap the other's MFMAs (with 4 waves per CU the matrix pipe idled
//     during all of those);
//   * the MFMA (v_mfma_f32_32x32x16_bf16) puts reference positions on accumulator rows and query positions on
//     lanes, so the running (max, argmax) over j is per lane, in registers; R (13.3 GB at 720p) never exists.
// lo == NULL: single bf16 products; lo != NULL: split products al*bh + ah*bl + ah*bh (bf16x3, f32-grade scores).
// TOP2 (spei_corr_slab_top2_bf16): single bf16 products, but every query keeps its TWO best candidates; the exact
// winner is then decided by spei_corr_rescore (search.hip) on fp32 features with fp64 accumulation — f32-grade arg-max
// and S at the cost of the bf16 kernel.  The fold costs what the top-1 fold costs: a candidate is ONE float whose low
// five mantissa bits carry (31 - row of the 32-row group), so "insert into a sorted pair" is v_max_f32 + v_med3_f32.
#include <type_traits>
#include "common.h"

namespace {

constexpr int CMAXSPLIT = 8;
constexpr int RB_W = 32;                     // reference block: (2 WJ) rows x 32 positions
constexpr int SLAB_W = RB_W + 2;             // 34 pixels per slab row (1-pixel halo each side)

template <typename LP>       // LP: __bf16 or _Float16
struct CorrSlabParams {
    const LP* lrh;
    const LP* lrl;
    const LP* refh;
    const LP* refl;
    const float* inv_lr;
    const float* inv_ref;
    float* pval;
    int32_t* pidx;
    int Hl, Wl, Hr, Wr, Nl;
    int itiles_x;        // query tiles per row
    int rblocks_x, rblocks;   // reference blocks per row / total
    int rb_per_split;
};

// v_mfma_f32_16x16x32: same FLOP per cycle as the 32x32x16 form; on a kernel that runs at the chip's power limit (this one: 70 % of the
// matrix pipe busy at 54 % of peak) the 16x16 shape holds a higher clock (MI355X guide, DVFS give-back item 7: 1.12-1.15x FLOP/s).
// Fragment: lane l holds A[row l & 15][k = 8 (l >> 4) + j] / B[k][col l & 15]; result: col = l & 15, row = 4 (l >> 4) + reg.
__device__ __forceinline__ f32x4 mfma16x16(lpv<__bf16>::x8 a, lpv<__bf16>::x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16x16(lpv<_Float16>::x8 a, lpv<_Float16>::x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// top-2 of the union of two pairs, each sorted under `better` (value descending, then index ascending)
__device__ __forceinline__ void merge2(float& a1, int& i1, float& a2, int& i2, float b1, int j1, float b2, int j2) {
    if (better(b1, j1, a1, i1)) {
        const bool s = better(b2, j2, a1, i1);
        a2 = s ? b2 : a1; i2 = s ? j2 : i1;
        a1 = b1; i1 = j1;
    } else {
        const bool s = better(b1, j1, a2, i2);
        a2 = s ? b1 : a2; i2 = s ? j1 : i2;
    }
}

// NI: query tile rows (x 32 columns); KC: reference channels per stage; SPLIT: bf16x3; WJ: waves along the reference rows
// M16: the 16x16x32 MFMA shape (TOP2, NI = 4 only): the wave's 64 reference x 64 query positions as 4 x 4 tiles of 16 x 16
template <int NI, int KC, bool SPLIT, int WJ, bool TOP2, typename LP, bool M16 = false>
__global__ __launch_bounds__(128 * WJ) void corr_slab_kernel(const CorrSlabParams<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    static_assert(!(SPLIT && TOP2), "TOP2 is the single-product form");
    static_assert(!M16 || (TOP2 && NI == 4 && KC % 32 == 0), "the 16x16x32 form is built for the top-2 kernel");
    constexpr int C = 128;
    constexpr int NT = 128 * WJ;                       // threads
    constexpr int RB_H = 2 * WJ;                       // reference block rows
    constexpr int TN = NI / 2;                         // 32-column n-tiles per wave (2 waves along i)
    constexpr int NPART = SPLIT ? 2 : 1;
    constexpr int PITCH_L = 2 * C + 16;                // lr slab pixel pitch (bytes)
    constexpr int PITCH_R = 2 * KC + 16;               // ref stage pixel pitch
    constexpr int LPIX = (NI + 2) * SLAB_W, RPIX = (RB_H + 2) * SLAB_W;
    constexpr int L_BYTES = ((LPIX * PITCH_L + 15) / 16) * 16;
    constexpr int R_BYTES = ((RPIX * PITCH_R + 15) / 16) * 16;
    constexpr int NCH = C / KC;                        // channel chunks per reference block
    constexpr int RCH16 = KC / 8;                      // 16-byte pieces per pixel per stage
    constexpr int RLOADS = (RPIX * RCH16 + NT - 1) / NT; // staged 16-byte loads per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lslab = smem;                                   // [NPART][L_BYTES]
    unsigned char* rslab = smem + NPART * L_BYTES;                 // [2 buffers][NPART][R_BYTES]
    float* inv_s = reinterpret_cast<float*>(smem + NPART * L_BYTES + 2 * NPART * R_BYTES);   // [2][RB_H*32] 1/|patch| of the block

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;           // WJ waves along j (64 rows each), 2 along i
    const int fr = lane & 31, fk = lane >> 5;
    const int ity = blockIdx.x / p.itiles_x, itx = blockIdx.x - ity * p.itiles_x;
    const int iy0 = ity * NI, ix0 = itx * RB_W;
    const int rb0 = blockIdx.y * p.rb_per_split;
    const int rb1 = min(p.rblocks, rb0 + p.rb_per_split);
    const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};

    // ---- resident query slab ----------------------------------------------------------------------------
    for (int idx = tid; idx < LPIX * (C / 8); idx += NT) {
        const int pix = idx >> 4, c16 = idx & 15;      // C/8 == 16 pieces per pixel
        const int sy = pix / SLAB_W, sx = pix - sy * SLAB_W;
        const int gy = iy0 + sy - 1, gx = ix0 + sx - 1;
        const bool ok = (gy >= 0) & (gy < p.Hl) & (gx >= 0) & (gx < p.Wl);
        const size_t o = ((size_t)gy * p.Wl + gx) * C + c16 * 8;
        *reinterpret_cast<u32x4*>(lslab + pix * PITCH_L + c16 * 16) = ok ? *reinterpret_cast<const u32x4*>(p.lrh + o) : zero4;
        if (SPLIT) *reinterpret_cast<u32x4*>(lslab + L_BYTES + pix * PITCH_L + c16 * 16) = ok ? *reinterpret_cast<const u32x4*>(p.lrl + o) : zero4;
    }

    // ---- reference stage loader -------------------------------------------------------------------------
    u32x4 rh[RLOADS], rl[SPLIT ? RLOADS : 1];
    float riv = 0.f;
    auto load_stage = [&](int stage) __attribute__((always_inline)) {                 // stage = (rb - rb0) * NCH + chunk
        const int rb = rb0 + stage / NCH, ch = stage % NCH;
        const int rby = rb / p.rblocks_x, rbx = rb - rby * p.rblocks_x;
        const int jy0 = rby * RB_H - 1, jx0 = rbx * RB_W - 1;
        if (ch == 0 && tid < RB_H * RB_W) {              // the block's normalisers ride along with its first stage
            const int jy = rby * RB_H + (tid >> 5), jx = rbx * RB_W + (tid & 31);
            // positions outside the map get a NaN normaliser: their scores become NaN and lose every `>` of the fold
            // (TOP2: a 0 normaliser marks them; real normalisers are > 0)
            riv = (jy < p.Hr && jx < p.Wr) ? p.inv_ref[jy * p.Wr + jx] : (TOP2 ? 0.f : __builtin_nanf(""));
        }
#pragma unroll
        for (int u = 0; u < RLOADS; ++u) {
            const int idx = tid + u * NT;
            const int pix = idx / RCH16, c16 = idx - pix * RCH16;
            const int sy = pix / SLAB_W, sx = pix - sy * SLAB_W;
            const int gy = jy0 + sy, gx = jx0 + sx;
            const bool ok = (idx < RPIX * RCH16) & (gy >= 0) & (gy < p.Hr) & (gx >= 0) & (gx < p.Wr);
            const size_t o = ((size_t)gy * p.Wr + gx) * C + ch * KC + c16 * 8;
            rh[u] = ok ? *reinterpret_cast<const u32x4*>(p.refh + o) : zero4;
            if (SPLIT) rl[u] = ok ? *reinterpret_cast<const u32x4*>(p.refl + o) : zero4;
        }
    };
    auto store_stage = [&](int buf, int stage) __attribute__((always_inline)) {
        unsigned char* base = rslab + buf * NPART * R_BYTES;
        if (stage % NCH == 0 && tid < RB_H * RB_W) inv_s[((stage / NCH) & 1) * (RB_H * RB_W) + tid] = riv;
#pragma unroll
        for (int u = 0; u < RLOADS; ++u) {
            const int idx = tid + u * NT;
            if (idx < RPIX * RCH16) {
                const int pix = idx / RCH16, c16 = idx - pix * RCH16;
                *reinterpret_cast<u32x4*>(base + pix * PITCH_R + c16 * 16) = rh[u];
                if (SPLIT) *reinterpret_cast<u32x4*>(base + R_BYTES + pix * PITCH_R + c16 * 16) = rl[u];
            }
        }
    };

    // ---- per-lane operand bases ---------------------------------------------------------------------------
    int abase[2], bbase[TN];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                       // reference rows: block pixel (py, px) = (wm*2+i, fr)
        abase[i] = ((wm * 2 + i) * SLAB_W + fr) * PITCH_R + fk * 16;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {                      // query columns: tile pixel (qy, qx) = (wn*TN+j, fr)
        bbase[j] = ((wn * TN + j) * SLAB_W + fr) * PITCH_L + fk * 16;
    }
    float bestv[TN], il[TN], best2v[TN];
    int besti[TN], best2i[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        bestv[j] = best2v[j] = -INFINITY;
        besti[j] = best2i[j] = 0x7fffffff;
        const int qy = iy0 + wn * TN + j, qx = ix0 + fr;
        il[j] = (qy < p.Hl && qx < p.Wl) ? p.inv_lr[qy * p.Wl + qx] : 0.f;
    }

    if constexpr (M16) {
        // ---- 16x16x32 form -------------------------------------------------------------------------------------------------------
        // tile a (reference): block row wm*2 + (a >> 1), pixels 16 (a & 1) ..+16;  tile b (query): tile row wn*TN + (b >> 1), pixels
        // 16 (b & 1) ..+16.  Lane l: fragment row / column l & 15, k group g = l >> 4 (8 channels of the 32 of a k-step).
        const int f16r = lane & 15, g4 = lane >> 4;
        int abase16[4], bbase16[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) abase16[a] = ((wm * 2 + (a >> 1)) * SLAB_W + 16 * (a & 1) + f16r) * PITCH_R + g4 * 16;
#pragma unroll
        for (int b = 0; b < 4; ++b) bbase16[b] = ((wn * TN + (b >> 1)) * SLAB_W + 16 * (b & 1) + f16r) * PITCH_L + g4 * 16;
        float bv1[4], bv2[4];
        int bi1[4], bi2[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) { bv1[b] = bv2[b] = -INFINITY; bi1[b] = bi2[b] = 0x7fffffff; }
        f32x4 acc16[4][4];
        const int nstages = (rb1 - rb0) * NCH;
        if (nstages > 0) {
            load_stage(0);
            store_stage(0, 0);
        }
        __syncthreads();
        for (int st = 0; st < nstages; ++st) {
            const int buf = st & 1;
            const int ch = st % NCH;
            if (ch == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (st + 1 < nstages) load_stage(st + 1);
            const unsigned char* ra = rslab + buf * NPART * R_BYTES;
            const unsigned char* lb = lslab + ch * (KC * 2);
            constexpr int KS = KC / 32, NS = 9 * KS;               // k-steps of 32 channels
            lp8 fa[2][4], fb[2][4];
            auto load_frags = [&](int s_, int slot) __attribute__((always_inline)) {
                const int t = s_ / KS, ks = s_ - t * KS;
                const int ty = t / 3, tx = t - ty * 3;
                const int aoff = (ty * SLAB_W + tx) * PITCH_R + ks * 64;
                const int boff = (ty * SLAB_W + tx) * PITCH_L + ks * 64;
#pragma unroll
                for (int a = 0; a < 4; ++a) fa[slot][a] = *reinterpret_cast<const lp8*>(ra + abase16[a] + aoff);
#pragma unroll
                for (int b = 0; b < 4; ++b) fb[slot][b] = *reinterpret_cast<const lp8*>(lb + bbase16[b] + boff);
            };
            load_frags(0, 0);
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                const int cur = s_ & 1;
                if (s_ + 1 < NS) load_frags(s_ + 1, cur ^ 1);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc16[a][b] = mfma16x16(fa[cur][a], fb[cur][b], acc16[a][b]);
                if (s_ + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    // DS reads of step s+1 ...
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                     // ... then the MFMAs of step s
            }
            if (st + 1 < nstages) store_stage(buf ^ 1, st + 1);
            if (ch == NCH - 1) {
                // fold the block into the running top-2 of this lane's 4 query columns.  The lane owns 16 of the wave's 64 reference
                // rows: R = 16 a + 4 g + reg = 32 (block row) + pixel; key = score with (63 - R) in its low 6 mantissa bits.
                const int rb = rb0 + st / NCH;
                const int rby = rb / p.rblocks_x, rbx = rb - rby * p.rblocks_x;
                const float* pinv = inv_s + ((st / NCH) & 1) * (RB_H * RB_W) + (wm * 2) * 32 + 4 * g4;
                float irv[16];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(pinv + (a >> 1) * 32 + 16 * (a & 1));
                    irv[4 * a] = q[0]; irv[4 * a + 1] = q[1]; irv[4 * a + 2] = q[2]; irv[4 * a + 3] = q[3];
                }
                const bool edge = (rby * RB_H + RB_H > p.Hr) | (rbx * RB_W + RB_W > p.Wr);     // block-uniform
                float lk1[4], lk2[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) lk1[b] = lk2[b] = -INFINITY;
                auto fold = [&](auto EDGE) __attribute__((always_inline)) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned tag = (unsigned)(63 - (16 * a + r)) - 4u * (unsigned)g4;     // 63 - R
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const float v = acc16[a][b][r] * irv[4 * a + r];
                                float key = __uint_as_float((__float_as_uint(v) & ~63u) | tag);
                                if (decltype(EDGE)::value) key = irv[4 * a + r] == 0.f ? -INFINITY : key;
                                lk2[b] = __builtin_amdgcn_fmed3f(lk1[b], lk2[b], key);
                                lk1[b] = fmaxf(lk1[b], key);
                            }
                        }
                };
                if (edge) fold(std::true_type{}); else fold(std::false_type{});
                const int jj0 = (rby * RB_H + wm * 2) * p.Wr + rbx * RB_W;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    auto pos = [&](float k) __attribute__((always_inline)) {
                        const int R = 63 - (int)(__float_as_uint(k) & 63u);
                        return jj0 + (R >> 5) * p.Wr + (R & 31);
                    };
                    if (lk1[b] > -INFINITY)
                        merge2(bv1[b], bi1[b], bv2[b], bi2[b], lk1[b], pos(lk1[b]), lk2[b], lk2[b] > -INFINITY ? pos(lk2[b]) : 0x7fffffff);
                }
            }
            __syncthreads();
        }
        // combine the four lane groups of a query column, then the WJ waves along the reference rows
        float* rv = reinterpret_cast<float*>(smem);
        int* ri = reinterpret_cast<int*>(smem) + 2 * WJ * NI * 32;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                const float o1 = __shfl_xor(bv1[b], m, 64), o2 = __shfl_xor(bv2[b], m, 64);
                const int p1 = __shfl_xor(bi1[b], m, 64), p2 = __shfl_xor(bi2[b], m, 64);
                merge2(bv1[b], bi1[b], bv2[b], bi2[b], o1, p1, o2, p2);
            }
            if (g4 == 0) {
                const int col = (wn * TN + (b >> 1)) * 32 + 16 * (b & 1) + f16r;
                rv[(wm * NI * 32 + col) * 2] = bv1[b];
                rv[(wm * NI * 32 + col) * 2 + 1] = bv2[b];
                ri[(wm * NI * 32 + col) * 2] = bi1[b];
                ri[(wm * NI * 32 + col) * 2 + 1] = bi2[b];
            }
        }
        __syncthreads();
        if (tid < NI * 32) {
            const int qy = iy0 + (tid >> 5), qx = ix0 + (tid & 31);
            if (qy < p.Hl && qx < p.Wl) {
                float v1 = rv[tid * 2], v2 = rv[tid * 2 + 1];
                int x1 = ri[tid * 2], x2 = ri[tid * 2 + 1];
#pragma unroll
                for (int w = 1; w < WJ; ++w) {
                    const int o = (w * NI * 32 + tid) * 2;
                    merge2(v1, x1, v2, x2, rv[o], ri[o], rv[o + 1], ri[o + 1]);
                }
                const size_t i = ((size_t)blockIdx.y * p.Nl + (qy * p.Wl + qx)) * 2;
                p.pval[i] = v1; p.pval[i + 1] = v2;
                p.pidx[i] = x1; p.pidx[i + 1] = x2;
            }
        }
        return;
    }

    f32x16 acc[2][TN];
    const int nstages = (rb1 - rb0) * NCH;
    if (nstages > 0) {
        load_stage(0);
        store_stage(0, 0);
    }
    __syncthreads();

    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        const int ch = st % NCH;
        if (ch == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        if (st + 1 < nstages) load_stage(st + 1);
        const unsigned char* ra = rslab + buf * NPART * R_BYTES;
        const unsigned char* lb = lslab + ch * (KC * 2);
        // Software-pipelined over the 9 taps x KC/16 k-steps: the fragments of step s+1 are requested from LDS before
        // the MFMAs of step s issue (two register sets), and sched_group_barrier pins that interleave — left alone the
        // compiler reuses three fragment registers and waits on every ds_read right before its MFMA (29 % MFMA rate).
        constexpr int KS = KC / 16, NS = 9 * KS;
        constexpr int NRD = (2 + TN) * NPART, NMF = 2 * TN * (SPLIT ? 3 : 1);
        lp8 fa[2][2], fb[2][TN], fal[2][2], fbl[2][TN];
        auto load_frags = [&](int s, int slot) __attribute__((always_inline)) {
            const int t = s / KS, ks = s - t * KS;
            const int ty = t / 3, tx = t - ty * 3;
            const int aoff = (ty * SLAB_W + tx) * PITCH_R + ks * 32;
            const int boff = (ty * SLAB_W + tx) * PITCH_L + ks * 32;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[slot][i] = *reinterpret_cast<const lp8*>(ra + abase[i] + aoff);
                if (SPLIT) fal[slot][i] = *reinterpret_cast<const lp8*>(ra + R_BYTES + abase[i] + aoff);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fb[slot][j] = *reinterpret_cast<const lp8*>(lb + bbase[j] + boff);
                if (SPLIT) fbl[slot][j] = *reinterpret_cast<const lp8*>(lb + L_BYTES + bbase[j] + boff);
            }
        };
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cur = s & 1;
            if (s + 1 < NS) load_frags(s + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (SPLIT) {
                        acc[i][j] = mfma16(fal[cur][i], fb[cur][j], acc[i][j]);
                        acc[i][j] = mfma16(fa[cur][i], fbl[cur][j], acc[i][j]);
                    }
                    acc[i][j] = mfma16(fa[cur][i], fb[cur][j], acc[i][j]);
                }
            if (s + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);   // DS reads of step s+1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);                    // ... then the MFMAs of step s
        }
        if (st + 1 < nstages) store_stage(buf ^ 1, st + 1);
        if (ch == NCH - 1) {
            // Reference block finished: fold its rows into the running (max, argmax) of this lane's columns.  Branch-free and
            // latency-free: all 32 normalisers are fetched from LDS first (the fragment registers are dead here), the rows
            // of a block are visited in increasing j, so inside the block a strict `>` keeps the lowest maximising index;
            // only the block winner meets the running best under the full (value, lowest index) rule.  This fold used to
            // take a quarter of the kernel (in-kernel stamps): a dependent LDS read and two branches per row.
            const int rb = rb0 + st / NCH;
            const int rby = rb / p.rblocks_x, rbx = rb - rby * p.rblocks_x;
            const float* pinv = inv_s + ((st / NCH) & 1) * (RB_H * RB_W) + (wm * 2) * 32 + 4 * fk;
            float irv[32];
#pragma unroll
            for (int row = 0; row < 32; ++row) irv[row] = pinv[(row >> 4) * 32 + (row & 3) + 8 * ((row & 15) >> 2)];
            const int jj0 = (rby * RB_H + wm * 2) * p.Wr + rbx * RB_W + 4 * fk;
            if constexpr (TOP2) {
                // sorted pair (lk1 >= lk2) of keys per query column: key = score with (31 - row) in its low 5 mantissa bits
                // (2^-19 relative, far below the bf16 product error; among equal truncated scores the earlier row wins)
                const bool edge = (rby * RB_H + RB_H > p.Hr) | (rbx * RB_W + RB_W > p.Wr);     // block-uniform
                float lk1[TN], lk2[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) lk1[j] = lk2[j] = -INFINITY;
                auto fold = [&](auto EDGE) __attribute__((always_inline)) {
#pragma unroll
                    for (int row = 0; row < 32; ++row) {
                        const int i = row >> 4, r = row & 15;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float v = acc[i][j][r] * irv[row];
                            float key = __uint_as_float((__float_as_uint(v) & ~31u) | (unsigned)(31 - row));
                            if (decltype(EDGE)::value) key = irv[row] == 0.f ? -INFINITY : key;      // row outside the map
                            lk2[j] = __builtin_amdgcn_fmed3f(lk1[j], lk2[j], key);
                            lk1[j] = fmaxf(lk1[j], key);
                        }
                    }
                };
                if (edge) fold(std::true_type{}); else fold(std::false_type{});
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    auto pos = [&](float k) __attribute__((always_inline)) {
                        const int row = 31 - (int)(__float_as_uint(k) & 31u);
                        return jj0 + (row >> 4) * p.Wr + (row & 3) + 8 * ((row & 15) >> 2);
                    };
                    if (lk1[j] > -INFINITY)
                        merge2(bestv[j], besti[j], best2v[j], best2i[j], lk1[j], pos(lk1[j]), lk2[j],
                               lk2[j] > -INFINITY ? pos(lk2[j]) : 0x7fffffff);
                }
            } else {
            float lbv[TN];
            int lbi[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) { lbv[j] = -INFINITY; lbi[j] = 0; }
#pragma unroll
            for (int row = 0; row < 32; ++row) {
                const int i = row >> 4, r = row & 15;
                const int jj = jj0 + i * p.Wr + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // SPLIT: f32-grade scores in the reference's order of the two normalisations; otherwise the query's own
                    // normaliser (> 0, constant per lane column) is applied once at the end
                    const float v = SPLIT ? acc[i][j][r] * irv[row] * il[j] : acc[i][j][r] * irv[row];
                    const bool take = v > lbv[j];
                    lbv[j] = take ? v : lbv[j];
                    lbi[j] = take ? jj : lbi[j];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (lbv[j] > -INFINITY && better(lbv[j], lbi[j], bestv[j], besti[j])) { bestv[j] = lbv[j]; besti[j] = lbi[j]; }
            }
        }
        __syncthreads();
    }

    // ---- combine lane halves, then the two waves (wm) sharing the same query columns ---------------------------
    float* rv = reinterpret_cast<float*>(smem);            // [WJ][NI*32] ([..][2] with TOP2)
    int* ri = reinterpret_cast<int*>(smem) + (TOP2 ? 2 : 1) * WJ * NI * 32;
    if constexpr (TOP2) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float o1 = __shfl_xor(bestv[j], 32, 64), o2 = __shfl_xor(best2v[j], 32, 64);
            const int p1 = __shfl_xor(besti[j], 32, 64), p2 = __shfl_xor(best2i[j], 32, 64);
            merge2(bestv[j], besti[j], best2v[j], best2i[j], o1, p1, o2, p2);
            if (fk == 0) {
                const int col = (wn * TN + j) * 32 + fr;
                rv[(wm * NI * 32 + col) * 2] = bestv[j];
                rv[(wm * NI * 32 + col) * 2 + 1] = best2v[j];
                ri[(wm * NI * 32 + col) * 2] = besti[j];
                ri[(wm * NI * 32 + col) * 2 + 1] = best2i[j];
            }
        }
        __syncthreads();
        if (tid < NI * 32) {
            const int qy = iy0 + (tid >> 5), qx = ix0 + (tid & 31);
            if (qy < p.Hl && qx < p.Wl) {
                float v1 = rv[tid * 2], v2 = rv[tid * 2 + 1];
                int x1 = ri[tid * 2], x2 = ri[tid * 2 + 1];
#pragma unroll
                for (int w = 1; w < WJ; ++w) {
                    const int o = (w * NI * 32 + tid) * 2;
                    merge2(v1, x1, v2, x2, rv[o], ri[o], rv[o + 1], ri[o + 1]);
                }
                const size_t i = ((size_t)blockIdx.y * p.Nl + (qy * p.Wl + qx)) * 2;
                p.pval[i] = v1; p.pval[i + 1] = v2;
                p.pidx[i] = x1; p.pidx[i + 1] = x2;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float ov = __shfl_xor(bestv[j], 32, 64);
        const int oi = __shfl_xor(besti[j], 32, 64);
        if (better(ov, oi, bestv[j], besti[j])) { bestv[j] = ov; besti[j] = oi; }
        if (fk == 0) {
            const int col = (wn * TN + j) * 32 + fr;
            rv[wm * NI * 32 + col] = SPLIT ? bestv[j] : bestv[j] * il[j];
            ri[wm * NI * 32 + col] = besti[j];
        }
    }
    __syncthreads();
    if (tid < NI * 32) {
        const int qy = iy0 + (tid >> 5), qx = ix0 + (tid & 31);
        if (qy < p.Hl && qx < p.Wl) {
            float v = rv[tid];
            int ix = ri[tid];
#pragma unroll
            for (int w = 1; w < WJ; ++w)
                if (better(rv[w * NI * 32 + tid], ri[w * NI * 32 + tid], v, ix)) { v = rv[w * NI * 32 + tid]; ix = ri[w * NI * 32 + tid]; }
            const int i = qy * p.Wl + qx;
            p.pval[(size_t)blockIdx.y * p.Nl + i] = v;
            p.pidx[(size_t)blockIdx.y * p.Nl + i] = ix;
        }
    }
}

// merge the per-split pairs of the TOP2 form: (S, arg) = best candidate, (S2, arg2) = runner-up (arg2 = -1: none)
__global__ __launch_bounds__(256) void corr_top2_final_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
                                                              int splits, int Nl, float* __restrict__ S, int32_t* __restrict__ arg,
                                                              float* __restrict__ S2, int32_t* __restrict__ arg2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nl) return;
    float v1 = pval[(size_t)i * 2], v2 = pval[(size_t)i * 2 + 1];
    int x1 = pidx[(size_t)i * 2], x2 = pidx[(size_t)i * 2 + 1];
    for (int s = 1; s < splits; ++s) {
        const size_t o = ((size_t)s * Nl + i) * 2;
        merge2(v1, x1, v2, x2, pval[o], pidx[o], pval[o + 1], pidx[o + 1]);
    }
    S[i] = v1; S2[i] = v2;
    arg[i] = x1 == 0x7fffffff ? 0 : x1;
    arg2[i] = x2 == 0x7fffffff ? -1 : x2;
}

__global__ __launch_bounds__(256) void corr_slab_final_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
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

template <typename LP, int NI, int KC, bool SPLIT, int WJ, bool TOP2 = false, bool M16 = false>
void launch_corr(const CorrSlabParams<LP>& p, int itiles, int splits, hipStream_t st) {
    constexpr int C = 128;
    constexpr int RB_H = 2 * WJ;
    constexpr int NPART = SPLIT ? 2 : 1;
    constexpr int L_BYTES = ((((NI + 2) * SLAB_W) * (2 * C + 16) + 15) / 16) * 16;
    constexpr int R_BYTES = ((((RB_H + 2) * SLAB_W) * (2 * KC + 16) + 15) / 16) * 16;
    const size_t lds = (size_t)NPART * L_BYTES + (size_t)2 * NPART * R_BYTES + (size_t)2 * RB_H * RB_W * sizeof(float);
    ensure_dyn_lds<&corr_slab_kernel<NI, KC, SPLIT, WJ, TOP2, LP, M16>>(lds);
    hipLaunchKernelGGL((corr_slab_kernel<NI, KC, SPLIT, WJ, TOP2, LP, M16>), dim3(itiles, splits), dim3(128 * WJ), lds, st, p);
}

}  // namespace

// shared host side of the forms: 16-bit top-1 (bf16 / half), bf16x3 top-1, 16-bit top-2
template <typename LP>
static int corr_slab_run(const void* lr_hi, const void* lr_lo, const void* ref_hi, const void* ref_lo, const float* inv_lr,
                         const float* inv_ref, int Hl, int Wl, int Hr, int Wr, int C, float* S, int32_t* arg, float* S2,
                         int32_t* arg2, float* ws, spei_stream_t stream, const char* who) {
    const bool top2 = S2 != nullptr;
    SPEI_REQUIRE(lr_hi && ref_hi && inv_lr && inv_ref && S && arg && ws && (!top2 || arg2), "%s: null pointer", who);
    SPEI_REQUIRE((lr_lo == nullptr) == (ref_lo == nullptr), "%s: lo parts must both be given or both be NULL", who);
    SPEI_REQUIRE(C == 128, "%s: C=%d (128 built)", who, C);
    SPEI_REQUIRE(Hl > 0 && Wl > 0 && Hr > 0 && Wr > 0, "%s: empty map", who);
    SPEI_REQUIRE((int64_t)Hl * Wl < (1ll << 30) && (int64_t)Hr * Wr < (1ll << 30), "%s: map too large", who);
    SPEI_REQUIRE(((uintptr_t)lr_hi | (uintptr_t)ref_hi | (uintptr_t)lr_lo | (uintptr_t)ref_lo) % 16 == 0, "%s: 16-byte alignment required", who);
    const bool split = lr_lo != nullptr;
    SPEI_REQUIRE(!(split && (top2 || !__is_same(LP, __bf16))), "%s: the split (bf16x3) form is bf16, top-1", who);
    static const int ni_knob = spei_knob("SPEI_CORR_NI", 4);          // tuning build: 2 = one query tile per wave
    const int NI = split ? 2 : (top2 && ni_knob == 2 ? 2 : 4);
    static const int m16 = spei_knob("SPEI_CORR_M16", 1);           // tuning build: 0 = the 32x32x16 MFMA shape of rounds 1-2
    CorrSlabParams<LP> p;
    p.lrh = (const LP*)lr_hi; p.lrl = (const LP*)lr_lo; p.refh = (const LP*)ref_hi; p.refl = (const LP*)ref_lo;
    p.inv_lr = inv_lr; p.inv_ref = inv_ref;
    p.Hl = Hl; p.Wl = Wl; p.Hr = Hr; p.Wr = Wr; p.Nl = Hl * Wl;
    p.itiles_x = cdiv(Wl, RB_W);
    const int itiles = p.itiles_x * cdiv(Hl, NI);
    p.rblocks_x = cdiv(Wr, RB_W);
    const int rb_h = split ? 4 : 8;                  // 2 * WJ
    p.rblocks = p.rblocks_x * cdiv(Hr, rb_h);
    // split the reference blocks so that the grid fills 256 CUs (one workgroup per CU) with little tail
    int best_s = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= CMAXSPLIT && s <= p.rblocks; ++s) {
        const double units = (double)itiles * s;
        const double eff = units / (256.0 * (double)cdiv((int64_t)units, 256));
        if (eff > best_eff + 0.02) { best_eff = eff; best_s = s; }
    }
    p.rb_per_split = cdiv(p.rblocks, best_s);
    const int splits = cdiv(p.rblocks, p.rb_per_split);
    // workspace (spei_corr_ws_floats = 4 * CMAXSPLIT * Nl words): values, then indices; pairs in the top-2 form
    p.pval = ws;
    p.pidx = reinterpret_cast<int32_t*>(ws + (size_t)(top2 ? 2 : 1) * CMAXSPLIT * p.Nl);
    hipStream_t st = (hipStream_t)stream;
    if constexpr (__is_same(LP, __bf16)) {
        if (split) launch_corr<LP, 2, 32, true, 2>(p, itiles, splits, st);
    }
    if (!split) {
        if (top2 && NI == 2) launch_corr<LP, 2, 64, false, 4, true>(p, itiles, splits, st);
        else if (top2 && m16) launch_corr<LP, 4, 64, false, 4, true, true>(p, itiles, splits, st);
        else if (top2) launch_corr<LP, 4, 64, false, 4, true>(p, itiles, splits, st);
        else launch_corr<LP, 4, 64, false, 4>(p, itiles, splits, st);
    }
    if (top2)
        hipLaunchKernelGGL(corr_top2_final_kernel, dim3(cdiv(p.Nl, 256)), dim3(256), 0, st, p.pval, p.pidx, splits, p.Nl, S, arg, S2, arg2);
    else
        hipLaunchKernelGGL(corr_slab_final_kernel, dim3(cdiv(p.Nl, 256)), dim3(256), 0, st, p.pval, p.pidx, splits, p.Nl, S, arg);
    SPEI_CHECK_LAUNCH(who);
    return 0;
}

extern "C" int spei_corr_slab16(int fmt, const void* lr_hi, const void* lr_lo, const void* ref_hi, const void* ref_lo,
                                const float* inv_lr, const float* inv_ref, int Hl, int Wl, int Hr, int Wr, int C,
                                float* S, int32_t* arg, float* ws, spei_stream_t stream) {
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_corr_slab16: fmt=%d", fmt);
    if (fmt == SPEI_F16)
        return corr_slab_run<_Float16>(lr_hi, lr_lo, ref_hi, ref_lo, inv_lr, inv_ref, Hl, Wl, Hr, Wr, C, S, arg, nullptr, nullptr, ws, stream,
                                       "spei_corr_slab16");
    return corr_slab_run<__bf16>(lr_hi, lr_lo, ref_hi, ref_lo, inv_lr, inv_ref, Hl, Wl, Hr, Wr, C, S, arg, nullptr, nullptr, ws, stream,
                                 "spei_corr_slab16");
}

extern "C" int spei_corr_slab_top2_16(int fmt, const void* lr16, const void* ref16, const float* inv_lr, const float* inv_ref,
                                      int Hl, int Wl, int Hr, int Wr, int C, float* S, int32_t* arg, float* S2, int32_t* arg2,
                                      float* ws, spei_stream_t stream) {
    SPEI_REQUIRE(S2 && arg2, "spei_corr_slab_top2_16: null pointer");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_corr_slab_top2_16: fmt=%d", fmt);
    if (fmt == SPEI_F16)
        return corr_slab_run<_Float16>(lr16, nullptr, ref16, nullptr, inv_lr, inv_ref, Hl, Wl, Hr, Wr, C, S, arg, S2, arg2, ws, stream,
                                       "spei_corr_slab_top2_16");
    return corr_slab_run<__bf16>(lr16, nullptr, ref16, nullptr, inv_lr, inv_ref, Hl, Wl, Hr, Wr, C, S, arg, S2, arg2, ws, stream,
                                 "spei_corr_slab_top2_16");
}

// K11d, diagonal-sliding form of the fused 3x3-patch correlation + top-2 arg-max (reference model/SearchTransfer.py:26-34,
// 61-69), for a reference map at least as high as the query map: SearchTransfer's maps of one size and SelfTransfer's
// rotated landscape map (Hl x Wl queries against Wl x Hl references).
//
// The reference multiplies 9*C-long unfolded patches: 2 * (H W)^2 * 9 C flops.  But the score of query (y, x) against
// reference (y', x') is a sum over the three patch rows,
//
//   R[(y, x), (y', x')] = sum_{dy = -1..1} D[y + dy, y' + dy][x, x'],   D[a, b][x, x'] = sum_{dx, c} F[a, x + dx, c] G[b, x' + dx, c]
//
// and D[a, b] — one query row against one reference row, the horizontal taps only — is shared by the three (y, y') pairs on
// its diagonal.  A workgroup therefore walks DOWN four neighbouring diagonals delta = (y' - y) mod Hr: at query row a it computes
// the D tiles of rows (a, a + delta) once (K = 3 C on the matrix pipe; the 3 horizontal taps are LDS addressing, as in
// corr_slab16.hip) and the scores of row a - 1 are D_{a-2} + D_{a-1} + D_a, two fused multiply-adds per score.  A third of the
// reference's flops reach the matrix pipe; every score is still the full 9 C-term sum in fp32.
//
//   * workgroup = 8 waves = 4 neighbouring cyclic diagonals x 2 query halves of one 64-query x 64-reference position tile pair
//     (see the kernel for the wave roles); per wave the D tile being computed, the previous row's and the pending two-term sums
//     (3 x 32 accumulator registers) never leave registers;
//   * LDS: the query row (double buffered) and a ring of 5 reference rows (4 in use, 1 loading), 66 pixels x 128 channels each,
//     pixel pitch 2 * 128 + 16 B; per step ONE new query row tile and ONE new reference row tile cross L2 -> LDS (34 KB per
//     12.6 MFLOP: the four diagonals share them);
//   * cyclic diagonals (reference row = (a + delta) mod Hr) make every workgroup's walk the same length; where the reference
//     row wraps to 0 the chain of a diagonal is cut (the term across the wrap is a zero-padded patch row);
//   * every step ends in the top-2 fold of corr_slab16.hip over the wave's 64 reference positions, and lane pairs write the
//     two best keys of each query position for (delta, reference tile) — a float pair whose low six mantissa bits carry
//     63 - position.  corr_diag_reduce_kernel folds the Hr * ceil(Wr / 64) pairs of a query under the (masked score, lowest index)
//     rule; the output (S, arg, S2, arg2) goes to spei_corr_rescore like the slab kernel's.
#include "common.h"

namespace {

constexpr int DT = 64;                     // positions per tile side
constexpr int DM = 4;                      // diagonals (= waves) per workgroup
constexpr int DSLAB = DT + 2;              // pixels per staged row tile (1-pixel halo each side)
constexpr int DPITCH = 2 * 128 + 16;       // bytes per pixel in LDS
constexpr int DROWB = DSLAB * DPITCH;      // 17952 bytes per row tile
constexpr int DRING = DM + 1;
constexpr int DINV = 8;                    // ring of normaliser rows (a row's normalisers outlive its features by two steps)
constexpr int DPIECES = DSLAB * 16;        // 16-byte pieces per row tile
constexpr int DRSPLIT = 8;                 // splits of the reduce kernel

template <typename LP>
struct CorrDiagParams {
    const LP* lr;
    const LP* ref;
    const float* inv_ref;
    float* part;                           // [Hr (delta)][ktiles][Hl * Wl] key pairs
    int Hl, Wl, Hr, Wr;                    // query map, reference map (Hr >= Hl)
    int qtiles, ktiles, ngroups, seg_len, nseg;
    int nwg, xcd;                          // workgroups that have work; xcd: consecutive logical workgroups share an XCD (and its L2)
    long long* stamps;                     // tuning build: per-workgroup phase-time sums (tools/stamp_corr_diag.py)
};

__device__ __forceinline__ float vmax(float a, float b) {          // v_max_f32 without the canonicalising v_max x, x, x of fmaxf
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Eight waves, two per SIMD, out of phase.  Wave w = (diagonal w & 3, query half w >> 2): 64 reference x 32 query positions, three tiles of 32 accumulator registers (the
// one being computed, the previous row's, the pending two-term sums) — 96 of the 256 registers a wave has at two waves per SIMD,
// so nothing lives in the accumulation file.  Waves w and w + 4 share a SIMD (a workgroup's waves go round the SIMDs in a fixed
// cyclic order); the upper four run their vector work (fold of the PREVIOUS row, sliding sums) BEFORE the MFMAs of a step and the
// lower four AFTER them, so on every SIMD one wave's 48 MFMAs leave their issue gaps to the other wave's ~200 vector instructions.
template <typename LP>
__global__ __launch_bounds__(512) void corr_diag_kernel(const CorrDiagParams<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    constexpr int C = 128, NT = 512;
    constexpr int NLD = (2 * DPIECES + NT - 1) / NT;                 // staged 16-byte pieces per thread and step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* fbuf = smem;                                      // [2][DROWB]
    unsigned char* gring = smem + 2 * DROWB;                         // [DRING][DROWB]
    float* inv_s = reinterpret_cast<float*>(smem + (2 + DRING) * DROWB);   // [DINV][DT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wave & 3, jq = wave >> 2;                         // diagonal inside the group, query half
    const bool lag = jq != 0;                                        // vector work of a row one step late, ahead of the MFMAs
    const int fr = lane & 31, fk = lane >> 5;
    // Workgroup b runs on XCD b % 8.  The 32 workgroups an XCD holds at a time should be NEIGHBOURING diagonal groups of one tile
    // pair: they read the same query row and, four steps apart, the same reference rows — one fetch per XCD instead of one per
    // workgroup (the grid is padded to a multiple of 8; logical ids past the last workgroup exit).
    int bid = p.xcd ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    if (bid >= p.nwg) return;
    // fastest to slowest: diagonal group, segment, query tile, reference tile — the tile pairs an XCD works through one after the other
    // keep their reference column (all Hr rows of one 64-pixel tile: 2.9 MB at 720p) in its L2
    const int group = bid % p.ngroups; bid /= p.ngroups;
    const int seg = bid % p.nseg; bid /= p.nseg;
    const int qxt = bid % p.qtiles;
    const int kxt = bid / p.qtiles;
    const int Hl = p.Hl, Wl = p.Wl, Hr = p.Hr, Wr = p.Wr;            // diagonals are cyclic in the reference height Hr >= Hl
    const int d0 = group * DM, delta = d0 + wd;
    const bool active = delta < Hr;
    const int a0 = seg * p.seg_len, a1 = min(Hl, a0 + p.seg_len);
    const int a_start = max(a0 - 1, 0), a_end = min(a1, Hl - 1);
    const int qx0 = qxt * DT, kx0 = kxt * DT;
    const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
    const float qnan = __builtin_nanf("");

    auto ref_row = [&](int rho) { return (rho + d0) % Hr; };

    for (int idx = tid; idx < (1 + DM) * DPIECES; idx += NT) {
        const int which = idx / DPIECES, rem = idx - which * DPIECES;
        const int pix = rem >> 4, c16 = rem & 15;
        const bool q = which == 0;
        const int rho = a_start + which - 1;
        const int row = q ? a_start : ref_row(rho);
        const int gx = (q ? qx0 : kx0) - 1 + pix;
        const LP* src = q ? p.lr : p.ref;
        const int Wm = q ? Wl : Wr;
        const bool ok = (gx >= 0) & (gx < Wm);
        unsigned char* dst = q ? fbuf : gring + (rho % DRING) * DROWB;
        *reinterpret_cast<u32x4*>(dst + pix * DPITCH + c16 * 16) =
            ok ? *reinterpret_cast<const u32x4*>(src + ((size_t)row * Wm + gx) * C + c16 * 8) : zero4;
    }
    for (int idx = tid; idx < DINV * DT; idx += NT) {                // every slot defined: a lagging wave's first fold reads one early
        const int k = idx >> 6, rho = a_start + k, x = kx0 + (idx & 63);
        inv_s[(rho % DINV) * DT + (idx & 63)] = (k < DM && x < Wr) ? p.inv_ref[(size_t)ref_row(rho) * Wr + x] : qnan;
    }
    __syncthreads();

    // per-thread staging pieces (the same every step: only the two rows change)
    int soff[NLD], sdst[NLD];
    bool sok[NLD], sq[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int idx = min(tid + u * NT, 2 * DPIECES - 1);          // the threads past the last piece repeat it (same bytes)
        sq[u] = idx < DPIECES;
        const int rem = sq[u] ? idx : idx - DPIECES;
        const int pix = rem >> 4, c16 = rem & 15;
        const int gx = (sq[u] ? qx0 : kx0) - 1 + pix;
        const int Wm = sq[u] ? Wl : Wr;
        sok[u] = (gx >= 0) & (gx < Wm);
        soff[u] = min(max(gx, 0), Wm - 1) * C + c16 * 8;
        sdst[u] = pix * DPITCH + c16 * 16;
    }
    const int abase = fr * DPITCH + fk * 16;
    const size_t pbase = ((size_t)delta * p.ktiles + kxt) * Hl;

    f32x16 Xa[2], Xb[2], P[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { Xa[i][r] = 0.f; Xb[i][r] = 0.f; P[i][r] = 0.f; }

    // scores of row aq = P + m X (X = D tile of row aq + 1), folded; then P <- m Pold + X
    auto post = [&](int aq, const f32x16 (&X)[2], const f32x16 (&Pold)[2], float m, bool store) __attribute__((always_inline)) {
        const float* pinv = inv_s + ((aq + wd) & (DINV - 1)) * DT + 4 * fk;
        float lk1 = -INFINITY, lk2 = -INFINITY;
#pragma unroll
        for (int g4 = 0; g4 < 8; ++g4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(pinv + (g4 >> 2) * 32 + (g4 & 3) * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 4 * g4 + e, i = row >> 4, r = row & 15;
                const float x = X[i][r];
                const float v = __builtin_fmaf(x, m, P[i][r]) * q[e];
                P[i][r] = __builtin_fmaf(Pold[i][r], m, x);
                const float key = __uint_as_float((__float_as_uint(v) & ~31u) | (unsigned)(31 - row));
                lk2 = __builtin_amdgcn_fmed3f(lk1, lk2, key);
                lk1 = vmax(lk1, key);
            }
        }
        auto retag = [&](float k) __attribute__((always_inline)) {
            const unsigned u = __float_as_uint(k);
            const int row = 31 - (int)(u & 31u);
            const int R = 2 * row - (row & 3) + 4 * fk;
            return k == -INFINITY ? k : __uint_as_float((u & ~63u) | (unsigned)(63 - R));
        };
        const float k1 = retag(lk1), k2 = retag(lk2);
        const float o1 = __shfl_xor(k1, 32, 64), o2 = __shfl_xor(k2, 32, 64);
        const float m1 = vmax(k1, o1), m2 = vmax(fminf(k1, o1), vmax(k2, o2));
        const int qx = qx0 + 32 * jq + fr;
        if (store && fk == 0 && qx < Wl) {
            float2 o; o.x = m1; o.y = m2;
            *reinterpret_cast<float2*>(p.part + 2 * ((pbase + aq) * Wl + qx)) = o;
        }
    };
    auto mfma_tile = [&](const unsigned char* ga, const unsigned char* fb, f32x16 (&X)[2]) __attribute__((always_inline)) {
        constexpr int NS = 24;
        lp8 fa[2][2], fq[2];
        auto load_frags = [&](int s, int slot) __attribute__((always_inline)) {
            const int off = (s >> 3) * DPITCH + (s & 7) * 32;
            fa[slot][0] = *reinterpret_cast<const lp8*>(ga + abase + off);
            fa[slot][1] = *reinterpret_cast<const lp8*>(ga + abase + 32 * DPITCH + off);
            fq[slot] = *reinterpret_cast<const lp8*>(fb + abase + jq * 32 * DPITCH + off);
        };
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = s & 1;
            if (s + 1 < NS) load_frags(s + 1, c ^ 1);
            X[0] = mfma16(fa[c][0], fq[c], s ? X[0] : z);
            X[1] = mfma16(fa[c][1], fq[c], s ? X[1] : z);
            if (s + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
    };

    auto cont = [&](int a, int b) { return ((a > a_start) & (b != 0)) ? 1.f : 0.f; };
    auto emits = [&](int a) { return active & (a - 1 >= a0) & (a > a_start); };
    int bcur = (a_start + delta) % Hr, bprev = 1;
    int grow_next = ref_row(a_start + DM);

#ifdef SPEI_TUNING
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_amdgcn_s_memrealtime();
#define DSTAMP(k) do { const long long t_ = __builtin_amdgcn_s_memrealtime(); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define DSTAMP(k) do { } while (0)
#endif
    auto step = [&](int a, f32x16 (&X)[2], f32x16 (&Xold)[2]) __attribute__((always_inline)) {
        const int n = a - a_start;
        const int cur = n & 1;
        const float mprev = cont(a - 1, bprev), mcur = cont(a, bcur);
        const int rho_new = a + DM;
        const int grow = grow_next, qrow = min(a + 1, Hl - 1);
        u32x4 st[NLD];
        const LP* qsrc = p.lr + (size_t)qrow * Wl * C;
        const LP* gsrc = p.ref + (size_t)grow * Wr * C;
#pragma unroll
        for (int u = 0; u < NLD; ++u) st[u] = *reinterpret_cast<const u32x4*>((sq[u] ? qsrc : gsrc) + soff[u]);
        float riv = p.inv_ref[(size_t)grow * Wr + min(kx0 + lane, Wr - 1)];
        riv = kx0 + lane < Wr ? riv : qnan;
        const unsigned char* ga = gring + ((a + wd) % DRING) * DROWB;
        const unsigned char* fb = fbuf + cur * DROWB;
        auto stores = [&]() __attribute__((always_inline)) {
            unsigned char* fdst = fbuf + (cur ^ 1) * DROWB;
            unsigned char* gdst = gring + (rho_new % DRING) * DROWB;
#pragma unroll
            for (int u = 0; u < NLD; ++u)
                *reinterpret_cast<u32x4*>((sq[u] ? fdst : gdst) + sdst[u]) = sok[u] ? st[u] : zero4;
            if (wave == 0) inv_s[(rho_new & (DINV - 1)) * DT + lane] = riv;
        };
        // the lagging half: scores of row a - 2 first (X = D tile of row a - 1 = Xold, its predecessor = the registers X still holds)
        DSTAMP(0);
        if (lag) post(a - 2, Xold, X, mprev, emits(a - 1));
        DSTAMP(1);
        mfma_tile(ga, fb, X);
        DSTAMP(2);
        if (!lag) post(a - 1, X, Xold, mcur, emits(a));
        DSTAMP(3);
        stores();
        bprev = bcur;
        bcur = bcur + 1 == Hr ? 0 : bcur + 1;
        grow_next = grow + 1 == Hr ? 0 : grow + 1;
        DSTAMP(4);
        __syncthreads();
        DSTAMP(5);
    };
    auto tail = [&](f32x16 (&X)[2], f32x16 (&Xold)[2]) __attribute__((always_inline)) {
        if (lag) post(a_end - 1, X, Xold, cont(a_end, bprev), emits(a_end));
        if (a_end < a1) post(a_end, P, P, 0.f, active);              // the map's last row: no successor, its scores are the pending sums
    };
    for (int a = a_start;;) {
        step(a, Xa, Xb);
        if (a == a_end) { tail(Xa, Xb); break; }
        ++a;
        step(a, Xb, Xa);
        if (a == a_end) { tail(Xb, Xa); break; }
        ++a;
    }
#ifdef SPEI_TUNING
    if (p.stamps && lane == 0) {                                     // [workgroup][wave][8]: six section sums + the step count
        long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int k = 0; k < 6; ++k) o[k] = tacc[k];
        o[6] = a_end - a_start + 1;
    }
#endif
#undef DSTAMP
}

__device__ __forceinline__ bool dbetter(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// fold the (delta, reference tile) pairs of every query position; blockIdx.y takes every DRSPLIT-th slot
__global__ __launch_bounds__(256) void corr_diag_reduce_kernel(const float* __restrict__ part, int Hl, int Wl, int Hr, int Wr, int ktiles,
                                                               float* __restrict__ pval, int32_t* __restrict__ pidx) {
    const int Nl = Hl * Wl;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nl) return;
    const int a = i / Wl;
    float v1 = -INFINITY, v2 = -INFINITY;
    int x1 = 0x7fffffff, x2 = 0x7fffffff;
    const int nslots = Hr * ktiles;
    for (int s = blockIdx.y; s < nslots; s += DRSPLIT) {
        const float2 k = *reinterpret_cast<const float2*>(part + 2 * ((size_t)s * Nl + i));
        const int delta = s / ktiles, kxt = s - delta * ktiles;
        int b = a + delta;
        b = b >= Hr ? b - Hr : b;
        const int base = b * Wr + kxt * DT;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float key = e ? k.y : k.x;
            if (key == -INFINITY) continue;
            const unsigned u = __float_as_uint(key);
            const float v = __uint_as_float(u & ~63u);
            const int idx = base + 63 - (int)(u & 63u);
            if (dbetter(v, idx, v1, x1)) { v2 = v1; x2 = x1; v1 = v; x1 = idx; }
            else if (dbetter(v, idx, v2, x2)) { v2 = v; x2 = idx; }
        }
    }
    const size_t o = ((size_t)blockIdx.y * Nl + i) * 2;
    pval[o] = v1; pval[o + 1] = v2;
    pidx[o] = x1; pidx[o + 1] = x2;
}

__global__ __launch_bounds__(256) void corr_diag_final_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx, int Nl,
                                                              float* __restrict__ S, int32_t* __restrict__ arg,
                                                              float* __restrict__ S2, int32_t* __restrict__ arg2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nl) return;
    float v1 = -INFINITY, v2 = -INFINITY;
    int x1 = 0x7fffffff, x2 = 0x7fffffff;
    for (int s = 0; s < DRSPLIT; ++s) {
        const size_t o = ((size_t)s * Nl + i) * 2;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = pval[o + e];
            const int idx = pidx[o + e];
            if (dbetter(v, idx, v1, x1)) { v2 = v1; x2 = x1; v1 = v; x1 = idx; }
            else if (dbetter(v, idx, v2, x2)) { v2 = v; x2 = idx; }
        }
    }
    S[i] = v1; S2[i] = v2;
    arg[i] = x1 == 0x7fffffff ? 0 : x1;
    arg2[i] = x2 == 0x7fffffff ? -1 : x2;
}

constexpr size_t DIAG_LDS = (size_t)(2 + DRING) * DROWB + (size_t)DINV * DT * sizeof(float);

template <typename LP>
int corr_diag_run(const void* lr16, const void* ref16, const float* inv_ref, int Hl, int Wl, int Hr, int Wr, float* S, int32_t* arg,
                  float* S2, int32_t* arg2, float* ws, hipStream_t st) {
    CorrDiagParams<LP> p;
    p.lr = (const LP*)lr16; p.ref = (const LP*)ref16; p.inv_ref = inv_ref;
    p.Hl = Hl; p.Wl = Wl; p.Hr = Hr; p.Wr = Wr;
    p.qtiles = cdiv(Wl, DT);
    p.ktiles = cdiv(Wr, DT);
    p.ngroups = cdiv(Hr, DM);
    const int Nl = Hl * Wl;
    // cut the walk down the query rows into segments (each pays two extra rows) until the grid fills 256 CUs with little tail
    const int64_t cols = (int64_t)p.qtiles * p.ktiles * p.ngroups;
    int best_n = 1;
    double best_eff = 0.0;
    for (int n = 1; n <= 8 && n <= Hl; ++n) {
        const int len = cdiv(Hl, n);
        const int64_t units = cols * cdiv(Hl, len);
        const double eff = ((double)len / (len + (n > 1 ? 2 : 0))) * (double)units / (256.0 * (double)cdiv(units, 256));
        if (eff > best_eff + 0.02) { best_eff = eff; best_n = n; }
    }
    p.seg_len = cdiv(Hl, best_n);
    const int nseg = p.nseg = cdiv(Hl, p.seg_len);
    p.part = ws;
    p.stamps = spei_stamp_buffer();
    float* pval = ws + (size_t)2 * Hr * p.ktiles * Nl;
    int32_t* pidx = reinterpret_cast<int32_t*>(pval + (size_t)2 * DRSPLIT * Nl);
    p.nwg = (int)(cols * nseg);
    p.xcd = spei_knob("SPEI_CORR_DIAG_XCD", 1);                     // tuning build: 0 = workgroups in launch order
    ensure_dyn_lds<&corr_diag_kernel<LP>>(DIAG_LDS);
    hipLaunchKernelGGL((corr_diag_kernel<LP>), dim3((unsigned)(p.xcd ? cdiv(p.nwg, 8) * 8 : p.nwg)), dim3(512), DIAG_LDS, st, p);
    hipLaunchKernelGGL(corr_diag_reduce_kernel, dim3(cdiv(Nl, 256), DRSPLIT), dim3(256), 0, st, p.part, Hl, Wl, Hr, Wr, p.ktiles, pval, pidx);
    hipLaunchKernelGGL(corr_diag_final_kernel, dim3(cdiv(Nl, 256)), dim3(256), 0, st, pval, pidx, Nl, S, arg, S2, arg2);
    return 0;
}

}  // namespace

extern "C" int64_t spei_corr_diag_ws_floats(int Hl, int Wl, int Hr, int Wr) {
    const int64_t n = (int64_t)Hl * Wl;
    return 2 * (int64_t)Hr * cdiv(Wr, DT) * n + 4 * (int64_t)DRSPLIT * n;
}

extern "C" int spei_corr_diag_top2_16(int fmt, const void* lr16, const void* ref16, const float* inv_ref, int Hl, int Wl, int Hr, int Wr,
                                      int C, float* S, int32_t* arg, float* S2, int32_t* arg2, float* ws, spei_stream_t stream) {
    SPEI_REQUIRE(lr16 && ref16 && inv_ref && S && arg && S2 && arg2 && ws, "spei_corr_diag_top2_16: null pointer");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_corr_diag_top2_16: fmt=%d", fmt);
    SPEI_REQUIRE(C == 128, "spei_corr_diag_top2_16: C=%d (128 built)", C);
    SPEI_REQUIRE(Hl > 0 && Wl > 0 && Hr > 0 && Wr > 0, "spei_corr_diag_top2_16: empty map");
    SPEI_REQUIRE(Hr >= Hl, "spei_corr_diag_top2_16: the diagonals are cyclic in the reference height: Hr=%d must be >= Hl=%d (use spei_corr_slab_top2_16)", Hr, Hl);
    SPEI_REQUIRE((int64_t)Hl * Wl < (1ll << 30) && (int64_t)Hr * Wr < (1ll << 30), "spei_corr_diag_top2_16: map too large");
    SPEI_REQUIRE(((uintptr_t)lr16 | (uintptr_t)ref16) % 16 == 0, "spei_corr_diag_top2_16: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (fmt == SPEI_F16) corr_diag_run<_Float16>(lr16, ref16, inv_ref, Hl, Wl, Hr, Wr, S, arg, S2, arg2, ws, st);
    else corr_diag_run<__bf16>(lr16, ref16, inv_ref, Hl, Wl, Hr, Wr, S, arg, S2, arg2, ws, st);
    SPEI_CHECK_LAUNCH("spei_corr_diag_top2_16");
    return 0;
}

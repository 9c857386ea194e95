// K11d, diagonal-sliding form of the fused 3x3-patch correlation + top-2 arg-max (reference model/SearchTransfer.py:26-34,
// 61-69), for query and reference maps of the same size (the only case the model produces).
//
// The reference multiplies 9*C-long unfolded patches: 2 * (H W)^2 * 9 C flops.  But the score of query (y, x) against
// reference (y', x') is a sum over the three patch rows,
//
//   R[(y, x), (y', x')] = sum_{dy = -1..1} D[y + dy, y' + dy][x, x'],   D[a, b][x, x'] = sum_{dx, c} F[a, x + dx, c] G[b, x' + dx, c]
//
// and D[a, b] — one query row against one reference row, the horizontal taps only — is shared by the three (y, y') pairs on
// its diagonal.  A workgroup therefore walks DOWN a diagonal delta = (y' - y) mod H: at query row a it computes the D tile of
// rows (a, a + delta) once (K = 3 C on the matrix pipe; the 3 horizontal taps are LDS addressing, as in corr_slab16.hip) and
// the scores of row a - 1 are D_{a-2} + D_{a-1} + D_a, two vector adds per element.  A third of the reference's flops reach
// the matrix pipe; every score is still the full 9 C-term sum in fp32.
//
//   * workgroup = 4 waves = 4 neighbouring (cyclic) diagonals x one 64-query x 64-reference position tile pair; wave w owns
//     diagonal delta0 + w: a 64 x 64 D tile (4 accumulators of 32 x 32), the previous D and the pending two-term sum stay in
//     registers (192 accumulator registers per lane; one wave per SIMD);
//   * LDS: the query row (double buffered) and a ring of 5 reference rows (4 in use, 1 loading), 66 pixels x 128 channels each,
//     pixel pitch 2 * 128 + 16 B; per step ONE new query row and ONE new reference row tile cross L2 -> LDS (34 KB per
//     12.6 MFLOP... the four diagonals share them);
//   * cyclic diagonals (reference row = (a + delta) mod H) make every workgroup's walk the same length; where the reference
//     row wraps to 0 the chain of a diagonal is cut (the term across the wrap is a zero-padded patch row);
//   * every step ends in the top-2 fold of corr_slab16.hip over the wave's 64 reference positions, and lane pairs write the
//     two best keys of each query position for (delta, reference tile) — a float pair whose low six mantissa bits carry
//     63 - position.  corr_diag_reduce_kernel folds the H * ceil(W / 64) pairs of a query; corr_top2_final-style output
//     (S, arg, S2, arg2) then goes to spei_corr_rescore like the slab kernel's.
#include <type_traits>
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
constexpr int DLOADS = (2 * DPIECES + 255) / 256;
constexpr int DRSPLIT = 8;                 // splits of the reduce kernel

template <typename LP>
struct CorrDiagParams {
    const LP* lr;
    const LP* ref;
    const float* inv_ref;
    float* part;                           // [H (delta)][xtiles][H * W] key pairs
    int H, W, xtiles, ngroups, seg_len;
};

template <typename LP>
__global__ __launch_bounds__(64 * DM) void corr_diag_kernel(const CorrDiagParams<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    constexpr int C = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* fbuf = smem;                                      // [2][DROWB]
    unsigned char* gring = smem + 2 * DROWB;                         // [DRING][DROWB]
    float* inv_s = reinterpret_cast<float*>(smem + (2 + DRING) * DROWB);   // [DINV][DT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fk = lane >> 5;
    int bid = blockIdx.x;
    const int group = bid % p.ngroups; bid /= p.ngroups;
    const int kxt = bid % p.xtiles; bid /= p.xtiles;
    const int qxt = bid % p.xtiles;
    const int seg = bid / p.xtiles;
    const int H = p.H, W = p.W;
    const int d0 = group * DM, delta = d0 + wave;
    const bool active = delta < H;                                   // the last group of a map whose height is not a multiple of 4
    const int a0 = seg * p.seg_len, a1 = min(H, a0 + p.seg_len);
    const int a_start = max(a0 - 1, 0), a_end = min(a1, H - 1);      // rows whose D tile this workgroup computes (inclusive)
    const int qx0 = qxt * DT, kx0 = kxt * DT;
    const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};

    auto ref_row = [&](int rho) { return (rho + d0) % H; };          // rho = query row + diagonal offset inside the group

    // ---- prologue: query row a_start, reference rows rho = a_start .. a_start + DM - 1 -------------------------------------------
    for (int idx = tid; idx < (1 + DM) * DPIECES; idx += 64 * DM) {
        const int which = idx / DPIECES, rem = idx - which * DPIECES;
        const int pix = rem >> 4, c16 = rem & 15;
        const bool q = which == 0;
        const int rho = a_start + which - 1;
        const int row = q ? a_start : ref_row(rho);
        const int gx = (q ? qx0 : kx0) - 1 + pix;
        const LP* src = q ? p.lr : p.ref;
        const bool ok = (gx >= 0) & (gx < W);
        unsigned char* dst = q ? fbuf : gring + (rho % DRING) * DROWB;
        *reinterpret_cast<u32x4*>(dst + pix * DPITCH + c16 * 16) =
            ok ? *reinterpret_cast<const u32x4*>(src + ((size_t)row * W + gx) * C + c16 * 8) : zero4;
    }
    for (int idx = tid; idx < DM * DT; idx += 64 * DM) {
        const int rho = a_start + (idx >> 6), x = kx0 + (idx & 63);
        inv_s[(rho % DINV) * DT + (idx & 63)] = x < W ? p.inv_ref[(size_t)ref_row(rho) * W + x] : 0.f;
    }
    __syncthreads();

    // ---- per-lane operand bases -------------------------------------------------------------------------------------------------
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = (32 * i + fr) * DPITCH + fk * 16;
    const bool edge = kx0 + DT > W;                                  // block-uniform: reference positions beyond the map

    f32x16 A[2][2], B[2][2], P1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { A[i][j][r] = 0.f; B[i][j][r] = 0.f; P1[i][j][r] = 0.f; }

    // fold the scores of query row `aq` on this wave's diagonal (sum = P1 + m * X) and write the pair of every query position
    auto emit = [&](int aq, const f32x16 (&X)[2][2], float m) __attribute__((always_inline)) {
        const int rho = aq + wave;
        const float* pinv = inv_s + (rho % DINV) * DT + 4 * fk;
        float irv[32];
#pragma unroll
        for (int g4 = 0; g4 < 8; ++g4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(pinv + (g4 >> 2) * 32 + (g4 & 3) * 8);
            irv[4 * g4] = q[0]; irv[4 * g4 + 1] = q[1]; irv[4 * g4 + 2] = q[2]; irv[4 * g4 + 3] = q[3];
        }
        float lk1[2], lk2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) lk1[j] = lk2[j] = -INFINITY;
        auto fold = [&](auto EDGE) __attribute__((always_inline)) {
#pragma unroll
            for (int row = 0; row < 32; ++row) {
                const int i = row >> 4, r = row & 15;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float v = __builtin_fmaf(X[i][j][r], m, P1[i][j][r]) * irv[row];
                    float key = __uint_as_float((__float_as_uint(v) & ~31u) | (unsigned)(31 - row));
                    if (decltype(EDGE)::value) key = irv[row] == 0.f ? -INFINITY : key;
                    lk2[j] = __builtin_amdgcn_fmed3f(lk1[j], lk2[j], key);
                    lk1[j] = fmaxf(lk1[j], key);
                }
            }
        };
        if (edge) fold(std::true_type{}); else fold(std::false_type{});
        // 5-bit register tag -> 6-bit position tag (position R = 32 i + 8 (r >> 2) + 4 fk + (r & 3) = 2 row - (row & 3) + 4 fk)
        auto retag = [&](float k) __attribute__((always_inline)) {
            const unsigned u = __float_as_uint(k);
            const int row = 31 - (int)(u & 31u);
            const int R = 2 * row - (row & 3) + 4 * fk;
            return k == -INFINITY ? k : __uint_as_float((u & ~63u) | (unsigned)(63 - R));
        };
        const int qrow = aq;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float k1 = retag(lk1[j]), k2 = retag(lk2[j]);
            const float o1 = __shfl_xor(k1, 32, 64), o2 = __shfl_xor(k2, 32, 64);
            const float m1 = fmaxf(k1, o1), m2 = fmaxf(fminf(k1, o1), fmaxf(k2, o2));
            const int qx = qx0 + 32 * j + fr;
            if (fk == 0 && qx < W) {
                float2 o; o.x = m1; o.y = m2;
                *reinterpret_cast<float2*>(p.part + 2 * ((((size_t)delta * p.xtiles + kxt) * H + qrow) * W + qx)) = o;
            }
        }
    };

    // one step: D tile of query row a (computed into X), scores of row a - 1, state update.  Pold = D of row a - 1.
    auto step = [&](int a, f32x16 (&X)[2][2], f32x16 (&Pold)[2][2]) __attribute__((always_inline)) {
        const int n = a - a_start;
        const int cur = n & 1;
        const bool more = a < a_end;
        // stage the next step's rows in registers: query row a + 1, reference row rho = a + DM
        u32x4 st[DLOADS];
        float riv = 0.f;
        const int rho_new = a + DM;
        if (more) {
            const int grow = ref_row(rho_new);
#pragma unroll
            for (int u = 0; u < DLOADS; ++u) {
                const int idx = tid + u * 256;
                const bool q = idx < DPIECES;
                const int rem = q ? idx : idx - DPIECES;
                const int pix = rem >> 4, c16 = rem & 15;
                const int gx = (q ? qx0 : kx0) - 1 + pix;
                const int row = q ? a + 1 : grow;
                const LP* src = q ? p.lr : p.ref;
                const bool ok = (idx < 2 * DPIECES) & (gx >= 0) & (gx < W);
                st[u] = ok ? *reinterpret_cast<const u32x4*>(src + ((size_t)row * W + gx) * C + c16 * 8) : zero4;
            }
            if (tid < DT) riv = kx0 + tid < W ? p.inv_ref[(size_t)grow * W + kx0 + tid] : 0.f;
        }
        // ---- D tile: 3 horizontal taps x 8 k-steps of 16 channels -----------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) X[i][j][r] = 0.f;
        if (active) {
            const unsigned char* ga = gring + ((a + wave) % DRING) * DROWB;
            const unsigned char* fb = fbuf + cur * DROWB;
            constexpr int NS = 24;
            lp8 fa[2][2], fq[2][2];
            auto load_frags = [&](int s, int slot) __attribute__((always_inline)) {
                const int dx = s >> 3, ks = s & 7;
                const int off = dx * DPITCH + ks * 32;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[slot][i] = *reinterpret_cast<const lp8*>(ga + abase[i] + off);
                    fq[slot][i] = *reinterpret_cast<const lp8*>(fb + abase[i] + off);
                }
            };
            load_frags(0, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int c = s & 1;
                if (s + 1 < NS) load_frags(s + 1, c ^ 1);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) X[i][j] = mfma16(fa[c][i], fq[c][j], X[i][j]);
                if (s + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            // ---- scores of row a - 1 and the sliding state ----------------------------------------------------------------------------
            const int b = (a + delta) % H;
            const float m = (n > 0 && b != 0) ? 1.f : 0.f;           // D of row a continues the diagonal of row a - 1
            if (a - 1 >= a0) emit(a - 1, X, m);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) P1[i][j][r] = __builtin_fmaf(Pold[i][j][r], m, X[i][j][r]);
        }
        if (more) {
            unsigned char* fdst = fbuf + (cur ^ 1) * DROWB;
            unsigned char* gdst = gring + (rho_new % DRING) * DROWB;
#pragma unroll
            for (int u = 0; u < DLOADS; ++u) {
                const int idx = tid + u * 256;
                if (idx < 2 * DPIECES) {
                    const bool q = idx < DPIECES;
                    const int rem = q ? idx : idx - DPIECES;
                    *reinterpret_cast<u32x4*>((q ? fdst : gdst) + (rem >> 4) * DPITCH + (rem & 15) * 16) = st[u];
                }
            }
            if (tid < DT) inv_s[(rho_new % DINV) * DT + tid] = riv;
        }
        __syncthreads();
    };

    for (int a = a_start; a <= a_end; a += 2) {
        step(a, A, B);
        if (a + 1 <= a_end) step(a + 1, B, A);
    }
    // the last row of the map has no successor: its scores are the pending two-term sum
    if (active && a_end < a1) emit(a_end, P1, 0.f);
}

__device__ __forceinline__ bool dbetter(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// fold the (delta, reference tile) pairs of every query position; blockIdx.y takes every DRSPLIT-th slot
__global__ __launch_bounds__(256) void corr_diag_reduce_kernel(const float* __restrict__ part, int H, int W, int xtiles,
                                                               float* __restrict__ pval, int32_t* __restrict__ pidx) {
    const int Nl = H * W;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Nl) return;
    const int a = i / W;
    float v1 = -INFINITY, v2 = -INFINITY;
    int x1 = 0x7fffffff, x2 = 0x7fffffff;
    const int nslots = H * xtiles;
    for (int s = blockIdx.y; s < nslots; s += DRSPLIT) {
        const float2 k = *reinterpret_cast<const float2*>(part + 2 * ((size_t)s * Nl + i));
        const int delta = s / xtiles, kxt = s - delta * xtiles;
        int b = a + delta;
        b = b >= H ? b - H : b;
        const int base = b * W + kxt * DT;
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
int corr_diag_run(const void* lr16, const void* ref16, const float* inv_ref, int H, int W, float* S, int32_t* arg, float* S2,
                  int32_t* arg2, float* ws, hipStream_t st) {
    CorrDiagParams<LP> p;
    p.lr = (const LP*)lr16; p.ref = (const LP*)ref16; p.inv_ref = inv_ref;
    p.H = H; p.W = W;
    p.xtiles = cdiv(W, DT);
    p.ngroups = cdiv(H, DM);
    const int Nl = H * W;
    // cut the walk into segments (each pays two extra rows) until the grid fills 256 CUs with little tail
    const int64_t cols = (int64_t)p.xtiles * p.xtiles * p.ngroups;
    int best_n = 1;
    double best_eff = 0.0;
    for (int n = 1; n <= 8 && n <= H; ++n) {
        const int len = cdiv(H, n);
        const int64_t units = cols * cdiv(H, len);
        const double eff = ((double)len / (len + (n > 1 ? 2 : 0))) * (double)units / (256.0 * (double)cdiv(units, 256));
        if (eff > best_eff + 0.02) { best_eff = eff; best_n = n; }
    }
    p.seg_len = cdiv(H, best_n);
    const int nseg = cdiv(H, p.seg_len);
    p.part = ws;
    float* pval = ws + (size_t)2 * H * p.xtiles * Nl;
    int32_t* pidx = reinterpret_cast<int32_t*>(pval + (size_t)2 * DRSPLIT * Nl);
    ensure_dyn_lds<&corr_diag_kernel<LP>>(DIAG_LDS);
    hipLaunchKernelGGL((corr_diag_kernel<LP>), dim3((unsigned)(cols * nseg)), dim3(64 * DM), DIAG_LDS, st, p);
    hipLaunchKernelGGL(corr_diag_reduce_kernel, dim3(cdiv(Nl, 256), DRSPLIT), dim3(256), 0, st, p.part, H, W, p.xtiles, pval, pidx);
    hipLaunchKernelGGL(corr_diag_final_kernel, dim3(cdiv(Nl, 256)), dim3(256), 0, st, pval, pidx, Nl, S, arg, S2, arg2);
    return 0;
}

}  // namespace

extern "C" int64_t spei_corr_diag_ws_floats(int H, int W) {
    const int64_t n = (int64_t)H * W;
    return 2 * (int64_t)H * cdiv(W, DT) * n + 4 * (int64_t)DRSPLIT * n;
}

extern "C" int spei_corr_diag_top2_16(int fmt, const void* lr16, const void* ref16, const float* inv_ref, int H, int W, int C,
                                      float* S, int32_t* arg, float* S2, int32_t* arg2, float* ws, spei_stream_t stream) {
    SPEI_REQUIRE(lr16 && ref16 && inv_ref && S && arg && S2 && arg2 && ws, "spei_corr_diag_top2_16: null pointer");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_corr_diag_top2_16: fmt=%d", fmt);
    SPEI_REQUIRE(C == 128, "spei_corr_diag_top2_16: C=%d (128 built)", C);
    SPEI_REQUIRE(H > 0 && W > 0, "spei_corr_diag_top2_16: empty map");
    SPEI_REQUIRE((int64_t)H * W < (1ll << 30), "spei_corr_diag_top2_16: map too large");
    SPEI_REQUIRE(((uintptr_t)lr16 | (uintptr_t)ref16) % 16 == 0, "spei_corr_diag_top2_16: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (fmt == SPEI_F16) corr_diag_run<_Float16>(lr16, ref16, inv_ref, H, W, S, arg, S2, arg2, ws, st);
    else corr_diag_run<__bf16>(lr16, ref16, inv_ref, H, W, S, arg, S2, arg2, ws, st);
    SPEI_CHECK_LAUNCH("spei_corr_diag_top2_16");
    return 0;
}

// K3 — ResBlock epilogue: SE + TripletAttention gates and the gated residual sum
// (reference model/block.py:8-24, 71-96, 108-140).  x1 is conv2's output, NHWC [H][W][C].
//
//   s[c]      = sigmoid(W2 relu(W1 mean_hw(x1) + b1) + b2)
//   g1[h][c]  = BN(conv7x7([max_w x1, mean_w x1]))   on the (H, C) plane, zero pad 3, no sigmoid
//   g2[w][c]  = BN(conv5x5([max_h x1, mean_h x1]))   on the (C, W) plane, zero pad 2, no sigmoid
//   out       = x + (x1*s + (x1*g1 + x1*g2))
//
// Three global reductions force a grid-wide dependency, so the gates take two launches (tile statistics; then the gate
// maps + SE, whose blocks first combine the tile partials of the rows / columns they need, 7x7 / 5x5 halo included) and
// the application one elementwise launch.  All partial sums are combined in a fixed order: results are bitwise
// reproducible run to run.
#include "common.h"

namespace {

// Tiling of the statistics pass: a 256-thread block owns a TR-row x TC-column pixel tile, thread = (pixel column, channel
// octet): TC = 256 / (C/8) columns.  Every thread issues all TR loads of its column up front (16 B each for 16-bit x1), keeps
// the column partials in registers, and parks the loaded vectors in LDS; the row partials are then summed by a TRANSPOSED
// thread mapping (thread = (row, octet, column segment), sequential over its columns) — no cross-lane reduction per element
// (round 1's version spent its time in them: 1.3 TB/s), fixed summation order, results bitwise reproducible.
constexpr int GATE_TR16 = 16;    // tile rows for 16-bit x1
constexpr int GATE_TR32 = 8;     // tile rows for fp32 x1 (same LDS footprint)
__host__ __device__ inline int gate_tr(bool lp) { return lp ? GATE_TR16 : GATE_TR32; }
__host__ __device__ inline int gate_tc(int C) { return 256 / (C / 8); }

struct GateWs {
    float* rowpmax;  // [ntx][H][C]
    float* rowpsum;  // [ntx][H][C]
    float* colpmax;  // [nty][W][C]
    float* colpsum;  // [nty][W][C]
    float* tilesum;  // [nty*ntx][C]  per-tile channel sums (for the SE mean)
    int ntx, nty;
};
__host__ __device__ inline GateWs carve(float* ws, int H, int W, int C, bool lp) {
    GateWs g;
    g.ntx = (W + gate_tc(C) - 1) / gate_tc(C);
    g.nty = (H + gate_tr(lp) - 1) / gate_tr(lp);
    g.rowpmax = ws;
    g.rowpsum = g.rowpmax + (size_t)g.ntx * H * C;
    g.colpmax = g.rowpsum + (size_t)g.ntx * H * C;
    g.colpsum = g.colpmax + (size_t)g.nty * W * C;
    g.tilesum = g.colpsum + (size_t)g.nty * W * C;
    return g;
}

template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* p) {                    // fp32, bf16 or half in HBM
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const f32x4*>(p);
    } else {
        const typename lpv<T>::x4 h = *reinterpret_cast<const typename lpv<T>::x4*>(p);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
}

__device__ __forceinline__ f32x4 max4(f32x4 a, f32x4 b) {
    return f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])};
}

// ---- launch 1: tile statistics -----------------------------------------------------------------------------
template <int C, typename TX>
__global__ __launch_bounds__(256) void gate_stats_kernel(const TX* __restrict__ x1, int H, int W, float* ws, int64_t wsf) {
    constexpr bool LP = sizeof(TX) == 2;
    // batched launch: blockIdx.y = map (one of the frame's encoder passes through this layer); its own map and workspace slice
    x1 += (size_t)blockIdx.y * H * W * C;
    const GateWs g = carve(ws + (size_t)blockIdx.y * wsf, H, W, C, LP);
    constexpr int OCT = C / 8;                   // threads per pixel (8 channels each)
    constexpr int TC = 256 / OCT;                // tile columns
    constexpr int TR = LP ? GATE_TR16 : GATE_TR32;
    constexpr int EB = LP ? 16 : 32;             // bytes of one (pixel, octet) entry
    constexpr int ROWB = TC * OCT * EB + 4 * EB; // LDS row pitch: + 4 entries, so the transposed reads of 16 rows spread over the banks
    constexpr int ITEMS = TR * OCT;              // (row, octet) sums of the transposed pass
    constexpr int PARTS = 256 / ITEMS;           // column segments per item: 4, 2, 1 (16-bit) / 8, 4, 2 (fp32)
    constexpr int SEG = TC / PARTS;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];      // [TR][ROWB] bytes, then the row sums [TR][C] floats
    float* rsum = reinterpret_cast<float*>(tile + TR * ROWB);
    const int tid = threadIdx.x;
    const int oct = tid % OCT, col = tid / OCT;
    const int tx = blockIdx.x % g.ntx, ty = blockIdx.x / g.ntx;
    const int x = tx * TC + col, y0 = ty * TR;

    // ---- pass 1: all loads of this thread's column, column partials in registers, raw vectors to LDS ------------------------
    u32x4 v[TR][LP ? 1 : 2];
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const bool ok = x < W && y0 + r < H;
        const TX* src = x1 + ((size_t)min(y0 + r, H - 1) * W + min(x, W - 1)) * C + oct * 8;
#pragma unroll
        for (int hlf = 0; hlf < (LP ? 1 : 2); ++hlf) {
            v[r][hlf] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(src) + 16 * hlf);
            if (!ok) v[r][hlf] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    auto unpack = [](const u32x4* q, float* f) __attribute__((always_inline)) {
        if constexpr (LP) {
            const typename lpv<TX>::x8 h = *reinterpret_cast<const typename lpv<TX>::x8*>(q);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = __uint_as_float(q[e >> 2][e & 3]);
        }
    };
    float cmax[8], csum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { cmax[e] = -INFINITY; csum[e] = 0.f; }
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        float f[8];
        unpack(v[r], f);
        if (y0 + r < H) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { cmax[e] = fmaxf(cmax[e], f[e]); csum[e] += f[e]; }
        }
#pragma unroll
        for (int hlf = 0; hlf < (LP ? 1 : 2); ++hlf)
            *reinterpret_cast<u32x4*>(tile + r * ROWB + (col * OCT + oct) * EB + 16 * hlf) = v[r][hlf];
    }
    if (x < W) {
        float* pm = g.colpmax + ((size_t)ty * W + x) * C + oct * 8;
        float* ps = g.colpsum + ((size_t)ty * W + x) * C + oct * 8;
        *reinterpret_cast<f32x4*>(pm) = f32x4{cmax[0], cmax[1], cmax[2], cmax[3]};
        *reinterpret_cast<f32x4*>(pm + 4) = f32x4{cmax[4], cmax[5], cmax[6], cmax[7]};
        *reinterpret_cast<f32x4*>(ps) = f32x4{csum[0], csum[1], csum[2], csum[3]};
        *reinterpret_cast<f32x4*>(ps + 4) = f32x4{csum[4], csum[5], csum[6], csum[7]};
    }
    __syncthreads();

    // ---- pass 2: row partials, thread = (row, octet, column phase), sequential over every PARTS-th column -----------------
    {
        const int part = tid % PARTS, item = tid / PARTS;
        const int ro = item % OCT, rr = item / OCT;             // octet, row
        float rmax[8], rs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { rmax[e] = -INFINITY; rs[e] = 0.f; }
        const int ncol = min(TC, W - tx * TC);                  // columns of this tile inside the map
#pragma unroll 4
        for (int j = 0; j < SEG; ++j) {
            const int c2 = j * PARTS + part;               // interleaved segments: 16 lanes read 256 contiguous bytes
            u32x4 q[LP ? 1 : 2];
#pragma unroll
            for (int hlf = 0; hlf < (LP ? 1 : 2); ++hlf) q[hlf] = *reinterpret_cast<const u32x4*>(tile + rr * ROWB + (c2 * OCT + ro) * EB + 16 * hlf);
            float f[8];
            unpack(q, f);
            if (c2 < ncol) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { rmax[e] = fmaxf(rmax[e], f[e]); rs[e] += f[e]; }
            }
        }
        // combine the PARTS adjacent lanes of an item in a fixed order (lane part 0 gathers): xor 1, 2, 4 butterflies
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (PARTS >= 2) { rmax[e] = fmaxf(rmax[e], dpp_quad<0xB1>(rmax[e])); rs[e] += dpp_quad<0xB1>(rs[e]); }
            if constexpr (PARTS >= 4) { rmax[e] = fmaxf(rmax[e], dpp_quad<0x4E>(rmax[e])); rs[e] += dpp_quad<0x4E>(rs[e]); }
            if constexpr (PARTS >= 8) { rmax[e] = fmaxf(rmax[e], dpp_quad<0x141>(rmax[e])); rs[e] += dpp_quad<0x141>(rs[e]); }
        }
        const int y = y0 + rr;
        if (part == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) rsum[rr * C + ro * 8 + e] = (y < H) ? rs[e] : 0.f;
            if (y < H) {
                float* pm = g.rowpmax + ((size_t)tx * H + y) * C + ro * 8;
                float* ps = g.rowpsum + ((size_t)tx * H + y) * C + ro * 8;
                *reinterpret_cast<f32x4*>(pm) = f32x4{rmax[0], rmax[1], rmax[2], rmax[3]};
                *reinterpret_cast<f32x4*>(pm + 4) = f32x4{rmax[4], rmax[5], rmax[6], rmax[7]};
                *reinterpret_cast<f32x4*>(ps) = f32x4{rs[0], rs[1], rs[2], rs[3]};
                *reinterpret_cast<f32x4*>(ps + 4) = f32x4{rs[4], rs[5], rs[6], rs[7]};
            }
        }
    }
    __syncthreads();
    if (tid < C) {                                   // tile total per channel (SE mean): the row sums, top to bottom
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < TR; ++r) t += rsum[r * C + tid];
        g.tilesum[(size_t)blockIdx.x * C + tid] = t;
    }
}

template <int C, typename TX>
void launch_gate_stats(const TX* x1, int H, int W, const GateWs& g, float* ws, int64_t wsf, int batch, hipStream_t st) {
    constexpr bool LP = sizeof(TX) == 2;
    constexpr int OCT = C / 8, TC = 256 / OCT, TR = LP ? GATE_TR16 : GATE_TR32, EB = LP ? 16 : 32;
    const size_t lds = (size_t)TR * (TC * OCT * EB + 4 * EB) + (size_t)TR * C * sizeof(float);
    ensure_dyn_lds<&gate_stats_kernel<C, TX>>(lds);
    hipLaunchKernelGGL((gate_stats_kernel<C, TX>), dim3(g.ntx * g.nty, batch), dim3(256), lds, st, x1, H, W, ws, wsf);
}

// Reduce the `nt` tile partials of `cnt` rows (or columns) starting at line `l0` (lines outside [0, L) give -inf / 0) into
// zmax / zmean in LDS.  A thread owns up to 4 (line, channel) items and walks the tiles with all of them in flight: the
// loads of one step are independent, only the per-item max / sum chains are serial (fixed tile order => reproducible).
__device__ __forceinline__ void reduce_partials(const float* __restrict__ pmax, const float* __restrict__ psum, int nt, int64_t stride,
                                                int l0, int cnt, int L, int C, float, float denom, float* zmax, float* zmean) {
    float mx[4], sm[4];
    int64_t off[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int it = threadIdx.x + 256 * j;
        const int line = l0 + it / C;
        ok[j] = it < cnt * C && line >= 0 && line < L;
        off[j] = ok[j] ? (int64_t)line * C + it % C : 0;
        mx[j] = -INFINITY;
        sm[j] = 0.f;
    }
    int k = 0;
    for (; k + 4 <= nt; k += 4) {                    // 32 independent loads in flight, then the four tiles in order
        float a[4][4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[u][j] = pmax[(size_t)(k + u) * stride + off[j]];                                        // clamped address when !ok
                b[u][j] = psum[(size_t)(k + u) * stride + off[j]];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) { mx[j] = fmaxf(mx[j], a[u][j]); sm[j] += b[u][j]; }
    }
    for (; k < nt; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = pmax[(size_t)k * stride + off[j]], b = psum[(size_t)k * stride + off[j]];
            mx[j] = fmaxf(mx[j], a);
            sm[j] += b;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int it = threadIdx.x + 256 * j;
        if (it < cnt * C) {
            zmax[it] = ok[j] ? mx[j] : -INFINITY;
            zmean[it] = ok[j] ? sm[j] / denom : 0.f;
        }
    }
}

// ---- launch 2: combine the tile partials, then the two 2->1 channel convolutions + eval BatchNorm(1); last block = SE ----
// A block owns 256 consecutive (row, channel) or (column, channel) outputs; it first reduces the tile partials of every
// row / column its 7x7 / 5x5 window touches into LDS (the reduction used to be a launch of its own: ~13 us of a tiny grid
// on the critical chain of every ResBlock), in the same fixed order over the tiles.
__global__ __launch_bounds__(256) void gate_maps_kernel(int H, int W, int C, float* ws, int64_t wsf, int lp, const float* __restrict__ cw_w,
                                                        const float* __restrict__ cw_bn, const float* __restrict__ hc_w,
                                                        const float* __restrict__ hc_bn, float* __restrict__ g1,
                                                        float* __restrict__ g2, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float* __restrict__ s) {
    const GateWs g = carve(ws + (size_t)blockIdx.y * wsf, H, W, C, lp != 0);      // blockIdx.y = map of a batched launch
    g1 += (size_t)blockIdx.y * H * C;
    g2 += (size_t)blockIdx.y * W * C;
    s += (size_t)blockIdx.y * C;
    __shared__ float wk[98];
    __shared__ float zmax[1024], zmean[1024];          // (256/C + 6) * C <= 1024 reduced rows / columns incl. halo
    const int64_t n1 = (int64_t)H * C;
    const int nb1 = (int)((n1 + 255) / 256);
    const int nb2 = (int)(((int64_t)W * C + 255) / 256);
    if ((int)blockIdx.x == nb1 + nb2) {
        // SE: channel means from the per-tile sums.  thread = (channel quad, part): 1024 / C interleaved tile subsets, eight
        // independent 16-byte loads in flight per thread (one block walks ~900 tiles: latency, not bandwidth, is the cost)
        __shared__ f32x4 part[256];
        __shared__ float mean[128], hid[32];
        const int ntiles = g.ntx * g.nty, cq = C / 4, parts = 256 / cq;
        const int q = threadIdx.x % cq, pt = threadIdx.x / cq;
        const f32x4* ts = reinterpret_cast<const f32x4*>(g.tilesum) + q;
        f32x4 acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        int t = pt;
        for (; t + 7 * parts < ntiles; t += 8 * parts) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += ts[(size_t)(t + u * parts) * cq];
        }
        for (; t < ntiles; t += parts) acc[0] += ts[(size_t)t * cq];
        part[threadIdx.x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        if ((int)threadIdx.x < C) {
            const int c = threadIdx.x;
            float a = 0.f;
            for (int k = 0; k < parts; ++k) a += part[k * cq + (c >> 2)][c & 3];
            mean[c] = a / ((float)H * (float)W);
        }
        __syncthreads();
        const int mid = C / 4;
        if ((int)threadIdx.x < mid) {
            float a = b1[threadIdx.x];
            for (int k = 0; k < C; ++k) a = fmaf(w1[threadIdx.x * C + k], mean[k], a);
            hid[threadIdx.x] = fmaxf(a, 0.f);
        }
        __syncthreads();
        if ((int)threadIdx.x < C) {
            float a = b2[threadIdx.x];
            for (int k = 0; k < mid; ++k) a = fmaf(w2[threadIdx.x * mid + k], hid[k], a);
            s[threadIdx.x] = 1.0f / (1.0f + expf(-a));
        }
        return;
    }
    if ((int)blockIdx.x < nb1) {
        if (threadIdx.x < 98) wk[threadIdx.x] = cw_w[threadIdx.x];
        const int64_t i0 = (int64_t)blockIdx.x * 256;
        const int y0 = (int)(i0 / C) - 3, rows = 256 / C + 6;           // 256 % C == 0: the block covers whole rows
        reduce_partials(g.rowpmax, g.rowpsum, g.ntx, n1, y0, rows, H, C, 1.0f, (float)W, zmax, zmean);
        __syncthreads();
        const int64_t i = i0 + threadIdx.x;
        if (i >= n1) return;
        const int c = (int)(i % C), y = (int)(i / C);
        float acc = 0.f;
        // conv "height" axis = H, "width" axis = C  (x.permute(0,3,2,1) -> [B,W,H,C], ZPool over dim 1)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float* z = ch == 0 ? zmax : zmean;
            for (int dy = 0; dy < 7; ++dy) {
                const int yy = y + dy - 3;
                if (yy < 0 || yy >= H) continue;
                for (int dc = 0; dc < 7; ++dc) {
                    const int cc = c + dc - 3;
                    if (cc < 0 || cc >= C) continue;
                    acc = fmaf(wk[ch * 49 + dy * 7 + dc], z[(yy - y0) * C + cc], acc);
                }
            }
        }
        g1[i] = fmaf(acc, cw_bn[0], cw_bn[1]);
    } else {
        if (threadIdx.x < 50) wk[threadIdx.x] = hc_w[threadIdx.x];
        const int64_t n2 = (int64_t)W * C;
        const int64_t i0 = (int64_t)(blockIdx.x - nb1) * 256;
        const int x0 = (int)(i0 / C) - 2, cols = 256 / C + 4;
        reduce_partials(g.colpmax, g.colpsum, g.nty, n2, x0, cols, W, C, 1.0f, (float)H, zmax, zmean);
        __syncthreads();
        const int64_t i = i0 + threadIdx.x;
        if (i >= n2) return;
        const int c = (int)(i % C), x = (int)(i / C);
        float acc = 0.f;
        // conv "height" axis = C, "width" axis = W  (x.permute(0,2,1,3) -> [B,H,C,W], ZPool over dim 1)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float* z = ch == 0 ? zmax : zmean;
            for (int dc = 0; dc < 5; ++dc) {
                const int cc = c + dc - 2;
                if (cc < 0 || cc >= C) continue;
                for (int dx = 0; dx < 5; ++dx) {
                    const int xx = x + dx - 2;
                    if (xx < 0 || xx >= W) continue;
                    acc = fmaf(wk[ch * 25 + dc * 5 + dx], z[(xx - x0) * C + cc], acc);
                }
            }
        }
        g2[i] = fmaf(acc, hc_bn[0], hc_bn[1]);
    }
}

// ---- apply ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void resblock_apply_kernel(const float* __restrict__ x, const T* __restrict__ x1,
                                                             const float* __restrict__ s, const float* __restrict__ g1,
                                                             const float* __restrict__ g2, const float* __restrict__ extra,
                                                             float* __restrict__ out, int ldo, int H, int W, int C) {
    const int cg = C / 4;
    const int64_t total = (int64_t)H * W * cg;
    {   // batched launch (blockIdx.y = map; dense maps, ldo == C, no `extra`): this map's slices
        const size_t b = blockIdx.y;
        x += b * H * W * C; x1 += b * H * W * C; out += b * H * W * ldo;
        s += b * C; g1 += b * H * C; g2 += b * W * C;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cg) * 4;
        const int64_t pix = i / cg;
        const int xx = (int)(pix % W), yy = (int)(pix / W);
        const float4 a = *reinterpret_cast<const float4*>(x + pix * C + c);
        const f32x4 bv = ld4<T>(x1 + pix * C + c);
        const float4 b = make_float4(bv[0], bv[1], bv[2], bv[3]);
        const float4 sv = *reinterpret_cast<const float4*>(s + c);
        const float4 u = *reinterpret_cast<const float4*>(g1 + (size_t)yy * C + c);
        const float4 v = *reinterpret_cast<const float4*>(g2 + (size_t)xx * C + c);
        float4 o;
        // x3 = se(x1) + (cw(x1) + hc(x1));  return x3 + x   (block.py:136-140)
        o.x = (b.x * sv.x + (b.x * u.x + b.x * v.x)) + a.x;
        o.y = (b.y * sv.y + (b.y * u.y + b.y * v.y)) + a.y;
        o.z = (b.z * sv.z + (b.z * u.z + b.z * v.z)) + a.z;
        o.w = (b.w * sv.w + (b.w * u.w + b.w * v.w)) + a.w;
        if (extra) {
            const float4 e = *reinterpret_cast<const float4*>(extra + pix * C + c);
            o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
        }
        *reinterpret_cast<float4*>(out + pix * ldo + c) = o;
    }
}

}  // namespace

extern "C" int64_t spei_gate_ws_floats(int H, int W, int C) {
    if (C != 32 && C != 64 && C != 128) return 0;
    const int64_t ntx = (W + gate_tc(C) - 1) / gate_tc(C), nty = (H + GATE_TR32 - 1) / GATE_TR32;      // the finer of the two tilings
    return 2 * ntx * H * C + 2 * nty * W * C + ntx * nty * C;
}

static int resblock_gates_run(const void* x1, int x1_fmt, int batch, int H, int W, int C, const float* se_w1, const float* se_b1,
                              const float* se_w2, const float* se_b2, const float* cw_w, const float* cw_bn,
                              const float* hc_w, const float* hc_bn, float* s, float* g1, float* g2, float* ws,
                              spei_stream_t stream) {
    SPEI_REQUIRE(batch >= 1 && batch <= 65535, "spei_resblock_gates_batched: batch=%d", batch);
    SPEI_REQUIRE(x1 && se_w1 && se_b1 && se_w2 && se_b2 && cw_w && cw_bn && hc_w && hc_bn && s && g1 && g2 && ws,
                 "spei_resblock_gates: null pointer");
    SPEI_REQUIRE(C == 32 || C == 64 || C == 128, "spei_resblock_gates: C=%d (32/64/128 built)", C);
    SPEI_REQUIRE(H > 0 && W > 0, "spei_resblock_gates: empty map");
    SPEI_REQUIRE(((uintptr_t)x1 | (uintptr_t)ws) % 16 == 0, "spei_resblock_gates: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    SPEI_REQUIRE(x1_fmt == SPEI_F32 || x1_fmt == SPEI_BF16 || x1_fmt == SPEI_F16, "spei_resblock_gates: x1_fmt=%d", x1_fmt);
    GateWs g = carve(ws, H, W, C, x1_fmt != SPEI_F32);
    const int64_t wsf = spei_gate_ws_floats(H, W, C);
    if (x1_fmt == SPEI_BF16) {
        const __bf16* xp = (const __bf16*)x1;
        if (C == 32) launch_gate_stats<32>(xp, H, W, g, ws, wsf, batch, st);
        else if (C == 64) launch_gate_stats<64>(xp, H, W, g, ws, wsf, batch, st);
        else launch_gate_stats<128>(xp, H, W, g, ws, wsf, batch, st);
    } else if (x1_fmt == SPEI_F16) {
        const _Float16* xp = (const _Float16*)x1;
        if (C == 32) launch_gate_stats<32>(xp, H, W, g, ws, wsf, batch, st);
        else if (C == 64) launch_gate_stats<64>(xp, H, W, g, ws, wsf, batch, st);
        else launch_gate_stats<128>(xp, H, W, g, ws, wsf, batch, st);
    } else {
        const float* xp = (const float*)x1;
        if (C == 32) launch_gate_stats<32>(xp, H, W, g, ws, wsf, batch, st);
        else if (C == 64) launch_gate_stats<64>(xp, H, W, g, ws, wsf, batch, st);
        else launch_gate_stats<128>(xp, H, W, g, ws, wsf, batch, st);
    }
    hipLaunchKernelGGL(gate_maps_kernel, dim3(cdiv((int64_t)H * C, 256) + cdiv((int64_t)W * C, 256) + 1, batch), dim3(256), 0, st, H, W, C, ws, wsf,
                       (int)(x1_fmt != SPEI_F32), cw_w, cw_bn, hc_w, hc_bn, g1, g2, se_w1, se_b1, se_w2, se_b2, s);
    SPEI_CHECK_LAUNCH("spei_resblock_gates");
    return 0;
}

extern "C" int spei_resblock_gates(const void* x1, int x1_fmt, int H, int W, int C, const float* se_w1, const float* se_b1,
                                   const float* se_w2, const float* se_b2, const float* cw_w, const float* cw_bn,
                                   const float* hc_w, const float* hc_bn, float* s, float* g1, float* g2, float* ws,
                                   spei_stream_t stream) {
    return resblock_gates_run(x1, x1_fmt, 1, H, W, C, se_w1, se_b1, se_w2, se_b2, cw_w, cw_bn, hc_w, hc_bn, s, g1, g2, ws, stream);
}

extern "C" int spei_resblock_gates_batched(const void* x1, int x1_fmt, int batch, int H, int W, int C, const float* se_w1, const float* se_b1,
                                           const float* se_w2, const float* se_b2, const float* cw_w, const float* cw_bn,
                                           const float* hc_w, const float* hc_bn, float* s, float* g1, float* g2, float* ws,
                                           spei_stream_t stream) {
    return resblock_gates_run(x1, x1_fmt, batch, H, W, C, se_w1, se_b1, se_w2, se_b2, cw_w, cw_bn, hc_w, hc_bn, s, g1, g2, ws, stream);
}

static int resblock_apply_run(const float* x, const void* x1, int x1_fmt, const float* s, const float* g1, const float* g2,
                              const float* extra, float* out, int ldo, int batch, int H, int W, int C, spei_stream_t stream) {
    SPEI_REQUIRE(x && x1 && s && g1 && g2 && out, "spei_resblock_apply: null pointer");
    SPEI_REQUIRE(batch >= 1 && batch <= 65535 && (batch == 1 || (!extra && ldo == C)), "spei_resblock_apply_batched: dense maps, no `extra`");
    SPEI_REQUIRE(C % 4 == 0 && ldo % 4 == 0 && ldo >= C && H > 0 && W > 0, "spei_resblock_apply: bad shape");
    const int64_t total = (int64_t)H * W * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    const dim3 grid(blocks, batch);
    SPEI_REQUIRE(x1_fmt == SPEI_F32 || x1_fmt == SPEI_BF16 || x1_fmt == SPEI_F16, "spei_resblock_apply: x1_fmt=%d", x1_fmt);
    if (x1_fmt == SPEI_BF16) hipLaunchKernelGGL(resblock_apply_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, x, (const __bf16*)x1, s, g1, g2, extra, out, ldo, H, W, C);
    else if (x1_fmt == SPEI_F16) hipLaunchKernelGGL(resblock_apply_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, x, (const _Float16*)x1, s, g1, g2, extra, out, ldo, H, W, C);
    else hipLaunchKernelGGL(resblock_apply_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (const float*)x1, s, g1, g2, extra, out, ldo, H, W, C);
    SPEI_CHECK_LAUNCH("spei_resblock_apply");
    return 0;
}

extern "C" int spei_resblock_apply(const float* x, const void* x1, int x1_fmt, const float* s, const float* g1, const float* g2,
                                   const float* extra, float* out, int ldo, int H, int W, int C, spei_stream_t stream) {
    return resblock_apply_run(x, x1, x1_fmt, s, g1, g2, extra, out, ldo, 1, H, W, C, stream);
}

extern "C" int spei_resblock_apply_batched(const float* x, const void* x1, int x1_fmt, const float* s, const float* g1, const float* g2,
                                           float* out, int batch, int H, int W, int C, spei_stream_t stream) {
    return resblock_apply_run(x, x1, x1_fmt, s, g1, g2, nullptr, out, C, batch, H, W, C, stream);
}

// Training step: the gate maps of a ResBlock from its plane statistics, forward and backward (reference model/block.py:8-24 SEBlock,
// :49-68 BasicConv1 = 2->1 channel conv + BatchNorm2d(1), :75-96 AttentionGate1 / 2 without their sigmoid, :116-119 TripletAttention as
// the sum of the two gates) — in train() mode, i.e. BatchNorm on BATCH statistics with its running buffers moved (momentum 0.01,
// unbiased variance), or in eval() mode on the running statistics.
//
// Round 3 wrote this glue in torch tensor arithmetic (parameter- and plane-sized, "plumbing") and differentiated it with
// torch.autograd: ~40 launches forward and ~100 backward per ResBlock call, 36 calls per step — a third of the step's 15 000 launches
// and most of its at::native kernel time (profiles/r03_train_kernel_stats.md).  Here: 3 launches forward, 6 backward.
//
//   s  [N][C]    = sigmoid(W2 relu(W1 mean + b1) + b2)
//   g1 [N][H][C] = BN_cw(conv7x7([rowmax, rowmean] as a 2-channel H x C image))
//   g2 [N][W][C] = BN_hc(conv5x5([colmax^T, colmean^T] as a 2-channel C x W image))^T
// N = groups x B samples: the statistics of BatchNorm are taken per GROUP of B consecutive samples (one group per encoder pass when the
// frames of a window go through the stack as one batch: the reference normalises each pass with its own batch statistics and moves the
// running buffers once per pass, in pass order).  All sums in float64, fixed order: results are bitwise reproducible.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int K1 = 7, K2 = 5, NW1 = 2 * K1 * K1, NW2 = 2 * K2 * K2;      // 98, 50 conv weights

struct GateP {
    const float *rowmax, *rowmean, *colmax, *colmean, *mean;              // [N,H,C] [N,H,C] [N,W,C] [N,W,C] [N,C]
    const float *se_w1, *se_b1, *se_w2, *se_b2;                           // [C/4,C] [C/4] [C,C/4] [C]
    const float *cw_w, *cw_g, *cw_b, *hc_w, *hc_g, *hc_b;                 // conv [2,K,K], BN affine (scalars)
    float *cw_rm, *cw_rv, *hc_rm, *hc_rv;                                 // BN running buffers (scalars)
    int N, G, B, H, W, C, train, update;
    float *s, *g1, *g2;                                                   // outputs
    float *t1, *t2, *hid, *bnstat;                                        // saved: [N,H,C] [N,C,W] [N,C/4] [2][G][2] = (mean, rstd)
    double* part;                                                         // [blocks][2]
    // backward
    const float *ds, *dg1, *dg2;
    float *d_rowmax, *d_rowmean, *d_colmax, *d_colmean, *d_mean;
    float *dt1, *dt2;                                                     // [N,H,C] [N,C,W]
    double* wpart;                                                        // [NW1 + NW2][WCH]
    float* separt;                                                        // [N][se_len]
    float* dprm;                                                          // se_w1 | se_b1 | se_w2 | se_b2 | cw_w | cw_g | cw_b | hc_w | hc_g | hc_b
    int nb1g, nb2g;                                                       // blocks per group, gate 1 / gate 2
};
constexpr int WCH = 32;                                                   // element chunks per conv weight in the weight-gradient sums

__device__ __forceinline__ double block_sum(double v, double* sh) {      // fixed order: lanes, then the 4 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// which (gate, group, first element) a block of the element grids works on: blocks [0, G nb1g) gate 1, then G nb2g of gate 2
__device__ __forceinline__ void locate(const GateP& p, int& gate, int& grp, long& e0, long& eg) {
    int b = blockIdx.x;
    gate = b >= p.G * p.nb1g;
    if (gate) b -= p.G * p.nb1g;
    const int nb = gate ? p.nb2g : p.nb1g;
    grp = b / nb;
    eg = (long)p.B * (gate ? (long)p.C * p.W : (long)p.H * p.C);          // elements per group
    e0 = (long)(b - grp * nb) * NT;
}

// t1[n,h,c] (gate 1) / t2[n,c,w] (gate 2) of element e of group grp: the 2 -> 1 channel convolution, zero padding
__device__ __forceinline__ float conv_at(const GateP& p, int gate, long idx, const float* wsh) {
    if (!gate) {
        const int c = idx % p.C, h = (idx / p.C) % p.H;
        const long n = idx / ((long)p.C * p.H);
        const float* a0 = p.rowmax + n * p.H * p.C;
        const float* a1 = p.rowmean + n * p.H * p.C;
        float acc = 0.f;
        for (int dy = 0; dy < K1; ++dy) {
            const int y = h + dy - K1 / 2;
            if (y < 0 || y >= p.H) continue;
            for (int dx = 0; dx < K1; ++dx) {
                const int x = c + dx - K1 / 2;
                if (x < 0 || x >= p.C) continue;
                acc += wsh[dy * K1 + dx] * a0[y * p.C + x] + wsh[K1 * K1 + dy * K1 + dx] * a1[y * p.C + x];
            }
        }
        return acc;
    }
    const int w = idx % p.W, c = (idx / p.W) % p.C;
    const long n = idx / ((long)p.C * p.W);
    const float* a0 = p.colmax + n * p.W * p.C;                            // z2[n,0,y,x] = colmax[n,x,y]
    const float* a1 = p.colmean + n * p.W * p.C;
    float acc = 0.f;
    for (int dy = 0; dy < K2; ++dy) {
        const int y = c + dy - K2 / 2;
        if (y < 0 || y >= p.C) continue;
        for (int dx = 0; dx < K2; ++dx) {
            const int x = w + dx - K2 / 2;
            if (x < 0 || x >= p.W) continue;
            acc += wsh[dy * K2 + dx] * a0[x * p.C + y] + wsh[K2 * K2 + dy * K2 + dx] * a1[x * p.C + y];
        }
    }
    return acc;
}

__global__ __launch_bounds__(NT) void gm_conv_kernel(const GateP p) {
    __shared__ float wsh[NW1];
    __shared__ double sh[4];
    int gate, grp;
    long e0, eg;
    locate(p, gate, grp, e0, eg);
    const float* wsrc = gate ? p.hc_w : p.cw_w;
    for (int i = threadIdx.x; i < (gate ? NW2 : NW1); i += NT) wsh[i] = wsrc[i];
    __syncthreads();
    const long e = e0 + threadIdx.x;
    float t = 0.f;
    if (e < eg) {
        const long idx = (long)grp * eg + e;
        t = conv_at(p, gate, idx, wsh);
        (gate ? p.t2 : p.t1)[idx] = t;
    }
    const double s1 = block_sum((double)t, sh);
    const double s2 = block_sum((double)t * (double)t, sh);
    if (threadIdx.x == 0) { p.part[2 * blockIdx.x] = s1; p.part[2 * blockIdx.x + 1] = s2; }
}

// (mean, rstd) of (gate, group): batch statistics from the conv kernel's partial sums, or the running buffers
__device__ __forceinline__ void bn_stats(const GateP& p, int gate, int grp, long eg, double* sh, float& mu, float& rstd, double& var_out) {
    if (!p.train) {
        mu = gate ? *p.hc_rm : *p.cw_rm;
        const float rv = gate ? *p.hc_rv : *p.cw_rv;
        rstd = 1.0f / sqrtf(rv + 1e-5f);
        var_out = rv;
        return;
    }
    const int nb = gate ? p.nb2g : p.nb1g;
    const int b0 = (gate ? p.G * p.nb1g : 0) + grp * nb;
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < nb; i += NT) { a += p.part[2 * (b0 + i)]; b += p.part[2 * (b0 + i) + 1]; }
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    const double m = a / (double)eg;
    const double var = fmax(b / (double)eg - m * m, 0.0);                 // biased, as BatchNorm normalises
    mu = (float)m;
    rstd = (float)(1.0 / sqrt(var + 1e-5));
    var_out = var;
}

__global__ __launch_bounds__(NT) void gm_norm_kernel(const GateP p) {
    __shared__ double sh[4];
    int gate, grp;
    long e0, eg;
    locate(p, gate, grp, e0, eg);
    float mu, rstd;
    double var;
    bn_stats(p, gate, grp, eg, sh, mu, rstd, var);
    const float g = gate ? *p.hc_g : *p.cw_g, b = gate ? *p.hc_b : *p.cw_b;
    const long e = e0 + threadIdx.x;
    if (e < eg) {
        const long idx = (long)grp * eg + e;
        const float t = (gate ? p.t2 : p.t1)[idx];
        const float v = (t - mu) * rstd * g + b;
        if (!gate) p.g1[idx] = v;
        else {                                                            // t2 [n,c,w] -> g2 [n,w,c]
            const int w = idx % p.W, c = (idx / p.W) % p.C;
            const long n = idx / ((long)p.C * p.W);
            p.g2[(n * p.W + w) * p.C + c] = v;
        }
    }
    if (e0 == 0 && threadIdx.x == 0) {
        p.bnstat[(gate * p.G + grp) * 2] = mu;
        p.bnstat[(gate * p.G + grp) * 2 + 1] = rstd;
    }
}

// the running buffers move once per group, in group order (momentum 0.01, unbiased variance): one block per gate, after gm_norm
__global__ __launch_bounds__(NT) void gm_running_kernel(const GateP p) {
    __shared__ double sh[4];
    const int gate = blockIdx.x;
    const long eg = (long)p.B * (gate ? (long)p.C * p.W : (long)p.H * p.C);
    float* rm = gate ? p.hc_rm : p.cw_rm;
    float* rv = gate ? p.hc_rv : p.cw_rv;
    for (int grp = 0; grp < p.G; ++grp) {
        float mu, rstd;
        double var;
        bn_stats(p, gate, grp, eg, sh, mu, rstd, var);
        if (threadIdx.x == 0) {
            const float unb = (float)(var * ((double)eg / (double)(eg > 1 ? eg - 1 : 1)));
            *rm = *rm * (1.0f - 0.01f) + 0.01f * mu;
            *rv = *rv * (1.0f - 0.01f) + 0.01f * unb;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(128) void gm_se_kernel(const GateP p) {
    __shared__ float m[128], h[32];
    const int n = blockIdx.x, c = threadIdx.x, C = p.C, R = p.C / 4;
    if (c < C) m[c] = p.mean[(long)n * C + c];
    __syncthreads();
    if (c < R) {
        float a = p.se_b1[c];
        for (int k = 0; k < C; ++k) a += p.se_w1[c * C + k] * m[k];
        a = fmaxf(a, 0.f);
        h[c] = a;
        p.hid[(long)n * R + c] = a;
    }
    __syncthreads();
    if (c < C) {
        float a = p.se_b2[c];
        for (int k = 0; k < R; ++k) a += p.se_w2[c * R + k] * h[k];
        p.s[(long)n * C + c] = 1.0f / (1.0f + expf(-a));
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dg_at(const GateP& p, int gate, long idx) {       // the gate's output gradient at element idx of t1 / t2
    if (!gate) return p.dg1[idx];
    const int w = idx % p.W, c = (idx / p.W) % p.C;
    const long n = idx / ((long)p.C * p.W);
    return p.dg2[(n * p.W + w) * p.C + c];
}

__global__ __launch_bounds__(NT) void gm_bwd_reduce_kernel(const GateP p) {          // partial sums of dg and dg * xhat
    __shared__ double sh[4];
    int gate, grp;
    long e0, eg;
    locate(p, gate, grp, e0, eg);
    const float mu = p.bnstat[(gate * p.G + grp) * 2], rstd = p.bnstat[(gate * p.G + grp) * 2 + 1];
    const long e = e0 + threadIdx.x;
    double a = 0, b = 0;
    if (e < eg) {
        const long idx = (long)grp * eg + e;
        const float d = dg_at(p, gate, idx);
        a = d;
        b = (double)d * (double)(((gate ? p.t2 : p.t1)[idx] - mu) * rstd);
    }
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    if (threadIdx.x == 0) { p.part[2 * blockIdx.x] = a; p.part[2 * blockIdx.x + 1] = b; }
}

__global__ __launch_bounds__(NT) void gm_bwd_dt_kernel(const GateP p) {              // BatchNorm backward: dt = dL/d(conv output)
    __shared__ double sh[4];
    int gate, grp;
    long e0, eg;
    locate(p, gate, grp, e0, eg);
    const int nb = gate ? p.nb2g : p.nb1g;
    const int b0 = (gate ? p.G * p.nb1g : 0) + grp * nb;
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < nb; i += NT) { a += p.part[2 * (b0 + i)]; b += p.part[2 * (b0 + i) + 1]; }
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    const float mu = p.bnstat[(gate * p.G + grp) * 2], rstd = p.bnstat[(gate * p.G + grp) * 2 + 1];
    const float g = gate ? *p.hc_g : *p.cw_g;
    const float mdg = p.train ? (float)(a / (double)eg) : 0.f, mdx = p.train ? (float)(b / (double)eg) : 0.f;
    const long e = e0 + threadIdx.x;
    if (e < eg) {
        const long idx = (long)grp * eg + e;
        const float d = dg_at(p, gate, idx);
        const float xh = ((gate ? p.t2 : p.t1)[idx] - mu) * rstd;
        (gate ? p.dt2 : p.dt1)[idx] = g * rstd * (d - mdg - xh * mdx);
    }
    // dgamma = sum over everything of dg * xhat, dbeta = sum of dg: the first block of a gate adds the groups' sums, in group order
    if (blockIdx.x == (gate ? p.G * p.nb1g : 0)) {
        const int se_len = (p.C / 4) * p.C + p.C / 4 + p.C * (p.C / 4) + p.C;
        double ga = 0, gb = 0;
        for (int gq = 0; gq < p.G; ++gq) {
            double a2 = 0, b2 = 0;
            const int bq = (gate ? p.G * p.nb1g : 0) + gq * nb;
            for (int i = threadIdx.x; i < nb; i += NT) { a2 += p.part[2 * (bq + i)]; b2 += p.part[2 * (bq + i) + 1]; }
            gb += block_sum(a2, sh);
            ga += block_sum(b2, sh);
        }
        if (threadIdx.x == 0) {
            float* o = p.dprm + se_len + (gate ? NW1 + 2 : 0);
            o[gate ? NW2 : NW1] = (float)ga;
            o[(gate ? NW2 : NW1) + 1] = (float)gb;
        }
    }
}

__global__ __launch_bounds__(NT) void gm_bwd_dz_kernel(const GateP p) {              // gradients of the four pooled planes
    __shared__ float wsh[NW1];
    int gate, grp;
    long e0, eg;
    locate(p, gate, grp, e0, eg);
    const float* wsrc = gate ? p.hc_w : p.cw_w;
    for (int i = threadIdx.x; i < (gate ? NW2 : NW1); i += NT) wsh[i] = wsrc[i];
    __syncthreads();
    const long e = e0 + threadIdx.x;
    if (e >= eg) return;
    const long idx = (long)grp * eg + e;
    if (!gate) {                                                                     // element (n, y, x) of the H x C planes
        const int x = idx % p.C, y = (idx / p.C) % p.H;
        const long n = idx / ((long)p.C * p.H);
        const float* dt = p.dt1 + n * p.H * p.C;
        float a0 = 0.f, a1 = 0.f;
        for (int dy = 0; dy < K1; ++dy) {
            const int h = y - dy + K1 / 2;
            if (h < 0 || h >= p.H) continue;
            for (int dx = 0; dx < K1; ++dx) {
                const int c = x - dx + K1 / 2;
                if (c < 0 || c >= p.C) continue;
                const float d = dt[h * p.C + c];
                a0 += wsh[dy * K1 + dx] * d;
                a1 += wsh[K1 * K1 + dy * K1 + dx] * d;
            }
        }
        p.d_rowmax[idx] = a0;
        p.d_rowmean[idx] = a1;
    } else {                                                                         // element (n, y = channel, x = column) of the C x W planes
        const int x = idx % p.W, y = (idx / p.W) % p.C;
        const long n = idx / ((long)p.C * p.W);
        const float* dt = p.dt2 + n * p.C * p.W;
        float a0 = 0.f, a1 = 0.f;
        for (int dy = 0; dy < K2; ++dy) {
            const int c = y - dy + K2 / 2;
            if (c < 0 || c >= p.C) continue;
            for (int dx = 0; dx < K2; ++dx) {
                const int w = x - dx + K2 / 2;
                if (w < 0 || w >= p.W) continue;
                const float d = dt[c * p.W + w];
                a0 += wsh[dy * K2 + dx] * d;
                a1 += wsh[K2 * K2 + dy * K2 + dx] * d;
            }
        }
        p.d_colmax[(n * p.W + x) * p.C + y] = a0;                                     // back to [n, w, c]
        p.d_colmean[(n * p.W + x) * p.C + y] = a1;
    }
}

// conv weight gradients: block (weight j, chunk k) sums dt * (shifted input) over the elements e = k (mod WCH)-th slice
__global__ __launch_bounds__(NT) void gm_bwd_dw_kernel(const GateP p) {
    __shared__ double sh[4];
    const int j = blockIdx.x, k = blockIdx.y;
    const int gate = j >= NW1;
    const int jj = gate ? j - NW1 : j;
    const int KK = gate ? K2 : K1;
    const int ch = jj / (KK * KK), dy = (jj / KK) % KK, dx = jj % KK;
    const long tot = (long)p.N * (gate ? (long)p.C * p.W : (long)p.H * p.C);
    const long per = (tot + WCH - 1) / WCH;
    const long lo = (long)k * per, hi = lo + per < tot ? lo + per : tot;
    double acc = 0;
    for (long idx = lo + threadIdx.x; idx < hi; idx += NT) {
        float z = 0.f;
        if (!gate) {
            const int c = idx % p.C, h = (idx / p.C) % p.H;
            const long n = idx / ((long)p.C * p.H);
            const int y = h + dy - K1 / 2, x = c + dx - K1 / 2;
            if (y >= 0 && y < p.H && x >= 0 && x < p.C) z = (ch ? p.rowmean : p.rowmax)[(n * p.H + y) * p.C + x];
            acc += (double)(p.dt1[idx] * z);
        } else {
            const int w = idx % p.W, c = (idx / p.W) % p.C;
            const long n = idx / ((long)p.C * p.W);
            const int y = c + dy - K2 / 2, x = w + dx - K2 / 2;
            if (y >= 0 && y < p.C && x >= 0 && x < p.W) z = (ch ? p.colmean : p.colmax)[(n * p.W + x) * p.C + y];
            acc += (double)(p.dt2[idx] * z);
        }
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) p.wpart[j * WCH + k] = acc;
}

// SE backward per sample: d_mean and the sample's contributions to the four SE parameter gradients
__global__ __launch_bounds__(128) void gm_bwd_se_kernel(const GateP p) {
    __shared__ float dz2[128], dz1[32], m[128], h[32];
    const int n = blockIdx.x, c = threadIdx.x, C = p.C, R = p.C / 4;
    const int se_len = R * C + R + C * R + C;
    float* o = p.separt + (long)n * se_len;
    if (c < C) {
        const float s = p.s[(long)n * C + c];
        dz2[c] = p.ds[(long)n * C + c] * s * (1.0f - s);
        m[c] = p.mean[(long)n * C + c];
    }
    if (c < R) h[c] = p.hid[(long)n * R + c];
    __syncthreads();
    if (c < R) {
        float a = 0.f;
        for (int k = 0; k < C; ++k) a += p.se_w2[k * R + c] * dz2[k];
        dz1[c] = h[c] > 0.f ? a : 0.f;
    }
    __syncthreads();
    if (c < C) {
        float a = 0.f;
        for (int k = 0; k < R; ++k) a += p.se_w1[k * C + c] * dz1[k];
        p.d_mean[(long)n * C + c] = a;
        for (int k = 0; k < R; ++k) o[R * C + R + c * R + k] = dz2[c] * h[k];        // d se_w2 [C][R]
        o[R * C + R + C * R + c] = dz2[c];                                           // d se_b2
        for (int k = 0; k < R; ++k) o[k * C + c] = dz1[k] * m[c];                     // d se_w1 [R][C]
    }
    if (c < R) o[R * C + c] = dz1[c];                                                // d se_b1
}

// sums over the samples (SE parameters) and over the element chunks (conv weights), in index order
__global__ __launch_bounds__(NT) void gm_bwd_final_kernel(const GateP p) {
    const int R = p.C / 4, se_len = R * p.C + R + p.C * R + p.C;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i < se_len) {
        double a = 0;
        for (int n = 0; n < p.N; ++n) a += (double)p.separt[(long)n * se_len + i];
        p.dprm[i] = (float)a;
    } else if (i < se_len + NW1 + NW2) {
        const int j = i - se_len;
        double a = 0;
        for (int k = 0; k < WCH; ++k) a += p.wpart[j * WCH + k];
        p.dprm[se_len + (j < NW1 ? j : j + 2)] = (float)a;                           // layout: cw_w [98] cw_g cw_b hc_w [50] hc_g hc_b
    }
}

int fill(GateP& p, int N, int G, int H, int W, int C) {
    p.N = N; p.G = G; p.B = N / G; p.H = H; p.W = W; p.C = C;
    p.nb1g = cdiv((int64_t)p.B * H * C, NT);
    p.nb2g = cdiv((int64_t)p.B * C * W, NT);
    return G * (p.nb1g + p.nb2g);
}

}  // namespace

// saved: t1 [N H C] | t2 [N C W] | hid [N C/4] | bnstat [2 G 2]       ws (doubles): part [blocks][2] | wpart [148][WCH]; then floats: separt, dt1, dt2
extern "C" int64_t spei_gate_train_saved_floats(int N, int G, int H, int W, int C) {
    return (int64_t)N * H * C + (int64_t)N * C * W + (int64_t)N * (C / 4) + 4 * G;
}
extern "C" int64_t spei_gate_train_ws_floats(int N, int G, int H, int W, int C) {
    GateP p;
    const int64_t blocks = fill(p, N, G, H, W, C);
    const int se_len = (C / 4) * C + C / 4 + C * (C / 4) + C;
    return 2 * (2 * blocks + (int64_t)(NW1 + NW2) * WCH) + (int64_t)N * se_len + (int64_t)N * H * C + (int64_t)N * C * W + 16;
}
extern "C" int spei_gate_train_nparams(int C) { return (C / 4) * C + C / 4 + C * (C / 4) + C + NW1 + 2 + NW2 + 2; }

static int gate_common(GateP& p, const float* rowmax, const float* rowmean, const float* colmax, const float* colmean, const float* mean,
                       const float* const* prm, float* const* run, int N, int G, int H, int W, int C, int train, float* s, float* saved,
                       float* ws) {
    SPEI_REQUIRE(rowmax && rowmean && colmax && colmean && mean && prm && run && s && saved && ws, "spei_gate_maps: null pointer");
    SPEI_REQUIRE(N > 0 && G > 0 && N % G == 0 && H > 0 && W > 0 && (C == 32 || C == 64 || C == 128), "spei_gate_maps: N=%d G=%d %dx%d C=%d", N, G, H, W, C);
    SPEI_REQUIRE((uintptr_t)ws % 8 == 0, "spei_gate_maps: workspace must be 8-byte aligned");
    const int blocks = fill(p, N, G, H, W, C);
    p.rowmax = rowmax; p.rowmean = rowmean; p.colmax = colmax; p.colmean = colmean; p.mean = mean;
    p.se_w1 = prm[0]; p.se_b1 = prm[1]; p.se_w2 = prm[2]; p.se_b2 = prm[3];
    p.cw_w = prm[4]; p.cw_g = prm[5]; p.cw_b = prm[6]; p.hc_w = prm[7]; p.hc_g = prm[8]; p.hc_b = prm[9];
    p.cw_rm = run[0]; p.cw_rv = run[1]; p.hc_rm = run[2]; p.hc_rv = run[3];
    p.train = train; p.s = s;
    p.t1 = saved; p.t2 = p.t1 + (int64_t)N * H * C; p.hid = p.t2 + (int64_t)N * C * W; p.bnstat = p.hid + (int64_t)N * (C / 4);
    p.part = reinterpret_cast<double*>(ws);
    p.wpart = p.part + 2 * (int64_t)blocks;
    const int se_len = (C / 4) * C + C / 4 + C * (C / 4) + C;
    p.separt = reinterpret_cast<float*>(p.wpart + (int64_t)(NW1 + NW2) * WCH);
    p.dt1 = p.separt + (int64_t)N * se_len; p.dt2 = p.dt1 + (int64_t)N * H * C;
    return blocks;
}

// prm: 10 pointers se_w1, se_b1, se_w2, se_b2, cw_w, cw_g, cw_b, hc_w, hc_g, hc_b; run: 4 pointers cw_rm, cw_rv, hc_rm, hc_rv
extern "C" int spei_gate_maps_fwd(const float* rowmax, const float* rowmean, const float* colmax, const float* colmean, const float* mean,
                                  const float* const* prm, float* const* run, int N, int groups, int H, int W, int C, int bn_train,
                                  int update_running, float* s, float* g1, float* g2, float* saved, float* ws, spei_stream_t stream) {
    GateP p = {};
    const int blocks = gate_common(p, rowmax, rowmean, colmax, colmean, mean, prm, run, N, groups, H, W, C, bn_train, s, saved, ws);
    if (blocks < 0) return blocks;
    SPEI_REQUIRE(g1 && g2, "spei_gate_maps_fwd: null output");
    p.g1 = g1; p.g2 = g2; p.update = update_running;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gm_se_kernel, dim3(N), dim3(128), 0, st, p);
    hipLaunchKernelGGL(gm_conv_kernel, dim3(blocks), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(gm_norm_kernel, dim3(blocks), dim3(NT), 0, st, p);
    if (bn_train && update_running) hipLaunchKernelGGL(gm_running_kernel, dim3(2), dim3(NT), 0, st, p);
    SPEI_CHECK_LAUNCH("spei_gate_maps_fwd");
    return 0;
}

// d_* outputs shaped like the statistics; dprm [spei_gate_train_nparams(C)]: se_w1 | se_b1 | se_w2 | se_b2 | cw_w | cw_g | cw_b | hc_w | hc_g | hc_b
extern "C" int spei_gate_maps_bwd(const float* rowmax, const float* rowmean, const float* colmax, const float* colmean, const float* mean,
                                  const float* const* prm, float* const* run, int N, int groups, int H, int W, int C, int bn_train,
                                  const float* s, const float* saved, const float* ds, const float* dg1, const float* dg2, float* d_rowmax,
                                  float* d_rowmean, float* d_colmax, float* d_colmean, float* d_mean, float* dprm, float* ws,
                                  spei_stream_t stream) {
    GateP p = {};
    const int blocks = gate_common(p, rowmax, rowmean, colmax, colmean, mean, prm, run, N, groups, H, W, C, bn_train, const_cast<float*>(s),
                                   const_cast<float*>(saved), ws);
    if (blocks < 0) return blocks;
    SPEI_REQUIRE(ds && dg1 && dg2 && d_rowmax && d_rowmean && d_colmax && d_colmean && d_mean && dprm, "spei_gate_maps_bwd: null pointer");
    p.ds = ds; p.dg1 = dg1; p.dg2 = dg2;
    p.d_rowmax = d_rowmax; p.d_rowmean = d_rowmean; p.d_colmax = d_colmax; p.d_colmean = d_colmean; p.d_mean = d_mean; p.dprm = dprm;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gm_bwd_reduce_kernel, dim3(blocks), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(gm_bwd_dt_kernel, dim3(blocks), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(gm_bwd_dz_kernel, dim3(blocks), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(gm_bwd_dw_kernel, dim3(NW1 + NW2, WCH), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(gm_bwd_se_kernel, dim3(N), dim3(128), 0, st, p);
    const int se_len = (C / 4) * C + C / 4 + C * (C / 4) + C;
    hipLaunchKernelGGL(gm_bwd_final_kernel, dim3(cdiv(se_len + NW1 + NW2, NT)), dim3(NT), 0, st, p);
    SPEI_CHECK_LAUNCH("spei_gate_maps_bwd");
    return 0;
}

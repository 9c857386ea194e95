// K7 LayerNorm(256) and K8 the 5x5 cross-window attention core of the reference's SwinIR variant
// (model/swinir.py:115-149 WindowAttention.forward, :215-236 calculate_mask, :238-281 block forward).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// LayerNorm over 256 channels: one wave per token, 4 channels per lane, two-pass moments in registers.
// ---------------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void layernorm256_kernel(const float* __restrict__ x, TO* __restrict__ y,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int64_t M) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gamma) g = reinterpret_cast<const float4*>(gamma)[lane];
    if (beta) b = reinterpret_cast<const float4*>(beta)[lane];
    for (int64_t m = wave; m < M; m += nwaves) {
        const float4 v = reinterpret_cast<const float4*>(x + m * 256)[lane];
        const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        float4 o;
        o.x = dx * rstd * g.x + b.x;
        o.y = dy * rstd * g.y + b.y;
        o.z = dz * rstd * g.z + b.z;
        o.w = dw * rstd * g.w + b.w;
        if (sizeof(TO) == 4) {
            reinterpret_cast<float4*>(y + m * 256)[lane] = o;
        } else {
            typename lpv<TO>::x4 h;
            h[0] = (TO)o.x; h[1] = (TO)o.y; h[2] = (TO)o.z; h[3] = (TO)o.w;
            reinterpret_cast<typename lpv<TO>::x4*>(y + m * 256)[lane] = h;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Window attention.  Two 256-thread workgroups per 5x5 window, one wave per head (8 heads x 32 dims).
// Tokens are padded 25 -> 32 so both products run on v_mfma_f32_32x32x2_f32:
//   S^T[key][query] = K Q^T          (A = K rows, B = Q rows; the query index lands on the MFMA lane)
//   softmax over keys = over the 16 accumulator registers of a lane + one cross-half shuffle
//   O[query][d]      = P V           (the S^T accumulators ARE the A operand, no data movement)
// The cyclic shift and window partition/reverse are pure index arithmetic on the token -> pixel map.
// ---------------------------------------------------------------------------------------------------
constexpr int WS = 5, NT = 25, HD = 32, LDT = HD + 1;

__device__ __forceinline__ int mask_region(int v, int n, int shift) {
    // img_mask slices (0,-ws), (-ws,-shift), (-shift,None) on the SHIFTED frame (swinir.py:219-230)
    return v < n - WS ? 0 : (v < n - shift ? 1 : 2);
}

template <typename T>
__device__ __forceinline__ float4 load4(const T* p) {                 // fp32, bf16 or half in HBM
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const float4*>(p);
    } else {
        const typename lpv<T>::x4 h = *reinterpret_cast<const typename lpv<T>::x4*>(p);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void window_attention_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                               const float* __restrict__ relbias, T* __restrict__ out,
                                                               int H, int W, int shift) {
    __shared__ float sQ[4][32][LDT], sK[4][32][LDT], sV[4][32][LDT];
    __shared__ int tok_pix[32], tok_reg[32];
    {   // blockIdx.y = sample of a batch of equally sized maps stored one after the other
        const size_t z = (size_t)blockIdx.y * H * W;
        q += z * 256; kv += z * 512; out += z * 256;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nwx = W / WS;
    const int win = blockIdx.x >> 1;
    const int wy = win / nwx, wx = win - wy * nwx;
    if (threadIdx.x < 32) {
        int pix = 0, reg = 0;
        if (threadIdx.x < NT) {
            const int ys = wy * WS + threadIdx.x / WS, xs = wx * WS + threadIdx.x % WS;   // shifted-frame coords
            int yo = ys + shift, xo = xs + shift;                                         // roll(-shift): shifted[y] = x[(y+shift)%H]
            if (yo >= H) yo -= H;
            if (xo >= W) xo -= W;
            pix = yo * W + xo;
            reg = shift > 0 ? 3 * mask_region(ys, H, shift) + mask_region(xs, W, shift) : 0;
        }
        tok_pix[threadIdx.x] = pix;
        tok_reg[threadIdx.x] = reg;
    }
    __syncthreads();
    const int h = (blockIdx.x & 1) * 4 + wave;   // global head
    const int hl = wave;                         // LDS slot
    // stage Q, K, V of this head: 32 rows x 8 float4 (rows >= 25 zero)
    for (int i = lane; i < 32 * 8; i += 64) {
        const int r = i >> 3, c4 = (i & 7) * 4;
        float4 vq = make_float4(0.f, 0.f, 0.f, 0.f), vk = vq, vv = vq;
        if (r < NT) {
            const size_t p = (size_t)tok_pix[r];
            vq = load4<T>(q + p * 256 + h * HD + c4);
            vk = load4<T>(kv + p * 512 + h * HD + c4);
            vv = load4<T>(kv + p * 512 + 256 + h * HD + c4);
        }
        float* d = &sQ[hl][r][c4]; d[0] = vq.x; d[1] = vq.y; d[2] = vq.z; d[3] = vq.w;
        d = &sK[hl][r][c4];        d[0] = vk.x; d[1] = vk.y; d[2] = vk.z; d[3] = vk.w;
        d = &sV[hl][r][c4];        d[0] = vv.x; d[1] = vv.y; d[2] = vv.z; d[3] = vv.w;
    }
    __syncthreads();
    const int fr = lane & 31, fk = lane >> 5;
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < HD; kk += 2)
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[hl][fr][kk + fk], sQ[hl][fr][kk + fk], st, 0, 0, 0);
    // lane holds query = fr, keys = (r&3) + 8*(r>>2) + 4*fk
    const int qi = fr < NT ? fr : 0;
    const int qreg = tok_reg[qi];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
        float v = -INFINITY;
        if (key < NT) {
            v = st[r] + relbias[(h * NT + qi) * NT + key];
            if (shift > 0 && tok_reg[key] != qreg) v += -100.0f;
        }
        st[r] = v;
        mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float e = expf(st[r] - mx);     // exp(-inf) = 0 for the padded keys
        st[r] = e;
        sum += e;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * fk;
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r] * inv, sV[hl][key][fr], o, 0, 0, 0);
    }
    // o: col d = fr, row query = (r&3) + 8*(r>>2) + 4*fk
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (qq < NT) out[(size_t)tok_pix[qq] * 256 + h * HD + fr] = (T)o[r];
    }
}

}  // namespace

extern "C" int spei_layernorm256(const float* x, void* y, int out_fmt, const float* gamma, const float* beta, int64_t M,
                                 spei_stream_t stream) {
    SPEI_REQUIRE(x && y && M > 0, "spei_layernorm256: bad arguments");
    SPEI_REQUIRE(out_fmt == SPEI_F32 || out_fmt == SPEI_BF16 || out_fmt == SPEI_F16, "spei_layernorm256: out_fmt=%d", out_fmt);
    const int64_t blocks = (M + 3) / 4;
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192));
    if (out_fmt == SPEI_BF16) hipLaunchKernelGGL(layernorm256_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, x, (__bf16*)y, gamma, beta, M);
    else if (out_fmt == SPEI_F16) hipLaunchKernelGGL(layernorm256_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, x, (_Float16*)y, gamma, beta, M);
    else hipLaunchKernelGGL(layernorm256_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (float*)y, gamma, beta, M);
    SPEI_CHECK_LAUNCH("spei_layernorm256");
    return 0;
}

static int window_attention_run(const void* q, const void* kv, int io_fmt, const float* relbias, void* out, int H, int W, int shift, int batch,
                                spei_stream_t stream) {
    SPEI_REQUIRE(q && kv && relbias && out, "spei_window_attention: null pointer");
    SPEI_REQUIRE(batch >= 1 && batch <= 65535, "spei_window_attention: batch=%d", batch);
    SPEI_REQUIRE(H > 0 && W > 0 && H % WS == 0 && W % WS == 0, "spei_window_attention: %dx%d is not a multiple of the 5x5 window", H, W);
    SPEI_REQUIRE(shift >= 0 && shift < WS, "spei_window_attention: shift=%d", shift);
    SPEI_REQUIRE(io_fmt == SPEI_F32 || io_fmt == SPEI_BF16 || io_fmt == SPEI_F16, "spei_window_attention: io_fmt=%d", io_fmt);
    const dim3 grid(2 * (H / WS) * (W / WS), batch);
    if (io_fmt == SPEI_BF16) hipLaunchKernelGGL(window_attention_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)q, (const __bf16*)kv, relbias, (__bf16*)out, H, W, shift);
    else if (io_fmt == SPEI_F16) hipLaunchKernelGGL(window_attention_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)q, (const _Float16*)kv, relbias, (_Float16*)out, H, W, shift);
    else hipLaunchKernelGGL(window_attention_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)q, (const float*)kv, relbias, (float*)out, H, W, shift);
    SPEI_CHECK_LAUNCH("spei_window_attention");
    return 0;
}

extern "C" int spei_window_attention(const void* q, const void* kv, int io_fmt, const float* relbias, void* out, int H, int W,
                                     int shift, spei_stream_t stream) {
    return window_attention_run(q, kv, io_fmt, relbias, out, H, W, shift, 1, stream);
}

extern "C" int spei_window_attention_batched(const void* q, const void* kv, int io_fmt, const float* relbias, void* out, int H, int W,
                                             int shift, int batch, spei_stream_t stream) {
    return window_attention_run(q, kv, io_fmt, relbias, out, H, W, shift, batch, stream);
}

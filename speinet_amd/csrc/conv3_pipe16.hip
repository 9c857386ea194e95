// 3x3 convolution, 256 -> 256 channels, on the fp32 token maps of the SwinIR body (reference model/swinir.py:467 RSTB.conv — the
// `conv(blocks(x)) + x` tail of every residual group, :483-484 — and :742 conv_after_body; stride 1, padding 1) on the gfx950 16-bit
// matrix pipe, as a PERSISTENT, software-pipelined kernel.
//
// Why a third conv kernel: in conv_slab_kernel a 64-pixel workgroup streams 590 KB of weight fragments for each 128-channel half of
// the output (9 taps x 256 x 256 weights: 18 KB of L2 -> CU traffic per pixel at ~70 GB/s per CU), stages a 141 KB fp32 slab first and
// runs alone on its CU: 190 us per launch for the frame's two stacked maps, 30 % matrix-pipe busy, the largest single item of the conv
// family (2.0 ms per frame).  Here one workgroup per CU walks 6 x 16 pixel tiles (tile = blockIdx.x + k gridDim.x), the structure of
// mlp_pipe_kernel (mlp_fused16.hip):
//   * eight waves, wave w = output channels [32 w, 32 w + 32) of ALL 96 pixels: a weight fragment (1 KiB from L2, through an 8-deep ring
//     of buffer loads that runs on from tile to tile) feeds three MFMAs, 12.3 KB of intake per pixel instead of 18;
//   * the tile's 8 x 18 pixel halo lives in LDS as 16-bit rows (76 KB), the NEXT tile's rows are loaded, converted and written to the
//     second slab in three chunks inside the 144 k-steps of the current tile; a tap is an immediate offset of the fragment read;
//   * results leave as one dword per lane in the accumulators' own layout (lanes 0-31 of a register = 32 consecutive channels of a pixel),
//     bias and residual added in registers; one barrier per tile.
// fp32 accumulation over (tap, k-step) with the k-steps rotated per tile; the same operand rounding as conv_slab_kernel, another
// summation order.  Maps whose height is not a multiple of 6 or width not a multiple of 16 stay on conv_slab_kernel (ops.py).
#include "common.h"
#include <type_traits>

namespace {

constexpr int C = 256;                          // input = output channels
constexpr int TH = 6, TW = 16;                   // output tile: 96 pixels = 3 MFMA row tiles of 2 x 16 pixels
constexpr int SH = TH + 2, SW = TW + 2, SP = SH * SW;      // slab: 8 x 18 = 144 pixels
constexpr int PA = 2 * C + 16;                  // LDS row pitch (bytes): conflict-free ds_read_b128 of 32 consecutive rows
constexpr int SLAB = SP * PA;                   // 76 032 bytes
constexpr int NSTEP = 9 * 16;                   // k-steps per tile: tap-major
constexpr int RG = 8;                           // weight fragments in flight per wave

// compile-time loop: `#pragma unroll` gives up on 144 steps of this size (the unroll threshold), and a loop left rolled turns every ring /
// fragment index into dynamic register indexing (scratch, waterfall loops: 59 000 lines of ISA)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <typename LP>
struct C3Params {
    const float* x;         // [batch][H*W][256]
    const LP* wfrag;        // fragment order [8 n-tiles][9 taps][16 k-steps][64][8]
    const float* bias;      // [256] or null
    const float* res;       // [batch][H*W][256] or null (may alias out)
    float* out;             // [batch][H*W][256]
    int H, W, batch, tiles_x, tiles_map, ntiles;
    long long* stamps;      // tuning build: phase stamps of every workgroup's SECOND tile (steady state), else NULL
};

template <typename LP>
__global__ __launch_bounds__(512) void conv3_pipe_kernel(const C3Params<LP> p) {
    typedef typename lpv<LP>::x8 lp8;
    typedef typename lpv<LP>::x4 lp4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int fr = lane & 31, fk = lane >> 5;               // (not const: laundered at the top of every tile, see mlp / attention pipe kernels)
    int l16 = tid & 15, rsub = tid >> 4;              // staging: 16 lanes per pixel, 32 pixels per pass
    const int HW = p.H * p.W;
    const int G = gridDim.x;

    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<LP*>(p.wfrag), 0, 9 * C * C * 2, 0x00020000);
    int lane16 = lane * 16;
    const int wbase = wave * NSTEP;                   // the wave's first fragment
    // fragment q of a tile's weight stream: tap q / 16, k-step (rot + q) & 15
    auto wload = [&](int q, int rot) -> lp8 {
        return __builtin_bit_cast(lp8, __builtin_amdgcn_raw_buffer_load_b128(rsw, lane16, (wbase + (q & ~15) + ((rot + q) & 15)) * 1024, 0));
    };
    float bias = p.bias ? p.bias[wave * 32 + fr] : 0.f;

    // ---- staging of a tile's halo: pass ps = slab pixels [32 ps, 32 ps + 32), fp32 rows -> 16-bit rows rotated by rb bytes; pixels
    // outside the map are zeros (padding 1) ----
    f32x4 xr[2][4];
    auto tile_origin = [&](int t, int& bmap, int& y0, int& x0) {
        bmap = __builtin_amdgcn_readfirstlane(t / p.tiles_map);
        const int r = t - bmap * p.tiles_map;
        const int ty = __builtin_amdgcn_readfirstlane(r / p.tiles_x);
        y0 = ty * TH;
        x0 = (r - ty * p.tiles_x) * TW;
    };
    auto stage_load = [&](int t, int ps, int slot) {
        int bmap, y0, x0;
        tile_origin(t, bmap, y0, x0);
        const int sp = min(ps * 32 + rsub, SP - 1);
        const int sy = sp / SW, sx = sp - sy * SW;
        const int gy = min(max(y0 - 1 + sy, 0), p.H - 1), gx = min(max(x0 - 1 + sx, 0), p.W - 1);
        const float* src = p.x + ((size_t)bmap * HW + (size_t)gy * p.W + gx) * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[slot][j] = reinterpret_cast<const f32x4*>(src)[l16 + 16 * j];
    };
    auto stage_write = [&](int t, int ps, int slot, unsigned char* dst, int rb) {
        int bmap, y0, x0;
        tile_origin(t, bmap, y0, x0);
        const int sp = ps * 32 + rsub;
        const int spc = min(sp, SP - 1);
        const int sy = spc / SW, sx = spc - sy * SW;
        const int gy = y0 - 1 + sy, gx = x0 - 1 + sx;
        const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        unsigned char* const row = dst + spc * PA;    // pass 4 holds 16 pixels: its upper lanes rewrite pixel 143 with the same values
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lp4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = to_lp<LP>(inside ? xr[slot][j][e] : 0.f);
            *reinterpret_cast<lp4*>(row + (((l16 + 16 * j) * 8 - rb) & 511)) = hv;
        }
        (void)sp;
    };

    int tile = blockIdx.x;
    // the K rotation of a tile follows its index WITHIN its map: a map's result does not depend on its place in the batch
    auto tile_rot = [&](int t) { return (t % p.tiles_map) & 15; };
    int rot = __builtin_amdgcn_readfirstlane(tile_rot(tile));
    // ---- prologue: the first tile's halo -> slab 0; the ring's first fragments ----
    lp8 ring[RG];
#pragma unroll
    for (int d = 0; d < RG; ++d) ring[d] = wload(d, rot);
#pragma unroll
    for (int ps = 0; ps < 5; ++ps) {
        stage_load(tile, ps, 0);
        stage_write(tile, ps, 0, smem, rot * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();

    int cur = 0;
    int nth = 0;
    for (; tile < p.ntiles; tile += G, ++nth) {
        if (nth == 1) { SPEI_STAMP(p.stamps, 0); SPEI_STAMP_CLK(p.stamps, 8); }
        asm volatile("" : "+v"(tid), "+v"(fr), "+v"(fk), "+v"(l16), "+v"(rsub), "+v"(lane16), "+v"(bias));
        const unsigned char* const sa = smem + (cur ? SLAB : 0);          // this tile's halo
        unsigned char* const sn = smem + (cur ? 0 : SLAB);                // the next tile's
        const int tnx = tile + G;
        const int rotn = __builtin_amdgcn_readfirstlane(tile_rot(tnx));
        const int tnc = tnx < p.ntiles ? tnx : tile;                      // a tile past the end re-stages this one (never read)
        int bmap, y0, x0;
        tile_origin(tile, bmap, y0, x0);
        const size_t moff = (size_t)bmap * HW * C;
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + moff, 0, HW * (C * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res + moff : p.out + moff), 0,
                                                                             p.res ? HW * (C * 4) : 0, 0x00020000);
        // register 4 k + e of row tile i <-> pixel (y0 + 2 i + (k >> 1), x0 + 8 (k & 1) + 4 fk + e), channel 32 wave + fr: the lane's part
        // of the byte offset in the vector register, the pixel's in the scalar offset
        const int voff = fk * 4096 + fr * 4;
        const int soff0 = (y0 * p.W + x0) * 1024 + wave * 128;

        f32x16 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = bias;
        f32x16 res[3];
        // per row tile i: slab row of the lane's pixel (tile row 2 i + (fr >> 4), column fr & 15) for tap (0, 0)
        int abase[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) abase[i] = ((2 * i + (fr >> 4)) * SW + (fr & 15)) * PA + fk * 16;

        lp8 tn[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) tn[i] = *reinterpret_cast<const lp8*>(sa + abase[i]);
        static_for<0, NSTEP>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            lp8 tc[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) tc[i] = tn[i];
            if constexpr (q + 1 < NSTEP) {
                constexpr int tap = (q + 1) >> 4, s = (q + 1) & 15;
                constexpr int toff = ((tap / 3) * SW + tap % 3) * PA + s * 32;
#pragma unroll
                for (int i = 0; i < 3; ++i) tn[i] = *reinterpret_cast<const lp8*>(sa + abase[i] + toff);
            }
            const lp8 w = ring[q % RG];
            ring[q % RG] = q + RG < NSTEP ? wload(q + RG, rot) : wload(q + RG - NSTEP, rotn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = mfma16(tc[i], w, acc[i]);
            // ---- fillers: the next tile's halo in three chunks (loads as a burst, the conversion 40 steps later); this tile's residual
            // rows near the end (vector-memory accesses that outlast the ring's eight steps go in bursts: mlp_fused16.hip) ----
            if (q == 0) { stage_load(tnc, 0, 0); stage_load(tnc, 1, 1); }
            if (q == 40) { stage_write(tnc, 0, 0, sn, rotn * 32); if (nth == 1) { SPEI_STAMP(p.stamps, 1); SPEI_STAMP_CLK(p.stamps, 9); } }
            if (q == 41) stage_write(tnc, 1, 1, sn, rotn * 32);
            if (q == 44) { stage_load(tnc, 2, 0); stage_load(tnc, 3, 1); }
            if (q == 84) { stage_write(tnc, 2, 0, sn, rotn * 32); if (nth == 1) { SPEI_STAMP(p.stamps, 2); SPEI_STAMP_CLK(p.stamps, 10); } }
            if (q == 85) stage_write(tnc, 3, 1, sn, rotn * 32);
            if (q == 88) stage_load(tnc, 4, 0);
            if (q == 124) stage_write(tnc, 4, 0, sn, rotn * 32);
            if (q == 128) {
                if (nth == 1) { SPEI_STAMP(p.stamps, 3); SPEI_STAMP_CLK(p.stamps, 11); }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            rsr, voff, soff0 + ((2 * i + (r >> 3)) * p.W + 8 * ((r >> 2) & 1) + (r & 3)) * 1024, 0));
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(acc[i]));
            __builtin_amdgcn_sched_barrier(0);
        });
        if (nth == 1) { SPEI_STAMP(p.stamps, 4); SPEI_STAMP_CLK(p.stamps, 12); }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][r] + res[i][r]), rso, voff,
                                                      soff0 + ((2 * i + (r >> 3)) * p.W + 8 * ((r >> 2) & 1) + (r & 3)) * 1024, 0);
        if (nth == 1) { SPEI_STAMP(p.stamps, 5); SPEI_STAMP_CLK(p.stamps, 13); }
        lds_barrier();                                // the next tile's halo is complete; this tile's slab is free
        if (nth == 1) { SPEI_STAMP(p.stamps, 6); SPEI_STAMP_CLK(p.stamps, 14); }
        cur ^= 1;
        rot = rotn;
    }
}

}  // namespace

template <typename LP>
static int conv3_launch(const float* x, const void* wfrag, const float* bias, const float* res, float* out, int batch, int H, int W,
                        hipStream_t st) {
    C3Params<LP> p;
    p.x = x; p.wfrag = (const LP*)wfrag; p.bias = bias; p.res = res; p.out = out;
    p.H = H; p.W = W; p.batch = batch;
    p.stamps = spei_stamp_buffer();
    p.tiles_x = W / TW; p.tiles_map = p.tiles_x * (H / TH); p.ntiles = p.tiles_map * batch;
    const size_t lds = (size_t)2 * SLAB;
    ensure_dyn_lds<&conv3_pipe_kernel<LP>>(lds);
    const int grid = p.ntiles < spei_num_cus() ? p.ntiles : spei_num_cus();
    hipLaunchKernelGGL((conv3_pipe_kernel<LP>), dim3(grid), dim3(512), lds, st, p);
    SPEI_CHECK_LAUNCH("spei_conv3x3_256_pipe16");
    return 0;
}

extern "C" int spei_conv3x3_256_pipe16(int fmt, const float* x, const void* w_frag, const float* bias, const float* residual, float* out,
                                       int batch, int H, int W, spei_stream_t stream) {
    SPEI_REQUIRE(x && w_frag && out, "spei_conv3x3_256_pipe16: null pointer");
    SPEI_REQUIRE(fmt == SPEI_BF16 || fmt == SPEI_F16, "spei_conv3x3_256_pipe16: fmt=%d", fmt);
    SPEI_REQUIRE(batch >= 1 && H > 0 && W > 0 && H % TH == 0 && W % TW == 0, "spei_conv3x3_256_pipe16: %d maps of %dx%d (height %% 6, width %% 16)",
                 batch, H, W);
    SPEI_REQUIRE((int64_t)H * W <= (1ll << 21), "spei_conv3x3_256_pipe16: map too large for 32-bit byte offsets");
    SPEI_REQUIRE(x != out, "spei_conv3x3_256_pipe16: the input may not alias the output (tiles read their neighbours' pixels)");
    SPEI_REQUIRE(((uintptr_t)x | (uintptr_t)w_frag | (uintptr_t)out | (uintptr_t)residual) % 16 == 0, "spei_conv3x3_256_pipe16: 16-byte alignment required");
    hipStream_t st = (hipStream_t)stream;
    if (fmt == SPEI_F16) return conv3_launch<_Float16>(x, w_frag, bias, residual, out, batch, H, W, st);
    return conv3_launch<__bf16>(x, w_frag, bias, residual, out, batch, H, W, st);
}

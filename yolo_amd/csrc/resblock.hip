// Fused 64-channel residual block for gfx950, bf16: y = x + leaky(conv3x3(leaky(conv1x1(x, w1) + b1), w2) + b2).
// Replaces ResidualBlock.forward after the down-sampling conv (yolo/modeling/backbone/darknet.py:117-126:
// residual = x.clone(); conv1 1x1 C -> C/2, BN, leaky; conv2 3x3 C/2 -> C, BN, leaky; out += residual) for
// C == 64, the full-resolution/2 block of Darknet-53, in one launch.
//
// Why: at 208x208 the block is bound by HBM, not by the matrix cores: as two launches it moves x (read), t
// (written, read back) and x again (residual) plus y.  Here the 32-channel intermediate t of a tile of 8x16
// output pixels (10x18 pixels with the 3x3 halo, 1.4x recompute of a conv that is 1/9 of the block's FLOPs)
// is produced straight from global-memory MFMA operands into LDS and consumed in place; x is read once from
// HBM (halo overlap and the residual re-read hit L2) and y is written once.
//
// Not in place: neighbouring tiles read x's halo pixels, so y must be a different buffer than x.
//
// LDS image of t: one 64-byte row per patch pixel, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 3):
// the ds_read_b128 of 16 consecutive rows (one MFMA operand of a tap) and the producer's ds_write_b128 are
// bank-conflict free.  A wave owns 32 output channels of four rows of the tile; its share of the conv2 weights
// (32 x 288) stays in registers for the whole launch.
#include "common.h"

namespace yolo {

constexpr int RB_TH = 8, RB_TW = 16;          // output tile
constexpr int RB_PH = RB_TH + 2, RB_PW = RB_TW + 2;
constexpr int RB_PIX = RB_PH * RB_PW;         // 180 patch pixels
constexpr int RB_NT = (RB_PIX + 15) / 16;     // 12 producer n-tiles
constexpr int RB_PER = RB_NT / 4;             // per wave
static_assert(RB_NT % 4 == 0, "producer tiles split evenly over the 4 waves");

struct ResBlockParams {
    const char* x;
    char* y;
    long long x_img_stride, x_origin, y_img_stride, y_origin;  // bytes
    int x_row_stride, x_pix_stride, y_row_stride, y_pix_stride;  // bytes
    const bf16_t* w1;  // [>=32][64]
    const float* b1;   // [32]
    const bf16_t* w2;  // [>=64][10 taps][32] (tap 9 is padding)
    const float* b2;   // [64]
    int N, H, W;
    int tiles_x, tiles_y, tiles;
    float slope;
};

#ifndef RB_WGS
#define RB_WGS 2
#endif
__global__ void __launch_bounds__(256, RB_WGS) resblock_c64_kernel(const ResBlockParams p) {
    __shared__ __attribute__((aligned(16))) char patch[RB_NT * 16 * 64];
    __shared__ __attribute__((aligned(16))) bf16x8 w1frag_s[4 * 64];
    __shared__ __attribute__((aligned(16))) float bias_s[32 + 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;

    // conv1 weights: row m of fragment f <-> channel 8*(m>>2) + 4*f + (m&3): after both MFMAs a lane holds 8
    // consecutive channels.  Kept in LDS, re-read per tile (dead during conv2).
    if (wave == 0) {
        const int chb = 8 * (l15 >> 2) + (l15 & 3);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                w1frag_s[(f * 2 + ks) * 64 + lane] = *(const bf16x8*)(p.w1 + (chb + 4 * f) * 64 + ks * 32 + q * 8);
    }
    if (tid < 32) bias_s[tid] = p.b1[tid];
    if (tid >= 64 && tid < 128) bias_s[32 + tid - 64] = p.b2[tid - 64];

    // conv2: wave = 32 channels (wc) x 4 output rows (4 * wp ..) of 16 pixels.  Its weights -- two m-tiles of the same
    // row permutation as conv1's, all nine taps -- stay in registers for the whole launch (18 fragments; a 64-channel
    // wave would need 36 and had to keep a third of them in LDS).
    const int wc = wave & 1, wp = wave >> 1;
    bf16x8 dfrag[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = 32 * wc + 8 * (l15 >> 2) + 4 * i + (l15 & 3);
#pragma unroll
        for (int t = 0; t < 9; ++t) dfrag[i][t] = *(const bf16x8*)(p.w2 + (size_t)ch * 320 + t * 32 + q * 8);
    }
    // pin the fragments in registers (after all loads are issued): otherwise the compiler re-loads them per tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(dfrag[i][t]));

    // tile-independent patch coordinates of this lane's producer pixels
    int ppy[RB_PER], ppx[RB_PER];
#pragma unroll
    for (int i = 0; i < RB_PER; ++i) {
        const int pi = 16 * (wave + 4 * i) + l15;
        ppy[i] = pi / RB_PW;
        ppx[i] = pi - ppy[i] * RB_PW;
    }

    u32x4 pre[RB_PER][2];  // conv1 operands of the next tile, loaded straight from global memory
    unsigned pre_in = 0;   // bit i: patch pixel i of this lane lies inside the image (else t = 0: zero padding)

    auto prefetch = [&](int tile) {
        const int tx = tile % p.tiles_x;
        const int t2 = tile / p.tiles_x;
        const int ty = t2 % p.tiles_y;
        const int n = t2 / p.tiles_y;
        const char* xn = p.x + p.x_origin + (long long)n * p.x_img_stride;
        pre_in = 0;
#pragma unroll
        for (int i = 0; i < RB_PER; ++i) {
            const int gy = ty * RB_TH - 1 + ppy[i], gx = tx * RB_TW - 1 + ppx[i];
            const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && ppy[i] < RB_PH;
            // pixels outside the image are never used: read pixel (0,0) instead of memory beyond the halo
            const int off = in ? gy * p.x_row_stride + gx * p.x_pix_stride : 0;
            pre[i][0] = *(const u32x4*)(xn + off + q * 16);
            pre[i][1] = *(const u32x4*)(xn + off + 64 + q * 16);
            pre_in |= (in ? 1u : 0u) << i;
        }
    };

    int tile = blockIdx.x;
    if (tile < p.tiles) prefetch(tile);
    __syncthreads();  // conv1 operands and biases in LDS
    for (; tile < p.tiles; tile += gridDim.x) {
        const int tx = tile % p.tiles_x;
        const int t2 = tile / p.tiles_x;
        const int ty = t2 % p.tiles_y;
        const int n = t2 / p.tiles_y;
        const int oy0 = ty * RB_TH, ox0 = tx * RB_TW;

        // ---- conv1 (1x1, 64 -> 32) on the 10x18 patch: accumulators start from the bias
        {
            bf16x8 w1f[2][2];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) w1f[f][ks] = w1frag_s[(f * 2 + ks) * 64 + lane];
            const f32x4 sb0 = *(const f32x4*)&bias_s[8 * q], sb1 = *(const f32x4*)&bias_s[8 * q + 4];
#pragma unroll
            for (int i = 0; i < RB_PER; ++i) {
                f32x4 acc0 = sb0, acc1 = sb1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 b = __builtin_bit_cast(bf16x8, pre[i][ks]);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[0][ks], b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[1][ks], b, acc1, 0, 0, 0);
                }
                const f32x4 l0 = leaky4(acc0, p.slope), l1 = leaky4(acc1, p.slope);
                u32x4 pk = {pack_bf16x2(l0[0], l0[1]), pack_bf16x2(l0[2], l0[3]), pack_bf16x2(l1[0], l1[1]),
                            pack_bf16x2(l1[2], l1[3])};
                if (!((pre_in >> i) & 1u)) pk = u32x4{0u, 0u, 0u, 0u};
                const int r = 16 * (wave + 4 * i) + l15;
                *(u32x4*)(patch + r * 64 + ((q ^ ((r >> 1) & 3)) << 4)) = pk;
            }
        }
        __syncthreads();

        // ---- residual operands of this tile (L2 hits), then the next tile's conv1 operands: the epilogue
        //      waits only for the former
        const char* xn = p.x + p.x_origin + (long long)n * p.x_img_stride;
        const int ox = ox0 + l15;
        u32x4 res[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + 4 * wp + j;
            const bool in = oy < p.H && ox < p.W;
            const int off = in ? oy * p.x_row_stride + ox * p.x_pix_stride : 0;
            res[j] = *(const u32x4*)(xn + off + 64 * wc + q * 16);
        }
        if (tile + (int)gridDim.x < p.tiles) prefetch(tile + gridDim.x);

        // ---- conv2 (3x3, 32 -> 64): wave = 32 channels x 64 pixels (output rows 4*wp .. 4*wp + 3 of the tile)
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 bi = *(const f32x4*)&bias_s[32 + 32 * wc + 8 * q + 4 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = bi;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - 3 * ky;
            bf16x8 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (4 * wp + j + ky) * RB_PW + l15 + kx;
                b[j] = *(const bf16x8*)(patch + r * 64 + ((q ^ ((r >> 1) & 3)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfrag[i][t], b[j], acc[i][j], 0, 0, 0);
        }

        // ---- epilogue: LeakyReLU, + residual, 8 consecutive channels (16 bytes) per lane
        char* yn = p.y + p.y_origin + (long long)n * p.y_img_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + 4 * wp + j;
            if (oy < p.H && ox < p.W) {
                f32x4 lo = leaky4(acc[0][j], p.slope), hi = leaky4(acc[1][j], p.slope);
                const u32x4 r = res[j];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    lo[2 * e] += __uint_as_float(r[e] << 16);
                    lo[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
                    hi[2 * e] += __uint_as_float(r[2 + e] << 16);
                    hi[2 * e + 1] += __uint_as_float(r[2 + e] & 0xffff0000u);
                }
                const u32x4 pk = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]),
                                  pack_bf16x2(hi[2], hi[3])};
                *(u32x4*)(yn + (oy * p.y_row_stride + ox * p.y_pix_stride + 64 * wc + q * 16)) = pk;
            }
        }
        __syncthreads();  // all conv2 reads of the patch are done before the next tile's conv1 writes it
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_resblock_c64_fwd(const yolo_view* x, const void* w1_packed, const float* b1,
                                     const void* w2_packed, const float* b2, const yolo_view* y, int N,
                                     float leaky_slope, void* stream) {
    YOLO_REQUIRE(x && y && x->ptr && y->ptr && w1_packed && b1 && w2_packed && b2, "resblock_c64: null argument");
    YOLO_REQUIRE(N > 0 && x->H > 0 && x->W > 0 && y->H == x->H && y->W == x->W, "resblock_c64: bad geometry");
    YOLO_REQUIRE(x->halo >= 1, "resblock_c64: input halo %d < 1", x->halo);
    YOLO_REQUIRE(x->ch_offset % 8 == 0 && x->ch_total % 8 == 0 && x->ch_offset + 64 <= x->ch_total &&
                     y->ch_offset % 8 == 0 && y->ch_total % 8 == 0 && y->ch_offset + 64 <= y->ch_total,
                 "resblock_c64: views need a 64-channel, 8-aligned slice");
    YOLO_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, "resblock_c64: leaky slope %g outside [0, 1]",
                 (double)leaky_slope);
    const ViewGeom gx = geom_of(*x), gy = geom_of(*y);
    {   // neighbouring tiles read x's halo pixels: in-place operation would race
        const char* xb = (const char*)x->ptr;
        const char* yb = (const char*)y->ptr;
        const long long xbytes = gx.img_stride * 2 * N, ybytes = gy.img_stride * 2 * N;
        YOLO_REQUIRE(yb + ybytes <= xb || xb + xbytes <= yb, "resblock_c64: x and y buffers overlap (not an in-place op)");
    }
    YOLO_REQUIRE(gx.img_stride * 2 < (1ll << 31) && gy.img_stride * 2 < (1ll << 31), "resblock_c64: image too large");
    ResBlockParams p{};
    p.x = (const char*)x->ptr;
    p.y = (char*)y->ptr;
    p.x_img_stride = gx.img_stride * 2;
    p.x_row_stride = gx.row_stride * 2;
    p.x_pix_stride = gx.pix_stride * 2;
    p.x_origin = (long long)gx.origin * 2;
    p.y_img_stride = gy.img_stride * 2;
    p.y_row_stride = gy.row_stride * 2;
    p.y_pix_stride = gy.pix_stride * 2;
    p.y_origin = (long long)gy.origin * 2;
    p.w1 = (const bf16_t*)w1_packed;
    p.b1 = b1;
    p.w2 = (const bf16_t*)w2_packed;
    p.b2 = b2;
    p.N = N;
    p.H = x->H;
    p.W = x->W;
    p.tiles_x = (p.W + RB_TW - 1) / RB_TW;
    p.tiles_y = (p.H + RB_TH - 1) / RB_TH;
    const long long tiles = (long long)N * p.tiles_x * p.tiles_y;
    YOLO_REQUIRE(tiles < (1ll << 31), "resblock_c64: problem too large");
    p.tiles = (int)tiles;
    p.slope = leaky_slope;
    // one workgroup per resident slot (two per CU) walks the tiles: the per-workgroup set-up (weights into registers,
    // tables into LDS) is paid once per slot -- 2048 workgroups, four per slot, measured 7-9 % slower
    const int blocks = p.tiles < 512 ? p.tiles : 512;
    hipLaunchKernelGGL(resblock_c64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

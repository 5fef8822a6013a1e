// Fused first two layers for gfx950: pixel normalisation + zero padding + 3x3/s1 stem conv (3 -> 32) +
// bias + LeakyReLU, then the 3x3/s2 down-sampling conv (32 -> 64) + bias + LeakyReLU, bf16 in LDS between
// the two.  Replaces DenseDetector.preprocess_image (configs/yolov3.yaml:5-6, yolo/modeling/backbone/fpn.py:
// 161-163), BasicStem.forward (yolo/modeling/backbone/darknet.py:57-60) and the `downsample` conv of the
// first ResidualBlock (darknet.py:113-116) in one launch.
//
// Why: the 32-channel full-resolution stem output is the largest tensor of the network (N*H*W*64 bytes);
// written and read back once it costs more HBM time than both convolutions cost on the matrix cores.  Here
// it never leaves the CU: per tile of 8x16 output pixels the 17x33 stem pixels under it are computed from a
// 19x36 image patch (1.10x recompute of a layer that is 3 % of the FLOPs) into LDS and consumed in place.
//
// LDS image of the stem patch: one 64-byte row per stem pixel (32 bf16 channels), even and odd columns of
// each patch row in separate blocks, so the 16 pixels a stride-2 tap reads for one MFMA operand are 16
// CONSECUTIVE rows; 16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 3), which makes both the
// ds_read_b128 operand reads and the ds_write_b128 of the producer bank-conflict free.
#include "common.h"

namespace yolo {

constexpr int SD_TH = 8, SD_TW = 16;             // output tile of the down conv
constexpr int SD_PH = 2 * SD_TH + 1;             // 17 stem rows under it
constexpr int SD_PW = 2 * SD_TW + 1;             // 33 stem columns
constexpr int SD_HALF = SD_TW + 1;               // 17 slots per column parity
constexpr int SD_ROWS = SD_PH * 2 * SD_HALF;     // 578 64-byte rows
constexpr int SD_NT = (SD_ROWS + 15) / 16;       // 37 stem n-tiles of 16 pixels
constexpr int SD_IH = SD_PH + 2, SD_IW = SD_PW + 3;  // 19 x 36 image pixels (the k chunk spans 4 pixels)
constexpr int SD_IPIX = SD_IH * SD_IW;           // 684
constexpr int SD_PRE = (SD_IPIX + 255) / 256;    // image pixels per thread
#ifndef SD_WGS
#define SD_WGS 2                                 // workgroups per CU the register budget is set for (3 would need <= 168 registers: it spills)
#endif

struct StemDownParams {
    const void* img;
    const float* w_stem;     // [27][32] (kh,kw,cin) major, output channel minor
    const float* b_stem;     // [32]
    const bf16_t* w_down;    // [Cout_pad][10 taps][32] bf16 (tap 9 is padding)
    const float* b_down;     // [64]
    char* y;
    long long y_img_stride, y_origin;  // bytes
    int y_row_stride, y_pix_stride;    // bytes
    int N, Hs, Ws;           // stored image size
    int Hst, Wst;            // stem output size (padded input size)
    int Ho, Wo;              // down-conv output size
    int tiles_x, tiles_y;
    int tiles;
    float mean[3], stdv[3];
    float slope;
};

template <typename IMG>
__global__ void __launch_bounds__(256, SD_WGS) stem_down_kernel(const StemDownParams p) {
    // RGB0 bf16 pixels; 3 slack rows stay zero (the zero-weight k chunks and the unused odd slot read them)
    __shared__ __attribute__((aligned(16))) unsigned long long img_t[SD_IPIX + 3 * SD_IW];
    __shared__ unsigned short lut[3 * 256];  // u8 images: bf16((v - mean[c]) / std[c]) for every byte value
    __shared__ __attribute__((aligned(16))) char patch[SD_NT * 16 * 64];
    __shared__ __attribute__((aligned(16))) bf16x8 sfrag_s[4 * 64];
    __shared__ __attribute__((aligned(16))) float bias_s[32 + 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;

    // ---- stem operands: built once per block into LDS, re-read per tile (they are dead during the down conv)
    // stem weights: row m of fragment f <-> channel 8*(m>>2) + 4*f + (m&3), so that after both MFMAs a lane
    // holds 8 consecutive channels; k chunk c = 4*s + q covers (ky = c>>1, pixels 2*(c&1), 2*(c&1)+1) x RGB0.
    if (wave == 0) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int ch = 8 * (l15 >> 2) + 4 * f + (l15 & 3);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c = 4 * s + q;
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ky = c >> 1, kx = 2 * (c & 1) + (j >> 2), ci = j & 3;
                    const bool used = ky < 3 && kx < 3 && ci < 3;  // unconditional load: no wait per element
                    const float w = p.w_stem[used ? ((ky * 3 + kx) * 3 + ci) * 32 + ch : 0];
                    v[j] = (bf16_t)(used ? w : 0.f);
                }
                sfrag_s[(f * 2 + s) * 64 + lane] = v;
            }
        }
    }
    if (tid < 3 * SD_IW) img_t[SD_IPIX + tid] = 0ull;
#pragma unroll
    for (int c = 0; c < 3; ++c) lut[c * 256 + tid] = f32_to_bf16_bits(((float)tid - p.mean[c]) / p.stdv[c]);
    if (tid < 32) bias_s[tid] = p.b_stem[tid];
    if (tid >= 64 && tid < 128) bias_s[32 + tid - 64] = p.b_down[tid - 64];

    // down conv: wave = 32 channels (wc) x 4 output rows (4 * wp ..) of 16 pixels.  Its weights, two m-tiles of the
    // same row permutation as the stem's, stay in registers for the whole launch: 18 fragments -- half of what a
    // 64-channel wave would hold, which is what lets three workgroups share a CU.
    const int wc = wave & 1, wp = wave >> 1;
    bf16x8 dfrag[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = 32 * wc + 8 * (l15 >> 2) + 4 * i + (l15 & 3);
#pragma unroll
        for (int t = 0; t < 9; ++t) dfrag[i][t] = *(const bf16x8*)(p.w_down + (size_t)ch * 320 + t * 32 + q * 8);
    }
    // pin the fragments in registers (after all loads are issued): otherwise the compiler re-loads them per tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(dfrag[i][t]));
    const int plane = p.Hs * p.Ws;
    IMG pre[SD_PRE][3];   // raw pixels of the next tile; converted only when they are written to LDS, so the
    unsigned pre_ok = 0;  // loads stay in flight across both conv phases

    // per-thread image-patch coordinates (tile independent)
    int pr_r[SD_PRE], pr_c[SD_PRE];
#pragma unroll
    for (int k = 0; k < SD_PRE; ++k) {
        const int i = tid + 256 * k;
        pr_r[k] = i / SD_IW;
        pr_c[k] = i - pr_r[k] * SD_IW;
    }

    auto prefetch = [&](int tile) {
        const int tx = tile % p.tiles_x;
        const int t2 = tile / p.tiles_x;
        const int ty = t2 % p.tiles_y;
        const int n = t2 / p.tiles_y;
        const IMG* im = (const IMG*)p.img + (long long)n * 3 * plane;
        const int iy0 = 2 * ty * SD_TH - 2, ix0 = 2 * tx * SD_TW - 2;
        pre_ok = 0;
#pragma unroll
        for (int k = 0; k < SD_PRE; ++k) {
            const int iy = iy0 + pr_r[k], ix = ix0 + pr_c[k];
            const bool ok = tid + 256 * k < SD_IPIX && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
            const int o = ok ? iy * p.Ws + ix : 0;  // out-of-image lanes load pixel 0, unused
            pre[k][0] = im[o];
            pre[k][1] = im[plane + o];
            pre[k][2] = im[2 * plane + o];
            pre_ok |= (ok ? 1u : 0u) << k;
        }
    };

    int tile = blockIdx.x;
    if (tile < p.tiles) prefetch(tile);
    __syncthreads();  // lut, slack rows, stem operands
    for (; tile < p.tiles; tile += gridDim.x) {
        const int tx = tile % p.tiles_x;
        const int t2 = tile / p.tiles_x;
        const int ty = t2 % p.tiles_y;
        const int n = t2 / p.tiles_y;
        const int oy0 = ty * SD_TH, ox0 = tx * SD_TW;

        // ---- image patch -> LDS (normalised, bf16, RGB0)
#pragma unroll
        for (int k = 0; k < SD_PRE; ++k) {
            const int i = tid + 256 * k;
            if (i < SD_IPIX) {
                unsigned long long v = 0ull;  // out-of-image pixels are 0 after normalisation
                if ((pre_ok >> k) & 1u) {
                    if constexpr (sizeof(IMG) == 1) {
                        v = (unsigned long long)((unsigned)lut[pre[k][0]] | ((unsigned)lut[256 + pre[k][1]] << 16)) |
                            ((unsigned long long)lut[512 + pre[k][2]] << 32);
                    } else {
                        const float r0 = ((float)pre[k][0] - p.mean[0]) / p.stdv[0];
                        const float r1 = ((float)pre[k][1] - p.mean[1]) / p.stdv[1];
                        const float r2 = ((float)pre[k][2] - p.mean[2]) / p.stdv[2];
                        v = (unsigned long long)pack_bf16x2(r0, r1) | ((unsigned long long)pack_bf16x2(r2, 0.f) << 32);
                    }
                }
                img_t[i] = v;
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < p.tiles) prefetch(tile + gridDim.x);  // in flight during both conv phases

        // ---- stem: 37 n-tiles of 16 patch rows over the 4 waves; accumulators start from the bias
        bf16x8 sfrag[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int s = 0; s < 2; ++s) sfrag[f][s] = sfrag_s[(f * 2 + s) * 64 + lane];
        const f32x4 sb0 = *(const f32x4*)&bias_s[8 * q], sb1 = *(const f32x4*)&bias_s[8 * q + 4];
        // stem pixels outside the stem output are the down conv's zero padding (0, not stem(0)): only tiles on
        // the border have any
        const bool border = oy0 == 0 || ox0 == 0 || 2 * (oy0 + SD_TH) >= p.Hst || 2 * (ox0 + SD_TW) >= p.Wst;
#pragma unroll
        for (int it = 0; it < (SD_NT + 3) / 4; ++it) {
            const int t = wave + 4 * it;
            if (t >= SD_NT) break;
            const int r = 16 * t + l15;
            const int py = (r * 1928) >> 16;  // r / 34 for r < 592
            const int rem = r - py * (2 * SD_HALF);
            const int par = rem >= SD_HALF ? 1 : 0;
            const int px = 2 * rem - (2 * SD_HALF - 1) * par;  // 33 for the unused slot of the odd block
            f32x4 acc0 = sb0, acc1 = sb1;
            const char* a0p = (const char*)img_t + ((py + (q >> 1)) * SD_IW + px + 2 * (q & 1)) * 8;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // k chunk c = 4*s + q: image row py + (c>>1) = py + 2*s + (q>>1); chunks 6, 7 are zero weights,
                // whatever they read (in bounds: the tile has slack rows)
                const char* a = a0p + s * (2 * SD_IW * 8);
                const u32x2 lo = *(const u32x2*)a;
                const u32x2 hi = *(const u32x2*)(a + 8);
                const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                const bf16x8 b = __builtin_bit_cast(bf16x8, v);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sfrag[0][s], b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sfrag[1][s], b, acc1, 0, 0, 0);
            }
            const f32x4 l0 = leaky4(acc0, p.slope), l1 = leaky4(acc1, p.slope);
            u32x4 pk = {pack_bf16x2(l0[0], l0[1]), pack_bf16x2(l0[2], l0[3]), pack_bf16x2(l1[0], l1[1]),
                        pack_bf16x2(l1[2], l1[3])};
            if (border) {
                const int sy = 2 * oy0 - 1 + py, sx = 2 * ox0 - 1 + px;
                if (!(sy >= 0 && sy < p.Hst && sx >= 0 && sx < p.Wst)) pk = u32x4{0u, 0u, 0u, 0u};
            }
            *(u32x4*)(patch + r * 64 + ((q ^ ((r >> 1) & 3)) << 4)) = pk;
        }
        __syncthreads();

        // ---- down conv: wave = 32 channels x 64 pixels (output rows 4*wp .. 4*wp + 3 of the tile)
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
                const int r = ((2 * (4 * wp + j) + ky) * 2 + (kx & 1)) * SD_HALF + l15 + (kx >> 1);
                b[j] = *(const bf16x8*)(patch + r * 64 + ((q ^ ((r >> 1) & 3)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfrag[i][t], b[j], acc[i][j], 0, 0, 0);
        }

        // ---- epilogue: LeakyReLU, 8 consecutive channels (16 bytes) per lane
        const int ox = ox0 + l15;
        char* ybase = p.y + p.y_origin + (long long)n * p.y_img_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + 4 * wp + j;
            if (oy < p.Ho && ox < p.Wo) {
                const f32x4 lo = leaky4(acc[0][j], p.slope), hi = leaky4(acc[1][j], p.slope);
                const u32x4 pk = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]),
                                  pack_bf16x2(hi[2], hi[3])};
                *(u32x4*)(ybase + (oy * p.y_row_stride + ox * p.y_pix_stride + 64 * wc + q * 16)) = pk;
            }
        }
        // the next iteration's image-patch barrier orders this tile's patch reads before the next writes
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_stem_down_fwd(const void* img, int img_dtype, int N, int Hs, int Ws, const float* mean_host,
                                  const float* std_host, const float* w_stem, const float* b_stem,
                                  const void* w_down_packed, const float* b_down, const yolo_view* y, int H, int W,
                                  float leaky_slope, void* stream) {
    YOLO_REQUIRE(img && mean_host && std_host && w_stem && b_stem && w_down_packed && b_down && y && y->ptr,
                 "stem_down: null argument");
    YOLO_REQUIRE(img_dtype == 0 || img_dtype == 1, "stem_down: img_dtype %d (0 = f32, 1 = u8)", img_dtype);
    YOLO_REQUIRE(N > 0 && N <= 65535 && Hs > 0 && Ws > 0, "stem_down: bad batch / image size");
    YOLO_REQUIRE(H % 2 == 0 && W % 2 == 0 && Hs <= H && Ws <= W, "stem_down: image %dx%d does not fit padded size %dx%d (even)",
                 Hs, Ws, H, W);
    YOLO_REQUIRE(y->H == H / 2 && y->W == W / 2, "stem_down: output view %dx%d, expected %dx%d", y->H, y->W, H / 2, W / 2);
    YOLO_REQUIRE(y->ch_offset % 8 == 0 && y->ch_total % 8 == 0 && y->ch_offset + 64 <= y->ch_total,
                 "stem_down: output needs a 64-channel, 8-aligned slice");
    const ViewGeom gy = geom_of(*y);
    StemDownParams p{};
    p.img = img;
    p.w_stem = w_stem;
    p.b_stem = b_stem;
    p.w_down = (const bf16_t*)w_down_packed;
    p.b_down = b_down;
    p.y = (char*)y->ptr;
    p.y_img_stride = gy.img_stride * 2;
    p.y_row_stride = gy.row_stride * 2;
    p.y_pix_stride = gy.pix_stride * 2;
    p.y_origin = (long long)gy.origin * 2;
    p.N = N;
    p.Hs = Hs;
    p.Ws = Ws;
    p.Hst = H;
    p.Wst = W;
    p.Ho = H / 2;
    p.Wo = W / 2;
    p.tiles_x = (p.Wo + SD_TW - 1) / SD_TW;
    p.tiles_y = (p.Ho + SD_TH - 1) / SD_TH;
    const long long tiles = (long long)N * p.tiles_x * p.tiles_y;
    YOLO_REQUIRE(tiles < (1ll << 31) && (long long)Hs * Ws * 3 < (1ll << 31), "stem_down: problem too large");
    YOLO_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, "stem_down: leaky slope %g outside [0, 1]", (double)leaky_slope);
    p.tiles = (int)tiles;
    for (int i = 0; i < 3; ++i) {
        p.mean[i] = mean_host[i];
        p.stdv[i] = std_host[i];
    }
    p.slope = leaky_slope;
    // one workgroup per resident slot (two per CU) walks the tiles: the per-workgroup set-up (weights into registers,
    // tables into LDS) is paid once per slot -- 2048 workgroups, four per slot, measured 7-9 % slower
    const int blocks = p.tiles < 512 ? p.tiles : 512;
    if (img_dtype == 0)
        hipLaunchKernelGGL((stem_down_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((stem_down_kernel<unsigned char>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

// First-layer kernel (pixel normalisation + zero padding + 3x3 stem conv + bias + LeakyReLU) and the
// stand-alone nearest-2x upsample, for gfx950.
//
// yolo_stem_fwd replaces DenseDetector.preprocess_image as configured by configs/yolov3.yaml:5-6 and
// yolo/modeling/backbone/fpn.py:161-163, plus BasicStem.forward (yolo/modeling/backbone/darknet.py:57-60).
// yolo_upsample2x_nhwc replaces nn.Upsample(2,"nearest") + torch.cat (fpn.py:31,36,179).
#include "common.h"

namespace yolo {

struct StemParams {
    const void* img;
    const float* w;     // [27][32]: (kh,kw,cin) major, output channel minor
    const float* bias;  // [32]
    char* y;
    long long y_img_stride, y_origin;  // bytes
    int y_row_stride, y_pix_stride;    // bytes
    int N, Hs, Ws, H, W;
    float mean[3], stdv[3];
    float slope;
};

// One wave = 64 consecutive output pixels of one row; lane = (pixel group pg, channel group cg): it
// computes 8 channels for the 4 pixels pg, pg+16, pg+32, pg+48, so every store instruction of the
// wave writes 16 pixels x 64 bytes = 1 KiB of contiguous NHWC output.
template <typename IMG, bool YF32>
__global__ void __launch_bounds__(256) stem_kernel(const StemParams p) {
    __shared__ __attribute__((aligned(16))) float wl[27 * 32];
    for (int i = threadIdx.x; i < 27 * 32; i += 256) wl[i] = p.w[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int spans = (p.W + 63) >> 6;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = (long long)p.N * p.H * spans;
    if (wid >= total) return;
    const int span = (int)(wid % spans);
    const long long t = wid / spans;
    const int y = (int)(t % p.H);
    const int n = (int)(t / p.H);
    const int cg = lane & 3, pg = lane >> 2;

    float acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;

    const IMG* im = (const IMG*)p.img + (long long)n * 3 * p.Hs * p.Ws;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = y + ky - 1;
        const bool yok = iy >= 0 && iy < p.Hs;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const int tapi = (ky * 3 + kx) * 3 + ci;
                const f32x4 w0 = *(const f32x4*)&wl[tapi * 32 + cg * 8];
                const f32x4 w1 = *(const f32x4*)&wl[tapi * 32 + cg * 8 + 4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ix = span * 64 + pg + 16 * j + kx - 1;
                    float v = 0.f;
                    if (yok && ix >= 0 && ix < p.Ws) {
                        const float raw = (float)im[((long long)ci * p.Hs + iy) * p.Ws + ix];
                        v = (raw - p.mean[ci]) / p.stdv[ci];
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[j][c] = fmaf(v, w0[c], acc[j][c]);
                        acc[j][4 + c] = fmaf(v, w1[c], acc[j][4 + c]);
                    }
                }
            }
        }
    }

    const f32x4 b0 = *(const f32x4*)&p.bias[cg * 8];
    const f32x4 b1 = *(const f32x4*)&p.bias[cg * 8 + 4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = span * 64 + pg + 16 * j;
        if (x >= p.W) continue;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float a = acc[j][c] + (c < 4 ? b0[c] : b1[c - 4]);
            v[c] = a >= 0.f ? a : a * p.slope;
        }
        char* o = p.y + p.y_origin + (long long)n * p.y_img_stride + (long long)y * p.y_row_stride +
                  (long long)x * p.y_pix_stride + cg * 8 * (YF32 ? 4 : 2);
        if constexpr (YF32) {
            *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4*)(o + 16) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            u32x4 pk;
            pk[0] = pack_bf16x2(v[0], v[1]);
            pk[1] = pack_bf16x2(v[2], v[3]);
            pk[2] = pack_bf16x2(v[4], v[5]);
            pk[3] = pack_bf16x2(v[6], v[7]);
            *(u32x4*)o = pk;
        }
    }
}

// bf16-output stem on the matrix cores.  The normalised image tile is staged in LDS as bf16 "RGB0"
// pixels (8 bytes each); with k ordered (ky, 4 pixels x-1..x+2, 4 channels) every 8-element MFMA k
// chunk of a pixel is 16 contiguous LDS bytes, so the im2col operand needs no gather.  k = 3*16 = 48 is
// padded to 64 (two 16x16x32 MFMAs); the 4th pixel and 4th channel meet zero weights.  Output channel
// rows are permuted so that each lane ends up with 8 consecutive channels: one 16-byte store per lane,
// 1 KiB contiguous per wave.
constexpr int ST_TH = 8, ST_TW = 64, ST_LW = ST_TW + 4;

template <typename IMG>
__global__ void __launch_bounds__(256) stem_mfma_kernel(const StemParams p) {
    __shared__ __attribute__((aligned(16))) unsigned long long tile[(ST_TH + 2) * ST_LW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * ST_TW, y0 = blockIdx.y * ST_TH, n = blockIdx.z;
    const IMG* im = (const IMG*)p.img + (long long)n * 3 * p.Hs * p.Ws;
    for (int i = tid; i < (ST_TH + 2) * ST_LW; i += 256) {
        const int r = i / ST_LW, c = i - r * ST_LW;
        const int iy = y0 + r - 1, ix = x0 + c - 1;
        unsigned long long v = 0;
        if (iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws) {
            const long long o = (long long)iy * p.Ws + ix;
            const long long plane = (long long)p.Hs * p.Ws;
            const float r0 = ((float)im[o] - p.mean[0]) / p.stdv[0];
            const float r1 = ((float)im[plane + o] - p.mean[1]) / p.stdv[1];
            const float r2 = ((float)im[2 * plane + o] - p.mean[2]) / p.stdv[2];
            v = (unsigned long long)pack_bf16x2(r0, r1) | ((unsigned long long)pack_bf16x2(r2, 0.f) << 32);
        }
        tile[i] = v;
    }
    // weight fragments: A[row m][k], row m of fragment f <-> channel 8*(m>>2) + 4*f + (m&3)
    const int l15 = lane & 15, q = lane >> 4;
    bf16x8 afrag[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int ch = 8 * (l15 >> 2) + 4 * f + (l15 & 3);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = 4 * s + q;  // k chunk: ky = c/2, pixels 2*(c&1), 2*(c&1)+1
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ky = c >> 1, kx = 2 * (c & 1) + (j >> 2), ci = j & 3;
                float w = 0.f;
                if (ky < 3 && kx < 3 && ci < 3) w = p.w[((ky * 3 + kx) * 3 + ci) * 32 + ch];
                afrag[f][s][j] = (bf16_t)w;
            }
        }
    }
    float bias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[j] = p.bias[8 * q + j];
    __syncthreads();

    const char* tb = (const char*)tile;
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int row = wave * 2 + (it >> 2);
        const int x = (it & 3) * 16 + l15;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = 4 * s + q;
            bf16x8 b;
            if (c < 6) {
                const char* a = tb + ((row + (c >> 1)) * ST_LW + x + 2 * (c & 1)) * 8;
                const u32x2 lo = *(const u32x2*)a;
                const u32x2 hi = *(const u32x2*)(a + 8);
                const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                b = __builtin_bit_cast(bf16x8, v);
            } else {
                const u32x4 z = {0u, 0u, 0u, 0u};
                b = __builtin_bit_cast(bf16x8, z);
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[0][s], b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[1][s], b, acc1, 0, 0, 0);
        }
        const int gx = x0 + x, gy = y0 + row;
        if (gx < p.W && gy < p.H) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc0[j] + bias[j];
                v[4 + j] = acc1[j] + bias[4 + j];
            }
            u32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = v[2 * j] >= 0.f ? v[2 * j] : v[2 * j] * p.slope;
                const float b2 = v[2 * j + 1] >= 0.f ? v[2 * j + 1] : v[2 * j + 1] * p.slope;
                pk[j] = pack_bf16x2(a, b2);
            }
            char* o = p.y + p.y_origin + (long long)n * p.y_img_stride + (long long)gy * p.y_row_stride +
                      (long long)gx * p.y_pix_stride + q * 16;
            *(u32x4*)o = pk;
        }
    }
}

struct UpParams {
    const char* x;
    char* y;
    long long x_img_stride, x_origin, y_img_stride, y_origin;  // bytes
    int x_row_stride, x_pix_stride, y_row_stride, y_pix_stride;
    int H, W;     // input size
    int chunks;   // 16-byte chunks per pixel
    long long total;  // N*H*W*chunks
};

__global__ void __launch_bounds__(256) upsample2x_kernel(const UpParams p) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % p.chunks);
        long long t = i / p.chunks;
        const int x = (int)(t % p.W);
        t /= p.W;
        const int y = (int)(t % p.H);
        const int n = (int)(t / p.H);
        const u32x4 v = *(const u32x4*)(p.x + p.x_origin + n * p.x_img_stride + (long long)y * p.x_row_stride +
                                        x * p.x_pix_stride + c * 16);
        char* o = p.y + p.y_origin + n * p.y_img_stride + (long long)(2 * y) * p.y_row_stride +
                  (2 * x) * p.y_pix_stride + c * 16;
        *(u32x4*)o = v;
        *(u32x4*)(o + p.y_pix_stride) = v;
        *(u32x4*)(o + p.y_row_stride) = v;
        *(u32x4*)(o + p.y_row_stride + p.y_pix_stride) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the stem (training with MODEL.BACKBONE.FREEZE_AT = 0; the reference leaves it to autograd of
// BasicStem.forward, darknet.py:40-60, when DarkNet.freeze(0) at darknet.py:236-258 freezes nothing):
//   dW[co][kh][kw][ci] = sum over (n, y, x) of dz[n][y][x][co] * xn[n][y + kh - 1][x + kw - 1][ci],
// xn = the normalised image, zero outside the stored image (and, as the forward kernel of the bf16 plan sees it,
// rounded to bf16 when round_bf16 is set).  HBM-bound: dz is read once (64-128 bytes per pixel), the image three
// bytes per pixel; 864 sums.  A workgroup walks 16 x 16 pixel tiles; thread = (output channel, group of four
// (tap, input channel) combinations) keeps its four sums in registers over all its tiles, then leaves them in a row
// of partial sums that a second kernel folds in a fixed order (no float atomics: results do not depend on timing).
struct StemWgradParams {
    const void* img;
    const char* dz;
    long long dz_img_stride, dz_origin;  // bytes
    int dz_row_stride, dz_pix_stride;    // bytes
    float* partial;                      // [gridDim.x][32 * 27]
    int N, Hs, Ws, H, W;
    int tiles_x, tiles_y;
    long long total_tiles;
    float mean[3], stdv[3];
    int round_bf16;
};

constexpr int SW_T = 16, SW_L = SW_T + 2;

template <typename IMG, typename DZ>
__global__ void __launch_bounds__(256) stem_wgrad_kernel(const StemWgradParams p) {
    __shared__ float xs[SW_L * SW_L * 3 + 8];
    __shared__ float dzs[SW_T * SW_T * 32];
    const int tid = threadIdx.x, co = tid & 31, grp = tid >> 5;
    int base[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = grp * 4 + i;  // (kh, kw, ci) flattened; 27..31 are padding
        live[i] = j < 27;
        const int jj = live[i] ? j : 0;
        base[i] = ((jj / 9) * SW_L + (jj / 3) % 3) * 3 + jj % 3;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const long long plane = (long long)p.Hs * p.Ws;
    for (long long t = blockIdx.x; t < p.total_tiles; t += gridDim.x) {
        const int tx = (int)(t % p.tiles_x);
        const long long r = t / p.tiles_x;
        const int ty = (int)(r % p.tiles_y), n = (int)(r / p.tiles_y);
        const int x0 = tx * SW_T, y0 = ty * SW_T;
        const IMG* im = (const IMG*)p.img + (long long)n * 3 * plane;
        __syncthreads();  // the previous tile's reads are done
        for (int i = tid; i < SW_L * SW_L * 3; i += 256) {
            const int ci = i % 3, c = (i / 3) % SW_L, rr = i / (3 * SW_L);
            const int iy = y0 + rr - 1, ix = x0 + c - 1;
            float v = 0.f;
            if (iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws) {
                v = ((float)im[ci * plane + (long long)iy * p.Ws + ix] - p.mean[ci]) / p.stdv[ci];
                if (p.round_bf16) v = __uint_as_float(pack_bf16x2(v, 0.f) << 16);
            }
            xs[i] = v;
        }
        for (int i = tid; i < SW_T * SW_T * 4; i += 256) {  // 8 channels per thread and pass
            const int c8 = i & 3, px = (i >> 2) & (SW_T - 1), py = i >> 6;
            const int gy = y0 + py, gx = x0 + px;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (gy < p.H && gx < p.W) {
                const char* a = p.dz + p.dz_origin + (long long)n * p.dz_img_stride + (long long)gy * p.dz_row_stride +
                                (long long)gx * p.dz_pix_stride + c8 * 8 * (int)sizeof(DZ);
                if constexpr (sizeof(DZ) == 2) {
                    const u32x4 u = *(const u32x4*)a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] = __uint_as_float(u[e] << 16);
                        v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
                    }
                } else {
                    const f32x4 u0 = *(const f32x4*)a, u1 = *(const f32x4*)(a + 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = u0[e];
                        v[4 + e] = u1[e];
                    }
                }
            }
            float* d = &dzs[(py * SW_T + px) * 32 + c8 * 8];
            *(f32x4*)d = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4*)(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        __syncthreads();
#pragma unroll 4
        for (int px = 0; px < SW_T * SW_T; ++px) {
            const float d = dzs[px * 32 + co];
            const int o = ((px >> 4) * SW_L + (px & (SW_T - 1))) * 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(d, xs[o + base[i]], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (live[i]) p.partial[(long long)blockIdx.x * 864 + co * 27 + grp * 4 + i] = acc[i];
}

__global__ void __launch_bounds__(128) stem_wgrad_fold_kernel(const float* partial, int rows, float* dw, int accumulate) {
    const int o = blockIdx.x * 128 + threadIdx.x;
    if (o >= 864) return;
    float a = 0.f;
    for (int r = 0; r < rows; ++r) a += partial[(long long)r * 864 + o];
    dw[o] = accumulate ? dw[o] + a : a;
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_stem_fwd(const void* img, int img_dtype, int N, int Hs, int Ws, const float* mean_host,
                             const float* std_host, const float* w, const float* bias, const yolo_view* y, int Cout,
                             float leaky_slope, int y_dtype, void* stream) {
    YOLO_REQUIRE(img && mean_host && std_host && w && bias && y, "stem: null argument");
    YOLO_REQUIRE(img_dtype == 0 || img_dtype == 1, "stem: img_dtype %d (0 = f32, 1 = u8)", img_dtype);
    YOLO_REQUIRE(Cout == 32, "stem: Cout %d, only 32 output channels are implemented", Cout);
    YOLO_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && Hs <= y->H && Ws <= y->W, "stem: image %dx%d does not fit output %dx%d",
                 Hs, Ws, y->H, y->W);
    YOLO_REQUIRE(y->ch_offset % 8 == 0 && y->ch_total % 8 == 0 && y->ch_offset + Cout <= y->ch_total,
                 "stem: bad output channel slice");
    const int yes = (int)elem_size(y_dtype);
    const ViewGeom gy = geom_of(*y);
    StemParams p{};
    p.img = img;
    p.w = w;
    p.bias = bias;
    p.y = (char*)y->ptr;
    p.y_img_stride = gy.img_stride * yes;
    p.y_row_stride = gy.row_stride * yes;
    p.y_pix_stride = gy.pix_stride * yes;
    p.y_origin = (long long)gy.origin * yes;
    p.N = N;
    p.Hs = Hs;
    p.Ws = Ws;
    p.H = y->H;
    p.W = y->W;
    for (int i = 0; i < 3; ++i) {
        p.mean[i] = mean_host[i];
        p.stdv[i] = std_host[i];
    }
    p.slope = leaky_slope;
    hipStream_t st = (hipStream_t)stream;
    if (y_dtype == YOLO_BF16) {
        const dim3 grid((y->W + ST_TW - 1) / ST_TW, (y->H + ST_TH - 1) / ST_TH, N);
        YOLO_REQUIRE(N <= 65535, "stem: batch %d > 65535", N);
        if (img_dtype == 0)
            hipLaunchKernelGGL((stem_mfma_kernel<float>), grid, dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((stem_mfma_kernel<unsigned char>), grid, dim3(256), 0, st, p);
    } else {
        const long long waves = (long long)N * y->H * ((y->W + 63) / 64);
        const dim3 grid((unsigned)((waves + 3) / 4));
        if (img_dtype == 0)
            hipLaunchKernelGGL((stem_kernel<float, true>), grid, dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((stem_kernel<unsigned char, true>), grid, dim3(256), 0, st, p);
    }
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

constexpr int STEM_WGRAD_ROWS = 1024;

extern "C" size_t yolo_stem_wgrad_workspace_bytes(void) { return (size_t)STEM_WGRAD_ROWS * 864 * sizeof(float); }

extern "C" int yolo_stem_wgrad(const void* img, int img_dtype, int N, int Hs, int Ws, const float* mean_host,
                               const float* std_host, const yolo_view* dz, int dz_dtype, int round_input_bf16, float* dw,
                               int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(img && mean_host && std_host && dz && dz->ptr && dw && workspace, "stem_wgrad: null argument");
    YOLO_REQUIRE(img_dtype == 0 || img_dtype == 1, "stem_wgrad: img_dtype %d (0 = f32, 1 = u8)", img_dtype);
    YOLO_REQUIRE(dz_dtype == YOLO_F32 || dz_dtype == YOLO_BF16, "stem_wgrad: dz dtype %d", dz_dtype);
    YOLO_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && Hs <= dz->H && Ws <= dz->W, "stem_wgrad: image %dx%d does not fit dz %dx%d",
                 Hs, Ws, dz->H, dz->W);
    YOLO_REQUIRE(dz->ch_offset % 8 == 0 && dz->ch_total % 8 == 0 && dz->ch_offset + 32 <= dz->ch_total,
                 "stem_wgrad: dz must hold 32 channels (8-aligned slice)");
    YOLO_REQUIRE(ws_bytes >= yolo_stem_wgrad_workspace_bytes(), "stem_wgrad: workspace %zu < %zu bytes", ws_bytes,
                 yolo_stem_wgrad_workspace_bytes());
    const int es = (int)elem_size(dz_dtype);
    const ViewGeom g = geom_of(*dz);
    StemWgradParams p{};
    p.img = img;
    p.dz = (const char*)dz->ptr;
    p.dz_img_stride = g.img_stride * es;
    p.dz_row_stride = (int)(g.row_stride * es);
    p.dz_pix_stride = (int)(g.pix_stride * es);
    p.dz_origin = (long long)g.origin * es;
    p.partial = (float*)workspace;
    p.N = N;
    p.Hs = Hs;
    p.Ws = Ws;
    p.H = dz->H;
    p.W = dz->W;
    p.tiles_x = (dz->W + SW_T - 1) / SW_T;
    p.tiles_y = (dz->H + SW_T - 1) / SW_T;
    p.total_tiles = (long long)N * p.tiles_x * p.tiles_y;
    for (int i = 0; i < 3; ++i) {
        p.mean[i] = mean_host[i];
        p.stdv[i] = std_host[i];
    }
    p.round_bf16 = round_input_bf16 ? 1 : 0;
    const int rows = (int)(p.total_tiles < STEM_WGRAD_ROWS ? p.total_tiles : STEM_WGRAD_ROWS);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)rows), block(256);
    if (img_dtype == 0) {
        if (dz_dtype == YOLO_BF16)
            hipLaunchKernelGGL((stem_wgrad_kernel<float, bf16_t>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((stem_wgrad_kernel<float, float>), grid, block, 0, st, p);
    } else {
        if (dz_dtype == YOLO_BF16)
            hipLaunchKernelGGL((stem_wgrad_kernel<unsigned char, bf16_t>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((stem_wgrad_kernel<unsigned char, float>), grid, block, 0, st, p);
    }
    YOLO_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(stem_wgrad_fold_kernel, dim3(7), dim3(128), 0, st, (const float*)workspace, rows, dw, accumulate ? 1 : 0);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_upsample2x_nhwc(const yolo_view* x, const yolo_view* y, int N, int C, int dtype, void* stream) {
    YOLO_REQUIRE(x && y, "upsample2x: null argument");
    const int es = (int)elem_size(dtype);
    const int per16 = 16 / es;
    YOLO_REQUIRE(C > 0 && C % per16 == 0, "upsample2x: C %d must be a multiple of %d", C, per16);
    YOLO_REQUIRE(y->H == 2 * x->H && y->W == 2 * x->W, "upsample2x: output must be 2x the input");
    YOLO_REQUIRE(x->ch_offset % per16 == 0 && x->ch_total % per16 == 0 && y->ch_offset % per16 == 0 &&
                     y->ch_total % per16 == 0 && x->ch_offset + C <= x->ch_total && y->ch_offset + C <= y->ch_total,
                 "upsample2x: bad channel slice");
    const ViewGeom gx = geom_of(*x), gy = geom_of(*y);
    UpParams p{};
    p.x = (const char*)x->ptr;
    p.y = (char*)y->ptr;
    p.x_img_stride = gx.img_stride * es;
    p.x_row_stride = gx.row_stride * es;
    p.x_pix_stride = gx.pix_stride * es;
    p.x_origin = (long long)gx.origin * es;
    p.y_img_stride = gy.img_stride * es;
    p.y_row_stride = gy.row_stride * es;
    p.y_pix_stride = gy.pix_stride * es;
    p.y_origin = (long long)gy.origin * es;
    p.H = x->H;
    p.W = x->W;
    p.chunks = C / per16;
    p.total = (long long)N * x->H * x->W * p.chunks;
    long long blocks = (p.total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

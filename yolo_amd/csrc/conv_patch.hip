// 3x3 / stride-1 convolution for gfx950 with the input patch resident in LDS.
//
// Same contract and epilogue as conv_igemm_kernel (conv.hip): folded-BatchNorm bias + LeakyReLU
// (+ residual add) for yolo/modeling/backbone/darknet.py:113-126 and fpn.py:66-71,96-98 of the reference.
//
// Why a second kernel: the plain implicit GEMM re-fetches every input pixel once per filter tap (9x)
// through the L2 -> LDS path, and on MI355X that path, not the matrix cores, bounds a 128x128 tile.  Here
// a workgroup loads, per 32-channel slice, the patch of input pixels its 128 output pixels touch ONCE
// (as 64-byte LDS rows), and the 9 taps become 9 k-steps that read the same LDS rows at a row offset
// (kh * patch_width + kw).  Only the 8 KiB weight slice changes per k-step.
//
//   k order            : (cin slice of 32, kh, kw, 32 channels); weights packed to match
//   tile               : TC channels x 128 output pixels, 4 waves as 2 x 2, 16x16x32 bf16 MFMA
//   pixel tile shapes  : "rows"  = 128 consecutive pixels in (n, y, x) order, patch = the whole padded
//                                  input rows they touch (small images: one tile spans several rows/images)
//                        "block" = th x tw pixels of one image, patch = (th+2) x (tw+2)
//   LDS                : 2 weight buffers + 1 or 2 patch buffers, sized per launch (2-3 workgroups / CU)
//   swizzle            : 16-byte chunk c of LDS row r holds source chunk c ^ (2 * ((r >> 2) & 1)); with it a
//                        ds_read_b128 of any 16 consecutive rows is bank-conflict free
#include <stdlib.h>

#include "common.h"

namespace yolo {

struct PatchParams {
    const char* x;  // + origin of padded position (row 0, col 0), channel offset applied
    const char* w;
    const float* bias;
    const char* res;
    char* y;
    long long x_row_stride;  // bytes between padded input rows (rows are contiguous across images)
    int x_pix_stride;        // bytes
    int Hp, Wp;              // padded input rows per image / positions per row
    int halo_m1;             // input halo - 1: padded coordinate of tap (0,0) of output pixel (0,0)
    long long y_img_stride, y_origin;
    int y_row_stride, y_pix_stride;
    long long r_img_stride, r_origin;
    int r_row_stride, r_pix_stride;
    int N, Ho, Wo, M;
    int nchunks;       // Cin / 32
    int w_row_bytes;   // npairs * 128
    int Cout;
    int n_ctiles, n_ptiles;
    int mode;          // 0 = rows, 1 = block
    int th, tw;        // block mode tile
    int tiles_x, tiles_y;
    int WL;            // LDS patch width in positions
    int NRmax;         // patch rows allocated
    int patch_bytes;   // per patch buffer (multiple of 1024)
    int patch_bufs;    // 1 or 2
    int npairs;        // pairs of k-steps = ceil(9 * nchunks / 2); an odd count is padded with zero weights
    float slope;
};

__device__ __forceinline__ void glds16p(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

constexpr int PP_TP = 128;    // pixels per tile
constexpr int PP_MAXPIECES = 8;  // 1-KiB patch pieces per wave (<= 512 patch rows)

template <int TC, bool SINGLE>
__global__ void __launch_bounds__(256) conv3x3_patch_kernel(const PatchParams p) {
    constexpr int FC = TC / 32, FP = 4;        // fragments per wave: (TC/2)/16 channels x 64/16 pixels
    constexpr int WBUF = TC * 128;             // bytes per weight buffer: one pair of k-steps
    constexpr int NWP = TC / 32;               // weight pieces per wave per pair of k-steps
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wbuf = smem;
    char* const pbuf = smem + 2 * WBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1;
    const int l15 = lane & 15, q = lane >> 4;

    int tile_c, tile_p;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int qq = nb >> 3, r = nb & 7, xcd = bid & 7, j = bid >> 3;
        const int logical = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + j;
        tile_c = logical % p.n_ctiles;
        tile_p = logical / p.n_ctiles;
    }

    // ---- tile geometry: G0 = first global padded input row of the patch, X0 = first column
    const int HoWo = p.Ho * p.Wo;
    int G0, X0, NR, bn = 0, by0 = 0, bx0 = 0;
    if (p.mode == 0) {
        const int m0 = tile_p * PP_TP;
        int m1 = m0 + PP_TP - 1;
        m1 = m1 < p.M ? m1 : p.M - 1;
        const int n0 = m0 / HoWo, y0 = (m0 - n0 * HoWo) / p.Wo;
        const int n1 = m1 / HoWo, y1 = (m1 - n1 * HoWo) / p.Wo;
        G0 = n0 * p.Hp + y0 + p.halo_m1;
        NR = (n1 * p.Hp + y1 + p.halo_m1 + 2) - G0 + 1;
        X0 = 0;
    } else {
        const int per_img = p.tiles_x * p.tiles_y;
        bn = tile_p / per_img;
        const int t = tile_p - bn * per_img;
        by0 = (t / p.tiles_x) * p.th;
        bx0 = (t % p.tiles_x) * p.tw;
        G0 = bn * p.Hp + by0 + p.halo_m1;
        NR = p.th + 2;
        X0 = bx0 + p.halo_m1;
    }
    const int npos = NR * p.WL;  // patch positions actually needed (<= NRmax * WL)

    // pixel i of the tile -> (n, y, x, valid)
    auto pixel_of = [&](int i, int& n, int& y, int& x) -> bool {
        if (p.mode == 0) {
            const int m = tile_p * PP_TP + i;
            const int mm = m < p.M ? m : p.M - 1;
            n = mm / HoWo;
            const int rem = mm - n * HoWo;
            y = rem / p.Wo;
            x = rem - y * p.Wo;
            return m < p.M;
        }
        const int ly = i / p.tw, lx = i - ly * p.tw;
        n = bn;
        y = by0 + ly;
        x = bx0 + lx;
        const bool ok = ly < p.th && y < p.Ho && x < p.Wo;
        if (!ok) {
            y = by0;
            x = bx0;
        }
        return ok;
    };

    // ---- fragment read addresses (bytes inside a patch buffer), one per (tap, pixel fragment): the
    //      k-loop below is unrolled over taps so these stay in registers and cost no arithmetic per k-step
    int baddr[9][FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        int n, y, x;
        pixel_of(wp * 64 + j * 16 + l15, n, y, x);
        const int prow = (n * p.Hp + y + p.halo_m1 - G0) * p.WL + (x + p.halo_m1 - X0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = prow + (t / 3) * p.WL + (t % 3);
            baddr[t][j] = r * 64 + ((q ^ (((r >> 2) & 1) << 1)) << 4);
        }
    }
    // weight tile rows are 128 bytes = two k-steps of 32; 16-byte chunk c of row r holds source chunk c ^ (r & 7)
    const int a_row = (wc * (TC / 2) + l15) * 128;  // + fc*16*128
    const int a_k0 = ((q ^ (lane & 7)) << 4);       // k-step 0 of the pair; k-step 1 = a_k0 ^ 64

    // ---- loader addresses
    const int lrow4 = lane >> 2;                 // patch: 16 rows of 64 bytes per 1-KiB piece
    const int sw = ((lane >> 4) & 1) << 1;       // (row >> 2) & 1 of this lane's row inside any 16-row piece
    const int pchunk = ((lane & 3) ^ sw) * 16;   // patch source byte offset inside the 64-byte slice
    const int lrow8 = lane >> 3;                 // weights: 8 rows of 128 bytes per piece
    const int wchunk = ((lane & 7) ^ lrow8) * 16;
    const char* wsrc[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i)
        wsrc[i] = p.w + (size_t)(tile_c * TC + 8 * (wave + 4 * i) + lrow8) * p.w_row_bytes + wchunk;
    const long long gmax = (long long)p.N * p.Hp - 1;
    const char* psrc[PP_MAXPIECES];  // this lane's 16 bytes of patch piece k of slice 0
#pragma unroll
    for (int k = 0; k < PP_MAXPIECES; ++k) {
        int pos = 16 * (wave + 4 * k) + lrow4;
        pos = pos < npos ? pos : npos - 1;
        const int pr = pos / p.WL, pc = pos - pr * p.WL;
        long long g = (long long)G0 + pr;
        g = g < gmax ? g : gmax;
        int c = X0 + pc;
        c = c < p.Wp ? c : p.Wp - 1;
        psrc[k] = p.x + g * p.x_row_stride + (long long)c * p.x_pix_stride + pchunk;
    }
    const int my_pieces = (((npos + 15) >> 4) - wave + 3) >> 2;  // pieces this wave loads per patch

    f32x4 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_weights = [&](int pair, char* dst) {
#pragma unroll
        for (int i = 0; i < NWP; ++i) glds16p(wsrc[i] + pair * 128, dst + (wave + 4 * i) * 1024);
    };
    auto multiply_pair = [&](const char* wcur, const char* pb0, const char* pb1, const int (&o0)[FP],
                             const int (&o1)[FP]) {
        bf16x8 a[2][FC], b[2][FP];
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            a[0][i] = *(const bf16x8*)(wcur + a_row + i * 16 * 128 + a_k0);
            a[1][i] = *(const bf16x8*)(wcur + a_row + i * 16 * 128 + (a_k0 ^ 64));
        }
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            b[0][j] = *(const bf16x8*)(pb0 + o0[j]);
            b[1][j] = *(const bf16x8*)(pb1 + o1[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[h][i], b[h][j], acc[i][j], 0, 0, 0);
    };

    // ---- main loop over PAIRS of k-steps (one barrier per pair, 32 MFMAs per wave in between).  The
    // weight DMA of pair m+1 and up to two pieces of a later slice's patch fly while pair m is multiplied.
    char* const pbuf0 = pbuf;
    char* const pbuf1 = pbuf + p.patch_bytes;
#pragma unroll
    for (int k = 0; k < PP_MAXPIECES; ++k)
        if (k < my_pieces) glds16p(psrc[k], pbuf0 + (wave + 4 * k) * 1024);
    load_weights(0, wbuf);
    int m = 0;  // pair index
    if constexpr (SINGLE) {
        // one 32-channel slice: 9 k-steps + 1 zero-weight padding step = 5 pairs
#pragma unroll
        for (int i = 0; i < 5; ++i, ++m) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (i + 1 < 5) load_weights(m + 1, wbuf + ((m + 1) & 1) * WBUF);
            multiply_pair(wbuf + (m & 1) * WBUF, pbuf0, pbuf0, baddr[2 * i], baddr[i == 4 ? 0 : 2 * i + 1]);
        }
    } else {
        // two slices per trip: 18 k-steps = 9 pairs; slice cc lives in patch buffer 0, slice cc+1 in buffer 1
        for (int cc = 0; cc < p.nchunks; cc += 2) {
#pragma unroll
            for (int i = 0; i < 9; ++i, ++m) {
                const int t0 = (2 * i) % 9, c0 = (2 * i) / 9, t1 = (2 * i + 1) % 9, c1 = (2 * i + 1) / 9;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (m + 1 < p.npairs) load_weights(m + 1, wbuf + ((m + 1) & 1) * WBUF);
                if (i < 4) {
                    // slice cc+1 -> buffer 1 (last read by the previous trip), two pieces per pair
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (2 * i + e < my_pieces)
                            glds16p(psrc[2 * i + e] + (cc + 1) * 64, pbuf1 + (wave + 4 * (2 * i + e)) * 1024);
                } else if (i >= 5 && cc + 2 < p.nchunks) {
                    // slice cc+2 -> buffer 0 (its last reader was pair 4 of this trip)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (2 * (i - 5) + e < my_pieces)
                            glds16p(psrc[2 * (i - 5) + e] + (cc + 2) * 64, pbuf0 + (wave + 4 * (2 * (i - 5) + e)) * 1024);
                }
                multiply_pair(wbuf + (m & 1) * WBUF, c0 ? pbuf1 : pbuf0, c1 ? pbuf1 : pbuf0, baddr[t0], baddr[t1]);
            }
        }
    }
    __syncthreads();

    // ---- epilogue: bias + LeakyReLU (+ residual) in registers, bf16 through a wave-private LDS tile so
    //      that pixels leave as 128 contiguous bytes per wave half-row
    char* const stage = smem + wave * (64 * 128);  // 64 pixels x 64 channels bf16 = 8 KiB per wave
    {
        const int ch0 = tile_c * TC + wc * (TC / 2) + q * 4;
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            int n, y, x;
            const bool ok = pixel_of(wp * 64 + j * 16 + l15, n, y, x);
            const char* rp = p.res ? p.res + p.r_origin + (long long)n * p.r_img_stride + (long long)y * p.r_row_stride +
                                         x * p.r_pix_stride
                                   : nullptr;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int ch = ch0 + i * 16;
                const f32x4 bv = *(const f32x4*)(p.bias + ch);
                f32x4 v = acc[i][j] + bv;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * p.slope;
                if (rp && ok && ch < p.Cout) {
                    const u32x2 r = *(const u32x2*)(rp + ch * 2);
                    v[0] += __uint_as_float(r[0] << 16);
                    v[1] += __uint_as_float(r[0] & 0xffff0000u);
                    v[2] += __uint_as_float(r[1] << 16);
                    v[3] += __uint_as_float(r[1] & 0xffff0000u);
                }
                u32x2 o;
                o[0] = pack_bf16x2(v[0], v[1]);
                o[1] = pack_bf16x2(v[2], v[3]);
                // pixel row (j*16 + l15), 8-byte slot (i*4 + q), slots XOR-swizzled by the pixel
                const int prw = j * 16 + l15;
                *(u32x2*)(stage + prw * 128 + (((i * 4 + q) ^ (prw & 15)) << 3)) = o;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    {
        constexpr int CPW = TC / 2;          // channels per wave
        constexpr int LPP = CPW / 8;         // lanes per pixel (16-byte pieces)
        constexpr int PPI = 64 / LPP;        // pixels per iteration
        const int piece = lane % LPP, pl = lane / LPP;
        const int ch = tile_c * TC + wc * CPW + piece * 8;
#pragma unroll 1
        for (int it = 0; it < 64 / PPI; ++it) {
            const int prw = it * PPI + pl;
            int n, y, x;
            const bool ok = pixel_of(wp * 64 + prw, n, y, x);
            const u32x2 lo = *(const u32x2*)(stage + prw * 128 + (((2 * piece) ^ (prw & 15)) << 3));
            const u32x2 hi = *(const u32x2*)(stage + prw * 128 + (((2 * piece + 1) ^ (prw & 15)) << 3));
            if (ok && ch < p.Cout) {
                char* o = p.y + p.y_origin + (long long)n * p.y_img_stride + (long long)y * p.y_row_stride +
                          x * p.y_pix_stride + ch * 2;
                *(u32x4*)o = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
    }
}

struct PatchPlan {
    int mode, th, tw, tiles_x, tiles_y, WL, NRmax, n_ptiles;
};

// Pixel-tile shape for one layer: whole padded rows when that patch is small, else a 2-D block.
static PatchPlan plan_patch(int N, int Ho, int Wo, int Wp, int halo) {
    PatchPlan pl{};
    const int rows_spanned = (PP_TP - 2) / Wo + 2;   // output rows a run of 128 pixels can touch
    const int nr_rows = rows_spanned + 2 + 2 * halo;  // + 2 filter rows + the halo rows between two images
    if (nr_rows * Wp <= 448) {
        pl.mode = 0;
        pl.WL = Wp;
        pl.NRmax = nr_rows;
        pl.n_ptiles = (N * Ho * Wo + PP_TP - 1) / PP_TP;
        return pl;
    }
    pl.mode = 1;
    auto tiles = [&](int th, int tw) { return ((Ho + th - 1) / th) * ((Wo + tw - 1) / tw); };
    const int cand[4][2] = {{8, 16}, {16, 8}, {4, 32}, {32, 4}};
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (tiles(cand[i][0], cand[i][1]) < tiles(cand[best][0], cand[best][1])) best = i;
    pl.th = cand[best][0];
    pl.tw = cand[best][1];
    pl.tiles_y = (Ho + pl.th - 1) / pl.th;
    pl.tiles_x = (Wo + pl.tw - 1) / pl.tw;
    pl.WL = pl.tw + 2;
    pl.NRmax = pl.th + 2;
    pl.n_ptiles = N * pl.tiles_x * pl.tiles_y;
    return pl;
}

template <int TC, bool SINGLE>
static int launch_patch(const PatchParams& p, hipStream_t stream) {
    const int lds = 2 * TC * 128 + p.patch_bufs * p.patch_bytes;
    const int lds_alloc = lds > 4 * 64 * 128 ? lds : 4 * 64 * 128;  // the epilogue needs 32 KiB
    auto kern = conv3x3_patch_kernel<TC, SINGLE>;
    static bool attr_set = false;
    if (!attr_set) {
        YOLO_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.n_ctiles * p.n_ptiles), dim3(256), lds_alloc, stream, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_conv3x3_patch_fwd(const yolo_view* x, const void* w_packed, const float* bias,
                                      const yolo_view* residual, const yolo_view* y, int N, int Cin, int Cout,
                                      float leaky_slope, void* stream_) {
    YOLO_REQUIRE(x && y && w_packed && bias, "conv3x3_patch: null argument");
    YOLO_REQUIRE(N > 0 && Cout > 0 && (Cin == 32 || (Cin > 0 && Cin % 64 == 0)),
                 "conv3x3_patch: Cin %d must be 32 or a multiple of 64", Cin);
    YOLO_REQUIRE(x->halo >= 1, "conv3x3_patch: input halo %d < 1", x->halo);
    YOLO_REQUIRE(y->H == x->H && y->W == x->W, "conv3x3_patch: output view must match the input size");
    YOLO_REQUIRE(x->ch_offset % 8 == 0 && x->ch_total % 8 == 0 && x->ch_offset + Cin <= x->ch_total,
                 "conv3x3_patch: bad input channel slice");
    const int Cout8 = (Cout + 7) / 8 * 8;
    YOLO_REQUIRE(y->ch_offset % 8 == 0 && y->ch_total % 8 == 0 && y->ch_offset + Cout8 <= y->ch_total,
                 "conv3x3_patch: bad output channel slice");
    YOLO_REQUIRE((long long)N * x->H * x->W < (1ll << 31), "conv3x3_patch: too many pixels");
    if (residual)
        YOLO_REQUIRE(residual->H == y->H && residual->W == y->W && residual->ch_offset % 8 == 0 &&
                         residual->ch_total % 8 == 0 && residual->ch_offset + Cout8 <= residual->ch_total,
                     "conv3x3_patch: residual view mismatch");
    const float slope = leaky_slope;
    hipStream_t stream = (hipStream_t)stream_;
    const ViewGeom gx = geom_of(*x), gy = geom_of(*y);
    PatchParams p{};
    const int Hp = x->H + 2 * x->halo, Wp = x->W + 2 * x->halo;
    p.x = (const char*)x->ptr + (size_t)x->ch_offset * 2;
    p.w = (const char*)w_packed;
    p.bias = bias;
    p.res = residual ? (const char*)residual->ptr : nullptr;
    p.y = (char*)y->ptr;
    p.x_row_stride = (long long)gx.row_stride * 2;
    p.x_pix_stride = gx.pix_stride * 2;
    p.Hp = Hp;
    p.Wp = Wp;
    p.halo_m1 = x->halo - 1;
    p.y_img_stride = gy.img_stride * 2;
    p.y_row_stride = gy.row_stride * 2;
    p.y_pix_stride = gy.pix_stride * 2;
    p.y_origin = (long long)gy.origin * 2;
    if (residual) {
        const ViewGeom gr = geom_of(*residual);
        p.r_img_stride = gr.img_stride * 2;
        p.r_row_stride = gr.row_stride * 2;
        p.r_pix_stride = gr.pix_stride * 2;
        p.r_origin = (long long)gr.origin * 2;
    }
    p.N = N;
    p.Ho = x->H;
    p.Wo = x->W;
    p.M = N * x->H * x->W;
    p.nchunks = Cin / 32;
    p.npairs = (p.nchunks * 9 + 1) / 2;
    p.w_row_bytes = p.npairs * 128;
    p.Cout = Cout;
    p.slope = slope;
    const PatchPlan pl = plan_patch(N, p.Ho, p.Wo, Wp, x->halo);
    p.mode = pl.mode;
    p.th = pl.th;
    p.tw = pl.tw;
    p.tiles_x = pl.tiles_x;
    p.tiles_y = pl.tiles_y;
    p.WL = pl.WL;
    p.NRmax = pl.NRmax;
    p.n_ptiles = pl.n_ptiles;
    p.patch_bytes = (pl.NRmax * pl.WL * 64 + 1023) / 1024 * 1024;
    YOLO_REQUIRE(p.patch_bytes <= PP_MAXPIECES * 4 * 1024, "conv2d: input patch of %d rows does not fit", pl.NRmax * pl.WL);
    p.patch_bufs = p.nchunks > 1 ? 2 : 1;
    const int tc = Cout > 64 ? 128 : 64;
    p.n_ctiles = (Cout + tc - 1) / tc;
    if (p.nchunks == 1) return tc == 128 ? launch_patch<128, true>(p, stream) : launch_patch<64, true>(p, stream);
    return tc == 128 ? launch_patch<128, false>(p, stream) : launch_patch<64, false>(p, stream);
}

// Implicit-GEMM convolution for gfx950 (MI355X): bf16 / f32 MFMA, LDS-staged operand tiles filled
// by direct global->LDS loads, folded-BatchNorm bias + LeakyReLU (+ residual add, + 2x nearest
// upsample, + write into a channel slice of a wider buffer) fused into the epilogue.
//
// Replaces, per call, detectron2 Conv2d.forward + F.leaky_relu (+ `out += residual`, + nn.Upsample,
// + torch.cat) of the reference: yolo/modeling/backbone/darknet.py:58-59,113-126 and
// yolo/modeling/backbone/fpn.py:33-37,66-71,96-100,179.
//
// GEMM view: D[channel][pixel] = sum_k W[channel][k] * X[pixel][k], k = (kh, kw, cin) with cin
// innermost.  Both operand tiles are "row x 128 bytes of k" images in LDS, so weights (MFMA A
// operand) and activations (MFMA B operand) share one loader and one fragment-read pattern.  With
// channels on the MFMA row index each lane ends up holding 4 consecutive channels of one pixel,
// which is what an NHWC epilogue wants.
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"

namespace yolo {

struct ConvParams {
    const char* x;
    const char* w;
    const float* bias;
    const char* res;
    char* y;
    long long x_img_stride;  // bytes
    long long x_origin;      // bytes: buffer position of logical pixel (-pad,-pad), channel 0
    int x_row_stride;        // bytes
    int x_pix_stride;        // bytes
    long long y_img_stride;  // bytes
    long long y_origin;      // bytes
    int y_row_stride, y_pix_stride;
    long long r_img_stride, r_origin;
    int r_row_stride, r_pix_stride;
    int Ho, Wo, M;      // conv output size, M = N*Ho*Wo
    float inv_HoWo, inv_Wo;  // reciprocals for the pixel -> (n, ho, wo) split; used when M < 2^24 (fast_div)
    int fast_div;
    int stride, ksize;  // conv stride, kernel size
    int cpt;            // 128-byte k-steps per tap (Cin*esize/128)
    int nsteps;         // k-steps in total
    int w_row_bytes;    // bytes per packed weight row
    int Cout;
    int n_ctiles, n_ptiles;
    float slope;
    int upsample;
    int y_f32;  // output element type is f32 (else bf16)
    int Mc;        // S2D kernels: pixels per parity class (N * Hi * Wi); n_ptiles = 4 * tiles per class
    int Wreal;     // STRIP kernels: real output width; p.Wo is then the row length INCLUDING the two halo columns
    int split_steps;            // split-K launches: k-steps per split (0 = the whole k range in one workgroup)
    long long y_split_stride;   // split-K launches: bytes between the f32 partial-sum slabs of consecutive splits
    float* stats;  // STATS kernels: per pixel tile, per channel: [n_ptiles][2][Cout] sums of y and y*y
    // STATS kernels, BatchNorm-backward form (bz != null): y is the gradient dL/da of an activation a = leaky(bn(z)) and
    // the sums are (sum g, sum g * z) with g = y * leaky'(z * bz_scale + bz_shift): the reduction pass of that
    // BatchNorm's backward, taken from the epilogue that writes the gradient.
    const char* bz;
    long long bz_img_stride, bz_origin;  // bytes
    int bz_row_stride, bz_pix_stride;
    const float* bz_scale;  // [Cout]: gamma * invstd
    const float* bz_shift;  // [Cout]: beta - mean * gamma * invstd
    float bz_slope;
};

constexpr int ROWB = 128;  // bytes of k per tile row and per k-step

#ifdef YOLO_CONV_TRACE
// Phase timestamps of every workgroup of the last launch (measurement builds only): entry, first operands landed,
// loop end, epilogue end, in 10 ns ticks of the constant clock, plus the hardware id; slots 5..8: per epilogue half,
// tile staged in LDS / stores issued; slot 9: addresses ready, before the first operand loads are issued.
constexpr int TRACE_MAX = 1 << 15;
__device__ unsigned long long g_trace[TRACE_MAX * 10];
// shader-clock counter at the loop's first and last stamp: cycles / (10 ns ticks) x 100 MHz = the clock the CU held
__device__ unsigned long long g_cycles[TRACE_MAX * 2];
#define TRACE_MARK(i)                                                                      \
    do {                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < TRACE_MAX) {                                  \
            g_trace[blockIdx.x * 10 + (i)] = wall_clock64();                               \
            if ((i) == 1 || (i) == 2) g_cycles[blockIdx.x * 2 + (i) - 1] = __builtin_readcyclecounter(); \
        }                                                                                  \
    } while (0)
#else
#define TRACE_MARK(i) \
    do {              \
    } while (0)
#endif

// m -> (n, ho, wo).  Integer division by a run-time divisor costs ~40 instructions; for m < 2^24 a float
// reciprocal and one correction step are exact and cost ~8.  Every lane does this for each of its tile rows in
// the prologue and once per epilogue pass, which is felt on short-K layers.
__device__ __forceinline__ void split_pixel(const ConvParams& p, int m, int HoWo, int& n, int& ho, int& wo) {
    if (p.fast_div) {
        n = (int)((float)m * p.inv_HoWo);
        int r = m - n * HoWo;
        if (r < 0) {
            --n;
            r += HoWo;
        } else if (r >= HoWo) {
            ++n;
            r -= HoWo;
        }
        ho = (int)((float)r * p.inv_Wo);
        wo = r - ho * p.Wo;
        if (wo < 0) {
            --ho;
            wo += p.Wo;
        } else if (wo >= p.Wo) {
            ++ho;
            wo -= p.Wo;
        }
    } else {
        n = m / HoWo;
        const int r = m - n * HoWo;
        ho = r / p.Wo;
        wo = r - ho * p.Wo;
    }
}

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// T: element type of x / w / residual.  TC x TP: channel x pixel tile of one workgroup (4 waves laid
// out WC x WP).  TAP2: bf16 with Cin == 32, one k-step spans two filter taps.
// STATS: additionally leave, per pixel tile, the per-channel sums of the output values as stored (after rounding
// to the output type) and of their squares -- the first pass of a training-mode BatchNorm over this output; or (p.bz
// set, bf16) the two sums of the BatchNorm-backward reduction of the layer whose output gradient the launch writes.
//
// S2D: data gradient of a 3x3 / stride-2 / pad-1 convolution without materialising the zero-dilated gradient.
// x is dz [Hi x Wi], y is dx [2Hi x 2Wi]; output pixel (2i+a, 2j+b) only meets the taps kh' with a+kh'-1 even
// (one for a = 0, two for a = 1; likewise kw'), at dz row i + (a+kh'-1)/2.  The pixels of a launch are
// enumerated class by class ((a,b) = 00, 01, 10, 11), every tile lies inside one class and runs 1, 2, 2 or 4 of
// the 9 taps: 9 tap-visits per 4 output pixels instead of 36.  p.Ho/p.Wo/p.Mc describe ONE class (= dz's grid).
//
// STRIP: 3x3 / stride-1 / pad-1 convolution whose three kw taps share ONE staged strip of pixel rows.  The pixels of
// a launch are enumerated over the rows of the halo'd input INCLUDING their two halo columns ("positions":
// v = (n * H + row) * (W + 2) + column, column 0 and W + 1 being halo); consecutive positions are consecutive pixels in
// memory, so the tap (kh, kw) of position v is the pixel of position v + kw - 1 one buffer row further down per kh.
// Per (kh, 64-channel chunk) the workgroup stages ONE strip of TP + 2 pixel rows and reads it three times at row
// offsets 0, 1, 2; only the weights change per k-step: (3 * TC + TP + 2) * 128 bytes cross the L2 -> LDS path per
// three k-steps instead of 3 * (TC + TP) * 128.  Outputs at the halo columns (2 of W + 2) are computed and dropped.
// The k order is (kh, channel chunk, kw); the weights stay packed (kh, kw, channel).  p.Ho / p.Wo / p.M describe the
// position grid (H, W + 2, N * H * (W + 2)); p.Wreal is W.
//
// ODD: a pixel tile with an ODD number NF of 16-pixel fragment columns (TP = 176: 11) on the 2 x 2 wave layout.  A 128 x 192
// tile needs 57 pixel tiles for the 64 x 13 x 13 pixels of a stride-32 layer (456 workgroups for 512 resident slots) and 226
// for a stride-16 layer (904 for 1024); with 176 pixels it is 62 (496) and 246 (984): the rounds are full.  Both
// pixel-waves of a channel half take (NF - 1) / 2 columns of their own -- interleaved, column 2j + wp, so that the
// first and the second half of the tile's columns each hold columns of both waves -- and SHARE the last column: wave
// wp computes channel fragments 2wp, 2wp + 1 of it.  Every wave then issues the same 22 MFMAs per 32 k (balanced, which
// an uneven 6 + 5 split of columns is not).  So that the shared column's two fragments have compile-time register
// indices in both waves, wave wp keeps channel fragment f in slot f ^ 2wp: slots 0 and 1 are always the shared ones.
template <typename T, int TC, int TP, int WC, int WP, bool TAP2, int NW = 4, bool STATS = false, bool S2D = false,
          bool STRIP = false, bool ODD = false>
// (the instances with sums keep the occupancy their LDS request allows: the second form of the sums needs registers the
// compiler otherwise takes without regard to it)
__global__ void __launch_bounds__(NW * 64, TP >= 128 ? 2 : (STATS ? (TP >= 96 ? 2 : 3) : 1)) conv_igemm_kernel(const ConvParams p) {
    static_assert(!(S2D && (TAP2 || STATS)), "S2D is a plain bf16/f32 convolution");
    static_assert(!(STRIP && (TAP2 || S2D || (STATS && !ODD) || sizeof(T) != 2)), "STRIP is a plain bf16 3x3 convolution");
    static_assert(WC * WP == NW, "waves");
    static_assert(!ODD || (sizeof(T) == 2 && TC == 128 && WC == 2 && WP == 2 && NW == 4 && (TP / 16) % 2 == 1 && TP % 16 == 0 &&
                           !TAP2 && !S2D),
                  "ODD is a plain bf16 convolution on 2 x 2 waves with an odd number of pixel fragment columns");
    constexpr bool BF = sizeof(T) == 2;
    constexpr int NF = TP / 16;  // pixel fragment columns of the tile
    constexpr int SP = (TP + 2 + 7) / 8;          // STRIP: 1-KiB pieces (8 pixel rows) of a strip
    constexpr int NXS = (SP + NW - 1) / NW;       // STRIP: strip pieces per wave
    constexpr int XS_BYTES = SP * 1024;
    constexpr int W_BYTES = TC * ROWB, X_BYTES = STRIP ? XS_BYTES : TP * ROWB, BUF = W_BYTES + X_BYTES;
    constexpr int FC = TC / (16 * WC), FP = ODD ? (NF - 1) / 2 : TP / (16 * WP);  // FP: a wave's own pixel columns
    constexpr int XPIECES = TP / 8;                                   // 1-KiB pieces (8 pixel rows) of the pixel operand
    constexpr int NWP = TC / (8 * NW), NXP = (XPIECES + NW - 1) / NW;  // pieces per wave per operand
    constexpr bool XEVEN = XPIECES % NW == 0;                         // else the last round of pieces is partial
    static_assert(TC % (8 * NW) == 0 && TP % 8 == 0 && (ODD || XEVEN), "whole pieces per wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    TRACE_MARK(0);
#ifdef YOLO_CONV_TRACE
    if (tid == 0 && blockIdx.x < TRACE_MAX) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_trace[blockIdx.x * 10 + 4] = ((unsigned long long)xcc << 32) | hw;
    }
#endif

    // Workgroup -> tile.  Blocks b and b+8 share an XCD (private L2): give each XCD a contiguous
    // run of tiles, channel tiles of one pixel tile adjacent, so the activation tile is fetched
    // from HBM once per XCD.
    int tile_c, tile_p;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
        int j = bid >> 3;
#ifndef YOLO_NO_PAIR_CU
        // An XCD's workgroups are dealt to its 32 CUs in turn, so the j-th and the (j + 32)-th workgroup of an XCD
        // share a CU: they get neighbouring tiles (same pixel tile, next channel tile), so that the pixel operand the
        // second one asks for is still in the CU's vector cache.  Placement is not guaranteed: speed only (measured
        // 2 % on the 26 x 26 and 13 x 13 3x3 layers; mixed on the strip form, which keeps the plain order).
        if (!STRIP && !p.y_f32) {  // (the f32 head outputs, bound by their stores, measured 1-3 % slower paired)
            const int run = xcd < r ? q + 1 : q;              // tiles of this XCD
            const int grp = j >> 6, in = j & 63;
            if ((grp + 1) * 64 <= run) j = grp * 64 + ((in & 31) << 1) + (in >> 5);
        }
#endif
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        tile_c = logical % p.n_ctiles;
        tile_p = logical / p.n_ctiles;
    }
    // split-K launch: the grid holds `splits` copies of the tile grid; copy s sums k-steps [s * split_steps, ...) into
    // its own f32 slab (no bias, no activation: conv_splitk_reduce_kernel finishes the layer)
    int split = 0;
    if (p.split_steps) {
        split = tile_p / p.n_ptiles;
        tile_p -= split * p.n_ptiles;
    }
    char* const ybase = p.y + (long long)split * p.y_split_stride;
    // S2D: parity class of this tile and its pixel range inside the class
    int par_a = 0, par_b = 0, pix_base = tile_p * TP, pix_end = p.M;
    if constexpr (S2D) {
        const int tpc = p.n_ptiles >> 2;
        const int cls = tile_p / tpc;
        par_a = cls >> 1;
        par_b = cls & 1;
        pix_base = (tile_p - cls * tpc) * TP;
        pix_end = p.Mc;
    }

    // ---- loader set-up: each lane owns one 16-byte chunk of NWP weight rows and NXP pixel rows
    const int lrow = lane >> 3;
    const int schunk = (lane & 7) ^ lrow;  // source chunk; LDS chunk (lane&7) of row r holds chunk (lane&7)^(r&7)
    const char* wsrc[NWP];
    const char* xsrc[STRIP ? NXS : NXP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int row = 8 * (wave + NW * i) + lrow;
        wsrc[i] = p.w + (size_t)(tile_c * TC + row) * p.w_row_bytes + schunk * 16;
    }
    // The first k-step's weight rows are requested now, ahead of the pixel-address arithmetic below (a pixel ->
    // (image, row, column) split and carried steps per tile row): their flight time covers it.
#ifndef YOLO_NO_EARLY_W
    constexpr bool EARLY_W = !S2D;
#else
    constexpr bool EARLY_W = false;
#endif
    if constexpr (EARLY_W) {
        const int woff0 = (p.split_steps ? split * p.split_steps : 0) * ROWB;
#pragma unroll
        for (int i = 0; i < NWP; ++i) glds16(wsrc[i] + woff0, smem + (wave + NW * i) * 1024);
    }
    // ... and the first k-step's pixel rows are requested row by row as their addresses come out of that arithmetic.
#ifndef YOLO_NO_EARLY_X
    constexpr bool EARLY_X = EARLY_W && !TAP2;
#else
    constexpr bool EARLY_X = false;
#endif
    int xoff0 = 0;
    if (EARLY_X && p.split_steps) {  // first k-step of this split -> (tap, channel chunk)
        const int s0 = split * p.split_steps;
        const int tap = s0 / p.cpt, cc0 = s0 - tap * p.cpt, kh0 = tap / p.ksize, kw0 = tap - kh0 * p.ksize;
        xoff0 = kh0 * p.x_row_stride + kw0 * p.x_pix_stride + cc0 * ROWB;
    }
    const int HoWo = p.Ho * p.Wo;
    {
        // this lane's pixel rows are 8 * NW pixels apart: one pixel -> (image, row, column) split, then carried steps
        constexpr int STEP = 8 * NW;
        const int xs_rows = STEP / p.Wo, xs_cols = STEP - xs_rows * p.Wo;
        const bool xtiny = xs_rows + 1 > p.Ho;  // images smaller than a step: split every row
        // STRIP: strip row r is position pix_base - 1 + r (the tile's first output needs its left neighbour)
        const int mfirst = pix_base + 8 * wave + lrow - (STRIP ? 1 : 0);
        const int mclamp = mfirst < 0 ? 0 : mfirst;
        int n, ho, wo;
        split_pixel(p, mclamp < pix_end ? mclamp : pix_end - 1, HoWo, n, ho, wo);
        const char* first = nullptr;
        constexpr int NX = STRIP ? NXS : NXP;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const char* a = p.x + p.x_origin + (long long)n * p.x_img_stride + (long long)(ho * p.stride) * p.x_row_stride +
                            (wo * p.stride) * p.x_pix_stride + (TAP2 ? (schunk & 3) : schunk) * 16;
            if (i == 0) {
                first = a;
                if (STRIP && mfirst < 0) --wo;  // position -1 read position 0 (its output is a halo column); step on from -1
            }
            xsrc[i] = mfirst + STEP * i < pix_end ? a : first;  // tail rows read a valid pixel; never stored
            if constexpr (EARLY_X) {
                if constexpr (STRIP) {
                    if (wave + NW * i < SP) glds16(xsrc[i] + xoff0, smem + 2 * W_BYTES + (wave + NW * i) * 1024);
                } else {
                    if (XEVEN || wave + NW * i < XPIECES) glds16(xsrc[i] + xoff0, smem + W_BYTES + (wave + NW * i) * 1024);
                }
            }
            if (i + 1 < NX) {
                if (xtiny) {
                    const int m = mfirst + STEP * (i + 1);
                    split_pixel(p, m < pix_end ? m : pix_end - 1, HoWo, n, ho, wo);
                } else {
                    wo += xs_cols;
                    ho += xs_rows;
                    if (wo >= p.Wo) {
                        wo -= p.Wo;
                        ++ho;
                    }
                    if (ho >= p.Ho) {
                        ho -= p.Ho;
                        ++n;
                    }
                }
            }
        }
    }
    const bool tapsel = (schunk >> 2) != 0;

    // k-step cursor (wave-uniform)
    int cur_cc = 0, cur_kh = 0, cur_kw = 0, cur_step = 0;
    if (p.split_steps) {  // first k-step of this split -> (tap, channel chunk)
        cur_step = split * p.split_steps;
        const int tap = cur_step / p.cpt;
        cur_cc = cur_step - tap * p.cpt;
        cur_kh = tap / p.ksize;
        cur_kw = tap - cur_kh * p.ksize;
    }
    auto stage = [&](char* buf, bool skip_w = false, bool skip_x = false) {
        int xoff;
        if constexpr (TAP2) {
            const int t0 = 2 * cur_step;
            int t1 = t0 + 1;
            t1 = t1 > 8 ? 8 : t1;  // tap 9 does not exist: its weights are zero, read tap 8 again
            const int o0 = (t0 / 3) * p.x_row_stride + (t0 % 3) * p.x_pix_stride;
            const int o1 = (t1 / 3) * p.x_row_stride + (t1 % 3) * p.x_pix_stride;
            xoff = tapsel ? o1 : o0;
        } else {
            xoff = cur_kh * p.x_row_stride + cur_kw * p.x_pix_stride + cur_cc * ROWB;
        }
        int woff = cur_step * ROWB;
        if constexpr (S2D) {  // cur_kh / cur_kw count the class's taps: tap kh' = 2*cur_kh (a = 1) or 1 (a = 0)
            const int kh = par_a ? 2 * cur_kh : 1, kw = par_b ? 2 * cur_kw : 1;
            woff = ((kh * 3 + kw) * p.cpt + cur_cc) * ROWB;
        }
        if (!skip_w) {
#pragma unroll
            for (int i = 0; i < NWP; ++i) glds16(wsrc[i] + woff, buf + (wave + NW * i) * 1024);
        }
        if (!skip_x) {
#pragma unroll
            for (int i = 0; i < NXP; ++i)
                if (XEVEN || wave + NW * i < XPIECES) glds16(xsrc[i] + xoff, buf + W_BYTES + (wave + NW * i) * 1024);
        }
        ++cur_step;
        if (++cur_cc == p.cpt) {
            cur_cc = 0;
            if (++cur_kw == (S2D ? 1 + par_b : p.ksize)) {
                cur_kw = 0;
                ++cur_kh;
            }
        }
    };

    // STRIP stagers: the weights of one k-step; one third (PART = 0..2, or all: -1) of the strip of (kh, chunk cc)
    [[maybe_unused]] auto stage_w = [&](char* wbuf, int kh, int kw, int cc) {
        const int woff = ((kh * 3 + kw) * p.cpt + cc) * ROWB;
#pragma unroll
        for (int i = 0; i < NWP; ++i) glds16(wsrc[i] + woff, wbuf + (wave + NW * i) * 1024);
    };
    [[maybe_unused]] auto stage_x = [&](char* xbuf, int kh, int cc, auto part_c) {
        constexpr int PART = decltype(part_c)::value;
        const int xoff = kh * p.x_row_stride + cc * ROWB;
#pragma unroll
        for (int i = 0; i < NXS; ++i) {
            if (PART >= 0 && i % 3 != PART) continue;
            if (wave + NW * i < SP) glds16(xsrc[i] + xoff, xbuf + (wave + NW * i) * 1024);
        }
    };

    // ---- fragment addressing
    const int wc = wave / WP, wp = wave % WP;
    const int l15 = lane & 15, q = lane >> 4, l7 = lane & 7;
    const int a_base = (wc * (TC / WC) + l15) * ROWB;            // + fc*16*ROWB
    // pixel columns of a wave: contiguous (column wp * FP + j), ODD: interleaved (column 2j + wp) + the shared last one
    constexpr int BSTEP = ODD ? 32 : 16;                         // tile rows between a wave's consecutive columns
    const int b_base = W_BYTES + ((ODD ? wp * 16 : wp * (TP / WP)) + l15) * ROWB;  // + j*BSTEP*ROWB
    [[maybe_unused]] const int bs_base = W_BYTES + ((NF - 1) * 16 + l15) * ROWB;  // ODD: the shared column
    const int ko0 = (q ^ l7) << 4;                               // k chunk q, swizzled
    const int ko1 = ko0 ^ 64;                                    // k chunk 4+q
    // channel fragment held in accumulator slot i of this wave (ODD: slot i = fragment i ^ 2wp), as a tile-row offset
    auto frag_of_slot = [&](int i) { return ODD ? (i ^ (2 * wp)) : i; };
    [[maybe_unused]] const int a_sw = ODD ? 2 * wp * 16 * ROWB : 0;  // ODD: slots 0,1 / 2,3 swap places when wp = 1
    auto a_off = [&](int i) { return ODD ? a_base + ((i * 16 * ROWB) ^ a_sw) : a_base + i * 16 * ROWB; };

    // accumulators start from the bias (lane holds channels ch0 + 16*frag + 0..3 of pixel column l15)
    f32x4 acc[FC][FP];
    [[maybe_unused]] f32x4 acc_sh[2];  // ODD: slots 0, 1 of the shared column
#pragma unroll
    for (int i = 0; i < FC; ++i) {
        f32x4 bv = *(const f32x4*)(p.bias + tile_c * TC + wc * (TC / WC) + q * 4 + frag_of_slot(i) * 16);
        if (p.split_steps) bv = f32x4{0.f, 0.f, 0.f, 0.f};  // partial sums: the bias is added once, by the reduction
#pragma unroll
        for (int j = 0; j < FP; ++j) acc[i][j] = bv;
        if (ODD && i < 2) acc_sh[i] = bv;
    }

    // All fragment reads of a k-step (2 sub-steps of 32 k) are issued up front; the second half lands
    // while the matrix cores work on the first.
    // STRIP: weights and strip live in separate buffers; tap kw reads the strip KW rows further down, and the
    // swizzle term follows the row actually read.
    [[maybe_unused]] auto compute_strip = [&](const char* wbuf, const char* xbuf, auto kw_c) {
        constexpr int KW = decltype(kw_c)::value;
        const int kb0 = (q ^ ((l15 + KW) & 7)) << 4;
        const char* xb = xbuf + ((ODD ? wp * 16 : wp * (TP / WP)) + l15 + KW) * ROWB;
        [[maybe_unused]] const char* xsh = xbuf + ((NF - 1) * 16 + l15 + KW) * ROWB;  // ODD: the shared column
        bf16x8 a[2][FC], b[2][FP];
        [[maybe_unused]] bf16x8 bsh[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ko = kk ? ko1 : ko0;
            const int kb = kk ? (kb0 ^ 64) : kb0;
#pragma unroll
            for (int i = 0; i < FC; ++i) a[kk][i] = *(const bf16x8*)(wbuf + a_off(i) + ko);
#pragma unroll
            for (int j = 0; j < FP; ++j) b[kk][j] = *(const bf16x8*)(xb + j * BSTEP * ROWB + kb);
            if constexpr (ODD) bsh[kk] = *(const bf16x8*)(xsh + kb);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
            if constexpr (ODD) {
                acc_sh[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][0], bsh[kk], acc_sh[0], 0, 0, 0);
                acc_sh[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][1], bsh[kk], acc_sh[1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto compute = [&](const char* buf) {
        if constexpr (BF) {
            bf16x8 a[2][FC], b[2][FP];
            [[maybe_unused]] bf16x8 bsh[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ko = kk ? ko1 : ko0;
#pragma unroll
                for (int i = 0; i < FC; ++i) a[kk][i] = *(const bf16x8*)(buf + a_off(i) + ko);
#pragma unroll
                for (int j = 0; j < FP; ++j) b[kk][j] = *(const bf16x8*)(buf + b_base + j * BSTEP * ROWB + ko);
                if constexpr (ODD) bsh[kk] = *(const bf16x8*)(buf + bs_base + ko);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMA block
            __builtin_amdgcn_s_setprio(1);  // matrix work first when both co-resident workgroups have some (+1.5-2 %)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
                if constexpr (ODD) {
                    acc_sh[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][0], bsh[kk], acc_sh[0], 0, 0, 0);
                    acc_sh[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][1], bsh[kk], acc_sh[1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        } else {
            // f32 MFMA 16x16x4 takes ONE k per lane group; element e of each lane's 16-byte chunk feeds
            // MFMA e, the same k permutation on both operands.
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ko = kk ? ko1 : ko0;
                f32x4 a[FC], b[FP];
#pragma unroll
                for (int i = 0; i < FC; ++i) a[i] = *(const f32x4*)(buf + a_base + i * 16 * ROWB + ko);
#pragma unroll
                for (int j = 0; j < FP; ++j) b[j] = *(const f32x4*)(buf + b_base + j * 16 * ROWB + ko);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < FC; ++i)
#pragma unroll
                        for (int j = 0; j < FP; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ODD: the residual values of the epilogue's first half are requested under the LAST k-step's multiplications (the
    // loops below are peeled for it: a branch inside the loop body costs the loop its schedule), so that their
    // latency -- a read that mostly hits the Infinity Cache, ~1 us under load -- is not waited for at the first
    // staging barrier.  One lane = 8 channels of one pixel per pass, as in the epilogue's store passes.
#ifdef YOLO_NO_RES_PREFETCH
    constexpr bool RES_PREFETCH = false;   // (measurement builds: the residual is requested in the epilogue)
#else
    constexpr bool RES_PREFETCH = ODD;
#endif
    constexpr int ODD_J0 = (FP + 1) / 2, ODD_C0 = 2 * ODD_J0;   // tile columns [0, C0) form the first half
    constexpr int ODD_NP0 = ODD ? ODD_C0 * 16 / (NW * 64 / (TC / 8)) : 1;
    [[maybe_unused]] u32x4 rres0[ODD_NP0];
    [[maybe_unused]] auto prefetch_residual = [&]() {
        if constexpr (ODD) {
            constexpr int G_ = TC / 8, PPP_ = NW * 64 / G_;
            const int ch_first_ = tile_c * TC + (tid % G_) * 8;
            const int m0 = pix_base + tid / G_;
            const int adv_r = PPP_ / p.Wo, adv_c = PPP_ - adv_r * p.Wo;
            int n, ho, wo;
            split_pixel(p, m0 < pix_end ? m0 : pix_end - 1, HoWo, n, ho, wo);
            auto col = [&](int w_) {
                if constexpr (STRIP) {
                    const int c = w_ - 1;
                    return c < 0 ? 0 : (c >= p.Wreal ? p.Wreal - 1 : c);
                } else {
                    return w_;
                }
            };
            if (ch_first_ < p.Cout) {
                const long long ro_first = p.r_origin + (long long)n * p.r_img_stride + (long long)ho * p.r_row_stride +
                                           col(wo) * p.r_pix_stride + (long long)ch_first_ * 2;
#pragma unroll
                for (int pass = 0; pass < ODD_NP0; ++pass) {
                    const long long ro = p.r_origin + (long long)n * p.r_img_stride + (long long)ho * p.r_row_stride +
                                         col(wo) * p.r_pix_stride + (long long)ch_first_ * 2;
                    rres0[pass] = *(const u32x4*)(p.res + (m0 + pass * PPP_ < pix_end ? ro : ro_first));
                    wo += adv_c;
                    ho += adv_r;
                    if (wo >= p.Wo) {
                        wo -= p.Wo;
                        ++ho;
                    }
                    if (ho >= p.Ho) {
                        ho -= p.Ho;
                        ++n;
                    }
                }
            }
        }
    };

    // ---- main loop: loads of step s+1 fly while step s is multiplied
    int nsteps = S2D ? (1 + par_a) * (1 + par_b) * p.cpt : p.nsteps;
    if (p.split_steps) {
        const int left = p.nsteps - split * p.split_steps;
        nsteps = left < p.split_steps ? left : p.split_steps;
    }
    TRACE_MARK(9);
    if constexpr (STRIP) {
        // LDS: two weight buffers, then two strip buffers.  Group g = (kh, cc) runs three k-steps (kw = 0, 1, 2) on one
        // strip; while it runs, the next step's weights and a third of the next group's strip are in flight.
        char* const wb0 = smem;
        char* const xb0 = smem + 2 * W_BYTES;
        const int ngroups = 3 * p.cpt;
        int kh = 0, cc = 0;        // current group
        if constexpr (!EARLY_X) stage_x(xb0, 0, 0, std::integral_constant<int, -1>{});
        if constexpr (!EARLY_W) stage_w(wb0, 0, 0, 0);
        [[maybe_unused]] auto group = [&](int g, auto more_c) {
            constexpr bool more = decltype(more_c)::value;
            int nkh = kh, ncc = cc + 1;  // next group
            if (ncc == p.cpt) {
                ncc = 0;
                ++nkh;
            }
            const char* xcur = xb0 + (g & 1) * XS_BYTES;
            char* xnext = xb0 + ((g + 1) & 1) * XS_BYTES;
            // the three steps of a group alternate the weight buffers starting from (3 * g) & 1 = g & 1
            char* const wA = wb0 + (g & 1) * W_BYTES;
            char* const wB = wb0 + ((g + 1) & 1) * W_BYTES;
            // kw = 0
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#ifdef YOLO_CONV_TRACE
            if (g == 0) TRACE_MARK(1);
#endif
            stage_w(wB, kh, 1, cc);
            if constexpr (more) stage_x(xnext, nkh, ncc, std::integral_constant<int, 0>{});
            compute_strip(wA, xcur, std::integral_constant<int, 0>{});
            // kw = 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            stage_w(wA, kh, 2, cc);
            if constexpr (more) stage_x(xnext, nkh, ncc, std::integral_constant<int, 1>{});
            compute_strip(wB, xcur, std::integral_constant<int, 1>{});
            // kw = 2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if constexpr (more) {
                stage_w(wB, nkh, 0, ncc);
                stage_x(xnext, nkh, ncc, std::integral_constant<int, 2>{});
            } else if constexpr (RES_PREFETCH) {
                if (p.res) prefetch_residual();
            }
            compute_strip(wA, xcur, std::integral_constant<int, 2>{});
            kh = nkh;
            cc = ncc;
        };
        if constexpr (ODD) {  // last group peeled: no branch in the loop body
            for (int g = 0; g + 1 < ngroups; ++g) group(g, std::true_type{});
            group(ngroups - 1, std::false_type{});
        } else {
            for (int g = 0; g < ngroups; ++g) {
                int nkh = kh, ncc = cc + 1;  // next group
                if (ncc == p.cpt) {
                    ncc = 0;
                    ++nkh;
                }
                const bool more = g + 1 < ngroups;
                const char* xcur = xb0 + (g & 1) * XS_BYTES;
                char* xnext = xb0 + ((g + 1) & 1) * XS_BYTES;
                char* const wA = wb0 + (g & 1) * W_BYTES;
                char* const wB = wb0 + ((g + 1) & 1) * W_BYTES;
                // kw = 0
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#ifdef YOLO_CONV_TRACE
                if (g == 0) TRACE_MARK(1);
#endif
                stage_w(wB, kh, 1, cc);
                if (more) stage_x(xnext, nkh, ncc, std::integral_constant<int, 0>{});
                compute_strip(wA, xcur, std::integral_constant<int, 0>{});
                // kw = 1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                stage_w(wA, kh, 2, cc);
                if (more) stage_x(xnext, nkh, ncc, std::integral_constant<int, 1>{});
                compute_strip(wB, xcur, std::integral_constant<int, 1>{});
                // kw = 2
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (more) {
                    stage_w(wB, nkh, 0, ncc);
                    stage_x(xnext, nkh, ncc, std::integral_constant<int, 2>{});
                }
                compute_strip(wA, xcur, std::integral_constant<int, 2>{});
                kh = nkh;
                cc = ncc;
            }
        }
    } else if constexpr (ODD) {  // last k-step peeled: the first residual values are requested under it
        stage(smem, EARLY_W, EARLY_X);
        int s = 0;
        for (; s + 1 < nsteps; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#ifdef YOLO_CONV_TRACE
            if (s == 0) TRACE_MARK(1);
#endif
            stage(smem + ((s + 1) & 1) * BUF);
            compute(smem + (s & 1) * BUF);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef YOLO_CONV_TRACE
        if (s == 0) TRACE_MARK(1);
#endif
        if constexpr (RES_PREFETCH) {
            if (p.res) prefetch_residual();
        }
        compute(smem + (s & 1) * BUF);
    } else {
    stage(smem, EARLY_W, EARLY_X);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef YOLO_CONV_TRACE
        if (s == 0) TRACE_MARK(1);
#endif
        if (s + 1 < nsteps) stage(smem + ((s + 1) & 1) * BUF);
        compute(smem + (s & 1) * BUF);
    }
    }
    TRACE_MARK(2);
    // ---- epilogue 1: LeakyReLU in registers, f32 tile -> LDS as [pixel][channel], 16-byte chunks XOR-swizzled by
    //      (pixel & 7); epilogue 2: one lane = 8 channels of one pixel, whole pixel rows leave as contiguous bytes.
    //      A tile whose f32 image would not fit the operand buffers goes through in two halves (each wave's first
    //      and second half of pixel columns).
    constexpr int NH = STRIP ? 1 : (TP * TC * 4 > 2 * BUF ? 2 : 1);  // (STRIP launches with LDS for the whole f32 tile)
    constexpr int HP = TP / NH;
    constexpr int G = TC / 8;       // 8-channel groups per pixel
    constexpr int PPP = NW * 64 / G;  // pixels per pass
    static_assert(ODD || HP % PPP == 0, "whole passes");
    const int g = tid % G;
    const int ch_first = tile_c * TC + g * 8;
    int prow = tid / G;  // this lane's pixel row inside a pass
    asm volatile("" : "+v"(prow));  // keeps the epilogue's address arithmetic out of the main loop's registers
    float st_s[8], st_q[8];
    if constexpr (STATS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
    }
    // BatchNorm-backward sums: g = dy * leaky'(z * scale + shift) per element; st_s += g, st_q += g * z (the
    // normalisation of z is applied to the sums afterwards, once per channel)
    [[maybe_unused]] float bsc[STATS ? 8 : 1], bsh[STATS ? 8 : 1];
    if constexpr (STATS) {
        if (p.bz && ch_first < p.Cout) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bsc[e] = p.bz_scale[ch_first + e];
                bsh[e] = p.bz_shift[ch_first + e];
            }
        }
    }
    [[maybe_unused]] auto bz_accumulate = [&](const u32x4 dy, const u32x4 zz) {
        if constexpr (STATS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = __uint_as_float(dy[e] << 16), d1 = __uint_as_float(dy[e] & 0xffff0000u);
                const float z0 = __uint_as_float(zz[e] << 16), z1 = __uint_as_float(zz[e] & 0xffff0000u);
                const float g0 = z0 * bsc[2 * e] + bsh[2 * e] >= 0.f ? d0 : d0 * p.bz_slope;
                const float g1 = z1 * bsc[2 * e + 1] + bsh[2 * e + 1] >= 0.f ? d1 : d1 * p.bz_slope;
                st_s[2 * e] += g0;
                st_q[2 * e] += g0 * z0;
                st_s[2 * e + 1] += g1;
                st_q[2 * e + 1] += g1 * z1;
            }
        }
    };
    [[maybe_unused]] auto bz_offset = [&](int n, int ho, int wo) {  // wo: a real pixel column
        return p.bz_origin + (long long)n * p.bz_img_stride + (long long)ho * p.bz_row_stride + wo * p.bz_pix_stride +
               (long long)ch_first * 2;
    };
    // sums of the stored values and of their squares (the forward form)
    [[maybe_unused]] auto sq_accumulate = [&](const u32x4 packed) {
        if constexpr (STATS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z0 = __uint_as_float(packed[e] << 16);
                const float z1 = __uint_as_float(packed[e] & 0xffff0000u);
                st_s[2 * e] += z0;
                st_q[2 * e] += z0 * z0;
                st_s[2 * e + 1] += z1;
                st_q[2 * e + 1] += z1 * z1;
            }
        }
    };
    // fold the pixel lanes of the workgroup: [which][pixel lane][channel] through LDS (the staged tile is done)
    [[maybe_unused]] auto finish_stats = [&]() {
        if constexpr (STATS) {
            constexpr int PL = NW * 64 / G;
            float* red = (float*)smem;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(tid / G) * TC + g * 8 + e] = st_s[e];
                red[(PL + tid / G) * TC + g * 8 + e] = st_q[e];
            }
            __syncthreads();
            for (int t = tid; t < 2 * TC; t += NW * 64) {
                const int which = t / TC, c = t - which * TC;
                float a = 0.f;
#pragma unroll 4
                for (int l = 0; l < PL; ++l) a += red[(which * PL + l) * TC + c];
                const int cg = tile_c * TC + c;
                if (cg < p.Cout) p.stats[((long long)tile_p * 2 + which) * p.Cout + cg] = a;
            }
        }
    };
    // A tile staged in two halves: every wave stages half of its own pixel columns in each half (all waves write,
    // none waits), so a half holds two runs of HP/2 tile pixels, HP/2 apart: staged row s is tile pixel
    // half * HP/2 + s (+ HP/2 from the second run on).  JUMP = that extra distance, met once per lane and half.
    constexpr int RUN = NH == 2 ? HP / 2 : HP, JUMP = NH == 2 ? HP / 2 : 0;
    static_assert(ODD || (RUN % PPP == 0 && (NH == 1 || FP % 2 == 0)), "runs of whole passes");
    const int adv_rows = PPP / p.Wo, adv_cols = PPP - adv_rows * p.Wo;  // PPP pixels = adv_rows rows + adv_cols columns
    const int jmp_rows = (PPP + JUMP) / p.Wo, jmp_cols = (PPP + JUMP) - jmp_rows * p.Wo;
    const bool tiny = jmp_rows + 1 > p.Ho;
    // ---- bf16 output without residual (two launches of three): the tile is rounded in registers and staged as bf16
    //      -- half the LDS bytes, and a 128 x 192 tile (48 KB) goes through in ONE piece instead of two halves: one
    //      barrier pair less.  Same values as the f32-staged path (LeakyReLU in f32, one rounding).
    if constexpr (BF && TC == 128 && (!STATS || ODD) && !TAP2 && !STRIP) {
        // (the ODD instance is only launched for bf16 output without upsample or split-K on images of at least a step)
        if (!p.res && (ODD || (!p.y_f32 && !p.upsample && !p.split_steps && adv_rows + 1 <= p.Ho))) {
            constexpr int NPB = TP / PPP;
            [[maybe_unused]] u32x4 rz[STATS ? NPB : 1];   // BatchNorm-backward sums: this lane's z values, requested
            if constexpr (STATS) {                          // before the tile is staged
                if (p.bz && ch_first < p.Cout) {
                    const int mz = pix_base + prow;
                    int n, ho, wo;
                    split_pixel(p, mz < pix_end ? mz : pix_end - 1, HoWo, n, ho, wo);
                    const long long zo_first = bz_offset(n, ho, wo);
#pragma unroll
                    for (int pass = 0; pass < NPB; ++pass) {
                        rz[pass] = *(const u32x4*)(p.bz + (mz + pass * PPP < pix_end ? bz_offset(n, ho, wo) : zo_first));
                        if (pass + 1 < NPB) {
                            wo += adv_cols;
                            ho += adv_rows;
                            if (wo >= p.Wo) {
                                wo -= p.Wo;
                                ++ho;
                            }
                            if (ho >= p.Ho) {
                                ho -= p.Ho;
                                ++n;
                            }
                        }
                    }
                }
            }
            __syncthreads();  // every wave is done with the operand buffers
            {
                const int ch0 = wc * (TC / WC) + q * 4;
                const int px0 = (ODD ? wp * 16 : wp * (TP / WP)) + l15;
                auto put = [&](const f32x4 a, int ch, int px) {
                    const f32x4 v = leaky4(a, p.slope);
                    u32x2 pk;
                    pk[0] = pack_bf16x2(v[0], v[1]);
                    pk[1] = pack_bf16x2(v[2], v[3]);
                    // pixel row = 256 bytes; 16-byte chunk c of pixel px sits at chunk c ^ (px & 15)
                    *(u32x2*)(smem + px * 256 + ((((ch >> 3) ^ px) & 15) << 4) + ((ch >> 2) & 1) * 8) = pk;
                };
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    const int ch = ch0 + frag_of_slot(i) * 16;
#pragma unroll
                    for (int j = 0; j < FP; ++j) put(acc[i][j], ch, px0 + j * BSTEP);
                    if constexpr (ODD) {
                        if (i < 2) put(acc_sh[i], ch, (NF - 1) * 16 + l15);
                    }
                }
            }
            __syncthreads();
            TRACE_MARK(5);
            const int m0 = pix_base + prow;
            int n, ho, wo;
            split_pixel(p, m0 < pix_end ? m0 : pix_end - 1, HoWo, n, ho, wo);
            u32x4 sv[NPB];
#pragma unroll
            for (int pass = 0; pass < NPB; ++pass) {
                const int px = pass * PPP + prow;
                sv[pass] = *(const u32x4*)(smem + px * 256 + (((g ^ px) & 15) << 4));
            }
#pragma unroll
            for (int pass = 0; pass < NPB; ++pass) {
                bool lv = m0 + pass * PPP < pix_end && ch_first < p.Cout;
                int oh = ho, ow = wo;
                if constexpr (STRIP) {  // position column -> pixel column; the two halo columns store nothing
                    lv = lv && wo >= 1 && wo <= p.Wreal;
                    ow = wo - 1 < 0 ? 0 : (wo - 1 >= p.Wreal ? p.Wreal - 1 : wo - 1);
                }
                if constexpr (S2D) {
                    oh = 2 * ho + par_a;
                    ow = 2 * wo + par_b;
                }
                char* yp = ybase + p.y_origin + (long long)n * p.y_img_stride + (long long)oh * p.y_row_stride +
                           ow * p.y_pix_stride + (size_t)ch_first * 2;
                if (lv) *(u32x4*)yp = sv[pass];
                if constexpr (STATS) {
                    if (lv) {
                        if (p.bz)
                            bz_accumulate(sv[pass], rz[pass]);
                        else
                            sq_accumulate(sv[pass]);
                    }
                }
                if (pass + 1 < NPB) {
                    wo += adv_cols;
                    ho += adv_rows;
                    if (wo >= p.Wo) {
                        wo -= p.Wo;
                        ++ho;
                    }
                    if (ho >= p.Ho) {
                        ho -= p.Ho;
                        ++n;
                    }
                }
            }
            TRACE_MARK(6);
#ifdef YOLO_CONV_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TRACE_MARK(3);
#endif
            finish_stats();
            return;
        }
    }
    // ---- ODD, with residual (and every STRIP launch): f32 staging in two uneven halves, fragment columns [0, C0) and
    //      [C0, NF) -- each holds columns of both pixel-waves (interleaved assignment), the second also the shared
    //      column; staged row s of half h is tile pixel h * C0 * 16 + s.  Straight-line passes as in the generic path's
    //      `fast` form.  STRIP: pixels are positions of the halo'd rows; the two halo columns of a row store nothing.
    if constexpr (ODD) {
        constexpr int J0 = (FP + 1) / 2, C0 = 2 * J0;  // own columns j < J0 of both waves = tile columns [0, C0)
        auto real_col = [&](int wo) {
            if constexpr (STRIP) {
                const int c = wo - 1;
                return c < 0 ? 0 : (c >= p.Wreal ? p.Wreal - 1 : c);
            } else {
                return wo;
            }
        };
        auto run_half = [&](auto half_c, auto res_c) {
            constexpr int HALF = decltype(half_c)::value;
            constexpr bool RES = decltype(res_c)::value;
            constexpr int COLS = HALF == 0 ? C0 : NF - C0;
            constexpr int NPH = COLS * 16 / PPP;
            static_assert(COLS * 16 % PPP == 0 && COLS * 16 * TC * 4 <= 2 * BUF, "ODD epilogue half");
            const int m0 = pix_base + HALF * C0 * 16 + prow;
            int n0, ho0, wo0;
            split_pixel(p, m0 < pix_end ? m0 : pix_end - 1, HoWo, n0, ho0, wo0);
            auto step = [&](int& n, int& ho, int& wo) {
                wo += adv_cols;
                ho += adv_rows;
                if (wo >= p.Wo) {
                    wo -= p.Wo;
                    ++ho;
                }
                if (ho >= p.Ho) {
                    ho -= p.Ho;
                    ++n;
                }
            };
            static_assert(HALF != 0 || NPH == ODD_NP0, "the first half's residual is requested before the loop ends");
            [[maybe_unused]] u32x4 rz[STATS ? NPH : 1];
            if constexpr (STATS) {
                if (p.bz && ch_first < p.Cout) {
                    int n = n0, ho = ho0, wo = wo0;
                    const long long zo_first = bz_offset(n, ho, real_col(wo));
#pragma unroll
                    for (int pass = 0; pass < NPH; ++pass) {
                        rz[pass] = *(const u32x4*)(p.bz + (m0 + pass * PPP < pix_end ? bz_offset(n, ho, real_col(wo)) : zo_first));
                        if (pass + 1 < NPH) step(n, ho, wo);
                    }
                }
            }
            [[maybe_unused]] u32x4 rres[NPH];
            if constexpr (RES && HALF == 0 && RES_PREFETCH) {
#pragma unroll
                for (int pass = 0; pass < NPH; ++pass) rres[pass] = rres0[pass];
            } else if constexpr (RES) {
                if (ch_first < p.Cout) {
                    int n = n0, ho = ho0, wo = wo0;
                    const long long ro_first = p.r_origin + (long long)n * p.r_img_stride + (long long)ho * p.r_row_stride +
                                               real_col(wo) * p.r_pix_stride + (long long)ch_first * 2;
#pragma unroll
                    for (int pass = 0; pass < NPH; ++pass) {
                        const long long ro = p.r_origin + (long long)n * p.r_img_stride + (long long)ho * p.r_row_stride +
                                             real_col(wo) * p.r_pix_stride + (long long)ch_first * 2;
                        rres[pass] = *(const u32x4*)(p.res + (m0 + pass * PPP < pix_end ? ro : ro_first));
                        if (pass + 1 < NPH) step(n, ho, wo);
                    }
                }
            }
            __syncthreads();
            {
                const int ch0 = wc * (TC / WC) + q * 4;
                auto put = [&](const f32x4 a, int ch, int px) {  // px: staged row
                    *(f32x4*)(smem + px * (TC * 4) + (((ch >> 2) ^ (px & 7)) << 4)) = leaky4(a, p.slope);
                };
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    const int ch = ch0 + frag_of_slot(i) * 16;
#pragma unroll
                    for (int j = HALF == 0 ? 0 : J0; j < (HALF == 0 ? J0 : FP); ++j)
                        put(acc[i][j], ch, (2 * j + wp) * 16 + l15 - HALF * C0 * 16);
                    if (HALF == 1 && i < 2) put(acc_sh[i], ch, (NF - 1 - C0) * 16 + l15);
                }
            }
            __syncthreads();
            TRACE_MARK(5 + 2 * HALF);
            f32x4 sv0[NPH], sv1[NPH];
#pragma unroll
            for (int pass = 0; pass < NPH; ++pass) {
                const int px = pass * PPP + prow;
                const char* sp = smem + px * (TC * 4);
                sv0[pass] = *(const f32x4*)(sp + (((2 * g) ^ (px & 7)) << 4));
                sv1[pass] = *(const f32x4*)(sp + (((2 * g + 1) ^ (px & 7)) << 4));
            }
            int n = n0, ho = ho0, wo = wo0;
#pragma unroll
            for (int pass = 0; pass < NPH; ++pass) {
                const bool lv = m0 + pass * PPP < pix_end && ch_first < p.Cout && (!STRIP || (wo >= 1 && wo <= p.Wreal));
                f32x4 v0 = sv0[pass], v1 = sv1[pass];
                if constexpr (RES) {
                    const u32x4 r = rres[pass];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        v0[2 * e] += __uint_as_float(r[e] << 16);
                        v0[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
                        v1[2 * e] += __uint_as_float(r[2 + e] << 16);
                        v1[2 * e + 1] += __uint_as_float(r[2 + e] & 0xffff0000u);
                    }
                }
                u32x4 packed;
                packed[0] = pack_bf16x2(v0[0], v0[1]);
                packed[1] = pack_bf16x2(v0[2], v0[3]);
                packed[2] = pack_bf16x2(v1[0], v1[1]);
                packed[3] = pack_bf16x2(v1[2], v1[3]);
                char* yp = ybase + p.y_origin + (long long)n * p.y_img_stride + (long long)ho * p.y_row_stride +
                           real_col(wo) * p.y_pix_stride + (size_t)ch_first * 2;
                if (lv) *(u32x4*)yp = packed;
                if constexpr (STATS) {
                    if (lv) {
                        if (p.bz)
                            bz_accumulate(packed, rz[pass]);
                        else
                            sq_accumulate(packed);
                    }
                }
                if (pass + 1 < NPH) step(n, ho, wo);
            }
            TRACE_MARK(6 + 2 * HALF);
        };
        if (p.res) {
            run_half(std::integral_constant<int, 0>{}, std::true_type{});
            run_half(std::integral_constant<int, 1>{}, std::true_type{});
        } else {
            run_half(std::integral_constant<int, 0>{}, std::false_type{});
            run_half(std::integral_constant<int, 1>{}, std::false_type{});
        }
#ifdef YOLO_CONV_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRACE_MARK(3);
#endif
        finish_stats();
        return;
    }
#pragma unroll
    for (int half = 0; half < NH; ++half) {
    // Output position of each of this lane's passes: one pixel -> (image, row, column) split for the first pass,
    // then steps of PPP pixels with carries (a split per pass cost more than the stores themselves).  The walk is
    // done twice, for the residual loads and for the stores, rather than kept in registers across the staging.
    // bf16 residual: all of the lane's values for the half are requested before the tile is staged, so their
    // latency hides behind the staging.
    constexpr int NP = HP / PPP;
    const int up = p.upsample ? 2 : 1;
    const int m0 = pix_base + half * RUN + prow;
    auto pixel_of_pass = [&](int pass) { return m0 + pass * PPP + (pass * PPP >= RUN ? JUMP : 0); };
    int n0, ho0, wo0;
    split_pixel(p, m0 < pix_end ? m0 : pix_end - 1, HoWo, n0, ho0, wo0);
    // `fast`: bf16 in, no upsample, images not smaller than a step (and bf16 out, or f32 out without residual: the
    // head tensors) -- the case of nearly every launch; its
    // passes are straight-line code (only the store itself is predicated), so the scheduler overlaps them.
    const bool fast = BF && !tiny && !p.upsample;
    auto advance = [&](auto fast_c, int pass, int& n, int& ho, int& wo) {  // (n, ho, wo) of pass -> of pass + 1
        if (!decltype(fast_c)::value && tiny) {  // images smaller than a step: it may cross several of them
            const int m = pixel_of_pass(pass + 1);
            split_pixel(p, m < pix_end ? m : pix_end - 1, HoWo, n, ho, wo);
        } else {
            const bool jump = JUMP != 0 && (pass + 1) * PPP == RUN;
            wo += jump ? jmp_cols : adv_cols;
            ho += jump ? jmp_rows : adv_rows;
            if (wo >= p.Wo) {
                wo -= p.Wo;
                ++ho;
            }
            if (ho >= p.Ho) {
                ho -= p.Ho;
                ++n;
            }
        }
    };
    auto real_col = [&](int wo) {  // STRIP: position column -> pixel column, clamped (halo-column lanes store nothing)
        if constexpr (STRIP) {
            const int c = wo - 1;
            return c < 0 ? 0 : (c >= p.Wreal ? p.Wreal - 1 : c);
        } else {
            return wo;
        }
    };
    auto is_real = [&](int wo) { return !STRIP || (wo >= 1 && wo <= p.Wreal); };
    auto res_offset = [&](int n, int ho, int wo) {
        const int oh = S2D ? 2 * ho + par_a : ho, ow = S2D ? 2 * wo + par_b : real_col(wo);
        return p.r_origin + (long long)n * p.r_img_stride + (long long)oh * p.r_row_stride + ow * p.r_pix_stride +
               (long long)ch_first * (int)sizeof(T);
    };
    [[maybe_unused]] u32x4 rres[NP];
    [[maybe_unused]] u32x4 rbz[STATS ? NP : 1];   // BatchNorm-backward sums: this lane's z values of the half
    if constexpr (STATS && BF) {
        if (p.bz && ch_first < p.Cout) {   // requested with the residual, before the tile is staged
            int n = n0, ho = ho0, wo = wo0;
            const long long zo_first = bz_offset(n, ho, wo);
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                rbz[pass] = *(const u32x4*)(p.bz + (pixel_of_pass(pass) < pix_end ? bz_offset(n, ho, wo) : zo_first));
                if (pass + 1 < NP) {
                    if (fast)
                        advance(std::true_type{}, pass, n, ho, wo);
                    else
                        advance(std::false_type{}, pass, n, ho, wo);
                }
            }
        }
    }
    auto load_residual = [&](auto fast_c) {
        int n = n0, ho = ho0, wo = wo0;
        const long long ro_first = res_offset(n, ho, wo);
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            // a lane's dead passes (its last ones) re-read its first position: valid memory, value unused
            const long long ro = pixel_of_pass(pass) < pix_end ? res_offset(n, ho, wo) : ro_first;
            rres[pass] = *(const u32x4*)(p.res + ro);
            if (pass + 1 < NP) advance(fast_c, pass, n, ho, wo);
        }
    };
    if constexpr (BF) {
        if (p.res && ch_first < p.Cout) {
            if (fast)
                load_residual(std::true_type{});
            else
                load_residual(std::false_type{});
        }
    }
    __syncthreads();
    {
        const int ch0 = wc * (TC / WC) + q * 4;
        constexpr int FPH = FP / NH;  // pixel fragment columns of a wave per half
        const int px0 = wp * (HP / WP) + l15;
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            const int ch = ch0 + i * 16;
#pragma unroll
            for (int jj = 0; jj < FPH; ++jj) {
                const int px = px0 + jj * 16;
                f32x4 a;
                if constexpr (NH == 2) {  // a compile-time register choice per half, not an indexed array
                    a = half == 0 ? acc[i][jj] : acc[i][jj + FP - FPH];
                } else {
                    a = acc[i][jj];
                }
                const f32x4 v = leaky4(a, p.slope);
                *(f32x4*)(smem + px * (TC * 4) + (((ch >> 2) ^ (px & 7)) << 4)) = v;
            }
        }
    }
    __syncthreads();
    TRACE_MARK(5 + 2 * half);
    // the staged values are fetched GP passes at a time
    constexpr int GP = NP <= 6 ? NP : (NP % 5 == 0 ? 5 : (NP % 4 == 0 ? 4 : 1));
    static_assert(NP % GP == 0, "whole groups of passes");
    auto store_passes = [&](auto fast_c, auto res_c, auto f32out_c) {
        constexpr bool FAST = decltype(fast_c)::value, RES = decltype(res_c)::value, YF = decltype(f32out_c)::value;
        int n = n0, ho = ho0, wo = wo0;
#pragma unroll
        for (int g0 = 0; g0 < NP; g0 += GP) {
            f32x4 sv0[GP], sv1[GP];
#pragma unroll
            for (int i = 0; i < GP; ++i) {
                const int px = (g0 + i) * PPP + prow;
                const char* sp = smem + px * (TC * 4);
                sv0[i] = *(const f32x4*)(sp + (((2 * g) ^ (px & 7)) << 4));
                sv1[i] = *(const f32x4*)(sp + (((2 * g + 1) ^ (px & 7)) << 4));
            }
#pragma unroll
            for (int pass = g0; pass < g0 + GP; ++pass) {
                const bool lv = pixel_of_pass(pass) < pix_end && ch_first < p.Cout && is_real(wo);
                f32x4 v0 = sv0[pass - g0], v1 = sv1[pass - g0];
                const int oh = S2D ? 2 * ho + par_a : ho, ow = S2D ? 2 * wo + par_b : real_col(wo);
                if constexpr (FAST) {
                    if constexpr (RES) {
                        const u32x4 r = rres[pass];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            v0[2 * e] += __uint_as_float(r[e] << 16);
                            v0[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
                            v1[2 * e] += __uint_as_float(r[2 + e] << 16);
                            v1[2 * e + 1] += __uint_as_float(r[2 + e] & 0xffff0000u);
                        }
                    }
                    char* yp = ybase + p.y_origin + (long long)n * p.y_img_stride + (long long)oh * p.y_row_stride +
                               ow * p.y_pix_stride + (size_t)ch_first * (YF ? 4 : 2);
                    [[maybe_unused]] u32x4 packed;
                    if constexpr (YF) {  // f32 output (the head tensors)
                        if (lv) {
                            *(f32x4*)yp = v0;
                            *(f32x4*)(yp + 16) = v1;
                        }
                    } else {
                        packed[0] = pack_bf16x2(v0[0], v0[1]);
                        packed[1] = pack_bf16x2(v0[2], v0[3]);
                        packed[2] = pack_bf16x2(v1[0], v1[1]);
                        packed[3] = pack_bf16x2(v1[2], v1[3]);
                        if (lv) *(u32x4*)yp = packed;
                    }
                    if constexpr (STATS && !YF) {
                        if (lv) {
                            if (p.bz) {
                                bz_accumulate(packed, rbz[pass]);
                            } else {
                                sq_accumulate(packed);
                            }
                        }
                    }
                } else if (lv) {
                    if (p.res) {
                        if constexpr (BF) {
                            const u32x4 r = rres[pass];
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                v0[2 * e] += __uint_as_float(r[e] << 16);
                                v0[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
                                v1[2 * e] += __uint_as_float(r[2 + e] << 16);
                                v1[2 * e + 1] += __uint_as_float(r[2 + e] & 0xffff0000u);
                            }
                        } else {
                            const char* rp = p.res + res_offset(n, ho, wo);
                            v0 += *(const f32x4*)rp;
                            v1 += *(const f32x4*)(rp + 16);
                        }
                    }
                    char* yp = ybase + p.y_origin + (long long)n * p.y_img_stride + (long long)(oh * up) * p.y_row_stride +
                               (ow * up) * p.y_pix_stride + (size_t)ch_first * (p.y_f32 ? 4 : 2);
                    u32x4 packed;
                    if (!p.y_f32) {
                        packed[0] = pack_bf16x2(v0[0], v0[1]);
                        packed[1] = pack_bf16x2(v0[2], v0[3]);
                        packed[2] = pack_bf16x2(v1[0], v1[1]);
                        packed[3] = pack_bf16x2(v1[2], v1[3]);
                    }
                    for (int dy = 0; dy < up; ++dy)
                        for (int dx = 0; dx < up; ++dx) {
                            char* o = yp + dy * p.y_row_stride + dx * p.y_pix_stride;
                            if (p.y_f32) {
                                *(f32x4*)o = v0;
                                *(f32x4*)(o + 16) = v1;
                            } else {
                                *(u32x4*)o = packed;
                            }
                        }
                    if constexpr (STATS && BF) {
                        if (p.bz && !p.y_f32) bz_accumulate(packed, rbz[pass]);
                    }
                    if (!(STATS && BF) || !p.bz) {
                      if constexpr (STATS) {
                        float z[8];
                        if (p.y_f32) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                z[e] = v0[e];
                                z[4 + e] = v1[e];
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                z[2 * e] = __uint_as_float(packed[e] << 16);
                                z[2 * e + 1] = __uint_as_float(packed[e] & 0xffff0000u);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            st_s[e] += z[e];
                            st_q[e] += z[e] * z[e];
                        }
                      }
                    }
                }
                if (pass + 1 < NP) advance(fast_c, pass, n, ho, wo);
            }
        }
    };
    if (fast && !p.y_f32) {
        if (p.res)
            store_passes(std::true_type{}, std::true_type{}, std::false_type{});
        else
            store_passes(std::true_type{}, std::false_type{}, std::false_type{});
    } else if (!STATS && fast && !p.res) {
        store_passes(std::true_type{}, std::false_type{}, std::true_type{});
    } else {
        store_passes(std::false_type{}, std::false_type{}, std::false_type{});
    }
    TRACE_MARK(6 + 2 * half);
    }  // half
#ifdef YOLO_CONV_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have left the wave's queue
    TRACE_MARK(3);
#endif
    finish_stats();
}

template <typename T, int TC, int TP, int WC, int WP, bool TAP2, int NW = 4, bool STATS = false, bool S2D = false,
          bool STRIP = false, bool ODD = false>
static int launch_conv(const ConvParams& p, hipStream_t stream, int copies = 1) {
    // two operand buffers, or the f32 epilogue staging tile if that is larger (STRIP: two weight buffers + two strips
    // of ceil((TP + 2) / 8) KiB, and the staging tile whole)
    constexpr int OPS = STRIP ? 2 * TC * ROWB + 2 * ((TP + 2 + 7) / 8) * 1024 : 2 * (TC + TP) * ROWB;
    // staging: the f32 tile whole where it fits (STRIP: always), else bf16 whole / f32 in halves (ODD: the larger
    // half, (TP / 32 + 1) * 16 pixels)
    constexpr int STG = ODD ? (TP / 32 + 1) * 16 * TC * 4 : (STRIP ? TP * TC * 4 : (TP * TC * 4 > OPS ? TP * TC * 2 : TP * TC * 4));
    constexpr int LDS0 = OPS > STG ? OPS : STG;
    static_assert(TP <= 128 || LDS0 <= 80 * 1024, "the wide tiles are built for two workgroups per CU");
#ifdef YOLO_CONV_TRACE
    static const int LDS = getenv("YOLO_CONV_LONE") ? 100 * 1024 : LDS0;  // one workgroup per CU (measurement)
#else
    constexpr int LDS = LDS0;
#endif
    auto kern = conv_igemm_kernel<T, TC, TP, WC, WP, TAP2, NW, STATS, S2D, STRIP, ODD>;
    // the kernel's dynamic-LDS limit is raised once per instance and device
    static DynLdsLimit lds_limit;
    const void* const kerns[] = {(const void*)kern};
    YOLO_CHECK_HIP(lds_limit.raise(kerns, 1, LDS));
    const int nblocks = p.n_ctiles * p.n_ptiles * copies;  // copies > 1: split-K (see the kernel)
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NW * 64), LDS, stream, p);
    YOLO_CHECK_HIP(hipGetLastError());
#ifdef YOLO_CONV_TRACE
    if (getenv("YOLO_CONV_TRACE_DUMP")) {
        YOLO_CHECK_HIP(hipDeviceSynchronize());
        const int n = nblocks < TRACE_MAX ? nblocks : TRACE_MAX;
        unsigned long long* h = (unsigned long long*)malloc((size_t)n * 10 * 8);
        YOLO_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace), (size_t)n * 10 * 8));
        unsigned long long* hc = (unsigned long long*)malloc((size_t)n * 2 * 8);
        YOLO_CHECK_HIP(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_cycles), (size_t)n * 2 * 8));
        FILE* f = fopen(getenv("YOLO_CONV_TRACE_DUMP"), "w");
        if (f) {
            fprintf(f, "# tile %dx%d nsteps %d blocks %d\n", TC, TP, p.nsteps, nblocks);
            for (int i = 0; i < n; ++i)
                fprintf(f, "%d %llu %llu %llu %llu %llx %llu %llu %llu %llu %llu %llu\n", i, h[10 * i], h[10 * i + 1],
                        h[10 * i + 2], h[10 * i + 3], h[10 * i + 4], h[10 * i + 5], h[10 * i + 6], h[10 * i + 7], h[10 * i + 8],
                        h[10 * i + 9], hc[2 * i + 1] - hc[2 * i]);
            fclose(f);
        }
        free(h);
        free(hc);
    }
#endif
    return YOLO_OK;
}

template <typename T, int TC, int TP>
static int launch_conv_copies(const ConvParams& p, int copies, hipStream_t stream) {
    return launch_conv<T, TC, TP, 2, 2, false>(p, stream, copies);
}

// Pixel-tile choice (64, 96, 128 or 192).  Two workgroups share a CU (three of the 64-pixel tile, 48 KB of LDS each),
// so a launch runs in rounds of 512 (768) tiles; pick the pixel tile with the least total tile-time (rounds x pixels
// per tile x relative cost per pixel).  The 192-pixel tile moves 17 % fewer operand bytes per FLOP than 128 x 128
// (measured 5-12 % faster per pixel); the 64-pixel tile moves the weights once per 64 pixels and pays the per-tile
// set-up three times as often (measured ~1.5 x the cost per pixel), but fills the chip on layers with few pixels;
// it is only considered for 1x1 layers with K <= 512, where nothing else does.
//
// The 176-pixel tile (11 fragment columns, the last shared by the two pixel-waves: see the kernel) issues 9 % more LDS
// reads per multiply than the 192-pixel tile, but measures the same time per pixel (tools/tp_sweep.sh: the loop is not
// bound by LDS reads); it exists for the layers whose tile count it brings under a whole number of rounds (64 images of
// 13 x 13 or 26 x 26 pixels: 496 / 984 tiles instead of 456 / 904 larger ones), and wins narrowly elsewhere.  Only offered
// (`odd_ok`) to launches its epilogue handles: bf16 in and out, no upsample, no split-K.
#ifndef YOLO_TP176_COST
#define YOLO_TP176_COST 92
#endif
static int pick_tp(int M, int n_ctiles, int nsteps, bool odd_ok = false) {
#ifdef YOLO_CONV_TRACE
    static const int force = getenv("YOLO_TP") ? atoi(getenv("YOLO_TP")) : 0;  // measurement builds: pin the tile
    if (force == 64 || force == 96 || force == 128 || force == 192 || (force == 176 && odd_ok)) return force;
#endif
    auto cost = [&](int tp, long long per_pixel, long long slots) {
        const long long tiles = (long long)((M + tp - 1) / tp) * n_ctiles;
        // in sixteenths of a round: whole rounds while there are few (the resident tiles run in lock-step); from
        // three rounds on the slots drift apart and the tail costs about half a round
        const long long r16 = tiles >= 3 * slots ? (16 * tiles + slots - 1) / slots + 8 : 16 * ((tiles + slots - 1) / slots);
        return r16 * tp * per_pixel;
    };
    int best = 128;
    long long c = cost(128, 100, 512);
    if (cost(96, 100, 512) < c) {
        best = 96;
        c = cost(96, 100, 512);
    }
    if (cost(192, 92, 512) < c) {
        best = 192;
        c = cost(192, 92, 512);
    }
#ifndef YOLO_NO_TP176
    if (odd_ok && cost(176, YOLO_TP176_COST, 512) < c) {
        best = 176;
        c = cost(176, YOLO_TP176_COST, 512);
    }
#endif
    if (nsteps <= 8 && cost(64, 150, 768) < c) best = 64;
    return best;
}

// Launches the 176-pixel tile can serve, as far as the layer's shape says (the caller adds what only it knows: output
// type, image size against the epilogue's step).
static bool odd_tile_ok(int dtype, int Cin, int Cout, int k, bool upsample) {
    return dtype == YOLO_BF16 && Cin % 64 == 0 && Cout % 128 == 0 && !upsample && (k == 1 || k == 3);
}

static int conv_tc(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }

// The strip form (kw taps share one staged strip, 128 x 160 tile) serves 3x3 / stride-1 bf16 layers with a one-pixel
// halo in the input and at least 128 output channels.  It computes (and drops) the two halo columns of every row, so
// narrow images pay more for it than it saves.
#ifndef YOLO_STRIP_MIN_W
#define YOLO_STRIP_MIN_W 40
#endif
constexpr int STRIP_TP = 160;
// ... or 176 positions (11 fragment columns, the last shared by the two pixel-waves: the ODD form) where the launch
// writes bf16: 64 x 52 x 54 positions are 1022 such tiles per channel tile (x 2: four rounds of the 512 resident
// workgroups) instead of 1124 (4.4 rounds)
constexpr int STRIP_TP_ODD = 176;
// (with BatchNorm sums only the 176-position form below has the epilogue for them)
static bool use_strip(int dtype, int Cin, int Cout, int k, int stride, int x_halo, int W, bool upsample, bool stats) {
    return dtype == YOLO_BF16 && k == 3 && stride == 1 && Cin % 64 == 0 && Cout > 64 && x_halo == 1 && !upsample &&
           (!stats || Cout % 128 == 0) && W >= YOLO_STRIP_MIN_W;
}

static int conv_nsteps(int Cin, int k, int dtype) {
    if (dtype == YOLO_BF16 && Cin == 32 && k == 3) return 5;
    return k * k * (Cin / (ROWB / (int)elem_size(dtype)));
}

// channel x pixel tile of a convolution with M output pixels (`odd_ok`: the launch may use the 176-pixel tile)
static void conv_tile(int M, int Cin, int Cout, int k, int dtype, int* tc, int* tp, bool odd_ok = false) {
    *tc = conv_tc(Cout);
    *tp = *tc == 128 ? pick_tp(M, (Cout + *tc - 1) / *tc, conv_nsteps(Cin, k, dtype), odd_ok) : 128;
}

static int check_sums_bytes(const ConvParams& p, size_t stats_bytes) {
    YOLO_REQUIRE(!p.stats || stats_bytes >= (size_t)p.n_ptiles * 2 * p.Cout * sizeof(float),
                 "conv2d with BatchNorm sums: buffer of %zu bytes is too small (%d rows of 2 x %d)", stats_bytes, p.n_ptiles,
                 p.Cout);
    return YOLO_OK;
}

template <typename T, bool TAP2>
static int dispatch_tc(ConvParams& p, int tc, hipStream_t stream, bool odd_ok = false, size_t stats_bytes = 0) {
    p.n_ctiles = (p.Cout + tc - 1) / tc;
    const int tp = tc == 128 ? pick_tp(p.M, p.n_ctiles, p.nsteps, odd_ok) : 128;
    p.n_ptiles = (p.M + tp - 1) / tp;
    if (int e = check_sums_bytes(p, stats_bytes)) return e;
    if constexpr (!TAP2 && sizeof(T) == 2) {
        if (tp == 176) {
            if (p.stats) return launch_conv<T, 128, 176, 2, 2, false, 4, true, false, false, true>(p, stream);
            return launch_conv<T, 128, 176, 2, 2, false, 4, false, false, false, true>(p, stream);
        }
    }
    if constexpr (!TAP2) {
        if (p.stats) {
            switch (tc) {
                case 128:
                    if (tp == 64) return launch_conv<T, 128, 64, 2, 2, false, 4, true>(p, stream);
                    if (tp == 96) return launch_conv<T, 128, 96, 2, 2, false, 4, true>(p, stream);
                    if (tp == 192) return launch_conv<T, 128, 192, 2, 2, false, 4, true>(p, stream);
                    return launch_conv<T, 128, 128, 2, 2, false, 4, true>(p, stream);
                case 64: return launch_conv<T, 64, 128, 2, 2, false, 4, true>(p, stream);
                default: YOLO_REQUIRE(false, "conv2d with BatchNorm sums: Cout %d (needs > 32)", p.Cout);
            }
        }
    }
    switch (tc) {
        case 128:
            if (tp == 64) return launch_conv<T, 128, 64, 2, 2, TAP2>(p, stream);
            if (tp == 96) return launch_conv<T, 128, 96, 2, 2, TAP2>(p, stream);
            if (tp == 192) return launch_conv<T, 128, 192, 2, 2, TAP2>(p, stream);
            return launch_conv<T, 128, 128, 2, 2, TAP2>(p, stream);
        case 64: return launch_conv<T, 64, 128, 2, 2, TAP2>(p, stream);
        default: return launch_conv<T, 32, 128, 1, 4, TAP2>(p, stream);
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_conv_cout_pad(int Cout) { return (Cout + 127) / 128 * 128; }

extern "C" int yolo_conv_tile_shape(int M, int Cin, int Cout, int k, int dtype, int* tile_channels, int* tile_pixels) {
    YOLO_REQUIRE(M > 0 && Cin > 0 && Cout > 0 && tile_channels && tile_pixels, "conv_tile_shape: bad argument");
    conv_tile(M, Cin, Cout, k, dtype, tile_channels, tile_pixels);
    return YOLO_OK;
}


// BatchNorm-backward form of the sums (see ConvParams::bz)
struct BnBackwardSums {
    const yolo_view* z;
    const float* scale;
    const float* shift;
    float slope;
};

static int conv2d_impl(const yolo_view* x, const void* w_packed, const float* bias, const yolo_view* residual,
                       const yolo_view* y, int N, int Cin, int Cout, int k, int stride, float leaky_slope, int upsample2x,
                       int dtype, int y_dtype, float* stats, size_t stats_bytes, void* stream,
                       const BnBackwardSums* bwd = nullptr) {
    YOLO_REQUIRE(x && y && w_packed && bias, "conv2d: null argument");
    YOLO_REQUIRE(dtype == YOLO_F32 || dtype == YOLO_BF16, "conv2d: dtype %d", dtype);
    YOLO_REQUIRE(y_dtype == YOLO_F32 || y_dtype == YOLO_BF16, "conv2d: y_dtype %d", y_dtype);
    YOLO_REQUIRE(k == 1 || k == 3, "conv2d: kernel size %d (1 or 3)", k);
    YOLO_REQUIRE(stride == 1 || stride == 2, "conv2d: stride %d (1 or 2)", stride);
    YOLO_REQUIRE(N > 0 && Cin > 0 && Cout > 0, "conv2d: N/Cin/Cout must be positive");
    YOLO_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, "conv2d: leaky slope %g outside [0, 1]", (double)leaky_slope);
    const int pad = (k - 1) / 2;
    const int es = (int)elem_size(dtype), yes = (int)elem_size(y_dtype);
    const bool tap2 = dtype == YOLO_BF16 && Cin == 32 && k == 3;
    if (dtype == YOLO_BF16)
        YOLO_REQUIRE(Cin % 64 == 0 || tap2, "conv2d bf16: Cin %d must be a multiple of 64 (or 32 with k=3)", Cin);
    else
        YOLO_REQUIRE(Cin % 32 == 0, "conv2d f32: Cin %d must be a multiple of 32", Cin);
    YOLO_REQUIRE(x->halo >= pad, "conv2d: input halo %d < pad %d", x->halo, pad);
    const int Ho = (x->H + 2 * pad - k) / stride + 1, Wo = (x->W + 2 * pad - k) / stride + 1;
    const int up = upsample2x ? 2 : 1;
    YOLO_REQUIRE(y->H == Ho * up && y->W == Wo * up, "conv2d: output view %dx%d, expected %dx%d", y->H, y->W, Ho * up,
                 Wo * up);
    const int Cout8 = (Cout + 7) / 8 * 8;
    YOLO_REQUIRE(x->ch_offset + Cin <= x->ch_total && x->ch_offset % 8 == 0 && x->ch_total % 8 == 0,
                 "conv2d: bad input channel slice");
    YOLO_REQUIRE(y->ch_offset + Cout8 <= y->ch_total && y->ch_offset % 8 == 0 && y->ch_total % 8 == 0,
                 "conv2d: output channel slice [%d,+%d) does not fit %d (8-aligned)", y->ch_offset, Cout8, y->ch_total);
    YOLO_REQUIRE((long long)N * Ho * Wo < (1ll << 31), "conv2d: too many output pixels");
    if (residual) {
        YOLO_REQUIRE(!upsample2x, "conv2d: residual with upsample is not supported");
        YOLO_REQUIRE(residual->H == Ho && residual->W == Wo && residual->ch_offset % 8 == 0 &&
                         residual->ch_total % 8 == 0 && residual->ch_offset + Cout8 <= residual->ch_total,
                     "conv2d: residual view mismatch");
    }

    ConvParams p{};
    p.x = (const char*)x->ptr;
    p.w = (const char*)w_packed;
    p.bias = bias;
    p.res = residual ? (const char*)residual->ptr : nullptr;
    p.y = (char*)y->ptr;
    const ViewGeom gx = geom_of(*x), gy = geom_of(*y);
    p.x_img_stride = gx.img_stride * es;
    p.x_row_stride = gx.row_stride * es;
    p.x_pix_stride = gx.pix_stride * es;
    p.x_origin = ((long long)gx.origin - (long long)pad * gx.row_stride - (long long)pad * gx.pix_stride) * es;
    p.y_img_stride = gy.img_stride * yes;
    p.y_row_stride = gy.row_stride * yes;
    p.y_pix_stride = gy.pix_stride * yes;
    p.y_origin = (long long)gy.origin * yes;
    if (residual) {
        const ViewGeom gr = geom_of(*residual);
        p.r_img_stride = gr.img_stride * es;
        p.r_row_stride = gr.row_stride * es;
        p.r_pix_stride = gr.pix_stride * es;
        p.r_origin = (long long)gr.origin * es;
    }
    p.Ho = Ho;
    p.Wo = Wo;
    p.M = N * Ho * Wo;
    p.inv_HoWo = 1.0f / (float)(Ho * Wo);
    p.inv_Wo = 1.0f / (float)Wo;
    p.fast_div = p.M < (1 << 24);
    p.stride = stride;
    p.ksize = k;
    const int row_elems = ROWB / es;
    if (tap2) {
        p.cpt = 1;
        p.nsteps = 5;
        p.w_row_bytes = 5 * ROWB;
    } else {
        p.cpt = Cin / row_elems;
        p.nsteps = k * k * p.cpt;
        p.w_row_bytes = p.nsteps * ROWB;
    }
    p.Cout = Cout;
    p.slope = leaky_slope;
    p.upsample = upsample2x ? 1 : 0;
    p.y_f32 = y_dtype == YOLO_F32;

    const int tc = conv_tc(Cout);
    if (stats) {
        YOLO_REQUIRE(!tap2 && !upsample2x && (!residual || bwd), "conv2d with BatchNorm sums: plain convolutions only");
        if (bwd) {
            YOLO_REQUIRE(dtype == YOLO_BF16 && y_dtype == YOLO_BF16 && Cout % 8 == 0,
                         "conv2d with BatchNorm-backward sums: bf16, Cout %d a multiple of 8", Cout);
            YOLO_REQUIRE(bwd->z && bwd->z->ptr && bwd->scale && bwd->shift && bwd->z->H == Ho && bwd->z->W == Wo &&
                             bwd->z->ch_offset % 8 == 0 && bwd->z->ch_total % 8 == 0 &&
                             bwd->z->ch_offset + Cout <= bwd->z->ch_total,
                         "conv2d with BatchNorm-backward sums: z view mismatch");
            const ViewGeom gz = geom_of(*bwd->z);
            p.bz = (const char*)bwd->z->ptr;
            p.bz_img_stride = gz.img_stride * 2;
            p.bz_row_stride = gz.row_stride * 2;
            p.bz_pix_stride = gz.pix_stride * 2;
            p.bz_origin = (long long)gz.origin * 2;
            p.bz_scale = bwd->scale;
            p.bz_shift = bwd->shift;
            p.bz_slope = bwd->slope;
        }
        p.stats = stats;   // (the buffer's size is checked against the tile count once the tile is chosen)
    }
    hipStream_t st = (hipStream_t)stream;
    if (use_strip(dtype, Cin, Cout, k, stride, x->halo, x->W, upsample2x != 0, stats != nullptr)) {
        // pixels are enumerated as positions of the halo'd rows (see the kernel's comment); with a one-pixel halo
        // p.x_origin (logical pixel (-1, -1)) is the buffer's first pixel, which is position (row 0, column 0) at kh = 0
        const int Wv = Wo + 2;
        YOLO_REQUIRE((long long)N * Ho * Wv < (1ll << 31), "conv2d: too many output positions");
        p.Wreal = Wo;
        p.Wo = Wv;
        p.M = N * Ho * Wv;
        p.inv_HoWo = 1.0f / (float)(Ho * Wv);
        p.inv_Wo = 1.0f / (float)Wv;
        p.fast_div = p.M < (1 << 24);
        p.n_ctiles = (Cout + 127) / 128;
#ifndef YOLO_NO_STRIP176
        if (!p.y_f32 && Cout % 128 == 0) {  // the 176-position strip tile (bf16 output: its epilogue packs)
            p.n_ptiles = (p.M + STRIP_TP_ODD - 1) / STRIP_TP_ODD;
            if (int e = check_sums_bytes(p, stats_bytes)) return e;
            if (p.stats) return launch_conv<bf16_t, 128, STRIP_TP_ODD, 2, 2, false, 4, true, false, true, true>(p, st);
            return launch_conv<bf16_t, 128, STRIP_TP_ODD, 2, 2, false, 4, false, false, true, true>(p, st);
        }
#endif
        YOLO_REQUIRE(!p.stats, "conv2d with BatchNorm sums: the 160-position strip form has no epilogue for them");
        p.n_ptiles = (p.M + STRIP_TP - 1) / STRIP_TP;
        return launch_conv<bf16_t, 128, STRIP_TP, 2, 2, false, 4, false, false, true>(p, st);
    }
    // the 176-pixel tile's epilogue writes bf16 and walks the output in steps of 16 pixels with one row carry
    const bool odd_ok = odd_tile_ok(dtype, Cin, Cout, k, upsample2x != 0) && !p.y_f32 && 16 / Wo + 1 <= Ho;
    if (dtype == YOLO_BF16)
        return tap2 ? dispatch_tc<bf16_t, true>(p, tc, st) : dispatch_tc<bf16_t, false>(p, tc, st, odd_ok, stats_bytes);
    return dispatch_tc<float, false>(p, tc, st, false, stats_bytes);
}

namespace yolo {

struct SplitReduceParams {
    const float* slabs;          // [splits][M][Cp] f32 partial sums
    long long slab_stride;       // elements
    const float* bias;
    const char* res;
    char* y;
    long long y_img_stride, y_origin, r_img_stride, r_origin;  // bytes
    int y_row_stride, y_pix_stride, r_row_stride, r_pix_stride;
    int M, Ho, Wo, Cout, Cp, splits;
    float slope;
    int upsample, y_f32, res_f32;
};

// Finishes a split-K convolution: sum of the partial-sum slabs + bias -> LeakyReLU -> (+ residual) -> output view
// (bf16 or f32, 2x upsample, channel slice).  One thread = 8 channels of one pixel.
__global__ void __launch_bounds__(256) conv_splitk_reduce_kernel(const SplitReduceParams p) {
    const int groups = (p.Cout + 7) / 8;
    const long long total = (long long)p.M * groups;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int m = (int)(t / groups), g = (int)(t - (long long)m * groups);
        const int c0 = g * 8;
        const float* src = p.slabs + (long long)m * p.Cp + c0;
        f32x4 v0 = *(const f32x4*)(p.bias + c0), v1 = *(const f32x4*)(p.bias + c0 + 4);
        for (int s = 0; s < p.splits; ++s) {  // fixed order: the result does not depend on the launch's timing
            v0 += *(const f32x4*)(src + s * p.slab_stride);
            v1 += *(const f32x4*)(src + s * p.slab_stride + 4);
        }
        v0 = leaky4(v0, p.slope);
        v1 = leaky4(v1, p.slope);
        const int n = m / (p.Ho * p.Wo), r = m - n * (p.Ho * p.Wo), ho = r / p.Wo, wo = r - ho * p.Wo;
        if (p.res) {
            const char* rp = p.res + p.r_origin + (long long)n * p.r_img_stride + (long long)ho * p.r_row_stride +
                             wo * p.r_pix_stride + (long long)c0 * (p.res_f32 ? 4 : 2);
            if (p.res_f32) {
                v0 += *(const f32x4*)rp;
                v1 += *(const f32x4*)(rp + 16);
            } else {
                const u32x4 rr = *(const u32x4*)rp;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    v0[2 * e] += __uint_as_float(rr[e] << 16);
                    v0[2 * e + 1] += __uint_as_float(rr[e] & 0xffff0000u);
                    v1[2 * e] += __uint_as_float(rr[2 + e] << 16);
                    v1[2 * e + 1] += __uint_as_float(rr[2 + e] & 0xffff0000u);
                }
            }
        }
        const int up = p.upsample ? 2 : 1;
        char* yp = p.y + p.y_origin + (long long)n * p.y_img_stride + (long long)(ho * up) * p.y_row_stride +
                   (wo * up) * p.y_pix_stride + (long long)c0 * (p.y_f32 ? 4 : 2);
        u32x4 packed;
        if (!p.y_f32) {
            packed[0] = pack_bf16x2(v0[0], v0[1]);
            packed[1] = pack_bf16x2(v0[2], v0[3]);
            packed[2] = pack_bf16x2(v1[0], v1[1]);
            packed[3] = pack_bf16x2(v1[2], v1[3]);
        }
        for (int dy = 0; dy < up; ++dy)
            for (int dx = 0; dx < up; ++dx) {
                char* o = yp + dy * p.y_row_stride + dx * p.y_pix_stride;
                if (p.y_f32) {
                    *(f32x4*)o = v0;
                    *(f32x4*)(o + 16) = v1;
                } else {
                    *(u32x4*)o = packed;
                }
            }
    }
}

// Split-K pays when a layer has too few tiles to occupy the chip and a long k loop: batch 1 (16-58 tiles of 96-192
// pixels for 256 CUs, 18-72 k-steps each).  splits = how many workgroups share one tile's k range.  Every split layer
// costs a second launch (the slab sum, ~5 us with its dependency gap), so the rule is tuned on the whole single-image
// pass (interleaved A/B of library builds, tools/ab_latency_libs.sh): layers of fewer than 12 k-steps stay whole
// (1x1 with 512 input channels: the halved loop saves less than the extra launch costs), ONE workgroup per CU (256
// slots: a workgroup alone on its CU runs its k-steps 1.3 x faster than two sharing it), at least six k-steps per
// workgroup, at most eight slabs to sum: 1.12 -> 1.03 ms per image, batch 4 1.50 -> 1.37 ms.
#ifndef YOLO_SPLITK_MIN_STEPS
#define YOLO_SPLITK_MIN_STEPS 12
#endif
#ifndef YOLO_SPLITK_SLOTS
#define YOLO_SPLITK_SLOTS 256
#endif
#ifndef YOLO_SPLITK_MIN_PER
#define YOLO_SPLITK_MIN_PER 6
#endif
#ifndef YOLO_SPLITK_MAX
#define YOLO_SPLITK_MAX 8
#endif
static int splitk_choice(int M, int Cin, int Cout, int k, int dtype, int* tp_out) {
    if (dtype != YOLO_BF16 || Cin % 64 != 0 || Cout <= 64) return 1;
    const int nsteps = conv_nsteps(Cin, k, dtype);
    const int n_ctiles = (Cout + 127) / 128;
    const int tp = M <= 96 * 4 ? 96 : 128;  // few pixels: the smaller tile wastes less of its last tile
    const int tiles = ((M + tp - 1) / tp) * n_ctiles;
    if (tiles >= 128 || nsteps < YOLO_SPLITK_MIN_STEPS) return 1;  // (a quarter of the resident slots: below it the chip is mostly idle)
    int splits = YOLO_SPLITK_SLOTS / tiles;             // fill the chip's workgroup slots once
    if (splits > nsteps / YOLO_SPLITK_MIN_PER) splits = nsteps / YOLO_SPLITK_MIN_PER;  // k-steps per workgroup
    if (splits > YOLO_SPLITK_MAX) splits = YOLO_SPLITK_MAX;
    *tp_out = tp;
    return splits < 2 ? 1 : splits;
}

}  // namespace yolo

extern "C" int yolo_conv2d_splitk_plan(int M, int Cin, int Cout, int k, int stride, int dtype, int* splits,
                                       size_t* workspace_bytes) {
    YOLO_REQUIRE(M > 0 && Cin > 0 && Cout > 0 && splits && workspace_bytes, "conv2d_splitk_plan: bad argument");
    int tp = 0;
    (void)stride;
    *splits = splitk_choice(M, Cin, Cout, k, dtype, &tp);
    *workspace_bytes = *splits > 1 ? (size_t)*splits * (size_t)M * (size_t)((Cout + 127) / 128 * 128) * sizeof(float) : 0;
    return YOLO_OK;
}

extern "C" int yolo_conv2d_fwd_splitk(const yolo_view* x, const void* w_packed, const float* bias,
                                      const yolo_view* residual, const yolo_view* y, int N, int Cin, int Cout, int k,
                                      int stride, float leaky_slope, int upsample2x, int dtype, int y_dtype, void* workspace,
                                      size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(x && y && w_packed && bias, "conv2d_splitk: null argument");
    YOLO_REQUIRE(k == 1 || k == 3, "conv2d_splitk: kernel size %d (1 or 3)", k);
    YOLO_REQUIRE(stride == 1 || stride == 2, "conv2d_splitk: stride %d (1 or 2)", stride);
    const int pad = (k - 1) / 2;
    YOLO_REQUIRE(x->halo >= pad, "conv2d_splitk: input halo %d < pad %d", x->halo, pad);
    const int Ho = (x->H + 2 * pad - k) / stride + 1, Wo = (x->W + 2 * pad - k) / stride + 1;
    const long long M = (long long)N * Ho * Wo;
    YOLO_REQUIRE(M > 0 && M < (1ll << 24), "conv2d_splitk: %lld output pixels", M);
    int tp = 0;
    const int splits = splitk_choice((int)M, Cin, Cout, k, dtype, &tp);
    if (splits <= 1)  // nothing to gain for this shape: the plain launch
        return yolo_conv2d_fwd(x, w_packed, bias, residual, y, N, Cin, Cout, k, stride, leaky_slope, upsample2x, dtype, y_dtype,
                               stream);
    const int up = upsample2x ? 2 : 1;
    YOLO_REQUIRE(y->H == Ho * up && y->W == Wo * up, "conv2d_splitk: output view %dx%d, expected %dx%d", y->H, y->W, Ho * up,
                 Wo * up);
    YOLO_REQUIRE(y_dtype == YOLO_F32 || y_dtype == YOLO_BF16, "conv2d_splitk: y_dtype %d", y_dtype);
    YOLO_REQUIRE(leaky_slope >= 0.f && leaky_slope <= 1.f, "conv2d_splitk: leaky slope %g outside [0, 1]", (double)leaky_slope);
    const int Cout8 = (Cout + 7) / 8 * 8, Cp = (Cout + 127) / 128 * 128;
    YOLO_REQUIRE(x->ch_offset + Cin <= x->ch_total && x->ch_offset % 8 == 0 && x->ch_total % 8 == 0,
                 "conv2d_splitk: bad input channel slice");
    YOLO_REQUIRE(y->ch_offset + Cout8 <= y->ch_total && y->ch_offset % 8 == 0 && y->ch_total % 8 == 0,
                 "conv2d_splitk: bad output channel slice");
    if (residual)
        YOLO_REQUIRE(!upsample2x && residual->H == Ho && residual->W == Wo && residual->ch_offset % 8 == 0 &&
                         residual->ch_total % 8 == 0 && residual->ch_offset + Cout8 <= residual->ch_total,
                     "conv2d_splitk: residual view mismatch");
    const size_t need = (size_t)splits * (size_t)M * Cp * sizeof(float);
    YOLO_REQUIRE(workspace && ws_bytes >= need, "conv2d_splitk: workspace %zu < %zu bytes", ws_bytes, need);

    ConvParams p{};
    p.x = (const char*)x->ptr;
    p.w = (const char*)w_packed;
    p.bias = bias;
    p.y = (char*)workspace;
    const ViewGeom gx = geom_of(*x), gy = geom_of(*y);
    p.x_img_stride = gx.img_stride * 2;
    p.x_row_stride = gx.row_stride * 2;
    p.x_pix_stride = gx.pix_stride * 2;
    p.x_origin = ((long long)gx.origin - (long long)pad * gx.row_stride - (long long)pad * gx.pix_stride) * 2;
    p.y_pix_stride = Cp * 4;  // the slab: [M][Cp] f32, no halo
    p.y_row_stride = Wo * Cp * 4;
    p.y_img_stride = (long long)Ho * Wo * Cp * 4;
    p.y_origin = 0;
    p.Ho = Ho;
    p.Wo = Wo;
    p.M = (int)M;
    p.inv_HoWo = 1.0f / (float)(Ho * Wo);
    p.inv_Wo = 1.0f / (float)Wo;
    p.fast_div = 1;
    p.stride = stride;
    p.ksize = k;
    p.cpt = Cin / 64;
    p.nsteps = k * k * p.cpt;
    p.w_row_bytes = p.nsteps * ROWB;
    p.Cout = Cp;  // every channel of the slab is written (padded weight rows are zero)
    p.slope = 1.0f;
    p.y_f32 = 1;
    p.n_ctiles = Cp / 128;
    p.n_ptiles = (int)((M + tp - 1) / tp);
    p.split_steps = (p.nsteps + splits - 1) / splits;
    const int real_splits = (p.nsteps + p.split_steps - 1) / p.split_steps;
    p.y_split_stride = (long long)M * Cp * 4;
    hipStream_t st = (hipStream_t)stream;
    const int rc = tp == 96 ? launch_conv_copies<bf16_t, 128, 96>(p, real_splits, st)
                            : launch_conv_copies<bf16_t, 128, 128>(p, real_splits, st);
    if (rc != YOLO_OK) return rc;

    SplitReduceParams r{};
    r.slabs = (const float*)workspace;
    r.slab_stride = (long long)M * Cp;
    r.bias = bias;
    r.res = residual ? (const char*)residual->ptr : nullptr;
    r.y = (char*)y->ptr;
    const int yes = (int)elem_size(y_dtype);
    r.y_img_stride = gy.img_stride * yes;
    r.y_row_stride = gy.row_stride * yes;
    r.y_pix_stride = gy.pix_stride * yes;
    r.y_origin = (long long)gy.origin * yes;
    if (residual) {
        const ViewGeom gr = geom_of(*residual);
        r.r_img_stride = gr.img_stride * 2;
        r.r_row_stride = gr.row_stride * 2;
        r.r_pix_stride = gr.pix_stride * 2;
        r.r_origin = (long long)gr.origin * 2;
    }
    r.M = (int)M;
    r.Ho = Ho;
    r.Wo = Wo;
    r.Cout = Cout;
    r.Cp = Cp;
    r.splits = real_splits;
    r.slope = leaky_slope;
    r.upsample = upsample2x ? 1 : 0;
    r.y_f32 = y_dtype == YOLO_F32;
    r.res_f32 = 0;
    const long long threads = M * ((Cout + 7) / 8);
    long long blocks = (threads + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, r);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_conv_kernel_name(const yolo_view* x, const yolo_view* y, int N, int Cin, int Cout, int k, int stride,
                                     int upsample2x, int dtype, char* out, int out_len) {
    YOLO_REQUIRE(x && y && out && out_len > 0 && N > 0 && Cin > 0 && Cout > 0, "conv_kernel_name: bad argument");
    const int up = upsample2x ? 2 : 1;
    const int Ho = y->H / up, Wo = y->W / up;
    const char* dt = dtype == YOLO_BF16 ? "bf16" : "float";
    if (use_strip(dtype, Cin, Cout, k, stride, x->halo, x->W, upsample2x != 0, false)) {
#ifndef YOLO_NO_STRIP176
        const int stp = Cout % 128 == 0 ? STRIP_TP_ODD : STRIP_TP;  // (bf16 output assumed, as for the plain 176 tile)
#else
        const int stp = STRIP_TP;
#endif
        snprintf(out, (size_t)out_len, "conv_igemm_kernel<%s,128,%d,2,2,strip>", dt, stp);
        return YOLO_OK;
    }
    int tc, tp;
    // (a launch with f32 output, which this query is not told about, keeps the 192-pixel tile where this says 176)
    conv_tile(N * Ho * Wo, Cin, Cout, k, dtype, &tc, &tp,
              odd_tile_ok(dtype, Cin, Cout, k, upsample2x != 0) && 16 / Wo + 1 <= Ho);
    const bool tap2 = dtype == YOLO_BF16 && Cin == 32 && k == 3;
    snprintf(out, (size_t)out_len, "conv_igemm_kernel<%s,%d,%d,%s,%s>", dt, tc, tp, tc == 32 ? "1,4" : "2,2",
             tap2 ? "tap2" : "plain");
    return YOLO_OK;
}

extern "C" int yolo_conv2d_fwd(const yolo_view* x, const void* w_packed, const float* bias,
                               const yolo_view* residual, const yolo_view* y, int N, int Cin, int Cout, int k,
                               int stride, float leaky_slope, int upsample2x, int dtype, int y_dtype, void* stream) {
    return conv2d_impl(x, w_packed, bias, residual, y, N, Cin, Cout, k, stride, leaky_slope, upsample2x, dtype, y_dtype,
                       nullptr, 0, stream);
}

// mirrors conv2d_impl's choice of form and tile for a launch with sums (bf16 / f32 in = out, no upsample)
extern "C" int yolo_conv2d_bnsums_rows(const yolo_view* x, const yolo_view* y, int N, int Cin, int Cout, int k, int stride,
                                       int dtype) {
    if (!x || !y || N <= 0 || Cin <= 0 || Cout <= 0 || y->H <= 0 || y->W <= 0) return 0;
    const int Ho = y->H, Wo = y->W;
    if (use_strip(dtype, Cin, Cout, k, stride, x->halo, x->W, false, true))
        return (int)(((long long)N * Ho * (Wo + 2) + STRIP_TP_ODD - 1) / STRIP_TP_ODD);
    int tc, tp;
    conv_tile(N * Ho * Wo, Cin, Cout, k, dtype, &tc, &tp, odd_tile_ok(dtype, Cin, Cout, k, false) && 16 / Wo + 1 <= Ho);
    return (N * Ho * Wo + tp - 1) / tp;
}

extern "C" int yolo_conv2d_fwd_bnsums(const yolo_view* x, const void* w_packed, const float* bias, const yolo_view* y,
                                      int N, int Cin, int Cout, int k, int stride, int dtype, float* sums,
                                      size_t sums_bytes, void* stream) {
    YOLO_REQUIRE(sums, "conv2d_fwd_bnsums: null argument");
    return conv2d_impl(x, w_packed, bias, nullptr, y, N, Cin, Cout, k, stride, 1.0f, 0, dtype, dtype, sums, sums_bytes,
                       stream);
}

extern "C" int yolo_conv2d_fwd_bnbwd_sums(const yolo_view* x, const void* w_packed, const float* bias,
                                          const yolo_view* residual, const yolo_view* y, int N, int Cin, int Cout, int k,
                                          int dtype, const yolo_view* z, const float* bn_scale, const float* bn_shift,
                                          float act_slope, float* sums, size_t sums_bytes, void* stream) {
    YOLO_REQUIRE(sums && z && bn_scale && bn_shift, "conv2d_fwd_bnbwd_sums: null argument");
    YOLO_REQUIRE(act_slope >= 0.f && act_slope <= 1.f, "conv2d_fwd_bnbwd_sums: leaky slope %g outside [0, 1]",
                 (double)act_slope);
    const BnBackwardSums bwd{z, bn_scale, bn_shift, act_slope};
    return conv2d_impl(x, w_packed, bias, residual, y, N, Cin, Cout, k, 1, 1.0f, 0, dtype, dtype, sums, sums_bytes,
                       stream, &bwd);
}

extern "C" int yolo_conv3x3s2_dgrad(const yolo_view* dz, const void* w_dgrad_packed, const float* bias,
                                    const yolo_view* residual, const yolo_view* dx, int N, int Cin, int Cout, int dtype,
                                    void* stream) {
    YOLO_REQUIRE(dz && dx && w_dgrad_packed && bias, "conv3x3s2_dgrad: null argument");
    YOLO_REQUIRE(dtype == YOLO_BF16, "conv3x3s2_dgrad: bf16 only (dtype %d)", dtype);
    YOLO_REQUIRE(N > 0 && Cin > 0 && Cin % 64 == 0 && Cout > 64, "conv3x3s2_dgrad: Cin %d (multiple of 64), Cout %d (> 64)", Cin, Cout);
    YOLO_REQUIRE(dz->halo >= 1, "conv3x3s2_dgrad: the gradient view needs a zero halo");
    YOLO_REQUIRE(dx->H == 2 * dz->H && dx->W == 2 * dz->W, "conv3x3s2_dgrad: output view %dx%d, expected %dx%d", dx->H, dx->W,
                 2 * dz->H, 2 * dz->W);
    const int Cout8 = (Cout + 7) / 8 * 8;
    YOLO_REQUIRE(dz->ch_offset + Cin <= dz->ch_total && dz->ch_offset % 8 == 0 && dz->ch_total % 8 == 0,
                 "conv3x3s2_dgrad: bad gradient channel slice");
    YOLO_REQUIRE(dx->ch_offset + Cout8 <= dx->ch_total && dx->ch_offset % 8 == 0 && dx->ch_total % 8 == 0,
                 "conv3x3s2_dgrad: bad output channel slice");
    if (residual)
        YOLO_REQUIRE(residual->H == dx->H && residual->W == dx->W && residual->ch_offset % 8 == 0 &&
                         residual->ch_total % 8 == 0 && residual->ch_offset + Cout8 <= residual->ch_total,
                     "conv3x3s2_dgrad: residual view mismatch");
    const long long Mc = (long long)N * dz->H * dz->W;
    YOLO_REQUIRE(4 * Mc < (1ll << 31), "conv3x3s2_dgrad: too many output pixels");
    ConvParams p{};
    p.x = (const char*)dz->ptr;
    p.w = (const char*)w_dgrad_packed;
    p.bias = bias;
    p.res = residual ? (const char*)residual->ptr : nullptr;
    p.y = (char*)dx->ptr;
    const ViewGeom gx = geom_of(*dz), gy = geom_of(*dx);
    p.x_img_stride = gx.img_stride * 2;
    p.x_row_stride = gx.row_stride * 2;
    p.x_pix_stride = gx.pix_stride * 2;
    p.x_origin = (long long)gx.origin * 2;   // tap offsets are 0 or +1 rows / pixels: no shift by the padding
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
    p.Ho = dz->H;
    p.Wo = dz->W;
    p.Mc = (int)Mc;
    p.M = (int)(4 * Mc);
    p.inv_HoWo = 1.0f / (float)(dz->H * dz->W);
    p.inv_Wo = 1.0f / (float)dz->W;
    p.fast_div = Mc < (1 << 24);
    p.stride = 1;
    p.ksize = 3;
    p.cpt = Cin / 64;
    p.nsteps = 9 * p.cpt;
    p.w_row_bytes = p.nsteps * ROWB;
    p.Cout = Cout;
    p.slope = 1.0f;
    p.n_ctiles = (Cout + 127) / 128;
    // tiles never straddle a class; the average tile runs 9/4 taps
    const int tp = pick_tp(p.M, p.n_ctiles, 9 * p.cpt / 4);
    const int tpc = (int)((Mc + tp - 1) / tp);
    p.n_ptiles = 4 * tpc;
    hipStream_t st = (hipStream_t)stream;
    switch (tp) {
        case 64: return launch_conv<bf16_t, 128, 64, 2, 2, false, 4, false, true>(p, st);
        case 96: return launch_conv<bf16_t, 128, 96, 2, 2, false, 4, false, true>(p, st);
        case 192: return launch_conv<bf16_t, 128, 192, 2, 2, false, 4, false, true>(p, st);
        default: return launch_conv<bf16_t, 128, 128, 2, 2, false, 4, false, true>(p, st);
    }
}

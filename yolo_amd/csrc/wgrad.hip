// Weight gradient of the convolutions for gfx950, bf16 (fast path) or fp32 operands, fp32 result:
//   dW[co][kh][kw][ci] = sum over (n, ho, wo) of dz[n][ho][wo][co] * x[n][ho*s + kh - pad][wo*s + kw - pad][ci]
// This is what autograd computes for F.conv2d's weight inside detectron2's Conv2d.forward (the convs built at
// yolo/modeling/backbone/darknet.py:45-53,74-106 and fpn.py:23-30,44-61,78-92) during the reference's
// training step (tools/train_net.py -> DefaultTrainer -> losses.backward()).
//
// GEMM view per tap: D[co][ci] = sum_k A[co][k] * B[k][ci] with k = output pixel.  Both operands are stored
// pixel-major (NHWC), i.e. k is the SLOW index of both, the opposite of what an MFMA operand register wants
// (8 consecutive k per lane).  The tiles are therefore staged as they lie in memory -- [64 pixels][128
// channels] rows of 256 bytes, LDS-DMA, 16 bytes per lane -- and read with ds_read_b64_tr_b16, which hands
// each lane one column (channel) of a 4-row (pixel) block: two such reads are one 16x16x32 operand.
// 16-byte chunk c of row r sits at chunk c ^ (((r & 3) << 2) | ((r >> 2) & 3)) (source-side swizzle, same
// involution on the read): conflict-free transposed reads.
//
// Work split: one workgroup = one 128(co) x 128(ci) tile of one tap over one contiguous range of pixels
// (split-K); partial tiles go to fp32 slabs, a second kernel adds the slabs.  Rows past the end of the pixel
// range are fetched from a zero pixel of dz's halo, so they add nothing.
//
// Three ways to walk the pixels.  GATHER: k = output pixel (n, ho, wo), each tile row's address from two integer
// divisions -- about 300 vector instructions per k-step against 32 MFMAs, which is what bounds that form.
// CURSORS (the default): the same k, but the divisions happen once per workgroup; after that each lane advances
// (wo, ho, two 32-bit byte offsets) of its four tile rows by 64 pixels per k-step with one row carry and one
// image wrap at most (needs 64 / Wo + 1 <= Ho and buffers below 4 GiB).
// LINEAR (stride 1, x and dz buffers of identical geometry, zero halo in dz): k = position in the halo'd
// buffer, halo pixels included; dz is zero there, so they add nothing, x[k + tap offset] is always inside the
// buffer, and every address is base + k * pixel stride.  Images are walked in segments of G consecutive
// images (position range: first interior pixel of the first image .. last interior pixel of the last), G
// chosen to minimise the number of 64-position k-steps.
#include <mutex>

#include "common.h"

namespace yolo {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
    const char* x;
    const char* dz;
    long long x_img_stride, x_origin, dz_img_stride, dz_origin;  // bytes; x_origin already shifted by -pad
    int x_row_stride, x_pix_stride, dz_row_stride, dz_pix_stride;  // bytes
    long long dz_zero;  // byte offset of 256 readable zero bytes... of a zero pixel inside dz's buffer
    int Ho, Wo, HoWo, stride, ksize, taps;
    int M;  // N*Ho*Wo
    float inv_HoWo, inv_Wo;
    int Cin, Cout, n_mt, n_nt, splits, steps_per_split;
    float* out;  // [splits][Cout][taps][Cin]
    long long slab;  // elements per split
    // LINEAR mode: a segment = lin_len positions = lin_sps k-steps; byte offsets of a segment's first position
    // (x_first: of tap (pad, pad)) and from one segment to the next
    int lin_len, lin_sps, total_steps, pad;
    long long dz_first, x_first, dz_seg_stride, x_seg_stride;
    // cursor mode: 64 pixels = adv_rows rows + adv_cols columns; byte-offset increments (mod 2^32) of that
    // advance, of a row carry (next row, Wo columns back) and of an image wrap (next image, Ho rows back)
    int adv_rows, adv_cols;
    unsigned adv_a, adv_b, carry_a, carry_b, wrap_a, wrap_b;
};

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// q = m / d, r = m % d for 0 <= m < 2^24 through a float reciprocal and one correction step
__device__ __forceinline__ void divmod(int m, int d, float inv, int& q, int& r) {
    q = (int)((float)m * inv);
    r = m - q * d;
    if (r < 0) {
        --q;
        r += d;
    } else if (r >= d) {
        ++q;
        r -= d;
    }
}

// NARROW (Cin <= 64): the 4 waves split the 128 output channels four ways and each covers the 64 live input
// channels, instead of 2 x 2 waves of 64 x 64 -- no matrix-core work on the dead half of the ci tile.
// MODE: 0 = gather with divisions, 1 = linear walk, 2 = gather with cursors (one division per tile row per
// workgroup, then (wo, ho, byte offsets) advance by 64 pixels per k-step with at most one row carry and one image
// wrap -- which holds when 64 / Wo + 1 <= Ho and the buffers are addressable with 32 bits)
template <bool NARROW, int MODE>
__global__ void __launch_bounds__(256, 2) wgrad_kernel(const WgradParams p) {
    constexpr bool LINEAR = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 buffers x (A 16 KiB + B 16 KiB)
    constexpr int KT = 64;            // pixels per k-step: two 16x16x32 MFMA slabs per barrier
    constexpr int TILE = KT * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;

    int id = blockIdx.x;
    const int nt = id % p.n_nt;
    id /= p.n_nt;
    const int mt = id % p.n_mt;
    id /= p.n_mt;
    const int tap = id % p.taps;
    const int split = id / p.taps;
    const int kh = tap / p.ksize, kw = tap - kh * p.ksize;

    // ---- loader role: rows (4*wave + g) and +16 of the tile, 16-byte slot l15 of the row
    const int row0 = 4 * wave + g;
    const int chunk = l15 ^ swz(row0);  // swz(row0 + 16) == swz(row0)
    // channels past the end of a slice are fetched from chunk 0: finite values that only reach discarded outputs
    const int a_ch = mt * 128 + chunk * 8 < p.Cout ? mt * 128 + chunk * 8 : 0;
    const int b_ch = nt * 128 + chunk * 8 < p.Cin ? nt * 128 + chunk * 8 : 0;
    const long long x_tap = (long long)kh * p.x_row_stride + (long long)kw * p.x_pix_stride;
    const int m_begin = split * p.steps_per_split * KT;
    int m_end = m_begin + p.steps_per_split * KT;
    if (m_end > p.M) m_end = p.M;
    int nsteps = (m_end - m_begin + KT - 1) / KT;
    // LINEAR: cursor (segment, k-step inside it) of the next stage() call; stage() is called for consecutive steps
    int seg = 0, t = 0;
    unsigned a_lane = 0, b_lane = 0;   // per-lane byte offsets (origin + tile row + channel chunk): below 2^32
    long long tap_off = 0;             // wave-uniform
    if constexpr (LINEAR) {
        const int s_begin = split * p.steps_per_split;
        int s_end = s_begin + p.steps_per_split;
        if (s_end > p.total_steps) s_end = p.total_steps;
        nsteps = s_end - s_begin;
        seg = s_begin / p.lin_sps;
        t = s_begin - seg * p.lin_sps;
        a_lane = (unsigned)(p.dz_first + (long long)row0 * p.dz_pix_stride + a_ch * 2);
        b_lane = (unsigned)(p.x_first + (long long)row0 * p.x_pix_stride + b_ch * 2);
        tap_off = (long long)(kh - p.pad) * p.x_row_stride + (long long)(kw - p.pad) * p.x_pix_stride;
    }

    // MODE 2 cursors of this lane's four tile rows
    int c_wo[KT / 16], c_ho[KT / 16];
    unsigned c_a[KT / 16], c_b[KT / 16];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int j = 0; j < KT / 16; ++j) {
            int m = m_begin + row0 + 16 * j;
            m = m < p.M ? m : 0;   // rows past the end are never dereferenced (and never advance past M)
            int n, r;
            divmod(m, p.HoWo, p.inv_HoWo, n, r);
            divmod(r, p.Wo, p.inv_Wo, c_ho[j], c_wo[j]);
            c_a[j] = (unsigned)(p.dz_origin + n * p.dz_img_stride + (long long)c_ho[j] * p.dz_row_stride +
                                (long long)c_wo[j] * p.dz_pix_stride + a_ch * 2);
            c_b[j] = (unsigned)(p.x_origin + n * p.x_img_stride + (long long)(c_ho[j] * p.stride) * p.x_row_stride +
                                (long long)(c_wo[j] * p.stride) * p.x_pix_stride + x_tap + b_ch * 2);
        }
    }

    auto stage = [&](int step, char* buf) {
        if constexpr (MODE == 2) {
            const int m0 = m_begin + step * KT + row0;
#pragma unroll
            for (int j = 0; j < KT / 16; ++j) {
                const bool ok = m0 + 16 * j < m_end;
                const char* a = p.dz + (ok ? c_a[j] : (unsigned)(p.dz_zero + (a_ch & 127) * 2));
                const char* b = p.x + (ok ? c_b[j] : (unsigned)(p.x_origin + x_tap + b_ch * 2));
                char* la = buf + (16 * j + 4 * wave) * 256;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                                 (__attribute__((address_space(3))) void*)la, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b,
                                                 (__attribute__((address_space(3))) void*)(la + TILE), 16, 0, 0);
                // advance by 64 pixels = adv_rows rows + adv_cols columns
                int wo = c_wo[j] + p.adv_cols, ho = c_ho[j] + p.adv_rows;
                unsigned oa = c_a[j] + p.adv_a, ob = c_b[j] + p.adv_b;
                const bool carry = wo >= p.Wo;
                wo -= carry ? p.Wo : 0;
                ho += carry ? 1 : 0;
                oa += carry ? p.carry_a : 0u;
                ob += carry ? p.carry_b : 0u;
                const bool wrap = ho >= p.Ho;
                ho -= wrap ? p.Ho : 0;
                oa += wrap ? p.wrap_a : 0u;
                ob += wrap ? p.wrap_b : 0u;
                c_wo[j] = wo;
                c_ho[j] = ho;
                c_a[j] = oa;
                c_b[j] = ob;
            }
            return;
        }
        if constexpr (LINEAR) {
            const long long ua = (long long)seg * p.dz_seg_stride + (long long)t * (KT * p.dz_pix_stride);
            const long long ub = (long long)seg * p.x_seg_stride + (long long)t * (KT * p.x_pix_stride);
            if (t * KT + KT <= p.lin_len) {
                // all 64 positions inside the segment (every k-step but a segment's last): wave-uniform base +
                // the lane's loop-invariant 32-bit offset, no per-lane address arithmetic
#pragma unroll
                for (int j = 0; j < KT / 16; ++j) {
                    const char* a = p.dz + (ua + 16 * j * p.dz_pix_stride) + a_lane;
                    const char* b = p.x + (ub + tap_off + 16 * j * p.x_pix_stride) + b_lane;
                    char* la = buf + (16 * j + 4 * wave) * 256;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                                     (__attribute__((address_space(3))) void*)la, 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b,
                                                     (__attribute__((address_space(3))) void*)(la + TILE), 16, 0, 0);
                }
            } else {
                const int idx0 = t * KT + row0;
#pragma unroll
                for (int j = 0; j < KT / 16; ++j) {
                    const bool ok = idx0 + 16 * j < p.lin_len;
                    const char* a = ok ? p.dz + (ua + 16 * j * p.dz_pix_stride) + a_lane : p.dz + p.dz_zero + (a_ch & 127) * 2;
                    const char* b = ok ? p.x + (ub + tap_off + 16 * j * p.x_pix_stride) + b_lane : p.x + p.x_first + b_ch * 2;
                    char* la = buf + (16 * j + 4 * wave) * 256;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                                     (__attribute__((address_space(3))) void*)la, 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b,
                                                     (__attribute__((address_space(3))) void*)(la + TILE), 16, 0, 0);
                }
            }
            if (++t == p.lin_sps) {
                t = 0;
                ++seg;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < KT / 16; ++j) {
            const int m = m_begin + step * KT + row0 + 16 * j;
            const bool ok = m < m_end;
            int n, r, ho, wo;
            divmod(ok ? m : 0, p.HoWo, p.inv_HoWo, n, r);
            divmod(r, p.Wo, p.inv_Wo, ho, wo);
            const char* a = ok ? p.dz + p.dz_origin + n * p.dz_img_stride + (long long)ho * p.dz_row_stride +
                                     (long long)wo * p.dz_pix_stride + a_ch * 2
                               : p.dz + p.dz_zero + (a_ch & 127) * 2;
            const char* b = p.x + p.x_origin + n * p.x_img_stride + (long long)(ho * p.stride) * p.x_row_stride +
                            (long long)(wo * p.stride) * p.x_pix_stride + x_tap + b_ch * 2;
            char* la = buf + (16 * j + 4 * wave) * 256;  // wave-uniform base; the DMA adds lane * 16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                             (__attribute__((address_space(3))) void*)la, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b,
                                             (__attribute__((address_space(3))) void*)(la + TILE), 16, 0, 0);
        }
    };

    // ---- consumer role: wave = 64 (co) x 64 (ci); transposed-read addresses (tile-independent)
    constexpr int FM = NARROW ? 2 : 4;                 // 16-channel fragments of co per wave
    const int wm = NARROW ? wave : wave >> 1, wn = NARROW ? 0 : wave & 1;
    const int q4 = l15 >> 2, p4 = l15 & 3;
    int a_off[2][4], b_off[2][4];  // [half of the 8 k rows][fragment], first 32-pixel slab; the second is + 32 rows
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 8 * g + 4 * h + q4;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int ca = wm * 2 * FM + f * 2 + (p4 >> 1), cb = wn * 8 + f * 2 + (p4 >> 1);
            a_off[h][f] = 256 * r + 16 * (ca ^ swz(r)) + 8 * (p4 & 1);
            b_off[h][f] = TILE + 256 * r + 16 * (cb ^ swz(r)) + 8 * (p4 & 1);
        }
    }

    f32x4 acc[FM][4];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto frag = [&](const char* buf, const int off[2][4], int f) {   // swz(r + 32) == swz(r): slabs differ by a constant
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(buf + off[0][f]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(buf + off[1][f]));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    if (nsteps > 0) stage(0, smem);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nsteps) stage(s + 1, smem + ((s + 1) & 1) * 2 * TILE);
        const char* buf = smem + (s & 1) * 2 * TILE;
        bf16x8 a[2][FM], b[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int f = 0; f < FM; ++f) a[kk][f] = frag(buf + kk * 32 * 256, a_off, f);
#pragma unroll
            for (int f = 0; f < 4; ++f) b[kk][f] = frag(buf + kk * 32 * 256, b_off, f);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // x on the MFMA row index: a lane ends with 4 consecutive input channels of one output channel,
                    // which are 16 contiguous bytes of dW
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kk][j], a[kk][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }

    // ---- partial tile -> this split's slab, 16 bytes per store
    float* out = p.out + (long long)split * p.slab;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int co = mt * 128 + wm * 16 * FM + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = nt * 128 + wn * 64 + j * 16 + 4 * g;
            if (co < p.Cout && ci < p.Cin) *(f32x4*)(out + ((long long)co * p.taps + tap) * p.Cin + ci) = acc[i][j];
        }
    }
}

// fp32 operands (the 1e-3 parity plan): plain FMA tiles of 16 x 16 outputs per workgroup, 16 pixels per step
// through LDS.  Correctness path; the bf16 kernel above is the fast one.
struct WgradF32Params {
    const char* x;
    const char* dz;
    long long x_img_stride, x_origin, dz_img_stride, dz_origin;  // bytes
    int x_row_stride, x_pix_stride, dz_row_stride, dz_pix_stride;
    int Ho, Wo, HoWo, stride, ksize, taps, M, Cin, Cout, splits, pix_per_split;
    float* out;
    long long slab;
};

__global__ void __launch_bounds__(256) wgrad_f32_kernel(const WgradF32Params p) {
    __shared__ float sa[16][17], sb[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * 16;
    const int tap = blockIdx.z % p.taps, split = blockIdx.z / p.taps;
    const int kh = tap / p.ksize, kw = tap - kh * p.ksize;
    const int m0 = split * p.pix_per_split;
    const int m1 = m0 + p.pix_per_split < p.M ? m0 + p.pix_per_split : p.M;
    float acc = 0.f;
    for (int mb = m0; mb < m1; mb += 16) {
        const int m = mb + ty;  // this thread stages pixel m: channel tx of both operands
        float a = 0.f, b = 0.f;
        if (m < m1) {
            const int n = m / p.HoWo, r = m - n * p.HoWo, ho = r / p.Wo, wo = r - ho * p.Wo;
            if (co0 + tx < p.Cout)
                a = *(const float*)(p.dz + p.dz_origin + n * p.dz_img_stride + (long long)ho * p.dz_row_stride +
                                    (long long)wo * p.dz_pix_stride + (co0 + tx) * 4);
            if (ci0 + tx < p.Cin)
                b = *(const float*)(p.x + p.x_origin + n * p.x_img_stride + (long long)(ho * p.stride + kh) * p.x_row_stride +
                                    (long long)(wo * p.stride + kw) * p.x_pix_stride + (ci0 + tx) * 4);
        }
        sa[ty][tx] = a;
        sb[ty][tx] = b;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(sa[k][ty], sb[k][tx], acc);
        __syncthreads();
    }
    const int co = co0 + ty, ci = ci0 + tx;
    if (co < p.Cout && ci < p.Cin) p.out[(long long)split * p.slab + ((long long)co * p.taps + tap) * p.Cin + ci] = acc;
}

static void wgrad_f32_plan(int M, int Cin, int Cout, int k, int* splits, int* pix_per_split) {
    const long long tiles = (long long)((Cout + 15) / 16) * ((Cin + 15) / 16) * k * k;
    long long s = (2048 + tiles - 1) / tiles;
    if (s > (M + 255) / 256) s = (M + 255) / 256;
    if (s < 1) s = 1;
    *pix_per_split = (int)((M + s - 1) / s);
    *splits = (M + *pix_per_split - 1) / *pix_per_split;
}

// dw[i] (+)= sum over splits of slabs[k][i].  Block = 32 consecutive elements x 8 split lanes (each lane strides
// over the splits), folded through LDS: short dependent-load chains even for hundreds of splits.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* slabs, float* dw, long long n, int splits,
                                                           int accumulate) {
    __shared__ float part[8][33];
    const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
    for (long long base = (long long)blockIdx.x * 32; base < n; base += (long long)gridDim.x * 32) {
        const long long i = base + e;
        float s = 0.f;
        if (i < n)
            for (int k = sl; k < splits; k += 8) s += slabs[(long long)k * n + i];
        part[sl][e] = s;
        __syncthreads();
        if (sl == 0 && i < n) {
            float t = accumulate ? dw[i] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += part[k][e];
            dw[i] = t;
        }
        __syncthreads();
    }
}

// The same sum for large gradients (n % 4 == 0): one thread per 4 consecutive elements, 16-byte loads, the splits
// in order, four loads in flight.
__global__ void __launch_bounds__(256) wgrad_reduce4_kernel(const f32x4* slabs, f32x4* dw, long long n4, int splits,
                                                            int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 t = accumulate ? dw[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
        const f32x4 a = slabs[(long long)k * n4 + i], b = slabs[(long long)(k + 1) * n4 + i];
        const f32x4 c = slabs[(long long)(k + 2) * n4 + i], d = slabs[(long long)(k + 3) * n4 + i];
        t += a;
        t += b;
        t += c;
        t += d;
    }
    for (; k < splits; ++k) t += slabs[(long long)k * n4 + i];
    dw[i] = t;
}

static void launch_wgrad_reduce(const void* ws, float* dw, long long n, int splits, int accumulate, hipStream_t st) {
    // enough 16-byte threads to fill the chip (and few enough splits to walk in one thread): the wide form
    if (n % 4 == 0 && n / 4 >= 64 * 1024 && splits <= 64) {
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (const f32x4*)ws,
                           (f32x4*)dw, n / 4, splits, accumulate);
        return;
    }
    long long rb = (n + 31) / 32;
    if (rb > 8192) rb = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, (const float*)ws, dw, n, splits, accumulate);
}

static int wgrad_target_blocks(int k, int tiles) {
    // One round of the 512 resident workgroups for the memory-bound 1x1 layers, and for the 3x3 layers whose output
    // tiles fit a round at least twice (every split costs a slab of partial sums to write and to add up: 128->256 at
    // 52x52 142 -> 126 us, 256->512 at 26x26 129 -> 118 us with half the slabs); two rounds where one would leave
    // slots empty (512->1024: 288 tiles -- one split per tile fills 56 % of a round, three fill 1.7 rounds: 180 vs 140 us).
#ifdef YOLO_WGRAD_TWO_ROUNDS
    return k == 1 ? 512 : 1024;
#else
    return k == 1 || 2 * tiles <= 512 ? 512 : 1024;
#endif
}

// `steps` k-steps of 64 pixels (or positions); at most `max_splits` slabs fit the workspace.  512 workgroups are
// resident at a time (two per CU): the split count is the largest that keeps the grid within a whole number of
// such rounds (wgrad_target_blocks) -- a few workgroups more than that would run as a round of their own.
static void wgrad_plan(int steps, int Cin, int Cout, int k, long long max_splits, int* n_mt, int* n_nt, int* splits,
                       int* steps_per_split) {
    *n_mt = (Cout + 127) / 128;
    *n_nt = (Cin + 127) / 128;
    const int tiles = *n_mt * *n_nt * k * k;
    long long s = wgrad_target_blocks(k, tiles) / tiles;
    if (s > steps / 8) s = steps / 8;     // at least 8 k-steps per workgroup: a partial tile costs 64 KiB of slab
    if (s > max_splits) s = max_splits;
    if (s < 1) s = 1;
    *steps_per_split = (int)((steps + s - 1) / s);
    *splits = (steps + *steps_per_split - 1) / *steps_per_split;
}

// LINEAR walk: images per segment G (a divisor of N) with the fewest k-steps; positions and steps per segment
static void wgrad_linear_plan(int N, int H, int W, int halo, int* G, int* len, int* sps) {
    const long long Hp = H + 2 * halo, Wp = W + 2 * halo;
    long long best = -1;
    for (int g = 1; g <= N; ++g) {
        if (N % g) continue;
        const long long l = (long long)(g - 1) * Hp * Wp + (long long)(H - 1) * Wp + W;
        const long long st = (l + 63) / 64, total = st * (N / g);
        if (l < (1 << 30) && (best < 0 || total < best)) {
            best = total;
            *G = g;
            *len = (int)l;
            *sps = (int)st;
        }
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" size_t yolo_conv2d_wgrad_workspace_bytes(int M, int Cin, int Cout, int k) {
    if (M <= 0 || Cin <= 0 || Cout <= 0 || (k != 1 && k != 3)) return 0;
    // the halo'd-buffer walk takes more k-steps than there are pixels (how many depends on the image size, which
    // this query does not see): sized for twice the steps; a launch never uses more slabs than fit what it is given
    int n_mt, n_nt, splits, sps, s32, pps;
    wgrad_plan(2 * ((M + 63) / 64), Cin, Cout, k, 1 << 20, &n_mt, &n_nt, &splits, &sps);
    wgrad_f32_plan(M, Cin, Cout, k, &s32, &pps);
    if (s32 > splits) splits = s32;
    return (size_t)splits * Cout * k * k * Cin * sizeof(float);
}

static int wgrad_f32(const yolo_view* x, const yolo_view* dz, int N, int Cin, int Cout, int k, int stride, float* dw,
                     int accumulate, void* workspace, size_t ws_bytes, hipStream_t st) {
    const int pad = (k - 1) / 2;
    const int Ho = dz->H, Wo = dz->W;
    const ViewGeom gx = geom_of(*x), gd = geom_of(*dz);
    WgradF32Params p{};
    p.x = (const char*)x->ptr;
    p.dz = (const char*)dz->ptr;
    p.x_img_stride = gx.img_stride * 4;
    p.x_row_stride = gx.row_stride * 4;
    p.x_pix_stride = gx.pix_stride * 4;
    p.x_origin = ((long long)gx.origin - (long long)pad * gx.row_stride - (long long)pad * gx.pix_stride) * 4;
    p.dz_img_stride = gd.img_stride * 4;
    p.dz_row_stride = gd.row_stride * 4;
    p.dz_pix_stride = gd.pix_stride * 4;
    p.dz_origin = (long long)gd.origin * 4;
    p.Ho = Ho;
    p.Wo = Wo;
    p.HoWo = Ho * Wo;
    p.stride = stride;
    p.ksize = k;
    p.taps = k * k;
    p.M = N * Ho * Wo;
    p.Cin = Cin;
    p.Cout = Cout;
    wgrad_f32_plan(p.M, Cin, Cout, k, &p.splits, &p.pix_per_split);
    p.slab = (long long)Cout * k * k * Cin;
    YOLO_REQUIRE(ws_bytes >= (size_t)p.splits * p.slab * sizeof(float), "conv2d_wgrad: workspace too small");
    YOLO_REQUIRE(p.taps * p.splits <= 65535, "conv2d_wgrad f32: too many splits");
    p.out = (float*)workspace;
    hipLaunchKernelGGL(wgrad_f32_kernel, dim3((Cin + 15) / 16, (Cout + 15) / 16, p.taps * p.splits), dim3(256), 0, st, p);
    launch_wgrad_reduce(workspace, dw, p.slab, p.splits, accumulate, st);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_conv2d_wgrad(const yolo_view* x, const yolo_view* dz, int N, int Cin, int Cout, int k, int stride,
                                 int dtype, float* dw, int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(x && dz && x->ptr && dz->ptr && dw && workspace, "conv2d_wgrad: null argument");
    YOLO_REQUIRE(k == 1 || k == 3, "conv2d_wgrad: kernel size %d (1 or 3)", k);
    YOLO_REQUIRE(stride == 1 || stride == 2, "conv2d_wgrad: stride %d (1 or 2)", stride);
    YOLO_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0, "conv2d_wgrad: N/Cin/Cout (Cin %% 8 == 0)");
    const int pad = (k - 1) / 2;
    YOLO_REQUIRE(x->halo >= pad, "conv2d_wgrad: input halo %d < pad %d", x->halo, pad);
    const int Ho = (x->H + 2 * pad - k) / stride + 1, Wo = (x->W + 2 * pad - k) / stride + 1;
    YOLO_REQUIRE(dz->H == Ho && dz->W == Wo, "conv2d_wgrad: dz view %dx%d, expected %dx%d", dz->H, dz->W, Ho, Wo);
    YOLO_REQUIRE(x->ch_offset % 8 == 0 && x->ch_total % 8 == 0 && x->ch_offset + Cin <= x->ch_total,
                 "conv2d_wgrad: bad input channel slice");
    const int Cout8 = (Cout + 7) / 8 * 8;
    YOLO_REQUIRE(dz->ch_offset % 8 == 0 && dz->ch_total % 8 == 0 && dz->ch_offset + Cout8 <= dz->ch_total,
                 "conv2d_wgrad: bad dz channel slice");
    const long long M = (long long)N * Ho * Wo;
    YOLO_REQUIRE(dtype == YOLO_F32 || dtype == YOLO_BF16, "conv2d_wgrad: dtype %d", dtype);
    YOLO_REQUIRE(M < (1 << 24), "conv2d_wgrad: %lld output pixels (limit 2^24)", M);
    if (dtype == YOLO_F32) return wgrad_f32(x, dz, N, Cin, Cout, k, stride, dw, accumulate, workspace, ws_bytes, (hipStream_t)stream);
    YOLO_REQUIRE(dz->halo >= 1 || M % 64 == 0, "conv2d_wgrad: dz needs a zero halo when N*Ho*Wo is not a multiple of 64");
    // a tile row is 128 channels: reads past a narrower slice stay inside the buffer only with rows to spare
    YOLO_REQUIRE(dz->ch_total >= 128 || dz->halo >= 1, "conv2d_wgrad: dz narrower than 128 channels needs a halo");
    YOLO_REQUIRE(x->ch_total >= 128 || x->halo >= 1, "conv2d_wgrad: x narrower than 128 channels needs a halo");
    const ViewGeom gx = geom_of(*x), gd = geom_of(*dz);
    WgradParams p{};
    p.x = (const char*)x->ptr;
    p.dz = (const char*)dz->ptr;
    p.x_img_stride = gx.img_stride * 2;
    p.x_row_stride = gx.row_stride * 2;
    p.x_pix_stride = gx.pix_stride * 2;
    p.x_origin = ((long long)gx.origin - (long long)pad * gx.row_stride - (long long)pad * gx.pix_stride) * 2;
    p.dz_img_stride = gd.img_stride * 2;
    p.dz_row_stride = gd.row_stride * 2;
    p.dz_pix_stride = gd.pix_stride * 2;
    p.dz_origin = (long long)gd.origin * 2;
    p.dz_zero = 0;  // buffer pixel (0,0) of image 0: a halo pixel when halo >= 1 (unused when M % 32 == 0)
    p.Ho = Ho;
    p.Wo = Wo;
    p.HoWo = Ho * Wo;
    p.stride = stride;
    p.ksize = k;
    p.taps = k * k;
    p.M = (int)M;
    p.inv_HoWo = 1.0f / (float)(Ho * Wo);
    p.inv_Wo = 1.0f / (float)Wo;
    p.Cin = Cin;
    p.Cout = Cout;
    p.slab = (long long)Cout * k * k * Cin;
    const long long max_splits = (long long)(ws_bytes / ((size_t)p.slab * sizeof(float)));
    YOLO_REQUIRE(max_splits >= 1, "conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, (size_t)p.slab * sizeof(float));
    // Which walk: linear needs the same geometry on both sides and a (zero) halo in dz, and walks the halo too;
    // cursors need images at least 64 / Wo + 1 rows high and 32-bit byte offsets; divisions always work.
#ifdef YOLO_CONV_TRACE
    static const int force_mode = getenv("YOLO_WGRAD_MODE") ? atoi(getenv("YOLO_WGRAD_MODE")) : -1;  // measurement builds
#else
    constexpr int force_mode = -1;
#endif
    const bool can_linear = stride == 1 && x->halo == dz->halo && dz->halo >= 1 && x->H == dz->H && x->W == dz->W;
    const long long x_bytes = (long long)N * gx.img_stride * 2, dz_bytes = (long long)N * gd.img_stride * 2;
    const bool can_cursor = 64 / Wo + 1 <= Ho && x_bytes < (1ll << 32) && dz_bytes < (1ll << 32);
    int steps = ((int)M + 63) / 64;
    int G = 1, lin_steps = 0;
    if (can_linear) {
        wgrad_linear_plan(N, Ho, Wo, dz->halo, &G, &p.lin_len, &p.lin_sps);
        lin_steps = p.lin_sps * (N / G);
    }
    // cursors measured equal or faster than the linear walk on every layer shape (no halo positions to walk)
    int mode = can_cursor ? 2 : (can_linear ? 1 : 0);
    if (force_mode == 0 || (force_mode == 1 && can_linear) || (force_mode == 2 && can_cursor)) mode = force_mode;
    if (mode == 1) {
        p.total_steps = lin_steps;
        p.pad = pad;
        p.dz_first = (long long)gd.origin * 2;
        p.x_first = (long long)gx.origin * 2;
        p.dz_seg_stride = (long long)G * p.dz_img_stride;
        p.x_seg_stride = (long long)G * p.x_img_stride;
        steps = p.total_steps;
    } else if (mode == 2) {
        p.adv_rows = 64 / Wo;
        p.adv_cols = 64 % Wo;
        p.adv_a = (unsigned)((long long)p.adv_rows * p.dz_row_stride + (long long)p.adv_cols * p.dz_pix_stride);
        p.adv_b = (unsigned)(((long long)p.adv_rows * p.x_row_stride + (long long)p.adv_cols * p.x_pix_stride) * stride);
        p.carry_a = (unsigned)((long long)p.dz_row_stride - (long long)Wo * p.dz_pix_stride);
        p.carry_b = (unsigned)(((long long)p.x_row_stride - (long long)Wo * p.x_pix_stride) * stride);
        p.wrap_a = (unsigned)(p.dz_img_stride - (long long)Ho * p.dz_row_stride);
        p.wrap_b = (unsigned)(p.x_img_stride - (long long)Ho * stride * p.x_row_stride);
    }
    wgrad_plan(steps, Cin, Cout, k, max_splits, &p.n_mt, &p.n_nt, &p.splits, &p.steps_per_split);
    p.out = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    constexpr int LDS = 2 * 2 * 64 * 256;
    static DynLdsLimit lds_limit;  // dynamic-LDS limit of the six instances, raised once per device
    const void* const kerns[] = {(const void*)wgrad_kernel<false, 0>, (const void*)wgrad_kernel<true, 0>,
                                 (const void*)wgrad_kernel<false, 1>, (const void*)wgrad_kernel<true, 1>,
                                 (const void*)wgrad_kernel<false, 2>, (const void*)wgrad_kernel<true, 2>};
    YOLO_CHECK_HIP(lds_limit.raise(kerns, 6, LDS));
    const int blocks = p.n_mt * p.n_nt * p.taps * p.splits;
    const bool narrow = Cin <= 64;
    auto kern = mode == 1 ? (narrow ? wgrad_kernel<true, 1> : wgrad_kernel<false, 1>)
              : mode == 2 ? (narrow ? wgrad_kernel<true, 2> : wgrad_kernel<false, 2>)
                          : (narrow ? wgrad_kernel<true, 0> : wgrad_kernel<false, 0>);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS, st, p);
    launch_wgrad_reduce(workspace, dw, p.slab, p.splits, accumulate, st);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

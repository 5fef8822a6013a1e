// Training-mode BatchNorm + LeakyReLU (forward and backward) and the small data-movement kernels of the
// backward pass, for gfx950.  NHWC views as everywhere in this library; all of these are HBM-bound streaming
// kernels: 16 bytes per lane per access, fp32 arithmetic, fp32/fp64 reductions.
//
// Replaces, for modules in training mode, the BatchNorm2d that detectron2's Conv2d.forward applies after
// F.conv2d (built by get_norm("BN"); yolo/modeling/backbone/darknet.py:45-53,74-106, fpn.py:23-30,44-61,78-90),
// the F.leaky_relu that follows it (darknet.py:59,116,120,123; fpn.py:35,69,98), the `out += residual` of
// darknet.py:125, and their autograd backward; plus the backward of nn.Upsample(2, "nearest") (fpn.py:31,36)
// and the zero insertion that turns the data gradient of a stride-2 convolution into a stride-1 one.
#include <stdlib.h>

#include "common.h"

namespace yolo {

struct RedParams {
    const char* a;  // z (stats) or dy (backward)
    const char* z;  // backward only
    long long a_img_stride, a_origin, z_img_stride, z_origin;  // bytes
    int a_row_stride, a_pix_stride, z_row_stride, z_pix_stride;  // bytes
    int H, W, C, G;          // G = C/8 channel groups
    long long M;             // N*H*W pixels
    int chunk;               // pixels per block
    const float* scale;      // backward: gamma * invstd  [C]
    const float* shift;      // backward: beta - mean * scale
    const float* mean;       // backward
    const float* invstd;     // backward
    float slope;
    float* partial;          // [blocks][2][C]
};

template <typename T>
__device__ __forceinline__ void load8(const char* p, float v[8]) {
    if constexpr (sizeof(T) == 2) {
        const u32x4 r = *(const u32x4*)p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __uint_as_float(r[e] << 16);
            v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u);
        }
    } else {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = a[e];
            v[4 + e] = b[e];
        }
    }
}

template <typename T>
__device__ __forceinline__ void store8(char* p, const float v[8]) {
    if constexpr (sizeof(T) == 2) {
        const u32x4 r = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                         pack_bf16x2(v[6], v[7])};
        *(u32x4*)p = r;
    } else {
        *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(p + 16) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// Per-channel partial sums over a chunk of pixels.  MODE 0: (sum z, sum z^2).  MODE 1: with
// g = dy * leaky'(z*scale + shift): (sum g, sum g * zhat), zhat = (z - mean) * invstd.
// Thread = (pixel lane, 8-channel group); the pixel lanes of a block are folded through LDS.
template <typename T, int MODE>
__global__ void __launch_bounds__(256) bn_reduce_kernel(const RedParams p) {
    __shared__ float red[2][256][8];
    const int tid = threadIdx.x;
    const int g = tid % p.G, pl = tid / p.G, PL = 256 / p.G;
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
    float sc[8], sh_[8], mu[8], is[8];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = p.scale[g * 8 + e];
            sh_[e] = p.shift[g * 8 + e];
            mu[e] = p.mean[g * 8 + e];
            is[e] = p.invstd[g * 8 + e];
        }
    }
    const long long m0 = (long long)blockIdx.x * p.chunk;
    const long long m1 = m0 + p.chunk < p.M ? m0 + p.chunk : p.M;
    const int HW = p.H * p.W;
    // two pixels per iteration, all loads issued before the arithmetic: twice the bytes in flight per thread.  The two
    // pixel positions are cursors: one (image, row, column) split each, then carried steps of 2 * PL pixels.
    struct Cursor {
        int n, h, w;
    };
    auto split = [&](long long m) {
        Cursor c;
        c.n = (int)(m / HW);
        const int r = (int)(m - (long long)c.n * HW);
        c.h = r / p.W;
        c.w = r - c.h * p.W;
        return c;
    };
    const int step = 2 * PL;
    const int sn = step / HW, sr = step - sn * HW, sh = sr / p.W, sw = sr - sh * p.W;
    auto advance = [&](Cursor& c) {
        c.w += sw;
        if (c.w >= p.W) {
            c.w -= p.W;
            ++c.h;
        }
        c.h += sh;
        if (c.h >= p.H) {
            c.h -= p.H;
            ++c.n;
        }
        c.n += sn;
    };
    auto addr = [&](const Cursor& c, const char* base, long long origin, long long img, int row, int pix) {
        return base + origin + c.n * img + (long long)c.h * row + c.w * pix + g * 8 * (int)sizeof(T);
    };
    auto accumulate = [&](const float a[8], const float z[8]) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s0[e] += a[e];
                s1[e] += a[e] * a[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pre = z[e] * sc[e] + sh_[e];
                const float gr = pre >= 0.f ? a[e] : a[e] * p.slope;
                s0[e] += gr;
                s1[e] += gr * ((z[e] - mu[e]) * is[e]);
            }
        }
    };
    long long m = m0 + pl;
    if (m < m1) {
        Cursor c0 = split(m), c1 = split(m + PL < p.M ? m + PL : m);
        for (; m + PL < m1; m += 2 * PL) {
            float a0[8], a1[8], z0[8], z1[8];
            load8<T>(addr(c0, p.a, p.a_origin, p.a_img_stride, p.a_row_stride, p.a_pix_stride), a0);
            load8<T>(addr(c1, p.a, p.a_origin, p.a_img_stride, p.a_row_stride, p.a_pix_stride), a1);
            if constexpr (MODE == 1) {
                load8<T>(addr(c0, p.z, p.z_origin, p.z_img_stride, p.z_row_stride, p.z_pix_stride), z0);
                load8<T>(addr(c1, p.z, p.z_origin, p.z_img_stride, p.z_row_stride, p.z_pix_stride), z1);
            }
            accumulate(a0, z0);
            accumulate(a1, z1);
            advance(c0);
            advance(c1);
        }
        if (m < m1) {
            float a0[8], z0[8];
            load8<T>(addr(c0, p.a, p.a_origin, p.a_img_stride, p.a_row_stride, p.a_pix_stride), a0);
            if constexpr (MODE == 1) load8<T>(addr(c0, p.z, p.z_origin, p.z_img_stride, p.z_row_stride, p.z_pix_stride), z0);
            accumulate(a0, z0);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[0][tid][e] = s0[e];
        red[1][tid][e] = s1[e];
    }
    __syncthreads();
    // thread t < 2*C: which = t / C, channel c = t % C: fold the PL pixel lanes
    for (int t = tid; t < 2 * p.C; t += 256) {
        const int which = t / p.C, c = t - which * p.C;
        const int gg = c >> 3, e = c & 7;
        float acc = 0.f;
        for (int l = 0; l < PL; ++l) acc += red[which][l * p.G + gg][e];
        p.partial[((long long)blockIdx.x * 2 + which) * p.C + c] = acc;
    }
}

struct FinParams {
    const float* partial;
    int blocks, C;
    double count;  // N*H*W
    // stats mode
    const float* gamma;
    const float* beta;
    float eps, momentum;
    float* mean;
    float* var;
    float* invstd;
    float* scale;
    float* shift;
    float* running_mean;  // may be null
    float* running_var;
    // backward mode
    float* dgamma;
    float* dbeta;
};

// One workgroup per 8 channels: 128 thread groups stride over the partial blocks (double precision, two independent
// chains per thread), then fold through LDS, so the serial chain per thread is blocks/256 loads instead of blocks.
template <int MODE>
__global__ void __launch_bounds__(1024) bn_finalize_kernel(const FinParams p) {
    constexpr int NG = 128;
    __shared__ double sa[NG][8], sb[NG][8];
    const int cl = threadIdx.x & 7, j = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double a = 0.0, b = 0.0, a2 = 0.0, b2 = 0.0;
    if (c < p.C) {
        int i = j;
        for (; i + NG < p.blocks; i += 2 * NG) {
            const float x0 = p.partial[((long long)i * 2 + 0) * p.C + c], y0 = p.partial[((long long)i * 2 + 1) * p.C + c];
            const float x1 = p.partial[((long long)(i + NG) * 2 + 0) * p.C + c];
            const float y1 = p.partial[((long long)(i + NG) * 2 + 1) * p.C + c];
            a += (double)x0;
            b += (double)y0;
            a2 += (double)x1;
            b2 += (double)y1;
        }
        if (i < p.blocks) {
            a += (double)p.partial[((long long)i * 2 + 0) * p.C + c];
            b += (double)p.partial[((long long)i * 2 + 1) * p.C + c];
        }
        a += a2;
        b += b2;
    }
    sa[j][cl] = a;
    sb[j][cl] = b;
    __syncthreads();
    // fold 128 -> 8 groups in parallel, then the last 8 serially
    if (j < 8) {
        for (int k = j + 8; k < NG; k += 8) {
            a += sa[k][cl];
            b += sb[k][cl];
        }
        sa[j][cl] = a;
        sb[j][cl] = b;
    }
    __syncthreads();
    if (j != 0 || c >= p.C) return;
    for (int k = 1; k < 8; ++k) {
        a += sa[k][cl];
        b += sb[k][cl];
    }
    if constexpr (MODE == 0) {
        const double mean = a / p.count;
        double var = b / p.count - mean * mean;  // biased, what the normalisation uses
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float sc = p.gamma[c] * invstd;
        p.mean[c] = (float)mean;
        p.var[c] = (float)var;
        p.invstd[c] = invstd;
        p.scale[c] = sc;
        p.shift[c] = p.beta[c] - (float)mean * sc;
        if (p.running_mean) {  // torch BatchNorm2d: momentum update with the UNBIASED variance
            const double unb = p.count > 1.0 ? var * (p.count / (p.count - 1.0)) : var;
            p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * (float)mean;
            p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unb;
        }
    } else if constexpr (MODE == 1) {
        p.dbeta[c] = (float)a;
        p.dgamma[c] = (float)b;
    } else {  // the sums are (sum g, sum g * z): sum g * zhat = (sum g * z - mean * sum g) * invstd
        p.dbeta[c] = (float)a;
        p.dgamma[c] = (float)((b - (double)p.mean[c] * a) * (double)p.invstd[c]);
    }
}

struct ApplyParams {
    const char* z;
    const char* r;  // forward: residual (may be null); backward: dy
    char* y;        // forward: y; backward: dz
    long long z_img_stride, z_origin, r_img_stride, r_origin, y_img_stride, y_origin;  // bytes
    int z_row_stride, z_pix_stride, r_row_stride, r_pix_stride, y_row_stride, y_pix_stride;
    int H, W, C, G;
    long long total;  // N*H*W*G
    const float* scale;
    const float* shift;
    const float* mean;
    const float* invstd;
    const float* dgamma;
    const float* dbeta;
    float slope, inv_count;
};

// MODE 0: y = leaky(z*scale + shift) (+ r).  MODE 1: dz = scale * (g - dbeta/M - zhat * dgamma/M),
// g = dy * leaky'(z*scale + shift).
template <typename T, int MODE>
__global__ void __launch_bounds__(256) bn_apply_kernel(const ApplyParams p) {
    const int HW = p.H * p.W;
    // per-channel terms of one 8-channel group (k1 = dbeta / M, k2 = dgamma / M in the backward form)
    struct Terms {
        float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
    };
    auto load_terms = [&](int g, Terms& t) {
        const f32x4 sc0 = *(const f32x4*)(p.scale + g * 8), sc1 = *(const f32x4*)(p.scale + g * 8 + 4);
        const f32x4 sh0 = *(const f32x4*)(p.shift + g * 8), sh1 = *(const f32x4*)(p.shift + g * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t.sc[e] = sc0[e];
            t.sc[4 + e] = sc1[e];
            t.sh[e] = sh0[e];
            t.sh[4 + e] = sh1[e];
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 mu = *(const f32x4*)(p.mean + g * 8 + 4 * h), is = *(const f32x4*)(p.invstd + g * 8 + 4 * h);
                const f32x4 db = *(const f32x4*)(p.dbeta + g * 8 + 4 * h), dg = *(const f32x4*)(p.dgamma + g * 8 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    t.mu[4 * h + e] = mu[e];
                    t.is[4 * h + e] = is[e];
                    t.k1[4 * h + e] = db[e] * p.inv_count;
                    t.k2[4 * h + e] = dg[e] * p.inv_count;
                }
            }
        }
    };
    auto one = [&](const Terms& t, int g, int n, int h, int w) {
        const int cb = g * 8 * (int)sizeof(T);
        float z[8], o[8];
        load8<T>(p.z + p.z_origin + n * p.z_img_stride + (long long)h * p.z_row_stride + w * p.z_pix_stride + cb, z);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pre = z[e] * t.sc[e] + t.sh[e];
                o[e] = pre >= 0.f ? pre : pre * p.slope;
            }
            if (p.r) {
                float r[8];
                load8<T>(p.r + p.r_origin + n * p.r_img_stride + (long long)h * p.r_row_stride + w * p.r_pix_stride + cb, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += r[e];
            }
        } else {
            float dy[8];
            load8<T>(p.r + p.r_origin + n * p.r_img_stride + (long long)h * p.r_row_stride + w * p.r_pix_stride + cb, dy);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pre = z[e] * t.sc[e] + t.sh[e];
                const float gr = pre >= 0.f ? dy[e] : dy[e] * p.slope;
                const float zh = (z[e] - t.mu[e]) * t.is[e];
                o[e] = t.sc[e] * (gr - t.k1[e] - zh * t.k2[e]);
            }
        }
        store8<T>(p.y + p.y_origin + n * p.y_img_stride + (long long)h * p.y_row_stride + w * p.y_pix_stride + cb, o);
    };
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (256 % p.G == 0) {
        // The grid's stride is a whole number of pixels: a thread keeps its channel group (per-channel terms loaded
        // once) and walks the pixels in equal steps: one (image, row, column) split, then carried additions.
        if (i >= p.total) return;
        const int g = (int)(i % p.G);
        Terms t;
        load_terms(g, t);
        long long m = i / p.G;
        const long long step = stride / p.G, M = p.total / p.G;
        int n = (int)(m / HW);
        const int rr = (int)(m - (long long)n * HW);
        int h = rr / p.W, w = rr - h * p.W;
        const int sn = (int)(step / HW), sr = (int)(step - (long long)sn * HW);
        const int sh = sr / p.W, sw = sr - sh * p.W;
        for (; m < M; m += step) {
            one(t, g, n, h, w);
            w += sw;
            if (w >= p.W) {
                w -= p.W;
                ++h;
            }
            h += sh;
            if (h >= p.H) {
                h -= p.H;
                ++n;
            }
            n += sn;
        }
        return;
    }
    for (; i < p.total; i += stride) {
        const int g = (int)(i % p.G);
        const long long m = i / p.G;
        const int n = (int)(m / HW);
        const int rr = (int)(m - (long long)n * HW);
        const int h = rr / p.W, w = rr - h * p.W;
        Terms t;
        load_terms(g, t);
        one(t, g, n, h, w);
    }
}

struct MoveParams {
    const char* x;
    char* y;
    long long x_img_stride, x_origin, y_img_stride, y_origin;  // bytes
    int x_row_stride, x_pix_stride, y_row_stride, y_pix_stride;
    int H, W, G;  // size of the SMALLER side (x for dilate, y for the 2x2 sum)
    long long total;
};

// y[2i][2j] = x[i][j], the other three positions of each 2x2 cell = 0
template <typename T>
__global__ void __launch_bounds__(256) dilate2x_kernel(const MoveParams p) {
    const int HW = p.H * p.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % p.G);
        const long long m = i / p.G;
        const int n = (int)(m / HW);
        const int rr = (int)(m - (long long)n * HW);
        const int h = rr / p.W, w = rr - h * p.W;
        const int cb = g * 8 * (int)sizeof(T);
        float v[8];
        const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        load8<T>(p.x + p.x_origin + n * p.x_img_stride + (long long)h * p.x_row_stride + w * p.x_pix_stride + cb, v);
        char* o = p.y + p.y_origin + n * p.y_img_stride + (long long)(2 * h) * p.y_row_stride + (2 * w) * p.y_pix_stride + cb;
        store8<T>(o, v);
        store8<T>(o + p.y_pix_stride, zero);
        store8<T>(o + p.y_row_stride, zero);
        store8<T>(o + p.y_row_stride + p.y_pix_stride, zero);
    }
}

// y[i][j] = x[2i][2j] + x[2i][2j+1] + x[2i+1][2j] + x[2i+1][2j+1]   (backward of nearest 2x upsampling)
template <typename T>
__global__ void __launch_bounds__(256) sum2x2_kernel(const MoveParams p) {
    const int HW = p.H * p.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % p.G);
        const long long m = i / p.G;
        const int n = (int)(m / HW);
        const int rr = (int)(m - (long long)n * HW);
        const int h = rr / p.W, w = rr - h * p.W;
        const int cb = g * 8 * (int)sizeof(T);
        const char* s = p.x + p.x_origin + n * p.x_img_stride + (long long)(2 * h) * p.x_row_stride + (2 * w) * p.x_pix_stride + cb;
        float a[8], b[8], c[8], d[8], o[8];
        load8<T>(s, a);
        load8<T>(s + p.x_pix_stride, b);
        load8<T>(s + p.x_row_stride, c);
        load8<T>(s + p.x_row_stride + p.x_pix_stride, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (a[e] + b[e]) + (c[e] + d[e]);
        store8<T>(p.y + p.y_origin + n * p.y_img_stride + (long long)h * p.y_row_stride + w * p.y_pix_stride + cb, o);
    }
}

static int check_slice(const yolo_view* v, int C, const char* what) {
    YOLO_REQUIRE(v && v->ptr && v->H > 0 && v->W > 0, "%s: bad view", what);
    YOLO_REQUIRE(C > 0 && C % 8 == 0 && v->ch_offset % 8 == 0 && v->ch_total % 8 == 0 && v->ch_offset + C <= v->ch_total,
                 "%s: needs an 8-aligned slice of %d channels", what, C);
    return YOLO_OK;
}

static int red_blocks(long long M, int* chunk) {
    long long c = (M + 1023) / 1024;
    if (c < 64) c = 64;
    *chunk = (int)c;
    return (int)((M + c - 1) / c);
}

}  // namespace yolo

using namespace yolo;

extern "C" size_t yolo_bn_workspace_bytes(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    int chunk;
    const int blocks = red_blocks((long long)N * H * W, &chunk);
    return (size_t)blocks * 2 * C * sizeof(float);
}

extern "C" int yolo_bn_train_stats(const yolo_view* z, int N, int C, int dtype, const float* gamma, const float* beta,
                                   float eps, float momentum, float* running_mean, float* running_var, float* stats,
                                   void* workspace, size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(gamma && beta && stats && workspace, "bn_train_stats: null argument");
    if (int e = check_slice(z, C, "bn_train_stats")) return e;
    YOLO_REQUIRE(C <= 2048 && (C / 8 & (C / 8 - 1)) == 0 && C / 8 <= 256, "bn_train_stats: C %d (8 * power of two, <= 2048)", C);
    YOLO_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_stats: running stats come in pairs");
    YOLO_REQUIRE(ws_bytes >= yolo_bn_workspace_bytes(N, z->H, z->W, C), "bn_train_stats: workspace too small");
    const int es = (int)elem_size(dtype);
    const ViewGeom g = geom_of(*z);
    RedParams p{};
    p.a = (const char*)z->ptr;
    p.a_img_stride = g.img_stride * es;
    p.a_row_stride = g.row_stride * es;
    p.a_pix_stride = g.pix_stride * es;
    p.a_origin = (long long)g.origin * es;
    p.H = z->H;
    p.W = z->W;
    p.C = C;
    p.G = C / 8;
    p.M = (long long)N * z->H * z->W;
    const int blocks = red_blocks(p.M, &p.chunk);
    p.partial = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((bn_reduce_kernel<unsigned short, 0>), dim3(blocks), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((bn_reduce_kernel<float, 0>), dim3(blocks), dim3(256), 0, st, p);
    FinParams f{};
    f.partial = p.partial;
    f.blocks = blocks;
    f.C = C;
    f.count = (double)p.M;
    f.gamma = gamma;
    f.beta = beta;
    f.eps = eps;
    f.momentum = momentum;
    f.mean = stats;
    f.var = stats + C;
    f.invstd = stats + 2 * C;
    f.scale = stats + 3 * C;
    f.shift = stats + 4 * C;
    f.running_mean = running_mean;
    f.running_var = running_var;
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3((C + 7) / 8), dim3(1024), 0, st, f);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_bn_train_finalize(const float* sums, int rows, long long count, int C, const float* gamma,
                                      const float* beta, float eps, float momentum, float* running_mean,
                                      float* running_var, float* stats, void* stream) {
    YOLO_REQUIRE(sums && gamma && beta && stats && rows > 0 && count > 0 && C > 0, "bn_train_finalize: bad argument");
    YOLO_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_finalize: running stats come in pairs");
    FinParams f{};
    f.partial = sums;
    f.blocks = rows;
    f.C = C;
    f.count = (double)count;
    f.gamma = gamma;
    f.beta = beta;
    f.eps = eps;
    f.momentum = momentum;
    f.mean = stats;
    f.var = stats + C;
    f.invstd = stats + 2 * C;
    f.scale = stats + 3 * C;
    f.shift = stats + 4 * C;
    f.running_mean = running_mean;
    f.running_var = running_var;
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)stream, f);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

static void fill_apply(ApplyParams& p, const yolo_view* z, const yolo_view* r, const yolo_view* y, int N, int C, int es) {
    const ViewGeom gz = geom_of(*z), gy = geom_of(*y);
    p.z = (const char*)z->ptr;
    p.z_img_stride = gz.img_stride * es;
    p.z_row_stride = gz.row_stride * es;
    p.z_pix_stride = gz.pix_stride * es;
    p.z_origin = (long long)gz.origin * es;
    p.y = (char*)y->ptr;
    p.y_img_stride = gy.img_stride * es;
    p.y_row_stride = gy.row_stride * es;
    p.y_pix_stride = gy.pix_stride * es;
    p.y_origin = (long long)gy.origin * es;
    if (r) {
        const ViewGeom gr = geom_of(*r);
        p.r = (const char*)r->ptr;
        p.r_img_stride = gr.img_stride * es;
        p.r_row_stride = gr.row_stride * es;
        p.r_pix_stride = gr.pix_stride * es;
        p.r_origin = (long long)gr.origin * es;
    }
    p.H = z->H;
    p.W = z->W;
    p.C = C;
    p.G = C / 8;
    p.total = (long long)N * z->H * z->W * p.G;
}

static unsigned apply_grid(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b < 8192 ? b : 8192);
}

extern "C" int yolo_bn_act_fwd(const yolo_view* z, const float* stats, const yolo_view* residual, const yolo_view* y,
                               int N, int C, float leaky_slope, int dtype, void* stream) {
    YOLO_REQUIRE(stats, "bn_act_fwd: null argument");
    if (int e = check_slice(z, C, "bn_act_fwd")) return e;
    if (int e = check_slice(y, C, "bn_act_fwd")) return e;
    YOLO_REQUIRE(y->H == z->H && y->W == z->W, "bn_act_fwd: z and y differ in size");
    if (residual) {
        if (int e = check_slice(residual, C, "bn_act_fwd")) return e;
        YOLO_REQUIRE(residual->H == z->H && residual->W == z->W, "bn_act_fwd: residual size");
    }
    ApplyParams p{};
    fill_apply(p, z, residual, y, N, C, (int)elem_size(dtype));
    p.scale = stats + 3 * C;
    p.shift = stats + 4 * C;
    p.slope = leaky_slope;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((bn_apply_kernel<unsigned short, 0>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((bn_apply_kernel<float, 0>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_bn_act_bwd(const yolo_view* dy, const yolo_view* z, const float* stats, const yolo_view* dz, int N,
                               int C, float leaky_slope, int dtype, float* dgamma, float* dbeta, void* workspace,
                               size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(stats && dgamma && dbeta && workspace, "bn_act_bwd: null argument");
    if (int e = check_slice(dy, C, "bn_act_bwd")) return e;
    if (int e = check_slice(z, C, "bn_act_bwd")) return e;
    if (int e = check_slice(dz, C, "bn_act_bwd")) return e;
    YOLO_REQUIRE(dy->H == z->H && dy->W == z->W && dz->H == z->H && dz->W == z->W, "bn_act_bwd: sizes differ");
    YOLO_REQUIRE(C <= 2048 && (C / 8 & (C / 8 - 1)) == 0 && C / 8 <= 256, "bn_act_bwd: C %d (8 * power of two, <= 2048)", C);
    YOLO_REQUIRE(ws_bytes >= yolo_bn_workspace_bytes(N, z->H, z->W, C), "bn_act_bwd: workspace too small");
    const int es = (int)elem_size(dtype);
    const ViewGeom ga = geom_of(*dy), gz = geom_of(*z);
    RedParams r{};
    r.a = (const char*)dy->ptr;
    r.a_img_stride = ga.img_stride * es;
    r.a_row_stride = ga.row_stride * es;
    r.a_pix_stride = ga.pix_stride * es;
    r.a_origin = (long long)ga.origin * es;
    r.z = (const char*)z->ptr;
    r.z_img_stride = gz.img_stride * es;
    r.z_row_stride = gz.row_stride * es;
    r.z_pix_stride = gz.pix_stride * es;
    r.z_origin = (long long)gz.origin * es;
    r.H = z->H;
    r.W = z->W;
    r.C = C;
    r.G = C / 8;
    r.M = (long long)N * z->H * z->W;
    const int blocks = red_blocks(r.M, &r.chunk);
    r.mean = stats;
    r.invstd = stats + 2 * C;
    r.scale = stats + 3 * C;
    r.shift = stats + 4 * C;
    r.slope = leaky_slope;
    r.partial = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((bn_reduce_kernel<unsigned short, 1>), dim3(blocks), dim3(256), 0, st, r);
    else
        hipLaunchKernelGGL((bn_reduce_kernel<float, 1>), dim3(blocks), dim3(256), 0, st, r);
    FinParams f{};
    f.partial = r.partial;
    f.blocks = blocks;
    f.C = C;
    f.count = (double)r.M;
    f.dgamma = dgamma;
    f.dbeta = dbeta;
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3((C + 7) / 8), dim3(1024), 0, st, f);
    ApplyParams p{};
    fill_apply(p, z, dy, dz, N, C, es);
    p.mean = stats;
    p.invstd = stats + 2 * C;
    p.scale = stats + 3 * C;
    p.shift = stats + 4 * C;
    p.dgamma = dgamma;
    p.dbeta = dbeta;
    p.slope = leaky_slope;
    p.inv_count = (float)(1.0 / (double)r.M);
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((bn_apply_kernel<unsigned short, 1>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((bn_apply_kernel<float, 1>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

// The same backward step when the reduction came out of the epilogue that wrote dy (yolo_conv2d_fwd_bnbwd_sums):
// `sums` holds `rows` rows of [2][C] partial sums (sum g, sum g * z); two launches instead of three, dy and z read once.
extern "C" int yolo_bn_act_bwd_presummed(const yolo_view* dy, const yolo_view* z, const float* stats, const yolo_view* dz,
                                         int N, int C, float leaky_slope, int dtype, float* dgamma, float* dbeta,
                                         const float* sums, int rows, void* stream) {
    YOLO_REQUIRE(stats && dgamma && dbeta && sums && rows > 0, "bn_act_bwd_presummed: bad argument");
    if (int e = check_slice(dy, C, "bn_act_bwd_presummed")) return e;
    if (int e = check_slice(z, C, "bn_act_bwd_presummed")) return e;
    if (int e = check_slice(dz, C, "bn_act_bwd_presummed")) return e;
    YOLO_REQUIRE(dy->H == z->H && dy->W == z->W && dz->H == z->H && dz->W == z->W, "bn_act_bwd_presummed: sizes differ");
    const int es = (int)elem_size(dtype);
    const long long M = (long long)N * z->H * z->W;
    hipStream_t st = (hipStream_t)stream;
    FinParams f{};
    f.partial = sums;
    f.blocks = rows;
    f.C = C;
    f.count = (double)M;
    f.mean = const_cast<float*>(stats);
    f.invstd = const_cast<float*>(stats) + 2 * C;
    f.dgamma = dgamma;
    f.dbeta = dbeta;
    hipLaunchKernelGGL((bn_finalize_kernel<2>), dim3((C + 7) / 8), dim3(1024), 0, st, f);
    ApplyParams p{};
    fill_apply(p, z, dy, dz, N, C, es);
    p.mean = stats;
    p.invstd = stats + 2 * C;
    p.scale = stats + 3 * C;
    p.shift = stats + 4 * C;
    p.dgamma = dgamma;
    p.dbeta = dbeta;
    p.slope = leaky_slope;
    p.inv_count = (float)(1.0 / (double)M);
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((bn_apply_kernel<unsigned short, 1>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((bn_apply_kernel<float, 1>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

static int move_common(MoveParams& p, const yolo_view* x, const yolo_view* y, int N, int C, int es, bool x_small) {
    const ViewGeom gx = geom_of(*x), gy = geom_of(*y);
    p.x = (const char*)x->ptr;
    p.x_img_stride = gx.img_stride * es;
    p.x_row_stride = gx.row_stride * es;
    p.x_pix_stride = gx.pix_stride * es;
    p.x_origin = (long long)gx.origin * es;
    p.y = (char*)y->ptr;
    p.y_img_stride = gy.img_stride * es;
    p.y_row_stride = gy.row_stride * es;
    p.y_pix_stride = gy.pix_stride * es;
    p.y_origin = (long long)gy.origin * es;
    p.H = x_small ? x->H : y->H;
    p.W = x_small ? x->W : y->W;
    p.G = C / 8;
    p.total = (long long)N * p.H * p.W * p.G;
    return YOLO_OK;
}

extern "C" int yolo_dilate2x_nhwc(const yolo_view* x, const yolo_view* y, int N, int C, int dtype, void* stream) {
    if (int e = check_slice(x, C, "dilate2x")) return e;
    if (int e = check_slice(y, C, "dilate2x")) return e;
    YOLO_REQUIRE(y->H == 2 * x->H && y->W == 2 * x->W, "dilate2x: output must be 2x the input");
    MoveParams p{};
    move_common(p, x, y, N, C, (int)elem_size(dtype), true);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((dilate2x_kernel<unsigned short>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((dilate2x_kernel<float>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_sum2x2_nhwc(const yolo_view* x, const yolo_view* y, int N, int C, int dtype, void* stream) {
    if (int e = check_slice(x, C, "sum2x2")) return e;
    if (int e = check_slice(y, C, "sum2x2")) return e;
    YOLO_REQUIRE(x->H == 2 * y->H && x->W == 2 * y->W, "sum2x2: input must be 2x the output");
    MoveParams p{};
    move_common(p, x, y, N, C, (int)elem_size(dtype), false);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((sum2x2_kernel<unsigned short>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((sum2x2_kernel<float>), dim3(apply_grid(p.total)), dim3(256), 0, st, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

namespace yolo {

// Per-step repacking of a trainable convolution's fp32 master weights [Cout][Cin][k][k] into the two operand
// layouts the kernels read: forward rows [co][kh][kw][ci] and data-gradient rows [ci][kh][kw][co] with the taps
// rotated by 180 degrees (co padded to CoutP with zeros by the caller's one-time memset).
template <typename T>
__global__ void __launch_bounds__(256) pack_train_kernel(const float* w, T* fwd, T* dgrad, int Cout, int Cin, int k,
                                                         int fwd_row, int dg_row, int CoutP) {
    const int kk = k * k;
    const long long n = (long long)Cout * Cin * kk;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        // i enumerates the forward layout (co, t, ci): coalesced writes there
        const int ci = (int)(i % Cin);
        const long long r = i / Cin;
        const int t = (int)(r % kk), co = (int)(r / kk);
        const float v = w[((long long)co * Cin + ci) * kk + t];
        T o;
        if constexpr (sizeof(T) == 2)
            o = f32_to_bf16_bits(v);
        else
            o = v;
        if (fwd) fwd[(long long)co * fwd_row + t * Cin + ci] = o;
        if (dgrad) dgrad[(long long)ci * dg_row + (kk - 1 - t) * CoutP + co] = o;
    }
}

}  // namespace yolo

extern "C" int yolo_pack_conv_train(const float* w, int Cout, int Cin, int k, int dtype, void* w_fwd, int fwd_row,
                                    void* w_dgrad, int dgrad_row, int Cout_pad, void* stream) {
    YOLO_REQUIRE(w && (w_fwd || w_dgrad), "pack_conv_train: null argument");
    YOLO_REQUIRE(Cout > 0 && Cin > 0 && (k == 1 || k == 3), "pack_conv_train: bad shape");
    YOLO_REQUIRE(dtype == YOLO_F32 || dtype == YOLO_BF16, "pack_conv_train: dtype %d", dtype);
    YOLO_REQUIRE((!w_fwd || fwd_row >= k * k * Cin) && (!w_dgrad || (Cout_pad >= Cout && dgrad_row >= k * k * Cout_pad)),
                 "pack_conv_train: row lengths too short");
    const long long n = (long long)Cout * Cin * k * k;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((yolo::pack_train_kernel<unsigned short>), dim3((unsigned)blocks), dim3(256), 0, st, w,
                           (unsigned short*)w_fwd, (unsigned short*)w_dgrad, Cout, Cin, k, fwd_row, dgrad_row, Cout_pad);
    else
        hipLaunchKernelGGL((yolo::pack_train_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, w, (float*)w_fwd,
                           (float*)w_dgrad, Cout, Cin, k, fwd_row, dgrad_row, Cout_pad);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

namespace yolo {

// All trainable convolutions of a model in ONE launch.  Master weights live pixel-major, [Cout][kh][kw][Cin] fp32
// (the storage order of a torch channels_last [Cout,Cin,kh,kw] tensor), so the forward operand is a cast copy;
// the data-gradient operand [ci][rotated tap][co] goes through a 64 x 64 LDS transpose.  One workgroup = one
// (64 co, tap, 64 ci) tile; the layer of a workgroup is found by bisection over the table's first-tile column.
template <typename T>
__global__ void __launch_bounds__(256) pack_train_batch_kernel(const yolo_pack_desc* descs, int n_layers) {
    __shared__ T tile[64][66];
    int lo = 0, hi = n_layers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= descs[mid].first_tile) lo = mid; else hi = mid - 1;
    }
    const yolo_pack_desc d = descs[lo];
    const int kk = d.k * d.k;
    int id = blockIdx.x - d.first_tile;
    const int n_ci = (d.Cin + 63) / 64;
    const int ct = id % n_ci;
    id /= n_ci;
    const int t = id % kk;
    const int cot = id / kk;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float* w = (const float*)d.w;
    T* fwd = (T*)d.w_fwd;
    T* dg = (T*)d.w_dgrad;
    const int ci = ct * 64 + tx;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int co = cot * 64 + r;
        T o = 0;
        if (co < d.Cout && ci < d.Cin) {
            const float v = w[((long long)co * kk + t) * d.Cin + ci];
            if constexpr (sizeof(T) == 2) o = f32_to_bf16_bits(v); else o = v;
            if (fwd) fwd[(long long)co * d.fwd_row + t * d.Cin + ci] = o;
        }
        tile[r][tx] = o;
    }
    if (!dg) return;
    __syncthreads();
    const int co = cot * 64 + tx;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int c = ct * 64 + r;
        if (c < d.Cin && co < d.Cout) dg[(long long)c * d.dgrad_row + (kk - 1 - t) * d.Cout_pad + co] = tile[tx][r];
    }
}

}  // namespace yolo

// ------------------------------------------------------------------------------------------------
// SGD with momentum over the flat parameter / gradient / momentum arenas: one launch instead of the three
// multi-tensor passes of torch.optim.SGD (weight decay, momentum, update).  The block table says which run of up to
// 4096 elements a workgroup owns and which of the two hyper-parameter groups (weights / normalisation) it is in.
// ------------------------------------------------------------------------------------------------
namespace yolo {
struct SgdParams {
    float* param;
    const float* grad;
    float* mom;
    const int* blocks;  // [n][3]: first element, count (<= 4096), group
    float lr[2], wd[2];
    float momentum;
};

__global__ void __launch_bounds__(256) sgd_step_kernel(const SgdParams p) {
    const int first = p.blocks[3 * blockIdx.x], count = p.blocks[3 * blockIdx.x + 1], g = p.blocks[3 * blockIdx.x + 2];
    const float lr = p.lr[g], wd = p.wd[g], mu = p.momentum;
    for (int i = threadIdx.x; i < count; i += 256) {
        const long long e = (long long)first + i;
        const float w = p.param[e];
        float d = p.grad[e];
        if (wd != 0.f) d = d + wd * w;
        const float m = p.mom[e] * mu + d;
        p.mom[e] = m;
        p.param[e] = w - lr * m;
    }
}
}  // namespace yolo

extern "C" int yolo_sgd_step(float* param, const float* grad, float* momentum_buf, const int* block_table, int n_blocks,
                             float lr0, float weight_decay0, float lr1, float weight_decay1, float momentum,
                             void* stream) {
    YOLO_REQUIRE(param && grad && momentum_buf && block_table && n_blocks > 0, "sgd_step: bad argument");
    SgdParams p{};
    p.param = param;
    p.grad = grad;
    p.mom = momentum_buf;
    p.blocks = block_table;
    p.lr[0] = lr0;
    p.lr[1] = lr1;
    p.wd[0] = weight_decay0;
    p.wd[1] = weight_decay1;
    p.momentum = momentum;
    hipLaunchKernelGGL(sgd_step_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_pack_conv_train_batch(const yolo_pack_desc* descs_device, int n_layers, int total_tiles, int dtype,
                                          void* stream) {
    YOLO_REQUIRE(descs_device && n_layers > 0 && total_tiles > 0, "pack_conv_train_batch: empty table");
    YOLO_REQUIRE(dtype == YOLO_F32 || dtype == YOLO_BF16, "pack_conv_train_batch: dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL((yolo::pack_train_batch_kernel<unsigned short>), dim3((unsigned)total_tiles), dim3(256), 0, st,
                           descs_device, n_layers);
    else
        hipLaunchKernelGGL((yolo::pack_train_batch_kernel<float>), dim3((unsigned)total_tiles), dim3(256), 0, st,
                           descs_device, n_layers);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

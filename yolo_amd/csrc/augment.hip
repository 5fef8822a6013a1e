// Training-image augmentation on the device: uint8 HWC (3 channels) in, uint8 HWC out.
//
//   yolo_aug_letterbox  OpenCV-style 8-bit linear resize into a bordered destination
//                       (LetterBoxTransform.apply_image, yolo/data/transforms.py:207-212)
//   yolo_aug_resample   Pillow-style antialiased bilinear resize, two integer passes
//                       (detectron2 ResizeShortestEdge, yolo/data/detection_utils.py:96-104)
//   yolo_aug_apply      one pass per image: affine / perspective warp with the constant 114 border
//                       (AffineTransform.apply_image, transforms.py:43-50) -> HSV colour jitter
//                       (ColorJitterTransform.apply_image, transforms.py:177-186) -> flip -> paste of a crop
//                       into a canvas (YOLOV3DatasetMapper._blend_mosaic, yolo/data/dataset_mapper.py:136-152)
//
// All arithmetic that decides an output byte is integer or correctly-rounded IEEE (no contraction: the file is
// built with -ffp-contract=off), so results equal a CPU evaluation of the same formulas bit for bit.
// The work is bound by memory: ~3 bytes written and 3-12 (cached) bytes read per pixel.
#include "common.h"

namespace yolo {

static constexpr int AUG_THREADS = 256;

// ------------------------------------------------------------------------------------------------
// letterbox: resize (11-bit coefficients per axis, the vertical blend on values shifted down by 4) + border
// ------------------------------------------------------------------------------------------------
struct LetterboxParams {
    const uint8_t* src;
    uint8_t* dst;
    long long s_stride, d_stride;
    int sh, sw, dh, dw, top, left, rh, rw, border;
    double scale_x, scale_y;
};

__device__ __forceinline__ void linear_axis(int d, double scale, int n_src, int& s, int& c0, int& c1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) {
        f = 0.f;
        s = 0;
    }
    if (s >= n_src - 1) {
        f = 0.f;
        s = n_src - 1;
    }
    c0 = (int)rintf((1.f - f) * 2048.f);
    c1 = (int)rintf(f * 2048.f);
}

__global__ void __launch_bounds__(AUG_THREADS) aug_letterbox_kernel(const LetterboxParams p) {
    const long long idx = (long long)blockIdx.x * AUG_THREADS + threadIdx.x;
    if (idx >= (long long)p.dh * p.dw) return;
    const int y = (int)(idx / p.dw), x = (int)(idx - (long long)y * p.dw);
    uint8_t* o = p.dst + y * p.d_stride + 3 * x;
    const int ry = y - p.top, rx = x - p.left;
    if (ry < 0 || ry >= p.rh || rx < 0 || rx >= p.rw) {
        o[0] = o[1] = o[2] = (uint8_t)p.border;
        return;
    }
    if (p.rh == p.sh && p.rw == p.sw) {
        const uint8_t* s = p.src + ry * p.s_stride + 3 * rx;
        o[0] = s[0];
        o[1] = s[1];
        o[2] = s[2];
        return;
    }
    int sx, cx0, cx1, sy, cy0, cy1;
    linear_axis(rx, p.scale_x, p.sw, sx, cx0, cx1);
    linear_axis(ry, p.scale_y, p.sh, sy, cy0, cy1);
    const int sx1 = min(sx + 1, p.sw - 1), sy1 = min(sy + 1, p.sh - 1);
    const uint8_t* r0 = p.src + sy * p.s_stride;
    const uint8_t* r1 = p.src + sy1 * p.s_stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int a0 = r0[3 * sx + c] * cx0 + r0[3 * sx1 + c] * cx1;
        const int a1 = r1[3 * sx + c] * cx0 + r1[3 * sx1 + c] * cx1;
        int v = (((cy0 * (a0 >> 4)) >> 16) + ((cy1 * (a1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        o[c] = (uint8_t)v;
    }
}

// ------------------------------------------------------------------------------------------------
// resample: out[o] = clip8((2^21 + sum_k in[first(o) + k] * coef[o][k]) >> 22) along one axis; up to 16 images per launch
// ------------------------------------------------------------------------------------------------
struct PassJob {
    const uint8_t* src;
    uint8_t* dst;
    long long s_stride, d_stride;
    int out_h, out_w;   // of this pass
    const int* bounds;  // [n_out][2]: first source index, count
    const int* coef;    // [n_out][ksize]
    int ksize;
    int mode;    // 0 horizontal, 1 vertical, 2 copy
    int in_len;  // source columns (horizontal) or rows (vertical)
};
static constexpr int AUG_MAX_PASS_JOBS = 16;
struct PassJobs {
    PassJob j[AUG_MAX_PASS_JOBS];
};

__device__ __forceinline__ uint8_t clip8(int a) {
    a >>= 22;
    return (uint8_t)(a < 0 ? 0 : (a > 255 ? 255 : a));
}

// KTAPS > 0: every job of the launch has exactly KTAPS coefficients per output (3 when enlarging, 5 / 7 when
// shrinking by up to 2x / 3x): all loads of a thread are issued together.  Coefficients beyond an output's own
// count are zero in the tables, so reading KTAPS in-range source elements is the same sum.  KTAPS == 0: any size.
template <int KTAPS>
__global__ void __launch_bounds__(AUG_THREADS) aug_resample_kernel(const PassJobs jobs) {
    const PassJob& p = jobs.j[blockIdx.y];
    const unsigned idx = blockIdx.x * AUG_THREADS + threadIdx.x;
    if (p.mode == 0) {
        // one output pixel per thread
        if (idx >= (unsigned)p.out_h * (unsigned)p.out_w) return;
        const unsigned y = idx / (unsigned)p.out_w, x = idx - y * (unsigned)p.out_w;
        const int first = p.bounds[2 * x], n = p.bounds[2 * x + 1];
        const int* k = p.coef + (long long)x * p.ksize;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
        const uint8_t* s = p.src + y * p.s_stride + 3 * first;
        bool done = false;
        if constexpr (KTAPS > 0) {
            if (first + KTAPS <= p.in_len) {
                uint8_t px[3 * KTAPS];
                int w[KTAPS];
                __builtin_memcpy(px, s, 3 * KTAPS);
#pragma unroll
                for (int i = 0; i < KTAPS; ++i) w[i] = k[i];
#pragma unroll
                for (int i = 0; i < KTAPS; ++i) {
                    a0 += px[3 * i] * w[i];
                    a1 += px[3 * i + 1] * w[i];
                    a2 += px[3 * i + 2] * w[i];
                }
                done = true;
            }
        }
        if (!done) {
            for (int i = 0; i < n; ++i) {
                const int w = k[i];
                a0 += s[0] * w;
                a1 += s[1] * w;
                a2 += s[2] * w;
                s += 3;
            }
        }
        uint8_t* d = p.dst + y * p.d_stride + 3 * x;
        d[0] = clip8(a0);
        d[1] = clip8(a1);
        d[2] = clip8(a2);
    } else {
        // eight consecutive bytes of an output row per thread: the taps are whole rows, read 8 bytes at a time
        const unsigned row_bytes = 3u * (unsigned)p.out_w, octs = (row_bytes + 7u) >> 3;
        if (idx >= (unsigned)p.out_h * octs) return;
        const unsigned y = idx / octs, b = (idx - y * octs) * 8u;
        const int nb = row_bytes - b < 8u ? (int)(row_bytes - b) : 8;
        int first = (int)y, n = 1;
        const int* k = nullptr;
        if (p.mode == 1) {
            first = p.bounds[2 * y];
            n = p.bounds[2 * y + 1];
            k = p.coef + (long long)y * p.ksize;
        }
        int a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = 1 << 21;
        const uint8_t* s = p.src + first * p.s_stride + b;
        bool done = false;
        if constexpr (KTAPS > 0) {
            if (p.mode == 1 && nb == 8 && first + KTAPS <= p.in_len) {
                uint64_t v[KTAPS];
                int w[KTAPS];
#pragma unroll
                for (int i = 0; i < KTAPS; ++i) {
                    __builtin_memcpy(&v[i], s + i * p.s_stride, 8);
                    w[i] = k[i];
                }
#pragma unroll
                for (int i = 0; i < KTAPS; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] += (int)((v[i] >> (8 * e)) & 255u) * w[i];
                done = true;
            }
        }
        if (!done) {
            for (int i = 0; i < n; ++i) {
                const int w = k ? k[i] : (1 << 22);
                uint64_t v = 0;
                if (nb == 8) {
                    __builtin_memcpy(&v, s, 8);
                } else {
                    for (int e = 0; e < nb; ++e) v |= (uint64_t)s[e] << (8 * e);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += (int)((v >> (8 * e)) & 255u) * w;
                s += p.s_stride;
            }
        }
        uint8_t* d = p.dst + y * p.d_stride + b;
        if (p.mode == 2) {  // copy: the accumulators hold 2^21 + v * 2^22
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] -= 1 << 21;
        }
        uint64_t o = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) o |= (uint64_t)clip8(a[e]) << (8 * e);
        if (nb == 8) {
            __builtin_memcpy(d, &o, 8);
        } else {
            for (int e = 0; e < nb; ++e) d[e] = (uint8_t)(o >> (8 * e));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// apply: warp -> colour jitter -> flip -> paste
// ------------------------------------------------------------------------------------------------
static constexpr int AUG_MAX_JOBS = 16;
static constexpr int AUG_PIX_PER_THREAD = 4;
struct AugJobs {
    yolo_aug_job j[AUG_MAX_JOBS];
};

// cv2's saturate_cast<int>(double): round to nearest even, clamp to the int range
__device__ __forceinline__ long long sat_int(double v) {
    v = rint(v);
    v = fmin(fmax(v, -2147483648.0), 2147483647.0);
    return (long long)(int)v;
}

// 12-bit reciprocal tables of OpenCV's 8-bit RGB -> HSV: round((255 << 12) / i) and round((180 << 12) / (6 i)),
// rounded to nearest even -- exact integer arithmetic at compile time.
struct DivTables {
    int s[256], h[256];
};
constexpr int rounded_quotient(int n, int d) {
    const int q = n / d, r = n % d;
    return 2 * r > d ? q + 1 : (2 * r == d ? q + (q & 1) : q);
}
constexpr DivTables make_div_tables() {
    DivTables t{};
    for (int i = 1; i < 256; ++i) {
        t.s[i] = rounded_quotient(255 << 12, i);
        t.h[i] = rounded_quotient(30 << 12, i);
    }
    return t;
}
__constant__ DivTables c_div = make_div_tables();

__global__ void __launch_bounds__(AUG_THREADS) aug_apply_kernel(const AugJobs jobs) {
    const yolo_aug_job& J = jobs.j[blockIdx.y];
    __shared__ uint8_t lut[3][256];
    __shared__ int sdiv[256], hdiv[256];
    const unsigned npix = (unsigned)J.crop_w * (unsigned)J.crop_h;
    const unsigned base = blockIdx.x * (unsigned)(AUG_THREADS * AUG_PIX_PER_THREAD);
    if (base >= npix) return;  // whole block
    if (J.jitter) {
        const int t = threadIdx.x;
        const double g[3] = {J.hue_gain, J.saturation_gain, J.exposure_gain};
        // lut_hue = (x * gain % 180) with the sign of the divisor, the others clip(x * gain, 0, 255); both truncated
        double h = fmod((double)t * g[0], 180.0);
        if (h != 0.0 && h < 0.0) h += 180.0;
        lut[0][t] = (uint8_t)(int)h;
#pragma unroll
        for (int c = 1; c < 3; ++c) {
            const double v = fmin(fmax((double)t * g[c], 0.0), 255.0);
            lut[c][t] = (uint8_t)(int)v;
        }
        sdiv[t] = c_div.s[t];
        hdiv[t] = c_div.h[t];
        __syncthreads();
    }
    const float inv_w = 1.0f / (float)J.crop_w;
    const uint8_t* S = (const uint8_t*)J.src;
    uint8_t* D = (uint8_t*)J.dst;
    const int W = J.src_width, H = J.src_height;
    for (int it = 0; it < AUG_PIX_PER_THREAD; ++it) {
        const unsigned idx = base + (unsigned)it * AUG_THREADS + threadIdx.x;
        if (idx >= npix) break;
        // idx / crop_w through a float reciprocal, corrected by one either way (idx < 2^31, any crop_w)
        int py = (int)((float)idx * inv_w), px = (int)idx - py * J.crop_w;
        if (px < 0) {
            --py;
            px += J.crop_w;
        } else if (px >= J.crop_w) {
            ++py;
            px -= J.crop_w;
        }
        int u = J.crop_x + px, v = J.crop_y + py;
        if (J.flip_horizontal) u = W - 1 - u;
        if (J.flip_vertical) v = H - 1 - v;
        int c0, c1, c2;
        if (J.warp == 0) {
            const uint8_t* s = S + v * J.src_stride + 3 * u;
            c0 = s[0];
            c1 = s[1];
            c2 = s[2];
        } else {
            long long X, Y;
            const double* m = J.inverse_map;
            if (J.warp == 1) {
                // 10-bit column and row terms rounded separately, half a table step added, 5 fractional bits kept
                const long long X0 = sat_int((m[1] * (double)v + m[2]) * 1024.0) + 16;
                const long long Y0 = sat_int((m[4] * (double)v + m[5]) * 1024.0) + 16;
                X = (X0 + sat_int(m[0] * (double)u * 1024.0)) >> 5;
                Y = (Y0 + sat_int(m[3] * (double)u * 1024.0)) >> 5;
            } else {
                // homogeneous coordinates evaluated from the first column of the pixel's 64-wide block
                const double bx = (double)((u >> 6) << 6), x1 = (double)(u & 63), y = (double)v;
                const double X0 = m[0] * bx + m[1] * y + m[2];
                const double Y0 = m[3] * bx + m[4] * y + m[5];
                const double W0 = m[6] * bx + m[7] * y + m[8];
                double Wd = W0 + m[6] * x1;
                Wd = Wd != 0.0 ? 32.0 / Wd : 0.0;
                X = sat_int(fmax(-2147483648.0, fmin(2147483647.0, (X0 + m[0] * x1) * Wd)));
                Y = sat_int(fmax(-2147483648.0, fmin(2147483647.0, (Y0 + m[3] * x1) * Wd)));
            }
            long long sxl = X >> 5, syl = Y >> 5;
            sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
            syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
            const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32,
                      w11 = fx * fy * 32;
            int acc[3];
            if (sx >= 0 && sy >= 0 && sx + 3 <= W && sy + 1 < H) {
                // whole 2 x 2 patch inside and 8 readable bytes in both rows: two (unaligned) 8-byte loads
                const uint8_t* r0 = S + (long long)sy * J.src_stride + 3 * sx;
                uint64_t q0, q1;
                __builtin_memcpy(&q0, r0, 8);
                __builtin_memcpy(&q1, r0 + J.src_stride, 8);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int t00 = (int)((q0 >> (8 * c)) & 255u), t01 = (int)((q0 >> (8 * (c + 3))) & 255u);
                    const int t10 = (int)((q1 >> (8 * c)) & 255u), t11 = (int)((q1 >> (8 * (c + 3))) & 255u);
                    acc[c] = (t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11 + (1 << 14)) >> 15;
                }
            } else {
                const bool x0in = sx >= 0 && sx < W, x1in = sx + 1 >= 0 && sx + 1 < W;
                const bool y0in = sy >= 0 && sy < H, y1in = sy + 1 >= 0 && sy + 1 < H;
                const uint8_t* r0 = S + (long long)sy * J.src_stride + 3 * sx;
                const uint8_t* r1 = r0 + J.src_stride;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int t00 = (x0in && y0in) ? r0[c] : J.border;
                    const int t01 = (x1in && y0in) ? r0[3 + c] : J.border;
                    const int t10 = (x0in && y1in) ? r1[c] : J.border;
                    const int t11 = (x1in && y1in) ? r1[3 + c] : J.border;
                    acc[c] = (t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11 + (1 << 14)) >> 15;
                }
            }
            c0 = acc[0];
            c1 = acc[1];
            c2 = acc[2];
        }
        if (J.jitter) {
            const int r = J.rgb_order ? c0 : c2, g = c1, b = J.rgb_order ? c2 : c0;
            // 8-bit RGB -> HSV (H in 0..179), integer
            const int vmax = max(max(b, g), r), vmin = min(min(b, g), r), diff = vmax - vmin;
            const int sat = (diff * sdiv[vmax] + (1 << 11)) >> 12;
            int hh = vmax == r ? g - b : (vmax == g ? b - r + 2 * diff : r - g + 4 * diff);
            hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
            hh += hh < 0 ? 180 : 0;
            hh = hh > 255 ? 255 : hh;
            const int h2 = lut[0][hh], s2 = lut[1][sat], v2 = lut[2][vmax];
            // 8-bit HSV -> RGB, float
            const float sf = (float)s2 * (1.0f / 255.0f), vf = (float)v2 * (1.0f / 255.0f);
            float bf, gf, rf;
            if (s2 == 0) {
                bf = gf = rf = vf;
            } else {
                float hf = (float)h2 * (6.0f / 180.0f);
                if (hf >= 6.0f) hf -= 6.0f;
                int sector = (int)floorf(hf);
                float fr = hf - (float)sector;
                if ((unsigned)sector >= 6u) {
                    sector = 0;
                    fr = 0.f;
                }
                const float t0 = vf, t1 = vf * (1.f - sf), t2 = vf * (1.f - sf * fr), t3 = vf * (1.f - sf * (1.f - fr));
                // sector -> (b, g, r) picks
                switch (sector) {
                    case 0: bf = t1; gf = t3; rf = t0; break;
                    case 1: bf = t1; gf = t0; rf = t2; break;
                    case 2: bf = t3; gf = t0; rf = t1; break;
                    case 3: bf = t0; gf = t2; rf = t1; break;
                    case 4: bf = t0; gf = t1; rf = t3; break;
                    default: bf = t2; gf = t1; rf = t0; break;
                }
            }
            const float bo = fminf(fmaxf(rintf(bf * 255.0f), 0.f), 255.f), go = fminf(fmaxf(rintf(gf * 255.0f), 0.f), 255.f),
                        ro = fminf(fmaxf(rintf(rf * 255.0f), 0.f), 255.f);
            c0 = (int)(J.rgb_order ? ro : bo);
            c1 = (int)go;
            c2 = (int)(J.rgb_order ? bo : ro);
        }
        if (J.reverse_channels) {
            const int t = c0;
            c0 = c2;
            c2 = t;
        }
        if (J.dst_plane_stride) {
            uint8_t* o = D + (long long)(J.dst_y + py) * J.dst_stride + (J.dst_x + px);
            o[0] = (uint8_t)c0;
            o[J.dst_plane_stride] = (uint8_t)c1;
            o[2 * J.dst_plane_stride] = (uint8_t)c2;
        } else {
            uint8_t* o = D + (long long)(J.dst_y + py) * J.dst_stride + 3 * (J.dst_x + px);
            o[0] = (uint8_t)c0;
            o[1] = (uint8_t)c1;
            o[2] = (uint8_t)c2;
        }
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" int yolo_aug_letterbox(const void* src, int src_height, int src_width, long long src_stride, void* dst,
                                  int dst_height, int dst_width, long long dst_stride, int top, int left,
                                  int resized_height, int resized_width, int border_value, void* stream) {
    YOLO_REQUIRE(src && dst, "aug_letterbox: null argument");
    YOLO_REQUIRE(src_height > 0 && src_width > 0 && dst_height > 0 && dst_width > 0 && resized_height > 0 &&
                     resized_width > 0,
                 "aug_letterbox: sizes must be positive");
    YOLO_REQUIRE(top >= 0 && left >= 0 && top + resized_height <= dst_height && left + resized_width <= dst_width,
                 "aug_letterbox: resized %dx%d at (%d,%d) does not fit %dx%d", resized_height, resized_width, top, left,
                 dst_height, dst_width);
    YOLO_REQUIRE(src_stride >= 3ll * src_width && dst_stride >= 3ll * dst_width, "aug_letterbox: row stride below 3 * width");
    YOLO_REQUIRE(border_value >= 0 && border_value <= 255, "aug_letterbox: border value %d", border_value);
    LetterboxParams p{};
    p.src = (const uint8_t*)src;
    p.dst = (uint8_t*)dst;
    p.s_stride = src_stride;
    p.d_stride = dst_stride;
    p.sh = src_height;
    p.sw = src_width;
    p.dh = dst_height;
    p.dw = dst_width;
    p.top = top;
    p.left = left;
    p.rh = resized_height;
    p.rw = resized_width;
    p.border = border_value;
    p.scale_x = 1.0 / ((double)resized_width / (double)src_width);
    p.scale_y = 1.0 / ((double)resized_height / (double)src_height);
    const long long n = (long long)dst_height * dst_width;
    hipLaunchKernelGGL(aug_letterbox_kernel, dim3((unsigned)((n + AUG_THREADS - 1) / AUG_THREADS)), dim3(AUG_THREADS), 0,
                       (hipStream_t)stream, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" size_t yolo_aug_resample_workspace_bytes(int src_height, int dst_width) {
    return src_height > 0 && dst_width > 0 ? (size_t)src_height * dst_width * 3 : 0;
}

static int launch_passes(const PassJob* jobs, int n, hipStream_t st) {
    for (int first = 0; first < n; first += AUG_MAX_PASS_JOBS) {
        const int m = n - first < AUG_MAX_PASS_JOBS ? n - first : AUG_MAX_PASS_JOBS;
        PassJobs batch{};
        long long most = 0;
        int taps = -1;  // the common coefficient count of the launch's jobs, 0 if they differ
        for (int i = 0; i < m; ++i) {
            batch.j[i] = jobs[first + i];
            const PassJob& p = batch.j[i];
            const long long work = p.mode == 0 ? (long long)p.out_h * p.out_w : (long long)p.out_h * ((3 * p.out_w + 7) >> 3);
            most = work > most ? work : most;
            const int k = p.mode == 2 ? 0 : p.ksize;
            taps = taps < 0 ? k : (taps == k ? taps : 0);
        }
        const dim3 grid((unsigned)((most + AUG_THREADS - 1) / AUG_THREADS), m), block(AUG_THREADS);
        if (taps == 3)
            hipLaunchKernelGGL(aug_resample_kernel<3>, grid, block, 0, st, batch);
        else if (taps == 5)
            hipLaunchKernelGGL(aug_resample_kernel<5>, grid, block, 0, st, batch);
        else if (taps == 7)
            hipLaunchKernelGGL(aug_resample_kernel<7>, grid, block, 0, st, batch);
        else
            hipLaunchKernelGGL(aug_resample_kernel<0>, grid, block, 0, st, batch);
        YOLO_CHECK_HIP(hipGetLastError());
    }
    return YOLO_OK;
}

extern "C" int yolo_aug_resample_batch(const yolo_aug_resample_job* jobs, int n_jobs, void* stream) {
    YOLO_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0), "aug_resample: null argument");
    // first the horizontal passes (and plain copies) of all images, then the vertical ones
    PassJob* pass1 = n_jobs ? new PassJob[n_jobs] : nullptr;
    PassJob* pass2 = n_jobs ? new PassJob[n_jobs] : nullptr;
    int n1 = 0, n2 = 0, rc = YOLO_OK;
    for (int i = 0; i < n_jobs && rc == YOLO_OK; ++i) {
        const yolo_aug_resample_job& j = jobs[i];
        const bool horiz = j.dst_width != j.src_width, vert = j.dst_height != j.src_height;
        if (!(j.src && j.dst))
            rc = fail(YOLO_E_ARG, "aug_resample: job %d: null image", i);
        else if (!(j.src_height > 0 && j.src_width > 0 && j.dst_height > 0 && j.dst_width > 0))
            rc = fail(YOLO_E_ARG, "aug_resample: job %d: sizes must be positive", i);
        else if (!(j.src_stride >= 3ll * j.src_width && j.dst_stride >= 3ll * j.dst_width))
            rc = fail(YOLO_E_ARG, "aug_resample: job %d: row stride below 3 * width", i);
        else if ((long long)(j.src_height > j.dst_height ? j.src_height : j.dst_height) * j.dst_width >= (1ll << 30))
            rc = fail(YOLO_E_ARG, "aug_resample: job %d: image too large", i);
        else if (horiz && !(j.x_bounds && j.x_coef && j.x_ksize > 0))
            rc = fail(YOLO_E_ARG, "aug_resample: job %d: horizontal tables missing", i);
        else if (vert && !(j.y_bounds && j.y_coef && j.y_ksize > 0))
            rc = fail(YOLO_E_ARG, "aug_resample: job %d: vertical tables missing", i);
        else if (horiz && vert &&
                 !(j.workspace && j.workspace_bytes >= yolo_aug_resample_workspace_bytes(j.src_height, j.dst_width)))
            rc = fail(YOLO_E_ARG, "aug_resample: job %d: workspace of %zu bytes is too small", i, (size_t)j.workspace_bytes);
        if (rc != YOLO_OK) break;
        const uint8_t* vsrc = (const uint8_t*)j.src;
        long long vstride = j.src_stride;
        if (horiz || !vert) {
            PassJob p{};
            p.src = (const uint8_t*)j.src;
            p.s_stride = j.src_stride;
            p.out_h = j.src_height;
            p.out_w = j.dst_width;
            p.in_len = j.src_width;
            p.mode = horiz ? 0 : 2;
            p.bounds = j.x_bounds;
            p.coef = j.x_coef;
            p.ksize = j.x_ksize;
            if (horiz && vert) {
                p.dst = (uint8_t*)j.workspace;
                p.d_stride = 3ll * j.dst_width;
                vsrc = p.dst;
                vstride = p.d_stride;
            } else {
                p.dst = (uint8_t*)j.dst;
                p.d_stride = j.dst_stride;
            }
            pass1[n1++] = p;
        }
        if (vert) {
            PassJob p{};
            p.src = vsrc;
            p.s_stride = vstride;
            p.dst = (uint8_t*)j.dst;
            p.d_stride = j.dst_stride;
            p.out_h = j.dst_height;
            p.out_w = j.dst_width;
            p.in_len = j.src_height;
            p.mode = 1;
            p.bounds = j.y_bounds;
            p.coef = j.y_coef;
            p.ksize = j.y_ksize;
            pass2[n2++] = p;
        }
    }
    if (rc == YOLO_OK) rc = launch_passes(pass1, n1, (hipStream_t)stream);
    if (rc == YOLO_OK) rc = launch_passes(pass2, n2, (hipStream_t)stream);
    delete[] pass1;
    delete[] pass2;
    return rc;
}

extern "C" int yolo_aug_resample(const void* src, int src_height, int src_width, long long src_stride, void* dst,
                                 int dst_height, int dst_width, long long dst_stride, const int* x_bounds,
                                 const int* x_coef, int x_ksize, const int* y_bounds, const int* y_coef, int y_ksize,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    yolo_aug_resample_job j{};
    j.src = src;
    j.dst = dst;
    j.workspace = workspace;
    j.workspace_bytes = (long long)workspace_bytes;
    j.src_stride = src_stride;
    j.dst_stride = dst_stride;
    j.src_height = src_height;
    j.src_width = src_width;
    j.dst_height = dst_height;
    j.dst_width = dst_width;
    j.x_bounds = x_bounds;
    j.x_coef = x_coef;
    j.y_bounds = y_bounds;
    j.y_coef = y_coef;
    j.x_ksize = x_ksize;
    j.y_ksize = y_ksize;
    return yolo_aug_resample_batch(&j, 1, stream);
}

extern "C" int yolo_aug_apply(const yolo_aug_job* jobs, int n_jobs, void* stream) {
    YOLO_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0), "aug_apply: null argument");
    for (int i = 0; i < n_jobs; ++i) {
        const yolo_aug_job& j = jobs[i];
        YOLO_REQUIRE(j.src && j.dst, "aug_apply: job %d: null image", i);
        YOLO_REQUIRE(j.src_height > 0 && j.src_width > 0 && j.src_stride >= 3ll * j.src_width, "aug_apply: job %d: bad source geometry", i);
        YOLO_REQUIRE(j.crop_w >= 0 && j.crop_h >= 0 && j.crop_x >= 0 && j.crop_y >= 0 && j.crop_x + j.crop_w <= j.src_width &&
                         j.crop_y + j.crop_h <= j.src_height,
                     "aug_apply: job %d: crop [%d,%d)+(%d,%d) outside the %dx%d image", i, j.crop_x, j.crop_y, j.crop_w, j.crop_h,
                     j.src_width, j.src_height);
        YOLO_REQUIRE(j.dst_x >= 0 && j.dst_y >= 0 && j.dst_x + j.crop_w <= j.dst_width && j.dst_y + j.crop_h <= j.dst_height &&
                         (j.dst_plane_stride ? j.dst_stride >= (long long)j.dst_width &&
                                                   j.dst_plane_stride >= j.dst_stride * j.dst_height
                                             : j.dst_stride >= 3ll * j.dst_width),
                     "aug_apply: job %d: paste at (%d,%d) of %dx%d outside the %dx%d canvas", i, j.dst_x, j.dst_y, j.crop_w,
                     j.crop_h, j.dst_width, j.dst_height);
        YOLO_REQUIRE((long long)j.crop_w * j.crop_h < (1ll << 31), "aug_apply: job %d: crop too large", i);
        YOLO_REQUIRE(j.warp >= 0 && j.warp <= 2, "aug_apply: job %d: warp mode %d", i, j.warp);
        YOLO_REQUIRE(j.border >= 0 && j.border <= 255, "aug_apply: job %d: border value %d", i, j.border);
    }
    for (int first = 0; first < n_jobs; first += AUG_MAX_JOBS) {
        const int n = n_jobs - first < AUG_MAX_JOBS ? n_jobs - first : AUG_MAX_JOBS;
        AugJobs batch{};
        long long most = 0;
        for (int i = 0; i < n; ++i) {
            batch.j[i] = jobs[first + i];
            const long long np = (long long)batch.j[i].crop_w * batch.j[i].crop_h;
            most = np > most ? np : most;
        }
        if (most == 0) continue;
        const int per_block = AUG_THREADS * AUG_PIX_PER_THREAD;
        hipLaunchKernelGGL(aug_apply_kernel, dim3((unsigned)((most + per_block - 1) / per_block), n), dim3(AUG_THREADS), 0,
                           (hipStream_t)stream, batch);
        YOLO_CHECK_HIP(hipGetLastError());
    }
    return YOLO_OK;
}

// Shared helpers for the gfx950 kernels of libyolo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

#include "../../include/yolo_hip.h"

namespace yolo {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Thread-local message of the last failure, returned by yolo_last_error().
char* last_error_buf();
int fail(int code, const char* fmt, ...);

#define YOLO_CHECK_HIP(expr)                                                               \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return ::yolo::fail(YOLO_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

#define YOLO_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ::yolo::fail(YOLO_E_ARG, __VA_ARGS__); \
    } while (0)

// Geometry of an NHWC view as the kernels use it (element units).
struct ViewGeom {
    long long img_stride;  // elements between images
    int row_stride;        // elements between buffer rows
    int pix_stride;        // = ch_total
    int origin;            // element offset of logical pixel (0,0), channel 0 of the view
};

inline ViewGeom geom_of(const yolo_view& v) {
    ViewGeom g;
    const int Hp = v.H + 2 * v.halo, Wp = v.W + 2 * v.halo;
    g.pix_stride = v.ch_total;
    g.row_stride = Wp * v.ch_total;
    g.img_stride = (long long)Hp * Wp * v.ch_total;
    g.origin = (v.halo * Wp + v.halo) * v.ch_total + v.ch_offset;
    return g;
}

inline size_t elem_size(int dtype) { return dtype == YOLO_BF16 ? 2 : 4; }

// A kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) is a property of the kernel ON ONE
// DEVICE: it is raised once per device for the kernels of one call site, with that device current, whichever thread
// launches there first.  One static DynLdsLimit per call site.
struct DynLdsLimit {
    std::mutex lock;
    unsigned long long raised = 0;  // bit d: done on device d
    hipError_t raise(const void* const* kernels, int n, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const unsigned long long bit = 1ull << (dev & 63);
        std::lock_guard<std::mutex> g(lock);
        if (raised & bit) return hipSuccess;
        for (int i = 0; i < n; ++i) {
            e = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
        }
        raised |= bit;
        return hipSuccess;
    }
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    bf16_t h = (bf16_t)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    return __builtin_bit_cast(unsigned short, h);
}
// one v_cvt_pk_bf16_f32: both values rounded to nearest even, lo in bits 0..15
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}

// LeakyReLU for a negative-side slope in [0, 1]: max(x, slope * x) is bit-identical to
// x >= 0 ? x : slope * x there (slope 1 = linear), and costs a packed multiply + one max per element
// instead of multiply + compare + select.  Raw v_max_f32 skips the compiler's NaN-canonicalising extra max.
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x4 leaky4(f32x4 v, float slope) {
    const f32x4 m = v * slope;
    return f32x4{vmax_raw(v[0], m[0]), vmax_raw(v[1], m[1]), vmax_raw(v[2], m[2]), vmax_raw(v[3], m[3])};
}

}  // namespace yolo

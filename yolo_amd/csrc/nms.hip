// Batched per-class NMS for gfx950 and the detection gather / post-process.
//
// Replaces batched_nms(boxes, scores, classes, thr) at yolo/modeling/meta_arch/yolo.py:305
// (detectron2.layers.batched_nms -> torchvision.ops.boxes.batched_nms -> torchvision::nms), the
// keep[:100] truncation at :307 and detectron2's detector_postprocess, for all images of a batch.
//
// Pipeline (all on the caller's stream, no host synchronisation):
//   1. order B: stable radix sort of all candidates by (image, score descending); ties keep index
//      order.  This is the order torchvision returns kept indices in.
//   2. order A: stable radix sort of order B by (image, class): per (image, class) segment the boxes
//      are in descending-score order.
//   3. one wave per segment runs the greedy sweep in chunks of 64 boxes: lane = box; a chunk is
//      first tested against the boxes already kept (kept list in LDS, spilling to global memory),
//      then resolved internally with wave ballots + lane broadcasts.  IoU arithmetic is fp32 in
//      torchvision's operation order; the file is built with -ffp-contract=off.
//   4. one workgroup per image walks order B, ranks the kept flags and emits the first top_k.
// The coordinate-trick branch of torchvision (boxes shifted by class * (max_coordinate + 1) before a
// single class-agnostic NMS) is reproduced by applying the same fp32 shift to the boxes of each
// segment: shifted boxes of different classes never overlap, so per-segment sweeps give the same
// kept set whenever all coordinates are finite.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace yolo {

struct NmsParams {
    const float* boxes;
    const float* scores;
    const int* classes;
    const int* img_offsets;
    int N, M;
    double iou_thr;
    int mode, auto_limit, top_k, cbits;
    unsigned long long* keyB_in;
    unsigned long long* keyB_out;
    unsigned* valB_in;
    unsigned* valB_out;  // order B: original index
    unsigned* keyA_in;
    unsigned* keyA_out;  // (image << cbits) | class in order A
    unsigned* valA_in;
    unsigned* valA_out;  // order A: position in order B
    unsigned* seg_start;
    unsigned* nseg;
    float* img_max;
    unsigned char* keepB;  // kept flag by position in order B
    f32x4* kept_spill;     // [M] kept boxes beyond the LDS list, addressed by segment start
    int* keep;
    int keep_stride;
    int* n_keep;
};

__device__ __forceinline__ int image_of(const int* off, int N, int i) {
    int lo = 0, hi = N;  // largest n with off[n] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256) nms_keysB_kernel(const NmsParams p) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.M; i += gridDim.x * 256) {
        const int n = image_of(p.img_offsets, p.N, i);
        unsigned u = __float_as_uint(p.scores[i]);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-orderable
        u = ~u;                                          // descending score = ascending key
        p.keyB_in[i] = ((unsigned long long)n << 32) | u;
        p.valB_in[i] = (unsigned)i;
    }
}

__global__ void __launch_bounds__(256) nms_keysA_kernel(const NmsParams p) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < p.M; j += gridDim.x * 256) {
        const unsigned n = (unsigned)(p.keyB_out[j] >> 32);
        const unsigned c = (unsigned)p.classes[p.valB_out[j]];
        p.keyA_in[j] = (n << p.cbits) | c;
        p.valA_in[j] = (unsigned)j;
    }
}

__global__ void __launch_bounds__(256) nms_heads_kernel(const NmsParams p) {
    for (int t = blockIdx.x * 256 + threadIdx.x; t < p.M; t += gridDim.x * 256) {
        if (t == 0 || p.keyA_out[t] != p.keyA_out[t - 1]) {
            const unsigned s = atomicAdd(p.nseg, 1u);
            p.seg_start[s] = (unsigned)t;
        }
    }
}

// boxes.max() per image, for the coordinate trick (NaNs are ignored by the comparison).
__global__ void __launch_bounds__(256) nms_imgmax_kernel(const NmsParams p) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    const int b = p.img_offsets[n], e = p.img_offsets[n + 1];
    float m = -INFINITY;
    for (int i = 4 * b + threadIdx.x; i < 4 * e; i += 256) {
        const float v = p.boxes[i];
        m = v > m ? v : m;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const float o = __shfl_down(m, d, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) m = red[k] > m ? red[k] : m;
        p.img_max[n] = m;
    }
}

__device__ __forceinline__ float smaxf(float a, float b) { return (a < b) ? b : a; }  // std::max(a,b)
__device__ __forceinline__ float sminf(float a, float b) { return (b < a) ? b : a; }  // std::min(a,b)

// torchvision nms_kernel_impl inner test: box i (kept, higher score) against box j.
__device__ __forceinline__ bool suppresses(const f32x4 bi, float area_i, const f32x4 bj, float area_j, double thr) {
    const float xx1 = smaxf(bi[0], bj[0]);
    const float yy1 = smaxf(bi[1], bj[1]);
    const float xx2 = sminf(bi[2], bj[2]);
    const float yy2 = sminf(bi[3], bj[3]);
    const float w = smaxf(0.0f, xx2 - xx1);
    const float h = smaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (area_i + area_j - inter);
    return (double)ovr > thr;
}

__device__ __forceinline__ f32x4 bcast_box(const f32x4 b, int srclane) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        r[e] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(b[e]), srclane));
    return r;
}

constexpr int KEPT_LDS = 448;  // kept boxes per wave held in LDS (7 KiB); the rest spills to global

__global__ void __launch_bounds__(256) nms_sweep_kernel(const NmsParams p) {
    __shared__ f32x4 kept_lds[4][KEPT_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nseg = *p.nseg;
    const unsigned cmask = (1u << p.cbits) - 1u;
    f32x4* klds = kept_lds[wave];
    for (unsigned seg = blockIdx.x * 4 + wave; seg < nseg; seg += gridDim.x * 4) {
        const unsigned start = p.seg_start[seg];
        const unsigned key = p.keyA_out[start];
        const int n = (int)(key >> p.cbits);
        const unsigned cls = key & cmask;
        const int m_img = p.img_offsets[n + 1] - p.img_offsets[n];
        const bool trick = p.mode == YOLO_NMS_COORD_TRICK || (p.mode == YOLO_NMS_AUTO && 4 * m_img <= p.auto_limit);
        // torchvision: offsets = idxs.to(boxes) * (max_coordinate + 1); boxes_for_nms = boxes + offsets
        const float shift = trick ? (float)cls * (p.img_max[n] + 1.0f) : 0.0f;
        f32x4* kspill = p.kept_spill + start;
        int nkept = 0;
        const int kcap = p.top_k > 0 ? p.top_k : 0x7fffffff;
        for (unsigned cs = start; cs < (unsigned)p.M && nkept < kcap; cs += 64) {
            const unsigned t = cs + lane;
            bool valid = t < (unsigned)p.M;
            if (valid) valid = p.keyA_out[t] == key;
            const unsigned long long vmask = __ballot(valid);
            if (vmask == 0) break;
            unsigned posB = 0;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                posB = p.valA_out[t];
                b = *(const f32x4*)(p.boxes + 4ull * p.valB_out[posB]);
                if (trick) {
                    b[0] = b[0] + shift;
                    b[1] = b[1] + shift;
                    b[2] = b[2] + shift;
                    b[3] = b[3] + shift;
                }
            }
            const float area = (b[2] - b[0]) * (b[3] - b[1]);
            bool alive = valid;
            // phase A: against the boxes kept from earlier chunks (all have higher or equal score)
            for (int k = 0; k < nkept; ++k) {
                const f32x4 kb = k < KEPT_LDS ? klds[k] : kspill[k - KEPT_LDS];
                const float ka = (kb[2] - kb[0]) * (kb[3] - kb[1]);
                if (suppresses(kb, ka, b, area, p.iou_thr)) alive = false;
            }
            // phase B: inside the chunk, in score order
            unsigned long long amask = __ballot(alive);
            unsigned long long todo = amask;
            while (todo) {
                const int i = __builtin_ctzll(todo);
                todo &= todo - 1;
                const f32x4 bi = bcast_box(b, i);
                const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
                if (lane > i && alive && suppresses(bi, ai, b, area, p.iou_thr)) alive = false;
                const unsigned long long now = __ballot(alive);
                todo &= now;
                amask = now;
            }
            // append survivors in score order
            const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
            const int r = nkept + __popcll(amask & lt);
            if (alive && r < kcap) {
                p.keepB[posB] = 1;
                if (r < KEPT_LDS) klds[r] = b; else kspill[r - KEPT_LDS] = b;
            }
            nkept += __popcll(amask);
            // the kept list is read by every lane in the next chunk
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (!(vmask >> 63)) break;  // the segment ended inside this chunk
        }
    }
}

// One workgroup per image: kept flags in order B -> ranks -> keep list.
__global__ void __launch_bounds__(256) nms_emit_kernel(const NmsParams p) {
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = p.img_offsets[n], e = p.img_offsets[n + 1];
    const int limit = p.top_k > 0 ? (p.top_k < p.keep_stride ? p.top_k : p.keep_stride) : p.keep_stride;
    int* out = p.keep + (long long)n * p.keep_stride;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = b; base < e; base += 256) {
        const int j = base + tid;
        const int v = (j < e && j < p.M) ? (int)p.keepB[j] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int pre = carry_s;
        for (int k = 0; k < wave; ++k) pre += wsum[k];
        const int rank = pre + incl - v;
        if (v && rank < limit) out[rank] = (int)p.valB_out[j] - b;
        __syncthreads();
        if (tid == 255) carry_s = pre + incl;
        __syncthreads();
        if (carry_s >= limit) break;
    }
    if (tid == 0) p.n_keep[n] = carry_s < limit ? carry_s : limit;
}

__global__ void nms_zero_counts_kernel(int* n_keep, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) n_keep[i] = 0;
}

struct GatherParams {
    const float* boxes;
    const float* scores;
    const int* classes;
    const int* img_offsets;
    const int* keep;
    const int* n_keep;
    const int* in_hw;
    const int* out_hw;
    float* det;
    int* n_det;
    int keep_stride, N;
};

// detector_postprocess: scale by (out_w / in_w, out_h / in_h), clip to the output frame, drop boxes
// whose clipped width or height is not positive; order preserved.  One wave per image.
__global__ void __launch_bounds__(64) gather_kernel(const GatherParams p) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int nk = p.n_keep[n];
    const int base = p.img_offsets[n];
    const float in_h = (float)p.in_hw[2 * n], in_w = (float)p.in_hw[2 * n + 1];
    const float out_h = (float)p.out_hw[2 * n], out_w = (float)p.out_hw[2 * n + 1];
    // python: scale = out / in evaluated in double, then the fp32 tensor is multiplied by it
    const float sx = (float)((double)p.out_hw[2 * n + 1] / (double)p.in_hw[2 * n + 1]);
    const float sy = (float)((double)p.out_hw[2 * n] / (double)p.in_hw[2 * n]);
    (void)in_h;
    (void)in_w;
    int written = 0;
    float* out = p.det + (long long)n * p.keep_stride * 6;
    for (int s = 0; s < nk; s += 64) {
        const int k = s + lane;
        bool ok = false;
        float x0 = 0, y0 = 0, x1 = 0, y1 = 0, sc = 0, cl = 0;
        if (k < nk) {
            const int i = base + p.keep[(long long)n * p.keep_stride + k];
            const f32x4 b = *(const f32x4*)(p.boxes + 4ll * i);
            x0 = fminf(fmaxf(b[0] * sx, 0.f), out_w);
            y0 = fminf(fmaxf(b[1] * sy, 0.f), out_h);
            x1 = fminf(fmaxf(b[2] * sx, 0.f), out_w);
            y1 = fminf(fmaxf(b[3] * sy, 0.f), out_h);
            sc = p.scores[i];
            cl = (float)p.classes[i];
            ok = (x1 - x0) > 0.f && (y1 - y0) > 0.f;
        }
        const unsigned long long m = __ballot(ok);
        const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (ok) {
            float* o = out + (long long)(written + __popcll(m & lt)) * 6;
            o[0] = x0;
            o[1] = y0;
            o[2] = x1;
            o[3] = y1;
            o[4] = sc;
            o[5] = cl;
        }
        written += __popcll(m);
    }
    if (lane == 0) p.n_det[n] = written;
}

static size_t al(size_t v) { return (v + 255) / 256 * 256; }

struct NmsLayout {
    size_t keyB_in, keyB_out, valB_in, valB_out, keyA_in, keyA_out, valA_in, valA_out, seg_start, nseg, img_max, keepB,
        kept_spill, sort_tmp, sort_tmp_bytes, total;
};

static int nms_layout(int N, int M, NmsLayout& L) {
    size_t o = 0;
    const size_t m = (size_t)(M > 0 ? M : 1);
    auto take = [&](size_t bytes) {
        const size_t r = o;
        o += al(bytes);
        return r;
    };
    L.keyB_in = take(m * 8);
    L.keyB_out = take(m * 8);
    L.valB_in = take(m * 4);
    L.valB_out = take(m * 4);
    L.keyA_in = take(m * 4);
    L.keyA_out = take(m * 4);
    L.valA_in = take(m * 4);
    L.valA_out = take(m * 4);
    L.seg_start = take(m * 4);
    L.nseg = take(256);
    L.img_max = take((size_t)N * 4);
    L.keepB = take(m);
    L.kept_spill = take(m * 16);
    size_t t1 = 0, t2 = 0;
    hipError_t e1 = rocprim::radix_sort_pairs(nullptr, t1, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                              (unsigned*)nullptr, (unsigned*)nullptr, m, 0u, 64u);
    hipError_t e2 = rocprim::radix_sort_pairs(nullptr, t2, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                                              (unsigned*)nullptr, m, 0u, 32u);
    if (e1 != hipSuccess || e2 != hipSuccess) return -1;
    L.sort_tmp_bytes = t1 > t2 ? t1 : t2;
    L.sort_tmp = take(L.sort_tmp_bytes);
    L.total = o;
    return 0;
}

static int bits_for(unsigned v) {  // bits needed to represent values 0..v-1
    int b = 1;
    while ((1u << b) < v && b < 31) ++b;
    return b;
}

}  // namespace yolo

using namespace yolo;

extern "C" size_t yolo_nms_workspace_bytes(int N, int M) {
    if (N <= 0 || M < 0) return 0;
    NmsLayout L;
    if (nms_layout(N, M, L) != 0) return 0;
    return L.total;
}

extern "C" int yolo_batched_nms(const float* boxes, const float* scores, const int* classes, const int* img_offsets,
                                int N, int M_host, double iou_thr, int mode, int auto_numel_limit, int top_k,
                                int num_classes, int* keep, int keep_stride, int* n_keep, void* workspace,
                                size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(img_offsets && keep && n_keep, "nms: null argument");
    YOLO_REQUIRE(N >= 1 && N <= 65535, "nms: batch %d (1..65535)", N);
    YOLO_REQUIRE(M_host >= 0, "nms: negative candidate count");
    YOLO_REQUIRE(mode >= YOLO_NMS_PER_CLASS && mode <= YOLO_NMS_AUTO, "nms: mode %d", mode);
    YOLO_REQUIRE(keep_stride > 0 && top_k >= 0, "nms: keep_stride %d / top_k %d", keep_stride, top_k);
    YOLO_REQUIRE(num_classes >= 1 && num_classes <= 65536, "nms: num_classes %d", num_classes);
    hipStream_t st = (hipStream_t)stream;
    if (M_host == 0) {
        hipLaunchKernelGGL(nms_zero_counts_kernel, dim3((N + 255) / 256), dim3(256), 0, st, n_keep, N);
        YOLO_CHECK_HIP(hipGetLastError());
        return YOLO_OK;
    }
    YOLO_REQUIRE(boxes && scores && classes && workspace, "nms: null argument");
    NmsLayout L;
    if (nms_layout(N, M_host, L) != 0) return fail(YOLO_E_HIP, "nms: rocprim temp-storage query failed");
    if (ws_bytes < L.total) return fail(YOLO_E_WORKSPACE, "nms: workspace %zu < %zu bytes", ws_bytes, L.total);
    const int cbits = bits_for((unsigned)num_classes);
    const int nbits = bits_for((unsigned)N);
    YOLO_REQUIRE(cbits + nbits <= 32, "nms: batch x classes too large for the segment key");

    char* ws = (char*)workspace;
    NmsParams p{};
    p.boxes = boxes;
    p.scores = scores;
    p.classes = classes;
    p.img_offsets = img_offsets;
    p.N = N;
    p.M = M_host;
    p.iou_thr = iou_thr;
    p.mode = mode;
    p.auto_limit = auto_numel_limit;
    p.top_k = top_k;
    p.cbits = cbits;
    p.keyB_in = (unsigned long long*)(ws + L.keyB_in);
    p.keyB_out = (unsigned long long*)(ws + L.keyB_out);
    p.valB_in = (unsigned*)(ws + L.valB_in);
    p.valB_out = (unsigned*)(ws + L.valB_out);
    p.keyA_in = (unsigned*)(ws + L.keyA_in);
    p.keyA_out = (unsigned*)(ws + L.keyA_out);
    p.valA_in = (unsigned*)(ws + L.valA_in);
    p.valA_out = (unsigned*)(ws + L.valA_out);
    p.seg_start = (unsigned*)(ws + L.seg_start);
    p.nseg = (unsigned*)(ws + L.nseg);
    p.img_max = (float*)(ws + L.img_max);
    p.keepB = (unsigned char*)(ws + L.keepB);
    p.kept_spill = (f32x4*)(ws + L.kept_spill);
    p.keep = keep;
    p.keep_stride = keep_stride;
    p.n_keep = n_keep;

    int blocks = (M_host + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    YOLO_CHECK_HIP(hipMemsetAsync(p.nseg, 0, 256, st));
    YOLO_CHECK_HIP(hipMemsetAsync(p.keepB, 0, al((size_t)M_host), st));
    hipLaunchKernelGGL(nms_keysB_kernel, dim3(blocks), dim3(256), 0, st, p);
    size_t tmp = L.sort_tmp_bytes;
    YOLO_CHECK_HIP(rocprim::radix_sort_pairs(ws + L.sort_tmp, tmp, p.keyB_in, p.keyB_out, p.valB_in, p.valB_out,
                                             (size_t)M_host, 0u, (unsigned)(32 + nbits), st));
    hipLaunchKernelGGL(nms_keysA_kernel, dim3(blocks), dim3(256), 0, st, p);
    tmp = L.sort_tmp_bytes;
    YOLO_CHECK_HIP(rocprim::radix_sort_pairs(ws + L.sort_tmp, tmp, p.keyA_in, p.keyA_out, p.valA_in, p.valA_out,
                                             (size_t)M_host, 0u, (unsigned)(cbits + nbits), st));
    hipLaunchKernelGGL(nms_heads_kernel, dim3(blocks), dim3(256), 0, st, p);
    if (mode != YOLO_NMS_PER_CLASS) hipLaunchKernelGGL(nms_imgmax_kernel, dim3(N), dim3(256), 0, st, p);
    // one wave per segment; at most N*num_classes segments
    long long segs = (long long)N * num_classes;
    if (segs > M_host) segs = M_host;
    long long sblocks = (segs + 3) / 4;
    if (sblocks > 4096) sblocks = 4096;
    hipLaunchKernelGGL(nms_sweep_kernel, dim3((unsigned)sblocks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(nms_emit_kernel, dim3(N), dim3(256), 0, st, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_gather_detections(const float* boxes, const float* scores, const int* classes,
                                      const int* img_offsets, const int* keep, int keep_stride, const int* n_keep,
                                      int N, const int* in_hw, const int* out_hw, float* det, int* n_det,
                                      void* stream) {
    YOLO_REQUIRE(boxes && scores && classes && img_offsets && keep && n_keep && in_hw && out_hw && det && n_det,
                 "gather: null argument");
    YOLO_REQUIRE(N >= 1 && keep_stride > 0, "gather: N %d keep_stride %d", N, keep_stride);
    GatherParams p{boxes, scores, classes, img_offsets, keep, n_keep, in_hw, out_hw, det, n_det, keep_stride, N};
    hipLaunchKernelGGL(gather_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

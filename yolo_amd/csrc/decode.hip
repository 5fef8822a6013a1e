// YOLOv3 head activation + anchor decode + score filter + order-preserving compaction for gfx950.
//
// Replaces YoloV3Head.forward (yolo/modeling/meta_arch/yolo.py:392-420), _transpose_dense_predictions
// (:276-278), the per-forward anchor generation (:279; yolo/modeling/anchor_generator.py:18-40) and the
// per-image / per-level Python loops of _decode_multi_level_predictions / _decode_per_level_predictions
// (:309-383) with Box2BoxTransform.apply_deltas (yolo/modeling/box_regression.py:44-64).
//
// Three launches for the whole batch:
//   count : one lane per grid cell reads the cell's A objectness logits; cells where an anchor passes the
//           first filter are then read whole by the wave (coalesced, 4 channels per lane) to count the
//           (anchor, class) pairs that pass both filters;
//   scan  : per image exclusive prefix sum of the cell counts;
//   write : cells with a non-zero count are decoded again and their candidates written at
//           image_base + cell_offset + rank, rank = position in (anchor, class) order -- the
//           reference's candidate order (level, y, x, anchor, class) without atomics.
// `count` touches one sector per anchor of every cell plus the full rows of the few live cells; `write`
// touches only cells that produced candidates.
//
// Built with -ffp-contract=off: the box arithmetic keeps the reference's rounding sequence.
#include <mutex>

#include "common.h"

namespace yolo {

constexpr int MAX_LEVELS = 4;
constexpr int MAX_ANCHORS = 4;

struct DecodeParams {
    const char* head[MAX_LEVELS];
    long long img_stride[MAX_LEVELS];  // bytes
    long long origin[MAX_LEVELS];      // bytes
    int row_stride[MAX_LEVELS];        // bytes
    int pix_stride[MAX_LEVELS];        // bytes
    int gh[MAX_LEVELS], gw[MAX_LEVELS];
    int cell_base[MAX_LEVELS + 1];  // first cell index of each level inside an image
    float stride[MAX_LEVELS];
    float aw[MAX_LEVELS][MAX_ANCHORS], ah[MAX_LEVELS][MAX_ANCHORS];
    int L, N, A, C, K, AK;
    int cells;  // cells per image
    float thr;
    int* cell_count;   // [N][cells]
    int* cell_off;     // [N][cells]
    int* img_total;    // [N]
    float* boxes;
    float* scores;
    int* classes;
    int* img_offsets;  // [N+1]
    int cap;
};

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename T>
__device__ __forceinline__ float load_elem(const char* p, int idx) {
    if constexpr (sizeof(T) == 2)
        return bf16_bits_to_f32(((const unsigned short*)p)[idx]);
    else
        return ((const float*)p)[idx];
}

template <typename T>
__device__ __forceinline__ void load4(const char* p, int idx, float v[4]) {
    if constexpr (sizeof(T) == 2) {
        const u32x2 r = *(const u32x2*)(p + (size_t)idx * 2);
        v[0] = __uint_as_float(r[0] << 16);
        v[1] = __uint_as_float(r[0] & 0xffff0000u);
        v[2] = __uint_as_float(r[1] << 16);
        v[3] = __uint_as_float(r[1] & 0xffff0000u);
    } else {
        const f32x4 r = *(const f32x4*)(p + (size_t)idx * 4);
        v[0] = r[0];
        v[1] = r[1];
        v[2] = r[2];
        v[3] = r[3];
    }
}

struct CellPos {
    const char* ptr;  // first channel of the cell
    int level, y, x;
};

__device__ __forceinline__ CellPos locate_cell(const DecodeParams& p, int n, int cell) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < MAX_LEVELS; ++i)
        if (i < p.L && cell >= p.cell_base[i]) l = i;
    const int r = cell - p.cell_base[l];
    CellPos c;
    c.level = l;
    c.y = r / p.gw[l];
    c.x = r - c.y * p.gw[l];
    c.ptr = p.head[l] + p.origin[l] + (long long)n * p.img_stride[l] + (long long)c.y * p.row_stride[l] +
            (long long)c.x * p.pix_stride[l];
    return c;
}

// Per-lane hit mask (bit e = element e of this lane's 4 channels of chunk `chunk` is a candidate)
// and the candidate scores.  Channel ch = chunk*256 + 4*lane + e -> anchor ch / K, field ch % K.
template <typename T>
__device__ __forceinline__ unsigned cell_hits(const DecodeParams& p, const char* cptr, int chunk, int lane,
                                              const float conf[MAX_ANCHORS], float sc[4]) {
    const int chb = chunk * 256 + 4 * lane;
    unsigned hits = 0;
    if (chb < p.AK) {
        // skip the load when no anchor touched by this lane passed the objectness filter
        const int a_lo = chb / p.K;
        int a_hi = (chb + 3) / p.K;
        a_hi = a_hi < p.A ? a_hi : p.A - 1;
        if (conf[a_lo] > p.thr || conf[a_hi] > p.thr) {
            float v[4];
            load4<T>(cptr, chb, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = chb + e;
                const int a = ch / p.K;
                const int f = ch - a * p.K;
                if (ch < p.AK && f >= 5 && conf[a] > p.thr) {  // yolo.py:356 first filter, strict
                    const float s = sigmoidf_ref(v[e]) * conf[a];  // yolo.py:363
                    sc[e] = s;
                    if (s > p.thr) hits |= 1u << e;  // yolo.py:367 second filter, strict
                }
            }
        }
    }
    return hits;
}

template <typename T>
__device__ __forceinline__ void cell_conf(const DecodeParams& p, const char* cptr, float conf[MAX_ANCHORS]) {
#pragma unroll
    for (int a = 0; a < MAX_ANCHORS; ++a)
        conf[a] = a < p.A ? sigmoidf_ref(load_elem<T>(cptr, a * p.K + 4)) : 0.f;  // yolo.py:413
}

// One wave per group of 64 consecutive (image, cell) slots: every lane reads the objectness logits of its own
// cell (A small loads in flight per lane instead of one dependent round trip per cell), then the wave walks
// only the cells where some anchor passed the first filter and counts their candidates cooperatively.
template <typename T>
__global__ void __launch_bounds__(256) decode_count_kernel(const DecodeParams p) {
    const int lane = threadIdx.x & 63;
    const long long total = (long long)p.N * p.cells;
    const long long groups = (total + 63) >> 6;
    const long long nw = (long long)gridDim.x * 4;
    const int nchunks = (p.AK + 255) >> 8;
    for (long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); g < groups; g += nw) {
        const long long w = g * 64 + lane;
        const bool valid = w < total;
        const long long wc = valid ? w : total - 1;
        const int n = (int)(wc / p.cells);
        const int cell = (int)(wc - (long long)n * p.cells);
        const CellPos c = locate_cell(p, n, cell);
        float conf[MAX_ANCHORS];
        cell_conf<T>(p, c.ptr, conf);
        bool any = false;
#pragma unroll
        for (int a = 0; a < MAX_ANCHORS; ++a) any = any || conf[a] > p.thr;
        unsigned long long todo = __ballot(valid && any);
        int mine = 0;
        while (todo) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const long long wb = g * 64 + b;  // wave-uniform
            const int nb = (int)(wb / p.cells);
            const CellPos cb = locate_cell(p, nb, (int)(wb - (long long)nb * p.cells));
            float confb[MAX_ANCHORS];
#pragma unroll
            for (int a = 0; a < MAX_ANCHORS; ++a) confb[a] = __shfl(conf[a], b, 64);
            int cnt = 0;
            for (int chunk = 0; chunk < nchunks; ++chunk) {
                float sc[4];
                const unsigned h = cell_hits<T>(p, cb.ptr, chunk, lane, confb, sc);
#pragma unroll
                for (int e = 0; e < 4; ++e) cnt += __popcll(__ballot((h >> e) & 1u));
            }
            if (lane == b) mine = cnt;
        }
        if (valid) p.cell_count[w] = mine;
    }
}

// One workgroup per image: exclusive scan of the image's cell counts.  Each thread owns a contiguous run of
// cells; one block-wide scan of the 256 run sums, then every thread rewrites its run.
__global__ void __launch_bounds__(256) decode_scan_kernel(const DecodeParams p) {
    __shared__ int wsum[4];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* cnt = p.cell_count + (long long)n * p.cells;
    int* off = p.cell_off + (long long)n * p.cells;
    const int per = (p.cells + 255) / 256;
    const int lo = tid * per;
    const int hi = lo + per < p.cells ? lo + per : p.cells;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int k = 0; k < wave; ++k) run += wsum[k];
    for (int i = lo; i < hi; ++i) {
        off[i] = run;
        run += cnt[i];
    }
    if (tid == 255) p.img_total[n] = run;
}

template <typename T>
__global__ void __launch_bounds__(256) decode_write_kernel(const DecodeParams p) {
    __shared__ long long red[4];
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // first candidate slot of image n = sum of the totals of images before it
    long long part = 0;
    for (int i = tid; i < n; i += 256) part += p.img_total[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const long long base_ll = red[0] + red[1] + red[2] + red[3];
    const long long cap = p.cap;
    if (blockIdx.x == 0 && tid == 0) {
        const long long clampv = 0x7fffffffll;
        p.img_offsets[n] = (int)(base_ll < clampv ? base_ll : clampv);
        if (n == p.N - 1) {
            const long long tot = base_ll + p.img_total[n];
            p.img_offsets[p.N] = (int)(tot < clampv ? tot : clampv);
        }
    }
    const int nchunks = (p.AK + 255) >> 8;
    const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int* cnt = p.cell_count + (long long)n * p.cells;
    const int* off = p.cell_off + (long long)n * p.cells;
    for (int cg = blockIdx.x * 4 + wave; cg * 64 < p.cells; cg += gridDim.x * 4) {
      // 64 cells per wave; only those with candidates are decoded again
      const int mycell = cg * 64 + lane;
      unsigned long long todo = __ballot(mycell < p.cells && cnt[mycell < p.cells ? mycell : 0] != 0);
      while (todo) {
        const int cell = cg * 64 + __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const CellPos c = locate_cell(p, n, cell);
        float conf[MAX_ANCHORS];
        cell_conf<T>(p, c.ptr, conf);
        long long pos = base_ll + off[cell];
        const float stride = p.stride[c.level];
        // DefaultAnchorGenerator grid shifts: offset*stride + i*stride with offset 0.5
        const float sx = 0.5f * stride + (float)c.x * stride;
        const float sy = 0.5f * stride + (float)c.y * stride;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            float sc[4];
            const unsigned h = cell_hits<T>(p, c.ptr, chunk, lane, conf, sc);
            unsigned long long bal[4];
            int before = 0, tot = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bal[e] = __ballot((h >> e) & 1u);
                before += __popcll(bal[e] & lt_mask);
                tot += __popcll(bal[e]);
            }
            if (h) {
                const int chb = chunk * 256 + 4 * lane;
                int r = before;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!((h >> e) & 1u)) continue;
                    const long long o = pos + r;
                    ++r;
                    if (o >= cap) continue;
                    const int ch = chb + e;
                    const int a = ch / p.K;
                    const int cls = ch - a * p.K - 5;
                    // YoloV3Head: dx,dy = sigmoid(t) - 0.5 (yolo.py:410-411); dw,dh raw
                    const float dx = sigmoidf_ref(load_elem<T>(c.ptr, a * p.K + 0)) - 0.5f;
                    const float dy = sigmoidf_ref(load_elem<T>(c.ptr, a * p.K + 1)) - 0.5f;
                    const float dw = load_elem<T>(c.ptr, a * p.K + 2);
                    const float dh = load_elem<T>(c.ptr, a * p.K + 3);
                    // anchor = shift + cell anchor (-w/2,-h/2,w/2,h/2)
                    const float hw = p.aw[c.level][a] / 2.0f, hh = p.ah[c.level][a] / 2.0f;
                    const float ax0 = sx + (-hw), ay0 = sy + (-hh), ax1 = sx + hw, ay1 = sy + hh;
                    // apply_deltas, box_regression.py:50-61
                    const float bw = ax1 - ax0, bh = ay1 - ay0;
                    const float bcx = ax0 + bw / 2.0f, bcy = ay0 + bh / 2.0f;
                    const float tcx = dx * stride + bcx, tcy = dy * stride + bcy;
                    const float tw = bw * expf(dw), th = bh * expf(dh);
                    f32x4 box;
                    box[0] = tcx - tw / 2.0f;
                    box[1] = tcy - th / 2.0f;
                    box[2] = tcx + tw / 2.0f;
                    box[3] = tcy + th / 2.0f;
                    *(f32x4*)(p.boxes + 4 * o) = box;
                    p.scores[o] = sc[e];
                    p.classes[o] = cls;
                }
            }
            pos += tot;
        }
      }
    }
}

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

static int cells_per_image(const yolo_view* head, int L) {
    int c = 0;
    for (int l = 0; l < L; ++l) c += head[l].H * head[l].W;
    return c;
}

}  // namespace yolo

using namespace yolo;

extern "C" size_t yolo_decode_workspace_bytes(const yolo_view* head, int L, int N) {
    if (!head || L <= 0 || N <= 0) return 0;
    const size_t cells = (size_t)cells_per_image(head, L) * (size_t)N;
    return 2 * align256(cells * sizeof(int)) + align256((size_t)N * sizeof(int));
}

static int decode_impl(const yolo_view* head, int L, const int* stride_host, const float* anchors_wh_host, int N,
                       int A, int C, float thr, int head_dtype, float* boxes, float* scores, int* classes,
                       int* img_offsets, int cap, void* workspace, size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(head && stride_host && anchors_wh_host && boxes && scores && classes && img_offsets && workspace,
                 "decode: null argument");
    YOLO_REQUIRE(L >= 1 && L <= MAX_LEVELS, "decode: %d levels (1..%d)", L, MAX_LEVELS);
    YOLO_REQUIRE(A >= 1 && A <= MAX_ANCHORS, "decode: %d anchors per cell (1..%d)", A, MAX_ANCHORS);
    YOLO_REQUIRE(C >= 2, "decode: num_classes %d; the reference implements only num_classes > 1 (yolo.py:364-376)", C);
    YOLO_REQUIRE(N >= 1 && N <= 65535, "decode: batch %d (1..65535)", N);
    YOLO_REQUIRE(cap >= 0, "decode: negative capacity");
    YOLO_REQUIRE(head_dtype == YOLO_F32 || head_dtype == YOLO_BF16, "decode: head dtype %d", head_dtype);
    YOLO_REQUIRE(ws_bytes >= yolo_decode_workspace_bytes(head, L, N), "decode: workspace %zu < %zu bytes", ws_bytes,
                 yolo_decode_workspace_bytes(head, L, N));
    const int es = (int)elem_size(head_dtype);
    DecodeParams p{};
    p.L = L;
    p.N = N;
    p.A = A;
    p.C = C;
    p.K = 5 + C;
    p.AK = A * p.K;
    int cells = 0;
    for (int l = 0; l < L; ++l) {
        const yolo_view& v = head[l];
        YOLO_REQUIRE(v.ptr && v.H > 0 && v.W > 0, "decode: bad head view %d", l);
        YOLO_REQUIRE(v.ch_total % 8 == 0 && v.ch_offset % 8 == 0 && v.ch_offset + (p.AK + 3) / 4 * 4 <= v.ch_total,
                     "decode: level %d needs %d channels (8-aligned slice), view has %d at offset %d", l, p.AK,
                     v.ch_total, v.ch_offset);
        const ViewGeom g = geom_of(v);
        p.head[l] = (const char*)v.ptr;
        p.img_stride[l] = g.img_stride * es;
        p.row_stride[l] = g.row_stride * es;
        p.pix_stride[l] = g.pix_stride * es;
        p.origin[l] = (long long)g.origin * es;
        p.gh[l] = v.H;
        p.gw[l] = v.W;
        p.cell_base[l] = cells;
        cells += v.H * v.W;
        p.stride[l] = (float)stride_host[l];
        for (int a = 0; a < A; ++a) {
            p.aw[l][a] = anchors_wh_host[(l * A + a) * 2 + 0];
            p.ah[l][a] = anchors_wh_host[(l * A + a) * 2 + 1];
        }
    }
    p.cell_base[L] = cells;
    p.cells = cells;
    p.thr = thr;
    char* ws = (char*)workspace;
    const size_t cb = align256((size_t)cells * N * sizeof(int));
    p.cell_count = (int*)ws;
    p.cell_off = (int*)(ws + cb);
    p.img_total = (int*)(ws + 2 * cb);
    p.boxes = boxes;
    p.scores = scores;
    p.classes = classes;
    p.img_offsets = img_offsets;
    p.cap = cap;

    hipStream_t st = (hipStream_t)stream;
    const long long waves = ((long long)cells * N + 63) / 64;
    long long cblocks = (waves + 3) / 4;
    if (cblocks > 4096) cblocks = 4096;
    int wblocks = ((cells + 63) / 64 + 3) / 4;
    if (wblocks > 64) wblocks = 64;
    if (head_dtype == YOLO_BF16) {
        hipLaunchKernelGGL((decode_count_kernel<unsigned short>), dim3((unsigned)cblocks), dim3(256), 0, st, p);
        hipLaunchKernelGGL(decode_scan_kernel, dim3(N), dim3(256), 0, st, p);
        hipLaunchKernelGGL((decode_write_kernel<unsigned short>), dim3(wblocks, N), dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL((decode_count_kernel<float>), dim3((unsigned)cblocks), dim3(256), 0, st, p);
        hipLaunchKernelGGL(decode_scan_kernel, dim3(N), dim3(256), 0, st, p);
        hipLaunchKernelGGL((decode_write_kernel<float>), dim3(wblocks, N), dim3(256), 0, st, p);
    }
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_decode(const yolo_view* head, int L, const int* stride_host, const float* anchors_wh_host, int N,
                           int A, int C, float thr, int head_dtype, float* boxes, float* scores, int* classes,
                           int* img_offsets, int cap, void* workspace, size_t ws_bytes, void* stream) {
    return decode_impl(head, L, stride_host, anchors_wh_host, N, A, C, thr, head_dtype, boxes, scores, classes,
                       img_offsets, cap, workspace, ws_bytes, stream);
}

// ---- yolo_decode_noalloc: the same launches with the workspace in a device array of this code object -----------------
//
// For callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc).  The per-cell counts, per-cell offsets and per-image totals of yolo_decode live in `scratch`, a
// file-scope device array that every GPU the library runs on holds (at its own address) until the process ends.
// In host code the array's NAME is only a stand-in; its address on the current device comes from
// hipGetSymbolAddress, once per device, and only that pointer is handed to the kernels.
namespace yolo {

constexpr long long NOALLOC_MAX_SLOTS = YOLO_DECODE_NOALLOC_MAX_SLOTS;  // batch x grid cells per image
constexpr int NOALLOC_MAX_BATCH = YOLO_DECODE_NOALLOC_MAX_BATCH;
constexpr size_t NOALLOC_BYTES = 2 * NOALLOC_MAX_SLOTS * sizeof(int) + NOALLOC_MAX_BATCH * sizeof(int);  // 4.02 MiB

static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

constexpr int MAX_DEVICES = 64;
static std::mutex scratch_mutex;
static void* scratch_ptr[MAX_DEVICES];  // per device: address of that GPU's copy of `scratch` (null = not asked yet)

// Address of the current device's copy of `scratch`; resolved on the first call on that device.
static int scratch_of_current_device(void** out) {
    int dev = -1;
    YOLO_CHECK_HIP(hipGetDevice(&dev));
    YOLO_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "decode_noalloc: device index %d (0..%d)", dev, MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_ptr[dev]) {
        void* ptr = nullptr;
        YOLO_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        YOLO_REQUIRE(ptr != nullptr, "decode_noalloc: the scratch array has no address on device %d", dev);
        scratch_ptr[dev] = ptr;
    }
    *out = scratch_ptr[dev];
    return YOLO_OK;
}

}  // namespace yolo

extern "C" int yolo_decode_noalloc(const yolo_view* head, int L, const int* stride_host, const float* anchors_wh_host,
                                   int N, int A, int C, float thr, int head_dtype, float* boxes, float* scores,
                                   int* classes, int* img_offsets, int cap, void* stream) {
    YOLO_REQUIRE(head && L >= 1 && L <= MAX_LEVELS && N >= 1, "decode_noalloc: bad argument");
    const long long slots = (long long)cells_per_image(head, L) * N;
    YOLO_REQUIRE(slots <= NOALLOC_MAX_SLOTS && N <= NOALLOC_MAX_BATCH,
                 "decode_noalloc: batch %d x %d grid cells = %lld slots exceeds the built-in scratch (%lld slots, batch <= %d); "
                 "use yolo_decode with a caller-owned workspace",
                 N, cells_per_image(head, L), slots, NOALLOC_MAX_SLOTS, NOALLOC_MAX_BATCH);
    void* ptr = nullptr;
    const int rc = scratch_of_current_device(&ptr);
    if (rc != YOLO_OK) return rc;
    // every scratch word the kernels read is written earlier in the same call (a count for every cell, an offset for
    // every cell, a total for every image): nothing has to start at zero, and what the last call left is never read
    return decode_impl(head, L, stride_host, anchors_wh_host, N, A, C, thr, head_dtype, boxes, scores, classes, img_offsets,
                       cap, ptr, NOALLOC_BYTES, stream);
}

// Training-time label assignment and the three YOLOv3 losses with their gradient w.r.t. the raw head tensors,
// for gfx950.  Replaces, for a whole batch and without a host read-back:
//   YoloV3Head.forward + _transpose_dense_predictions      yolo/modeling/meta_arch/yolo.py:107-109,392-420
//   label_anchors + pairwise_iou + Matcher.__call__         yolo.py:214-264, yolo/modeling/matcher.py:18-55
//   losses / _compute_*_loss                                yolo.py:131-208
//   Box2BoxTransform.get_deltas                             yolo/modeling/box_regression.py:18-42
// and the autograd backward of all of these down to the head convolutions' outputs.
//
// Anchor index r of an image = level offset + (y * W + x) * A + a, levels from the finest stride (the order of
// DenseDetector._transpose_dense_predictions and of the anchor generator).  Rules restated from the reference:
//   mq[k][r] = IoU(gt k, anchor r) (0 where the intersection is empty)
//   per anchor: best = max_k mq, match = first arg max; label = ignore if best > threshold else negative
//   per gt k: among the anchors with mq[k][r] == max_r mq[k][.], the one with the smallest L1 centre distance
//             (first on ties) becomes positive and is matched to k (highest k wins when two gts pick one anchor,
//             which is what the reference's sequential index assignment does on the CPU)
//   conf loss = mean BCE(sigmoid(t_obj)[pos], 1) * w_obj + mean BCE(sigmoid(t_obj)[neg], 0) * w_noobj
//   class loss = mean over pos x C of BCE(sigmoid(t_cls), one-hot);  box loss = mean over pos x 4 of smooth-L1
//   (beta 1) between (sigmoid(tx) - 0.5, sigmoid(ty) - 0.5, tw, th) and get_deltas(gt, anchor, stride)
//   BCE on probabilities with torch's clamps: log >= -100 forward, p(1-p) >= 1e-12 in the backward quotient.
// All sums are formed per workgroup and folded in a fixed order in double precision: results do not depend on
// scheduling.  An empty positive (or negative) set gives NaN, as torch's mean over an empty tensor does.
#include "common.h"

// products and sums round separately, as the torch ops of the reference do (exact-equality tests on IoUs below)
#pragma clang fp contract(off)

namespace yolo {

constexpr int MAXL = 4;

struct LossLevels {
    const char* head[MAXL];   // f32 NHWC head tensors
    char* dhead[MAXL];        // gradient buffers (bf16 or f32)
    long long h_img[MAXL], h_origin[MAXL], d_img[MAXL], d_origin[MAXL];   // bytes
    int h_row[MAXL], h_pix[MAXL], d_row[MAXL], d_pix[MAXL];               // bytes
    int H[MAXL], W[MAXL], stride[MAXL], first[MAXL + 1];  // first anchor index of the level; first[L] = R
    int d_ch[MAXL];           // channels of the gradient slice to write (>= A*(5+C); the rest is zero)
    int L;
};

struct LossParams {
    LossLevels lv;
    const float* anchors;  // [R][4]
    const float* gt_boxes;  // [N][K][4]
    const int* gt_classes;  // [N][K]
    const int* gt_count;    // [N]
    int N, K, R, A, C;
    float thr, w_obj, w_noobj;
    int* pos;        // [N][K] anchor picked by each gt (-1: none)
    signed char* label;  // [N][R]  -1 ignore, 0 negative, 1 positive
    short* match;    // [N][R]  matched gt
    int* counts;     // [2] positives, negatives
    double* partial; // [blocks][4] obj, noobj, cls, box
    int partial_blocks;
    float* losses;   // [3] conf, cls, box
    const float* gscale;  // [3] upstream gradients of the three losses (backward)
};

__device__ __forceinline__ float iou_gt_anchor(const float* g, const float* a) {
    const float area_g = (g[2] - g[0]) * (g[3] - g[1]);
    const float area_a = (a[2] - a[0]) * (a[3] - a[1]);
    float w = fminf(g[2], a[2]) - fmaxf(g[0], a[0]);
    float h = fminf(g[3], a[3]) - fmaxf(g[1], a[1]);
    w = w > 0.f ? w : 0.f;
    h = h > 0.f ? h : 0.f;
    const float inter = w * h;
    return inter > 0.f ? inter / (area_g + area_a - inter) : 0.f;
}

__device__ __forceinline__ float sigmoidf_(float t) { return 1.f / (1.f + expf(-t)); }

// One workgroup per (image, gt): max IoU over the anchors, then the closest centre among the anchors that reach it.
__global__ void __launch_bounds__(256) loss_gt_pick_kernel(const LossParams p) {
    __shared__ float sbest[256];
    __shared__ float sdist[256];
    __shared__ int sidx[256];
    const int n = blockIdx.x / p.K, k = blockIdx.x - n * p.K;
    const int tid = threadIdx.x;
    if (k >= p.gt_count[n]) {
        if (tid == 0) p.pos[blockIdx.x] = -1;
        return;
    }
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = p.gt_boxes[((long long)n * p.K + k) * 4 + e];
    float best = -1.f;
    for (int r = tid; r < p.R; r += 256) best = fmaxf(best, iou_gt_anchor(g, p.anchors + 4 * r));
    sbest[tid] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sbest[tid] = fmaxf(sbest[tid], sbest[tid + s]);
        __syncthreads();
    }
    best = sbest[0];
    const float gcx = (g[0] + g[2]) / 2, gcy = (g[1] + g[3]) / 2;
    float dmin = INFINITY;
    int imin = 0x7fffffff;
    for (int r = tid; r < p.R; r += 256) {   // ascending r per thread: the first minimum is kept
        const float* a = p.anchors + 4 * r;
        if (iou_gt_anchor(g, a) == best) {
            const float d = fabsf(gcx - (a[0] + a[2]) / 2) + fabsf(gcy - (a[1] + a[3]) / 2);
            if (d < dmin) {
                dmin = d;
                imin = r;
            }
        }
    }
    sdist[tid] = dmin;
    sidx[tid] = imin;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float d2 = sdist[tid + s];
            const int i2 = sidx[tid + s];
            if (d2 < sdist[tid] || (d2 == sdist[tid] && i2 < sidx[tid])) {
                sdist[tid] = d2;
                sidx[tid] = i2;
            }
        }
        __syncthreads();
    }
    if (tid == 0) p.pos[blockIdx.x] = sidx[0] == 0x7fffffff ? -1 : sidx[0];
}

// One thread per (image, anchor): label and matched gt; positives / negatives counted per workgroup.
__global__ void __launch_bounds__(256) loss_label_kernel(const LossParams p) {
    __shared__ int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long long)p.N * p.R) {
        const int n = (int)(i / p.R), r = (int)(i - (long long)n * p.R);
        const int kc = p.gt_count[n];
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = p.anchors[4 * r + e];
        float best = -1.f;
        int m = 0;
        for (int k = 0; k < kc; ++k) {
            const float q = iou_gt_anchor(p.gt_boxes + ((long long)n * p.K + k) * 4, a);
            if (q > best) {
                best = q;
                m = k;
            }
        }
        int lab = best > p.thr ? -1 : 0;
        for (int k = kc - 1; k >= 0; --k)
            if (p.pos[n * p.K + k] == r) {
                lab = 1;
                m = k;
                break;
            }
        p.label[i] = (signed char)lab;
        p.match[i] = (short)m;
        if (lab == 1) atomicAdd(&cnt[0], 1);
        if (lab == 0) atomicAdd(&cnt[1], 1);
    }
    __syncthreads();
    if (threadIdx.x < 2 && cnt[threadIdx.x]) atomicAdd(&p.counts[threadIdx.x], cnt[threadIdx.x]);
}

struct AnchorSite {
    int l, y, x, a;
};

__device__ __forceinline__ AnchorSite site_of(const LossLevels& lv, int r, int A) {
    int l = 0;
    while (l + 1 < lv.L && r >= lv.first[l + 1]) ++l;
    const int q = r - lv.first[l];
    const int cell = q / A;
    AnchorSite s;
    s.l = l;
    s.a = q - cell * A;
    s.y = cell / lv.W[l];
    s.x = cell - s.y * lv.W[l];
    return s;
}

__device__ __forceinline__ void box_targets(const float* g, const float* a, float stride, float t[4]) {
    const float gw = g[2] - g[0], gh = g[3] - g[1];
    const float cx = (g[0] + gw / 2) / stride, cy = (g[1] + gh / 2) / stride;
    t[0] = cx - floorf(cx) - 0.5f;
    t[1] = cy - floorf(cy) - 0.5f;
    t[2] = logf(gw / (a[2] - a[0]) + 1e-16f);
    t[3] = logf(gh / (a[3] - a[1]) + 1e-16f);
}

__device__ __forceinline__ float bce(float pr, float y) {
    const float lp = fmaxf(logf(pr), -100.f), lq = fmaxf(logf(1.f - pr), -100.f);
    return -(y * lp + (1.f - y) * lq);
}

// d/dt of BCE(sigmoid(t), y) as torch's two backward steps compute it
__device__ __forceinline__ float bce_grad(float pr, float y) {
    const float v = (1.f - pr) * pr;
    return (pr - y) / fmaxf(v, 1e-12f) * v;
}

// One thread per (image, anchor): the anchor's terms of the four sums.
__global__ void __launch_bounds__(256) loss_value_kernel(const LossParams p) {
    __shared__ double red[4][4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (i < (long long)p.N * p.R) {
        const int lab = p.label[i];
        if (lab >= 0) {
            const int n = (int)(i / p.R), r = (int)(i - (long long)n * p.R);
            const AnchorSite st = site_of(p.lv, r, p.A);
            const int K1 = 5 + p.C;
            const float* h = (const float*)(p.lv.head[st.l] + p.lv.h_origin[st.l] + n * p.lv.h_img[st.l] +
                                            (long long)st.y * p.lv.h_row[st.l] + (long long)st.x * p.lv.h_pix[st.l]) + st.a * K1;
            const float pc = sigmoidf_(h[4]);
            if (lab == 0) {
                s[1] = bce(pc, 0.f);
            } else {
                s[0] = bce(pc, 1.f);
                const int k = p.match[i];
                const int cls = p.gt_classes[n * p.K + k];
                float acc = 0.f;
                for (int c = 0; c < p.C; ++c) acc += bce(sigmoidf_(h[5 + c]), c == cls ? 1.f : 0.f);
                s[2] = acc;
                float t[4];
                box_targets(p.gt_boxes + ((long long)n * p.K + k) * 4, p.anchors + 4 * r, (float)p.lv.stride[st.l], t);
                float b = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pred = j < 2 ? sigmoidf_(h[j]) - 0.5f : h[j];
                    const float d = fabsf(pred - t[j]);
                    b += d < 1.f ? 0.5f * d * d : d - 0.5f;
                }
                s[3] = b;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double v = s[e];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        p.partial[(long long)blockIdx.x * 4 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void __launch_bounds__(256) loss_finish_kernel(const LossParams p) {
    __shared__ double red[256][4];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < p.partial_blocks; b += 256)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += p.partial[(long long)b * 4 + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[threadIdx.x][e] += red[threadIdx.x + st][e];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double np = (double)p.counts[0], nn = (double)p.counts[1];
        p.losses[0] = (float)(red[0][0] / np) * p.w_obj + (float)(red[0][1] / nn) * p.w_noobj;
        p.losses[1] = (float)(red[0][2] / (np * p.C));
        p.losses[2] = (float)(red[0][3] / (np * 4.0));
    }
}

// One thread per (image, pixel, 8 channels) of a level: the gradient of gscale . losses w.r.t. the head tensor,
// zeros included, so the buffer needs no clearing between steps.
template <typename T>
__global__ void __launch_bounds__(256) loss_grad_kernel(const LossParams p, int l, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int G = p.lv.d_ch[l] / 8;
    const int g = (int)(i % G);
    const long long m = i / G;
    const int HW = p.lv.H[l] * p.lv.W[l];
    const int n = (int)(m / HW);
    const int cell = (int)(m - (long long)n * HW);
    const int y = cell / p.lv.W[l], x = cell - y * p.lv.W[l];
    const int K1 = 5 + p.C, AK = p.A * K1;
    const float np = (float)p.counts[0], nn = (float)p.counts[1];
    const float s_obj = p.gscale[0] * p.w_obj / np, s_noobj = p.gscale[0] * p.w_noobj / nn;
    const float s_cls = p.gscale[1] / (np * (float)p.C), s_box = p.gscale[2] / (np * 4.f);
    const float* h = (const float*)(p.lv.head[l] + p.lv.h_origin[l] + n * p.lv.h_img[l] + (long long)y * p.lv.h_row[l] +
                                    (long long)x * p.lv.h_pix[l]);
    float o[8];
    int c = g * 8;
    int a = c / K1, j = c - a * K1;
    int cur_a = -1, lab = 0, cls = 0;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e, ++c) {
        float v = 0.f;
        if (c < AK) {
            if (a != cur_a) {
                cur_a = a;
                const int r = p.lv.first[l] + cell * p.A + a;
                const long long li = (long long)n * p.R + r;
                lab = p.label[li];
                if (lab == 1) {
                    const int k = p.match[li];
                    cls = p.gt_classes[n * p.K + k];
                    box_targets(p.gt_boxes + ((long long)n * p.K + k) * 4, p.anchors + 4 * r, (float)p.lv.stride[l], t);
                }
            }
            if (lab == 0) {
                if (j == 4) v = s_noobj * bce_grad(sigmoidf_(h[c]), 0.f);
            } else if (lab == 1) {
                const float hv = h[c];
                if (j < 4) {
                    const float pr = sigmoidf_(hv);
                    const float d = (j < 2 ? pr - 0.5f : hv) - t[j];
                    const float sl = d < -1.f ? -1.f : (d > 1.f ? 1.f : d);   // smooth-L1', beta = 1
                    v = s_box * sl * (j < 2 ? (1.f - pr) * pr : 1.f);
                } else if (j == 4) {
                    v = s_obj * bce_grad(sigmoidf_(hv), 1.f);
                } else {
                    v = s_cls * bce_grad(sigmoidf_(hv), j - 5 == cls ? 1.f : 0.f);
                }
            }
            if (++j == K1) {
                j = 0;
                ++a;
            }
        }
        o[e] = v;
    }
    char* dst = p.lv.dhead[l] + p.lv.d_origin[l] + n * p.lv.d_img[l] + (long long)y * p.lv.d_row[l] + (long long)x * p.lv.d_pix[l] +
                (long long)g * 8 * (int)sizeof(T);
    if constexpr (sizeof(T) == 2) {
        const u32x4 r = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        *(u32x4*)dst = r;
    } else {
        *(f32x4*)dst = f32x4{o[0], o[1], o[2], o[3]};
        *(f32x4*)(dst + 16) = f32x4{o[4], o[5], o[6], o[7]};
    }
}

struct LossWs {
    size_t pos, label, match, counts, partial, total;
    int blocks;
};

static LossWs loss_ws(int N, int K, int R) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    LossWs w;
    const size_t NR = (size_t)N * R;
    w.blocks = (int)((NR + 255) / 256);
    w.pos = 0;
    w.label = up(w.pos + (size_t)N * K * 4);
    w.match = up(w.label + NR);
    w.counts = up(w.match + NR * 2);
    w.partial = up(w.counts + 16);
    w.total = up(w.partial + (size_t)w.blocks * 4 * 8);
    return w;
}

static int fill_loss(LossParams& p, const yolo_view* head, const yolo_view* dhead, int dhead_dtype, int L, const int* stride_host,
                     int N, int A, int C, void* workspace, size_t ws_bytes, int K, int R) {
    YOLO_REQUIRE(L >= 1 && L <= MAXL, "yolo_loss: %d levels (1..%d)", L, MAXL);
    YOLO_REQUIRE(N > 0 && A > 0 && C > 0 && K > 0 && K < 32768, "yolo_loss: bad N/A/C/K");
    const int AK = A * (5 + C);
    int first = 0;
    for (int l = 0; l < L; ++l) {
        YOLO_REQUIRE(head[l].ptr && head[l].ch_offset + AK <= head[l].ch_total, "yolo_loss: head %d has too few channels", l);
        const ViewGeom g = geom_of(head[l]);
        p.lv.head[l] = (const char*)head[l].ptr;
        p.lv.h_img[l] = g.img_stride * 4;
        p.lv.h_row[l] = g.row_stride * 4;
        p.lv.h_pix[l] = g.pix_stride * 4;
        p.lv.h_origin[l] = (long long)g.origin * 4;
        p.lv.H[l] = head[l].H;
        p.lv.W[l] = head[l].W;
        p.lv.stride[l] = stride_host[l];
        p.lv.first[l] = first;
        first += head[l].H * head[l].W * A;
        if (dhead) {
            const int es = (int)elem_size(dhead_dtype);
            YOLO_REQUIRE(dhead[l].ptr && dhead[l].H == head[l].H && dhead[l].W == head[l].W, "yolo_loss: gradient view %d", l);
            const int dch = dhead[l].ch_total - dhead[l].ch_offset;
            YOLO_REQUIRE(dch >= AK && dch % 8 == 0 && dhead[l].ch_offset % 8 == 0, "yolo_loss: gradient view %d channels", l);
            const ViewGeom d = geom_of(dhead[l]);
            p.lv.dhead[l] = (char*)dhead[l].ptr;
            p.lv.d_img[l] = d.img_stride * es;
            p.lv.d_row[l] = d.row_stride * es;
            p.lv.d_pix[l] = d.pix_stride * es;
            p.lv.d_origin[l] = (long long)d.origin * es;
            p.lv.d_ch[l] = dch;
        }
    }
    p.lv.first[L] = first;
    p.lv.L = L;
    YOLO_REQUIRE(first == R, "yolo_loss: %d anchors in the head views, %d in the anchor table", first, R);
    const LossWs w = loss_ws(N, K, R);
    YOLO_REQUIRE(workspace && ws_bytes >= w.total, "yolo_loss: workspace %zu < %zu bytes", ws_bytes, w.total);
    char* ws = (char*)workspace;
    p.pos = (int*)(ws + w.pos);
    p.label = (signed char*)(ws + w.label);
    p.match = (short*)(ws + w.match);
    p.counts = (int*)(ws + w.counts);
    p.partial = (double*)(ws + w.partial);
    p.partial_blocks = w.blocks;
    p.N = N;
    p.K = K;
    p.R = R;
    p.A = A;
    p.C = C;
    return YOLO_OK;
}

}  // namespace yolo

using namespace yolo;

extern "C" size_t yolo_train_loss_workspace_bytes(int N, int K, int R) {
    if (N <= 0 || K <= 0 || R <= 0) return 0;
    return loss_ws(N, K, R).total;
}

extern "C" int yolo_train_loss_fwd(const yolo_view* head, int L, const int* stride_host, const float* anchors, int R, int N,
                                   int A, int C, const float* gt_boxes, const int* gt_classes, const int* gt_count, int K,
                                   float match_thr, float obj_weight, float noobj_weight, float* losses, int* counts_out,
                                   void* workspace, size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(head && stride_host && anchors && gt_boxes && gt_classes && gt_count && losses, "train_loss_fwd: null argument");
    LossParams p{};
    if (int e = fill_loss(p, head, nullptr, 0, L, stride_host, N, A, C, workspace, ws_bytes, K, R)) return e;
    p.anchors = anchors;
    p.gt_boxes = gt_boxes;
    p.gt_classes = gt_classes;
    p.gt_count = gt_count;
    p.thr = match_thr;
    p.w_obj = obj_weight;
    p.w_noobj = noobj_weight;
    p.losses = losses;
    hipStream_t st = (hipStream_t)stream;
    YOLO_CHECK_HIP(hipMemsetAsync(p.counts, 0, 16, st));
    hipLaunchKernelGGL(loss_gt_pick_kernel, dim3(N * K), dim3(256), 0, st, p);
    hipLaunchKernelGGL(loss_label_kernel, dim3(p.partial_blocks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(loss_value_kernel, dim3(p.partial_blocks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, p);
    if (counts_out) YOLO_CHECK_HIP(hipMemcpyAsync(counts_out, p.counts, 8, hipMemcpyDeviceToDevice, st));
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_train_loss_bwd(const yolo_view* head, const yolo_view* dhead, int dhead_dtype, int L, const int* stride_host,
                                   const float* anchors, int R, int N, int A, int C, const float* gt_boxes,
                                   const int* gt_classes, const int* gt_count, int K, float obj_weight, float noobj_weight,
                                   const float* grad_losses, void* workspace, size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(head && dhead && stride_host && anchors && gt_boxes && gt_classes && gt_count && grad_losses,
                 "train_loss_bwd: null argument");
    YOLO_REQUIRE(dhead_dtype == YOLO_F32 || dhead_dtype == YOLO_BF16, "train_loss_bwd: dtype %d", dhead_dtype);
    LossParams p{};
    if (int e = fill_loss(p, head, dhead, dhead_dtype, L, stride_host, N, A, C, workspace, ws_bytes, K, R)) return e;
    p.anchors = anchors;
    p.gt_boxes = gt_boxes;
    p.gt_classes = gt_classes;
    p.gt_count = gt_count;
    p.w_obj = obj_weight;
    p.w_noobj = noobj_weight;
    p.gscale = grad_losses;
    hipStream_t st = (hipStream_t)stream;
    for (int l = 0; l < L; ++l) {
        const long long total = (long long)N * p.lv.H[l] * p.lv.W[l] * (p.lv.d_ch[l] / 8);
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (dhead_dtype == YOLO_BF16)
            hipLaunchKernelGGL((loss_grad_kernel<unsigned short>), dim3(blocks), dim3(256), 0, st, p, l, total);
        else
            hipLaunchKernelGGL((loss_grad_kernel<float>), dim3(blocks), dim3(256), 0, st, p, l, total);
    }
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

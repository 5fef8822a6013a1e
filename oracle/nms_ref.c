/*
 * Scalar C restatement of the NMS the reference reaches through
 *   yolo/modeling/meta_arch/yolo.py:305  batched_nms(boxes, scores, classes, thr)
 *     -> detectron2.layers.batched_nms -> torchvision.ops.boxes.batched_nms -> torchvision::nms (CPU).
 *
 * TEST INFRASTRUCTURE ONLY: linked by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * PARITY UNPINNED: torchvision is not installed in the build container and the reference holds no
 * NMS vectors; this follows torchvision's published CPU algorithm: stable descending score sort,
 * greedy sweep, fp32 IoU = inter / (area_i + area_j - inter) with std::max / std::min comparison
 * semantics, suppression when (double)iou > iou_threshold (strict).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float smaxf(float a, float b) { return (a < b) ? b : a; } /* std::max(a,b) */
static inline float sminf(float a, float b) { return (b < a) ? b : a; } /* std::min(a,b) */

/* stable merge sort of idx[0..n) by key descending */
static void msort_desc(int64_t *idx, int64_t *tmp, const float *key, int64_t n) {
    if (n < 2) return;
    int64_t h = n / 2;
    msort_desc(idx, tmp, key, h);
    msort_desc(idx + h, tmp, key, n - h);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take from the right run only when strictly greater: equal keys keep input order */
        if (key[idx[j]] > key[idx[i]]) tmp[k++] = idx[j++];
        else tmp[k++] = idx[i++];
    }
    while (i < h) tmp[k++] = idx[i++];
    while (j < n) tmp[k++] = idx[j++];
    memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
}

/* torchvision::nms CPU kernel.  boxes [n,4] xyxy fp32.  keep receives indices in descending score
 * order; returns their count. */
int64_t oracle_nms(const float *boxes, const float *scores, int64_t n, double iou_thr, int64_t *keep) {
    if (n <= 0) return 0;
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)n * 2);
    float *areas = (float *)malloc(sizeof(float) * (size_t)n);
    unsigned char *sup = (unsigned char *)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; ++i) {
        order[i] = i;
        areas[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    }
    msort_desc(order, order + n, scores, n);
    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (sup[j]) continue;
            float xx1 = smaxf(ix1, boxes[4 * j]);
            float yy1 = smaxf(iy1, boxes[4 * j + 1]);
            float xx2 = sminf(ix2, boxes[4 * j + 2]);
            float yy2 = sminf(iy2, boxes[4 * j + 3]);
            float w = smaxf(0.0f, xx2 - xx1);
            float h = smaxf(0.0f, yy2 - yy1);
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if ((double)ovr > iou_thr) sup[j] = 1;
        }
    }
    free(order);
    free(areas);
    free(sup);
    return nk;
}

/* torchvision _batched_nms_vanilla: per class on raw boxes, result in stable descending score order. */
int64_t oracle_batched_nms_vanilla(const float *boxes, const float *scores, const int64_t *cls, int64_t n,
                                   double iou_thr, int64_t *keep) {
    if (n <= 0) return 0;
    int64_t maxc = 0;
    for (int64_t i = 0; i < n; ++i)
        if (cls[i] > maxc) maxc = cls[i];
    unsigned char *mask = (unsigned char *)calloc((size_t)n, 1);
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)n * 2);
    float *cb = (float *)malloc(sizeof(float) * (size_t)n * 4);
    float *cs = (float *)malloc(sizeof(float) * (size_t)n);
    int64_t *ck = cur + n;
    for (int64_t c = 0; c <= maxc; ++c) {
        int64_t m = 0;
        for (int64_t i = 0; i < n; ++i)
            if (cls[i] == c) {
                cur[m] = i;
                memcpy(cb + 4 * m, boxes + 4 * i, 16);
                cs[m] = scores[i];
                ++m;
            }
        if (!m) continue;
        int64_t k = oracle_nms(cb, cs, m, iou_thr, ck);
        for (int64_t t = 0; t < k; ++t) mask[cur[ck[t]]] = 1;
    }
    int64_t nk = 0;
    for (int64_t i = 0; i < n; ++i)
        if (mask[i]) keep[nk++] = i;
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nk ? nk : 1));
    msort_desc(keep, tmp, scores, nk);
    free(tmp);
    free(mask);
    free(cur);
    free(cb);
    free(cs);
    return nk;
}

/* torchvision _batched_nms_coordinate_trick: boxes + cls * (max_coordinate + 1), one NMS. */
int64_t oracle_batched_nms_trick(const float *boxes, const float *scores, const int64_t *cls, int64_t n,
                                 double iou_thr, int64_t *keep) {
    if (n <= 0) return 0;
    float maxc = boxes[0];
    for (int64_t i = 1; i < 4 * n; ++i)
        maxc = (boxes[i] > maxc) ? boxes[i] : maxc; /* boxes.max(); NaN-free inputs assumed */
    float step = maxc + 1.0f;
    float *sb = (float *)malloc(sizeof(float) * (size_t)n * 4);
    for (int64_t i = 0; i < n; ++i) {
        float off = (float)cls[i] * step;
        for (int k = 0; k < 4; ++k) sb[4 * i + k] = boxes[4 * i + k] + off;
    }
    int64_t nk = oracle_nms(sb, scores, n, iou_thr, keep);
    free(sb);
    return nk;
}

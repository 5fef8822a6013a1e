// Batched per-class NMS for gfx950 and the detection gather / post-process.
//
// Replaces batched_nms(boxes, scores, classes, thr) at yolo/modeling/meta_arch/yolo.py:305
// (detectron2.layers.batched_nms -> torchvision.ops.boxes.batched_nms -> torchvision::nms), the
// keep[:100] truncation at :307 and detectron2's detector_postprocess, for all images of a batch.
//
// Pipeline (all on the caller's stream; the candidate counts are read ON THE DEVICE from img_offsets, the host only
// passes an upper bound, so nothing is read back and nothing synchronises):
//   1. sort kernel, one workgroup per image: a stable radix sort (8-bit digits, counting-sort passes whose ranks come
//      from wave ballots) of the image's candidates by descending score -- "order B", the order torchvision returns
//      kept indices in (ties: ascending index) -- then a stable radix sort of order B by class -- "order A", where
//      every (image, class) segment lists its boxes in descending-score order.  Passes whose digit is the same for
//      every key are skipped.  The same kernel lists the segment starts, takes the image's largest coordinate
//      (coordinate trick) and clears the kept flags.
//   2. sweep kernel, one wave per segment: greedy sweep in chunks of 64 boxes (lane = box); a chunk is first tested
//      against the boxes already kept (kept list in LDS, spilling to global memory), then resolved internally with
//      wave ballots + lane broadcasts.  IoU arithmetic is fp32 in torchvision's operation order; the file is built
//      with -ffp-contract=off.
//   3. emit kernel, one workgroup per image: walks order B, ranks the kept flags and emits the first top_k.
//   With top_k > 0 and no class of the image above 512 candidates, 1-3 take a shorter form: the sort kernel only groups
//   the candidates by class (counting sort through LDS atomics), the sweeping wave puts its segment into (score,
//   candidate index) order itself before the sweep and leaves its kept candidates as a list, and the emit kernel merges
//   the image's lists up to top_k.  Same kept indices in the same order.
//
// The coordinate-trick branch of torchvision (boxes shifted by class * (max_coordinate + 1) before a single
// class-agnostic NMS) is reproduced by applying the same fp32 shift to the boxes of each (image, class) segment.
// Per-segment sweeps equal the class-agnostic sweep as long as shifted boxes of different classes cannot suppress
// each other.  They cannot when no coordinate of the image is negative (the shifted classes occupy disjoint squares);
// and with negative corners they cannot when every box centre is non-negative and iou_thr >= 0.25: the overlap of a
// class-c box with a class-(c-1) box lies in the part of the former left of AND above its class origin, less than a
// quarter of its area, so the IoU stays below 0.25.  (The decode kernel's boxes always have their centres inside the
// frame.)  An image that meets neither condition is swept as ONE segment in score order with per-box shifts: the
// class-agnostic sweep itself, exact but slower.
#include <cstring>
#include <mutex>

#include "common.h"

namespace yolo {

struct NmsParams {
    const float* boxes;
    const float* scores;
    const int* classes;
    const int* img_offsets;
    int N, cap;  // cap: candidates that exist in the arrays (img_offsets beyond it are clamped)
    double iou_thr;
    int mode, auto_limit, top_k, cbits;
    unsigned *sort_k0, *sort_k1, *sort_v0, *sort_v1;  // [cap] each: (key, value) pairs of the radix passes, ping-pong
    unsigned* valB;            // [cap] order B: original index
    unsigned* keyA;            // [cap] (image << cbits) | class in order A
    unsigned* valA;            // [cap] order A: position in order B
    unsigned* seg_start;       // [cap]
    unsigned* nseg;
    float* img_max;            // [N]
    unsigned char* agnostic;   // [N] image swept as one class-agnostic segment
    unsigned char* keepB;      // [cap] kept flag by position in order B
    f32x4* kept_spill;         // [cap] kept boxes beyond the LDS list, addressed by segment start
    int* keep;
    int keep_stride;
    int* n_keep;
};

__device__ __forceinline__ int img_begin(const NmsParams& p, int n) {
    const int b = p.img_offsets[n];
    return b < p.cap ? b : p.cap;
}

constexpr int SORT_THREADS = 1024;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int RADIX_BITS = 8, RADIX_BINS = 1 << RADIX_BITS;

constexpr int SORT_E = 4;                          // pairs per thread and tile
constexpr int SORT_TILE = SORT_E * SORT_THREADS;   // 4096 pairs: sub-tile e holds pairs tile + e * 1024 + thread

struct SortShared {
    unsigned hist[4][RADIX_BINS];                   // digit histograms of up to four passes, taken in one sweep
    unsigned base[RADIX_BINS];                      // the running output position of every digit
    unsigned tile_total[RADIX_BINS];                // keys of every digit in the current tile
    // per (sub-tile, wave), per digit: count in the tile -> exclusive prefix in (sub-tile, wave) order
    unsigned short wave_count[SORT_E * SORT_WAVES][RADIX_BINS];
    unsigned tile_first[RADIX_BINS];                // position of every digit's first pair inside the reordered tile
    unsigned digit_wave_sum[RADIX_BINS / 64];       // (per 64 digits; tile_first is relative to its group of 64)
    unsigned tile_k[SORT_TILE], tile_v[SORT_TILE];  // the tile reordered by digit, so that runs leave as contiguous stores
    int uniform_digit;                              // every key has the same digit: the pass moves nothing
};

// One stable counting-sort pass of m (key, value) pairs on digit `pass` (bits 8 * pass ..) by ONE workgroup, from the
// digit's histogram in sh.hist[pass]: exclusive scan, then tile after tile (4096 consecutive pairs as four sub-tiles of 1024, one pair per thread and
// sub-tile, all loaded up front) every pair goes to
//   base[digit] + (pairs of that digit in lower (sub-tile, wave) groups of the tile) + (... in lower lanes of its wave),
// the last two from wave ballots -- no atomics decide an order, so equal digits keep their input order.
// vin == nullptr: the values are the input positions.  Returns false (and moves nothing) when all digits are equal.
// Histograms of the 8-bit digits at shifts 0, 8, ... (npass of them) of keys[0..m) in one sweep, four loads in flight
// per thread (a workgroup on its own is bound by load latency, not by bytes).
__device__ void radix_histograms(const unsigned* keys, int m, int npass, SortShared& sh) {
    const int tid = threadIdx.x;
    for (int t = tid; t < 4 * RADIX_BINS; t += SORT_THREADS) (&sh.hist[0][0])[t] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < m; i0 += SORT_TILE) {
        unsigned k[SORT_E];
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int i = i0 + e * SORT_THREADS + tid;
            k[e] = keys[i < m ? i : m - 1];
        }
#pragma unroll
        for (int e = 0; e < SORT_E; ++e)
            if (i0 + e * SORT_THREADS + tid < m)
                for (int ps = 0; ps < npass; ++ps) atomicAdd(&sh.hist[ps][(k[e] >> (ps * RADIX_BITS)) & (RADIX_BINS - 1)], 1u);
    }
    __syncthreads();
}

__device__ bool radix_pass(const unsigned* kin, const unsigned* vin, unsigned* kout, unsigned* vout, int m, int pass,
                           SortShared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int shift = pass * RADIX_BITS;
    if (tid == 0) sh.uniform_digit = 0;
    __syncthreads();
    if (tid < RADIX_BINS && sh.hist[pass][tid] == (unsigned)m) sh.uniform_digit = 1;
    __syncthreads();
    const bool uniform = sh.uniform_digit != 0;
    __syncthreads();  // every thread has read the flag before the next pass's thread 0 clears it
    if (uniform) return false;
    // exclusive scan of the 256 counts (4 waves)
    unsigned mine = 0, incl = 0;
    if (tid < RADIX_BINS) {
        mine = sh.hist[pass][tid];
        incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) sh.tile_total[wave] = incl;  // (scratch use: wave sums)
    }
    __syncthreads();
    if (tid < RADIX_BINS) {
        unsigned pre = 0;
        for (int w = 0; w < wave; ++w) pre += sh.tile_total[w];
        sh.base[tid] = pre + incl - mine;
    }
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    // Inside the loop the barriers order LDS traffic only (raw s_barrier behind a wait for the LDS counter): a
    // __syncthreads() would also wait for every global store of the tile before, and for the next tile's loads, which
    // are requested a whole tile ahead.  Global data is only handed over between passes (the barrier at the end).
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    unsigned kn[SORT_E], vn[SORT_E];
    auto request = [&](int t0) {  // all loads of a tile in flight together
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int i = t0 + e * SORT_THREADS + tid;
            const int ic = i < m ? i : m - 1;
            kn[e] = kin[ic];
            vn[e] = vin ? vin[ic] : (unsigned)ic;
        }
    };
    request(0);
    for (int t0 = 0; t0 < m; t0 += SORT_TILE) {
        unsigned k[SORT_E], v[SORT_E];
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            k[e] = kn[e];
            v[e] = vn[e];
        }
        lds_barrier();  // base[] is final for this tile; the last tile's wave_count reads are done
        {
            unsigned* z = (unsigned*)&sh.wave_count[0][0];  // SORT_E * 16 * 256 u16 = SORT_E * 2048 u32
#pragma unroll
            for (int e = 0; e < 2 * SORT_E; ++e)
                if (e < 2 * ((m - t0 + SORT_THREADS - 1) / SORT_THREADS)) z[tid + e * SORT_THREADS] = 0;
        }
        lds_barrier();
        int rank[SORT_E];
        unsigned dg[SORT_E];
        // sub-tiles in use (wave-uniform): a small image fills only the first one
        const int ne = (m - t0 + SORT_THREADS - 1) / SORT_THREADS < SORT_E ? (m - t0 + SORT_THREADS - 1) / SORT_THREADS : SORT_E;
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const bool valid = t0 + e * SORT_THREADS + tid < m;
            const unsigned d = (k[e] >> shift) & (RADIX_BINS - 1);
            dg[e] = d;
            rank[e] = -1;
            if (e >= ne) continue;
            unsigned long long same = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < RADIX_BITS; ++bit) {
                const bool set = (d >> bit) & 1u;
                const unsigned long long bal = __ballot(set);
                same &= set ? bal : ~bal;
            }
            rank[e] = valid ? __popcll(same & lt) : -1;
            if (valid && rank[e] == 0) sh.wave_count[e * SORT_WAVES + wave][d] = (unsigned short)__popcll(same);
        }
        lds_barrier();
        if (tid < RADIX_BINS) {
            // all counts of the digit into registers first: reads and writes of one array in one loop would be kept
            // in order by the compiler, one LDS round trip per group
            unsigned short c[SORT_E * SORT_WAVES];
#pragma unroll
            for (int w = 0; w < SORT_E * SORT_WAVES; ++w) c[w] = w < ne * SORT_WAVES ? sh.wave_count[w][tid] : (unsigned short)0;
            unsigned run = 0;
#pragma unroll
            for (int w = 0; w < SORT_E * SORT_WAVES; ++w) {
                if (w < ne * SORT_WAVES) sh.wave_count[w][tid] = (unsigned short)run;
                run += c[w];
            }
            sh.tile_total[tid] = run;
            // exclusive scan of the tile's digit counts: where each digit's run starts inside the reordered tile
            unsigned inc = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned t = __shfl_up(inc, d, 64);
                if (lane >= d) inc += t;
            }
            sh.tile_first[tid] = inc - run;  // within this wave of 64 digits; the wave sums are added below
            if (lane == 63) sh.digit_wave_sum[wave] = inc;
        }
        lds_barrier();
        if (t0 + SORT_TILE < m) request(t0 + SORT_TILE);
        // scatter into the reordered tile (LDS), then write the tile out run by run: consecutive lanes store
        // consecutive addresses inside a run instead of 64 unrelated words
        unsigned wave_pre[4];
        wave_pre[0] = 0;
#pragma unroll
        for (int w = 1; w < 4; ++w) wave_pre[w] = wave_pre[w - 1] + sh.digit_wave_sum[w - 1];
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            if (rank[e] >= 0) {
                const unsigned d = dg[e];
                const unsigned lp = wave_pre[d >> 6] + sh.tile_first[d] + sh.wave_count[e * SORT_WAVES + wave][d] + (unsigned)rank[e];
                sh.tile_k[lp] = k[e];
                sh.tile_v[lp] = v[e];
            }
        }
        lds_barrier();
        const int in_tile = m - t0 < SORT_TILE ? m - t0 : SORT_TILE;
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int j = e * SORT_THREADS + tid;
            if (j < in_tile) {
                const unsigned kk = sh.tile_k[j];
                const unsigned d = (kk >> shift) & (RADIX_BINS - 1);
                const unsigned pos = sh.base[d] + ((unsigned)j - (wave_pre[d >> 6] + sh.tile_first[d]));
                kout[pos] = kk;
                vout[pos] = sh.tile_v[j];
            }
        }
        lds_barrier();
        if (tid < RADIX_BINS) sh.base[tid] += sh.tile_total[tid];
    }
    __syncthreads();  // this workgroup's global stores are visible to its own later loads
    return true;
}

constexpr int SORT_SMALL = 1024;     // table size of the counting path (one entry per thread)
constexpr int MERGE_MAX = 1024;     // offers per merge round in the emit kernel
#ifndef NMS_SEGMENT_SORT_MIN
#define NMS_SEGMENT_SORT_MIN 0        // candidates per image from which the grouping-by-class form is used (tunable)
#endif
#ifndef NMS_MERGE_SHARE
#define NMS_MERGE_SHARE 4            // offers per segment and round, in units of a segment's fair share of top_k (tunable)
#endif
constexpr int SEG_SORT_MAX = 512;    // (image, class) segments up to this size are put into score order by their sweeping wave
// values of NmsParams::agnostic
constexpr unsigned char IMG_PER_CLASS = 0, IMG_AGNOSTIC = 1, IMG_SEGMENT_SORTED = 2;
constexpr int SORT_COUNT_MAX = 512;  // candidates per image up to which the orders are found by counting: its work
                                     // is quadratic (3 us at 200, 75 us at 1000; the radix passes take ~30 us)

__global__ void __launch_bounds__(SORT_THREADS) nms_sort_kernel(const NmsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    SortShared& sh = *reinterpret_cast<SortShared*>(sort_lds);
    __shared__ float red_max[SORT_WAVES], red_min[SORT_WAVES], red_cmin[SORT_WAVES];
    __shared__ unsigned seg_count, seg_base, seg_next;
    __shared__ int agnostic_s;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = img_begin(p, n), e = img_begin(p, n + 1);
    const int m = e - b;
    if (m <= 0) return;
    // ---- image statistics for the coordinate trick: max / min coordinate, min centre coordinate
    const bool trick = p.mode == YOLO_NMS_COORD_TRICK || (p.mode == YOLO_NMS_AUTO && 4ll * m <= p.auto_limit);
    if (trick) {
        float mx = -INFINITY, mn = INFINITY, cmn = INFINITY;
        for (int i = tid; i < m; i += SORT_THREADS) {
            const f32x4 bx = *(const f32x4*)(p.boxes + 4ll * (b + i));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mx = bx[k] > mx ? bx[k] : mx;  // NaNs are ignored by the comparison, as by torch's max()
                mn = bx[k] < mn ? bx[k] : mn;
            }
            const float cx = bx[0] + bx[2], cy = bx[1] + bx[3];  // twice the centre
            cmn = cx < cmn ? cx : cmn;
            cmn = cy < cmn ? cy : cmn;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const float o = __shfl_down(mx, d, 64), q = __shfl_down(mn, d, 64), r = __shfl_down(cmn, d, 64);
            mx = o > mx ? o : mx;
            mn = q < mn ? q : mn;
            cmn = r < cmn ? r : cmn;
        }
        if (lane == 0) {
            red_max[wave] = mx;
            red_min[wave] = mn;
            red_cmin[wave] = cmn;
        }
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < SORT_WAVES; ++k) {
                mx = red_max[k] > mx ? red_max[k] : mx;
                mn = red_min[k] < mn ? red_min[k] : mn;
                cmn = red_cmin[k] < cmn ? red_cmin[k] : cmn;
            }
            p.img_max[n] = mx;
            // per-class segments are exact unless a negative corner could reach into the class before (see the top)
            agnostic_s = !(mn >= 0.0f || (cmn >= 0.0f && p.iou_thr >= 0.25));
            p.agnostic[n] = (unsigned char)agnostic_s;
        }
    } else if (tid == 0) {
        agnostic_s = 0;
        p.agnostic[n] = IMG_PER_CLASS;
    }
    if (tid == 0) seg_count = seg_next = 0;
    __syncthreads();   // agnostic_s decides the path below for every thread
    // ---- few candidates (the usual case: a few hundred per image): both orders by counting.  Every thread owns one
    //      candidate and counts the candidates that precede it -- m broadcast reads of LDS per order, no passes, no
    //      scatter: the same two orders as the radix passes below (same keys, same tie-breaks), in a few microseconds.
    if (m <= SORT_COUNT_MAX) {
        unsigned* const key_s = reinterpret_cast<unsigned*>(sort_lds);
        unsigned* const cls_s = key_s + SORT_SMALL;
        unsigned* const rb_s = cls_s + SORT_SMALL;
        unsigned* const ca_s = rb_s + SORT_SMALL;
        const bool mine = tid < m;
        unsigned key = 0, cls = 0;
        if (mine) {
            unsigned u = __float_as_uint(p.scores[b + tid]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            key = ~u;  // ascending key = descending score
            cls = (unsigned)p.classes[b + tid];
            p.keepB[b + tid] = 0;
        }
        // the tables are read four entries per LDS instruction; entries past m never count (largest key / class)
        key_s[tid] = mine ? key : 0xFFFFFFFFu;
        __syncthreads();  // (also: agnostic_s is written)
        if (agnostic_s != 0) cls = 0;
        cls_s[tid] = mine ? cls : 0xFFFFFFFFu;
        const int m4 = (m + 3) & ~3;
        unsigned rb = 0;  // position in order B: (score descending, index ascending)
        if (mine) {
#pragma unroll 4
            for (int j = 0; j < m4; j += 4) {
                const u32x4 k = *(const u32x4*)(key_s + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) rb += (k[e] < key || (k[e] == key && j + e < tid)) ? 1u : 0u;
            }
        }
        rb_s[tid] = mine ? rb : 0xFFFFFFFFu;
        if (mine) p.valB[b + rb] = (unsigned)(b + tid);
        __syncthreads();
        unsigned ra = 0;  // position in order A: (class ascending, order B)
        if (mine) {
#pragma unroll 4
            for (int j = 0; j < m4; j += 4) {
                const u32x4 c = *(const u32x4*)(cls_s + j);
                const u32x4 r = *(const u32x4*)(rb_s + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) ra += (c[e] < cls || (c[e] == cls && r[e] < rb)) ? 1u : 0u;
            }
        }
        if (mine) {
            ca_s[ra] = cls;
            p.keyA[b + ra] = ((unsigned)n << p.cbits) | cls;
            p.valA[b + ra] = (unsigned)b + rb;
        }
        __syncthreads();
        const bool head = mine && (tid == 0 || ca_s[tid - 1] != ca_s[tid]);  // thread t looks at position t of order A
        const unsigned long long hb = __ballot(head);
        if (lane == 0 && hb) atomicAdd(&seg_count, (unsigned)__popcll(hb));
        __syncthreads();
        if (tid == 0) seg_base = atomicAdd(p.nseg, seg_count);
        __syncthreads();
        if (head) p.seg_start[seg_base + atomicAdd(&seg_next, 1u)] = (unsigned)(b + tid);
        return;
    }
    unsigned* kbuf[2] = {p.sort_k0 + b, p.sort_k1 + b};
    unsigned* vbuf[2] = {p.sort_v0 + b, p.sort_v1 + b};
    // ---- top-k, per-class segments of moderate size: the candidates are only GROUPED by class here (a counting sort
    //      through LDS atomics: order inside a class is whatever the atomics give); each segment is put into (score,
    //      candidate index) order by the wave that sweeps it, and the emit kernel merges the segments' kept lists.  The
    //      four passes over the scores -- what a single workgroup per image spends most of its time on when an image
    //      has tens of thousands of candidates -- are not needed for the first top_k.
    if (p.top_k > 0 && p.cbits <= RADIX_BITS && agnostic_s == 0 && m >= NMS_SEGMENT_SORT_MIN) {
        __shared__ unsigned cls_cnt[RADIX_BINS], scan_cnt[SORT_WAVES], scan_seg[SORT_WAVES];
        __shared__ int seg_too_long;
        if (tid < RADIX_BINS) cls_cnt[tid] = 0;
        if (tid == 0) seg_too_long = 0;
        __syncthreads();
        for (int i = tid; i < m; i += SORT_THREADS) atomicAdd(&cls_cnt[(unsigned)p.classes[b + i] & (RADIX_BINS - 1)], 1u);
        __syncthreads();
        if (tid < RADIX_BINS && cls_cnt[tid] > (unsigned)SEG_SORT_MAX) seg_too_long = 1;
        __syncthreads();
        if (seg_too_long == 0) {
            // exclusive scans over the classes: candidates (where a class starts) and non-empty classes (its segment)
            unsigned cnt = 0, incl = 0, has = 0, sincl = 0;
            if (tid < RADIX_BINS) {
                cnt = cls_cnt[tid];
                has = cnt ? 1u : 0u;
                incl = cnt;
                sincl = has;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned t = __shfl_up(incl, d, 64), u = __shfl_up(sincl, d, 64);
                    if (lane >= d) {
                        incl += t;
                        sincl += u;
                    }
                }
                if (lane == 63) {
                    scan_cnt[wave] = incl;
                    scan_seg[wave] = sincl;
                }
            }
            __syncthreads();
            unsigned base = 0, sbase = 0;
            if (tid < RADIX_BINS) {
                for (int w = 0; w < wave; ++w) {
                    base += scan_cnt[w];
                    sbase += scan_seg[w];
                }
                base += incl - cnt;
                sbase += sincl - has;
            }
            if (tid == 0) {
                unsigned nsegs = 0;
                for (int w = 0; w < RADIX_BINS / 64; ++w) nsegs += scan_seg[w];
                seg_count = nsegs;
                seg_base = atomicAdd(p.nseg, nsegs);
            }
            __syncthreads();   // (every count has been read: the table now holds each class's next free position)
            if (tid < RADIX_BINS) {
                cls_cnt[tid] = base;
                if (has) p.seg_start[seg_base + sbase] = (unsigned)b + base;
            }
            __syncthreads();
            const unsigned nkey = (unsigned)n << p.cbits;
            for (int i = tid; i < m; i += SORT_THREADS) {
                const unsigned c = (unsigned)p.classes[b + i];
                const unsigned pos = atomicAdd(&cls_cnt[c & (RADIX_BINS - 1)], 1u);
                p.keyA[b + pos] = nkey | c;
                p.valA[b + pos] = (unsigned)(b + i);   // the candidate itself (there is no order B)
            }
            if (tid == 0) {
                p.agnostic[n] = IMG_SEGMENT_SORTED;
                p.sort_v0[b] = seg_base;       // this image's entries of seg_start: the emit kernel merges them
                p.sort_v0[b + 1] = seg_count;
            }
            return;
        }
    }
    // ---- order B: stable sort by descending score (= by ascending index among equal scores)
    for (int i0 = 0; i0 < m; i0 += SORT_TILE) {
        float sc[SORT_E];
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int i = i0 + e * SORT_THREADS + tid;
            sc[e] = p.scores[b + (i < m ? i : m - 1)];
        }
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int i = i0 + e * SORT_THREADS + tid;
            if (i < m) {
                unsigned u = __float_as_uint(sc[e]);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-orderable
                kbuf[0][i] = ~u;                                  // descending score = ascending key
                p.keepB[b + i] = 0;
            }
        }
    }
    __syncthreads();
    radix_histograms(kbuf[0], m, 4, sh);
    int cur = 0;
    bool have_vals = false;  // until a pass has moved something, the values are the positions themselves
    for (int pass = 0; pass < 4; ++pass) {
        if (radix_pass(kbuf[cur], have_vals ? vbuf[cur] : nullptr, kbuf[cur ^ 1], vbuf[cur ^ 1], m, pass, sh)) {
            cur ^= 1;
            have_vals = true;
        }
    }
    // ---- order A: stable sort of order B by class; a class-agnostic image stays one segment
    const bool agn = agnostic_s != 0;
    const int nxt = cur ^ 1;
    for (int j0 = 0; j0 < m; j0 += SORT_TILE) {
        unsigned orig[SORT_E], cl[SORT_E];
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int j = j0 + e * SORT_THREADS + tid;
            const int jc = j < m ? j : m - 1;
            orig[e] = (unsigned)b + (have_vals ? vbuf[cur][jc] : (unsigned)jc);
        }
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) cl[e] = agn ? 0u : (unsigned)p.classes[orig[e]];
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int j = j0 + e * SORT_THREADS + tid;
            if (j < m) {
                p.valB[b + j] = orig[e];
                kbuf[nxt][j] = cl[e];
            }
        }
    }
    __syncthreads();
    cur = nxt;
    have_vals = false;
    const int cpasses = (p.cbits + RADIX_BITS - 1) / RADIX_BITS;  // <= 2 (at most 65536 classes)
    radix_histograms(kbuf[cur], m, cpasses, sh);
    for (int pass = 0; pass < cpasses; ++pass) {
        if (radix_pass(kbuf[cur], have_vals ? vbuf[cur] : nullptr, kbuf[cur ^ 1], vbuf[cur ^ 1], m, pass, sh)) {
            cur ^= 1;
            have_vals = true;
        }
    }
    // ---- segment keys / positions, segment starts
    const unsigned* ck = kbuf[cur];
    const unsigned nkey = (unsigned)n << p.cbits;
    unsigned heads = 0;
    for (int t0 = 0; t0 < m; t0 += SORT_TILE) {
        unsigned c[SORT_E], cp[SORT_E], pb[SORT_E];
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int t = t0 + e * SORT_THREADS + tid;
            const int tc = t < m ? t : m - 1;
            c[e] = ck[tc];
            cp[e] = tc > 0 ? ck[tc - 1] : ~0u;
            pb[e] = have_vals ? vbuf[cur][tc] : (unsigned)tc;
        }
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int t = t0 + e * SORT_THREADS + tid;
            if (t < m) {
                p.keyA[b + t] = nkey | c[e];
                p.valA[b + t] = (unsigned)b + pb[e];
                heads += cp[e] != c[e] ? 1u : 0u;
            }
        }
    }
    if (heads) atomicAdd(&seg_count, heads);
    __syncthreads();
    if (tid == 0) seg_base = atomicAdd(p.nseg, seg_count);
    __syncthreads();
    for (int t0 = 0; t0 < m; t0 += SORT_TILE) {
        unsigned c[SORT_E], cp[SORT_E];
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int t = t0 + e * SORT_THREADS + tid;
            const int tc = t < m ? t : m - 1;
            c[e] = ck[tc];
            cp[e] = tc > 0 ? ck[tc - 1] : ~0u;
        }
#pragma unroll
        for (int e = 0; e < SORT_E; ++e) {
            const int t = t0 + e * SORT_THREADS + tid;
            if (t < m && cp[e] != c[e]) p.seg_start[seg_base + atomicAdd(&seg_next, 1u)] = (unsigned)(b + t);
        }
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

// Ranks of a segment's candidates by counting, for the wave that owns the segment.  skey[0 .. len4) holds the keys
// (ascending key = descending score; padded to a multiple of 4 with 0xFFFFFFFF), lane l owns the candidates at
// positions l + 64 q, q < NOWN, whose keys are mykey[q].  rank_e = #{j : key_j < key_e}: ONE unsigned compare per pair.
// Candidates with equal keys end with EQUAL ranks, which the caller detects and resolves by candidate index.
template <int NOWN>
__device__ __forceinline__ void rank_by_counting(const unsigned* skey, int len4, const unsigned (&mykey)[SEG_SORT_MAX / 64],
                                                 unsigned (&rank)[SEG_SORT_MAX / 64]) {
#pragma unroll 2
    for (int j = 0; j < len4; j += 4) {
        const u32x4 k = *(const u32x4*)(skey + j);
#pragma unroll
        for (int q = 0; q < NOWN; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rank[q] += k[i] < mykey[q] ? 1u : 0u;
        }
    }
}

constexpr int KEPT_LDS = 448;  // kept boxes per wave held in LDS (7 KiB); the rest spills to global

__global__ void __launch_bounds__(256) nms_sweep_kernel(const NmsParams p) {
    __shared__ f32x4 kept_lds[4][KEPT_LDS];
    __shared__ __attribute__((aligned(16))) unsigned seg_key[4][SEG_SORT_MAX + 4], seg_idx[4][SEG_SORT_MAX];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nseg = *p.nseg;
    const unsigned cmask = (1u << p.cbits) - 1u;
    f32x4* klds = kept_lds[wave];
    for (unsigned seg = blockIdx.x * 4 + wave; seg < nseg; seg += gridDim.x * 4) {
        const unsigned start = p.seg_start[seg];
        const unsigned key = p.keyA[start];
        const int n = (int)(key >> p.cbits);
        const unsigned cls = key & cmask;
        const unsigned img_end = (unsigned)img_begin(p, n + 1);
        const int m_img = (int)img_end - img_begin(p, n);
        const bool trick = p.mode == YOLO_NMS_COORD_TRICK || (p.mode == YOLO_NMS_AUTO && 4ll * m_img <= p.auto_limit);
        const bool agn = p.agnostic[n] == IMG_AGNOSTIC;  // the whole image is this one segment; every box shifts by its own class
        const bool seg_sorted = p.agnostic[n] == IMG_SEGMENT_SORTED;  // grouped by class only: this wave sorts the segment by score
        // torchvision: offsets = idxs.to(boxes) * (max_coordinate + 1); boxes_for_nms = boxes + offsets
        const float unit = p.img_max[n] + 1.0f;
        float shift = trick ? (float)cls * unit : 0.0f;
        f32x4* kspill = p.kept_spill + start;
        int nkept = 0;
        const int kcap = p.top_k > 0 ? p.top_k : 0x7fffffff;
        // ---- a segment that is only grouped by class: this wave puts it into score order first.  Keys (descending score
        //      = ascending key) and candidate indices go to LDS; every lane ranks its candidates by COUNTING the ones
        //      that precede them, reading the keys four per LDS instruction as broadcasts; the candidates are then
        //      written back in rank order (equal scores: by candidate index, torch's stable order).
        unsigned* const skey = seg_key[wave];
        unsigned* const sidx = seg_idx[wave];
        int seg_len = 0;
        if (seg_sorted) {
            // (at most SEG_SORT_MAX candidates: the sort kernel checked; all chunks' loads are issued together: three
            //  round trips to memory for the segment instead of three per chunk)
            constexpr int CH = SEG_SORT_MAX / 64;
            unsigned kq[CH], oq[CH];
            float sq[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const unsigned t = start + 64 * q + lane;
                kq[q] = t < img_end ? p.keyA[t] : ~key;
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const unsigned t = start + 64 * q + lane;
                oq[q] = kq[q] == key ? p.valA[t] : 0u;
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) sq[q] = kq[q] == key ? p.scores[oq[q]] : 0.f;
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const bool valid = kq[q] == key;   // the segment is contiguous: valid lanes form a prefix
                if (valid) {
                    unsigned u = __float_as_uint(sq[q]);
                    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                    skey[64 * q + lane] = ~u;
                    sidx[64 * q + lane] = oq[q];
                }
                seg_len += __popcll(__ballot(valid));
            }
            const int len4 = (seg_len + 3) & ~3;
            if (seg_len + lane < len4) skey[seg_len + lane] = 0xFFFFFFFFu;   // padding never precedes anything
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            constexpr int OWN = SEG_SORT_MAX / 64;
            unsigned mykey[OWN], myidx[OWN], rank[OWN];
#pragma unroll
            for (int q = 0; q < OWN; ++q) {
                const int e = lane + 64 * q;
                mykey[q] = e < seg_len ? skey[e] : 0u;
                myidx[q] = e < seg_len ? sidx[e] : 0u;
                rank[q] = 0;
            }
            const int nown = (seg_len + 63) / 64;   // wave-uniform: groups of 64 candidates in use
            switch (nown) {
                case 1: rank_by_counting<1>(skey, len4, mykey, rank); break;
                case 2: rank_by_counting<2>(skey, len4, mykey, rank); break;
                case 3: rank_by_counting<3>(skey, len4, mykey, rank); break;
                case 4: rank_by_counting<4>(skey, len4, mykey, rank); break;
                case 5: rank_by_counting<5>(skey, len4, mykey, rank); break;
                case 6: rank_by_counting<6>(skey, len4, mykey, rank); break;
                case 7: rank_by_counting<7>(skey, len4, mykey, rank); break;
                default: rank_by_counting<8>(skey, len4, mykey, rank); break;
            }
            // the candidates in rank order (every lane read its own candidate indices above)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int q = 0; q < OWN; ++q)
                if (lane + 64 * q < seg_len) sidx[rank[q]] = myidx[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // Candidates with equal keys got the SAME rank: all but one of them do not find themselves in their slot.
            // Then (rare), key by key: the candidates that hold the key are enumerated through ballots, and each of them
            // moves up by the number of holders with a lower candidate index -- torch's order among equal scores.
            bool lostq[OWN];
            unsigned long long any_lost = 0;
#pragma unroll
            for (int q = 0; q < OWN; ++q) {
                lostq[q] = lane + 64 * q < seg_len && sidx[rank[q]] != myidx[q];
                any_lost |= __ballot(lostq[q]);
            }
            if (any_lost) {
                while (true) {
                    int fq = -1;
                    unsigned long long fm = 0;
#pragma unroll
                    for (int q = 0; q < OWN; ++q) {
                        const unsigned long long bm = __ballot(lostq[q]);
                        if (fq < 0 && bm) {
                            fq = q;
                            fm = bm;
                        }
                    }
                    if (fq < 0) break;   // every pass clears at least the candidate it started from
                    const int fl = __builtin_amdgcn_readfirstlane(__builtin_ctzll(fm));
                    unsigned K = 0;
#pragma unroll
                    for (int q = 0; q < OWN; ++q)
                        if (q == fq) K = __builtin_amdgcn_readlane(mykey[q], fl);
                    bool holds[OWN];
#pragma unroll
                    for (int q = 0; q < OWN; ++q) holds[q] = lane + 64 * q < seg_len && mykey[q] == K;
#pragma unroll
                    for (int qa = 0; qa < OWN; ++qa) {
                        unsigned long long tm = __ballot(holds[qa]);
                        while (tm) {
                            const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(tm));
                            tm &= tm - 1;
                            const unsigned other = __builtin_amdgcn_readlane(myidx[qa], l);
#pragma unroll
                            for (int qb = 0; qb < OWN; ++qb)
                                if (holds[qb] && other < myidx[qb]) ++rank[qb];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < OWN; ++q)
                        if (holds[q]) lostq[q] = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int q = 0; q < OWN; ++q)
                    if (lane + 64 * q < seg_len) sidx[rank[q]] = myidx[q];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
        for (unsigned cs = start; cs < img_end && nkept < kcap; cs += 64) {
            const unsigned t = cs + lane;
            bool valid = t < img_end;
            if (seg_sorted)
                valid = (int)(cs - start) + lane < seg_len;
            else if (valid)
                valid = p.keyA[t] == key;
            const unsigned long long vmask = __ballot(valid);
            if (vmask == 0) break;
            unsigned posB = 0;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                unsigned orig;
                if (seg_sorted) {
                    orig = sidx[cs - start + lane];
                    posB = orig;   // what is recorded for a kept box: the candidate itself
                } else {
                    posB = p.valA[t];
                    orig = p.valB[posB];
                }
                b = *(const f32x4*)(p.boxes + 4ull * orig);
                if (agn) shift = (float)p.classes[orig] * unit;
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
                if (seg_sorted)
                    p.valB[start + r] = posB;   // the segment's kept candidates in score order (valB is free: no order B)
                else
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
        if (seg_sorted && lane == 0) p.sort_k0[start] = (unsigned)(nkept < kcap ? nkept : kcap);   // length of that list
    }
}

// rk[q] += the number of offers in offer[0 .. T4) (T4 a multiple of 4) that are smaller than mine[q], q < NQ.
template <int NQ>
__device__ __forceinline__ void count_smaller_offers(const unsigned long long* offer, int T4,
                                                     const unsigned long long (&mine)[MERGE_MAX / 256],
                                                     unsigned (&rk)[MERGE_MAX / 256]) {
    for (int j = 0; j < T4; j += 4) {
        unsigned long long k[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) k[i] = offer[j + i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rk[q] += k[i] < mine[q] ? 1u : 0u;
        }
    }
}

// One workgroup per image: kept flags in order B -> ranks -> keep list.
__global__ void __launch_bounds__(256) nms_emit_kernel(const NmsParams p) {
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = img_begin(p, n), e = img_begin(p, n + 1);
    const int limit = p.top_k > 0 ? (p.top_k < p.keep_stride ? p.top_k : p.keep_stride) : p.keep_stride;
    int* out = p.keep + (long long)n * p.keep_stride;
    if (e - b > 0 && p.agnostic[n] == IMG_SEGMENT_SORTED) {
        // ---- merge of the segments' kept lists (each in score order, at most top_k long) into the image's first
        //      `limit` detections, in rounds: every segment OFFERS its next R kept candidates (R = as many as fit MERGE_MAX
        //      offers over all segments); the offers are ranked by counting ((key, candidate index): torch's order); they
        //      are accepted in that order up to and including the first one that is the LAST offer of a segment which still
        //      has more -- what follows in that segment may precede the remaining offers, so the next round starts there.
        //      An offer is ONE 64-bit value, key in the high word and candidate index in the low word: real offers are all
        //      different, an empty slot is all ones and precedes nothing.
        __shared__ __attribute__((aligned(16))) unsigned long long offer[MERGE_MAX + 4];
        __shared__ unsigned seg_ptr[256], seg_len[256], seg_at[256];
        __shared__ int cut_s, emitted_s;
        const unsigned seg_base = p.sort_v0[b];
        const int nsegs = (int)p.sort_v0[b + 1];   // <= 2^cbits <= 256
        if (tid < nsegs) {
            const unsigned at = p.seg_start[seg_base + tid];
            seg_at[tid] = at;
            seg_len[tid] = p.sort_k0[at];
            seg_ptr[tid] = 0;
        }
        if (tid == 0) emitted_s = 0;
        __syncthreads();
        // offers per segment and round: about four times a segment's fair share of the output (a round costs a ranking
        // that is quadratic in the offers; a segment that contributes more than its offers costs another round)
        int R = NMS_MERGE_SHARE * limit / (nsegs > 0 ? nsegs : 1) + 2;
        R = R < MERGE_MAX / (nsegs > 0 ? nsegs : 1) ? R : MERGE_MAX / (nsegs > 0 ? nsegs : 1);
        R = R < limit ? R : (limit > 0 ? limit : 1);
        const int T = R * nsegs;   // offer slots of a round: slot o = segment o / R, its (o % R)-th next candidate
        while (true) {
            const int emitted = emitted_s;
            if (emitted >= limit) break;
            // offers (two dependent loads each, all of a thread's slots requested together)
            constexpr int PER = MERGE_MAX / 256;
            unsigned oi[PER];
            unsigned long long mine[PER];
            bool have[PER], last_more[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int o = tid + 256 * q;
                have[q] = false;
                last_more[q] = false;
                oi[q] = 0;
                if (o < T) {
                    const int sg = o / R, r = o - sg * R;
                    const unsigned pos = seg_ptr[sg] + (unsigned)r;
                    if (pos < seg_len[sg]) {
                        have[q] = true;
                        last_more[q] = r == R - 1 && pos + 1 < seg_len[sg];
                        oi[q] = p.valB[seg_at[sg] + pos];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                mine[q] = ~0ull;
                if (have[q]) {
                    unsigned u = __float_as_uint(p.scores[oi[q]]);
                    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                    mine[q] = ((unsigned long long)(~u) << 32) | oi[q];
                }
            }
            if (tid == 0) cut_s = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int o = tid + 256 * q;
                if (o < T) offer[o] = mine[q];
            }
            if (tid < 4) offer[T + tid] = ~0ull;   // padding up to a multiple of four
            __syncthreads();
            // rank of every real offer among the offers
            const int T4 = (T + 3) & ~3;
            const int nq = (T + 255) >> 8;   // slots in use per thread
            unsigned rk[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) rk[q] = 0;
            switch (nq) {
                case 1: count_smaller_offers<1>(offer, T4, mine, rk); break;
                case 2: count_smaller_offers<2>(offer, T4, mine, rk); break;
                case 3: count_smaller_offers<3>(offer, T4, mine, rk); break;
                default: count_smaller_offers<4>(offer, T4, mine, rk); break;
            }
#pragma unroll
            for (int q = 0; q < PER; ++q)
                if (have[q] && last_more[q]) atomicMin(&cut_s, (int)rk[q]);
            __syncthreads();
            int total = 0;   // real offers of this round
            {
                int cnt = 0;
#pragma unroll
                for (int q = 0; q < PER; ++q) cnt += have[q] ? 1 : 0;
                // workgroup sum through the wave sums
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
                if (lane == 0) wsum[wave] = cnt;
                __syncthreads();
                total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            }
            if (total == 0) break;   // every list is used up
            int accept = cut_s == 0x7fffffff ? total : cut_s + 1;
            if (accept > limit - emitted) accept = limit - emitted;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                if (have[q] && (int)rk[q] < accept) {
                    out[emitted + (int)rk[q]] = (int)oi[q] - b;
                    atomicAdd(&seg_ptr[(tid + 256 * q) / R], 1u);
                }
            }
            __syncthreads();
            if (tid == 0) emitted_s = emitted + accept;
            __syncthreads();
        }
        if (tid == 0) p.n_keep[n] = emitted_s < limit ? emitted_s : limit;
        return;
    }
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = b; base < e; base += 256) {
        const int j = base + tid;
        const int v = j < e ? (int)p.keepB[j] : 0;
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
        if (v && rank < limit) out[rank] = (int)p.valB[j] - b;
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
    int cap;  // candidates that exist in the arrays: image starts are clamped to it, as in the NMS
};

// detector_postprocess: scale by (out_w / in_w, out_h / in_h), clip to the output frame, drop boxes
// whose clipped width or height is not positive; order preserved.  One wave per image.
__global__ void __launch_bounds__(64) gather_kernel(const GatherParams p) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int nk = p.n_keep[n];
    const int base = p.img_offsets[n] < p.cap ? p.img_offsets[n] : p.cap;
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
    size_t k0, k1, v0, v1, valB, keyA, valA, seg_start, nseg, img_max, agnostic, keepB, kept_spill, total;
};

static void nms_layout(int N, int cap, NmsLayout& L) {
    size_t o = 0;
    const size_t m = (size_t)(cap > 0 ? cap : 1);
    auto take = [&](size_t bytes) {
        const size_t r = o;
        o += al(bytes);
        return r;
    };
    L.k0 = take(m * 4);
    L.k1 = take(m * 4);
    L.v0 = take(m * 4);
    L.v1 = take(m * 4);
    L.valB = take(m * 4);
    L.keyA = take(m * 4);
    L.valA = take(m * 4);
    L.seg_start = take(m * 4);
    L.nseg = take(256);
    L.img_max = take((size_t)N * 4);
    L.agnostic = take((size_t)N);
    L.keepB = take(m);
    L.kept_spill = take(m * 16);
    L.total = o;
}

static int bits_for(unsigned v) {  // bits needed to represent values 0..v-1
    int b = 1;
    while ((1u << b) < v && b < 31) ++b;
    return b;
}

}  // namespace yolo

using namespace yolo;

extern "C" size_t yolo_nms_workspace_bytes(int N, int M_cap) {
    if (N <= 0 || M_cap < 0) return 0;
    NmsLayout L;
    nms_layout(N, M_cap, L);
    return L.total;
}

extern "C" int yolo_batched_nms(const float* boxes, const float* scores, const int* classes, const int* img_offsets,
                                int N, int M_cap, double iou_thr, int mode, int auto_numel_limit, int top_k,
                                int num_classes, int* keep, int keep_stride, int* n_keep, void* workspace,
                                size_t ws_bytes, void* stream) {
    YOLO_REQUIRE(img_offsets && keep && n_keep, "nms: null argument");
    YOLO_REQUIRE(N >= 1 && N <= 65535, "nms: batch %d (1..65535)", N);
    YOLO_REQUIRE(M_cap >= 0, "nms: negative candidate capacity");
    YOLO_REQUIRE(mode >= YOLO_NMS_PER_CLASS && mode <= YOLO_NMS_AUTO, "nms: mode %d", mode);
    YOLO_REQUIRE(keep_stride > 0 && top_k >= 0, "nms: keep_stride %d / top_k %d", keep_stride, top_k);
    YOLO_REQUIRE(num_classes >= 1 && num_classes <= 65536, "nms: num_classes %d", num_classes);
    hipStream_t st = (hipStream_t)stream;
    if (M_cap == 0) {
        hipLaunchKernelGGL(nms_zero_counts_kernel, dim3((N + 255) / 256), dim3(256), 0, st, n_keep, N);
        YOLO_CHECK_HIP(hipGetLastError());
        return YOLO_OK;
    }
    YOLO_REQUIRE(boxes && scores && classes && workspace, "nms: null argument");
    NmsLayout L;
    nms_layout(N, M_cap, L);
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
    p.cap = M_cap;
    p.iou_thr = iou_thr;
    p.mode = mode;
    p.auto_limit = auto_numel_limit;
    p.top_k = top_k;
    p.cbits = cbits;
    p.sort_k0 = (unsigned*)(ws + L.k0);
    p.sort_k1 = (unsigned*)(ws + L.k1);
    p.sort_v0 = (unsigned*)(ws + L.v0);
    p.sort_v1 = (unsigned*)(ws + L.v1);
    p.valB = (unsigned*)(ws + L.valB);
    p.keyA = (unsigned*)(ws + L.keyA);
    p.valA = (unsigned*)(ws + L.valA);
    p.seg_start = (unsigned*)(ws + L.seg_start);
    p.nseg = (unsigned*)(ws + L.nseg);
    p.img_max = (float*)(ws + L.img_max);
    p.agnostic = (unsigned char*)(ws + L.agnostic);
    p.keepB = (unsigned char*)(ws + L.keepB);
    p.kept_spill = (f32x4*)(ws + L.kept_spill);
    p.keep = keep;
    p.keep_stride = keep_stride;
    p.n_keep = n_keep;

    static DynLdsLimit lds_limit;  // the sort kernel's tables and its reordered tile: 70 KiB of dynamic LDS
    constexpr int SORT_LDS = (int)sizeof(SortShared);
    const void* const sort_kern[] = {(const void*)nms_sort_kernel};
    YOLO_CHECK_HIP(lds_limit.raise(sort_kern, 1, SORT_LDS));
    YOLO_CHECK_HIP(hipMemsetAsync(p.nseg, 0, 256, st));
    hipLaunchKernelGGL(nms_sort_kernel, dim3(N), dim3(SORT_THREADS), SORT_LDS, st, p);
    // one wave per segment; at most N * num_classes segments, and no more than candidates
    long long segs = (long long)N * num_classes;
    if (segs > M_cap) segs = M_cap;
    long long sblocks = (segs + 3) / 4;
    if (sblocks > 4096) sblocks = 4096;
    hipLaunchKernelGGL(nms_sweep_kernel, dim3((unsigned)sblocks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(nms_emit_kernel, dim3(N), dim3(256), 0, st, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_gather_detections(const float* boxes, const float* scores, const int* classes,
                                      const int* img_offsets, int M_cap, const int* keep, int keep_stride,
                                      const int* n_keep, int N, const int* in_hw, const int* out_hw, float* det,
                                      int* n_det, void* stream) {
    YOLO_REQUIRE(boxes && scores && classes && img_offsets && keep && n_keep && in_hw && out_hw && det && n_det,
                 "gather: null argument");
    YOLO_REQUIRE(N >= 1 && keep_stride > 0 && M_cap >= 0, "gather: N %d keep_stride %d capacity %d", N, keep_stride, M_cap);
    GatherParams p{boxes, scores, classes, img_offsets, keep, n_keep, in_hw, out_hw, det, n_det, keep_stride, N, M_cap};
    hipLaunchKernelGGL(gather_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, p);
    YOLO_CHECK_HIP(hipGetLastError());
    return YOLO_OK;
}

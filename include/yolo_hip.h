/*
 * yolo_hip.h -- C ABI of libyolo_hip.so: the gfx950 (MI355X) implementation of the YOLOv3
 * inference hot path of shiboyang/YOLO.
 *
 * The reference is pure Python and has no FFI of its own; each entry point below replaces the
 * torch / detectron2 / torchvision call chain cited next to it (paths relative to the reference
 * checkout).  Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - every function enqueues on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *     never synchronises and never allocates: the caller owns all buffers incl. the workspace;
 *   - return value 0 = ok, negative = error (YOLO_E_*); yolo_last_error() gives the message of the
 *     calling thread's last failure;
 *   - activations are NHWC ("pixel-major") with an optional zero halo of `halo` pixels on each
 *     spatial side, and may be a channel slice of a wider buffer:
 *       element (n, y, x, c) of a view {ptr, H, W, halo, ch_total, ch_offset} lives at
 *       ptr[((n*(H+2*halo) + y+halo)*(W+2*halo) + x+halo)*ch_total + ch_offset + c].
 *     Kernels only ever write the interior, so a halo zeroed once stays zero.
 */
#ifndef YOLO_HIP_H
#define YOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YOLO_OK 0
#define YOLO_E_ARG (-1)      /* invalid argument / unsupported shape */
#define YOLO_E_WORKSPACE (-2) /* workspace too small */
#define YOLO_E_HIP (-3)      /* a HIP runtime call failed */

#define YOLO_F32 0
#define YOLO_BF16 1

/* NMS modes, see yolo_batched_nms */
#define YOLO_NMS_PER_CLASS 0   /* torchvision _batched_nms_vanilla */
#define YOLO_NMS_COORD_TRICK 1 /* torchvision _batched_nms_coordinate_trick */
#define YOLO_NMS_AUTO 2        /* torchvision batched_nms branch rule, per image */

const char* yolo_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int yolo_abi_version(void);

/* A strided NHWC view (see the layout rule above). */
typedef struct yolo_view {
    void* ptr;
    int H, W;      /* logical spatial size */
    int halo;      /* zero border in the buffer, pixels per side */
    int ch_total;  /* channels per pixel in the buffer */
    int ch_offset; /* first channel of this view */
} yolo_view;

/*
 * Fused convolution: y = act(conv(x, w) + bias) [+ residual], optionally written 2x nearest-
 * upsampled.  Replaces detectron2 Conv2d.forward (F.conv2d + eval BatchNorm, folded by the caller
 * into w/bias) + F.leaky_relu at yolo/modeling/backbone/darknet.py:58-59,115-116,119-123 and
 * fpn.py:34-35,68-69,97-99; the `out += residual` of darknet.py:118,125 (added AFTER the activation);
 * nn.Upsample(2, nearest) of fpn.py:36; and, through y.ch_offset, the torch.cat of fpn.py:179.
 *
 *   x        input view, N images, Cin channels; x.halo >= pad is required (zero padding comes from
 *            the halo), Cin % 8 == 0 for bf16 (Cin % 4 for f32); Cin == 32 or Cin % 64 == 0 (bf16),
 *            Cin % 32 == 0 (f32)
 *   w_packed [Cout_pad][k][k][Cin] in `dtype`, Cout_pad = Cout rounded up to 128, rows >= Cout zero
 *   bias     [Cout_pad] f32
 *   residual NULL, or a view with y's geometry (its own ptr / ch_total / ch_offset / halo)
 *   y        output view with H, W = conv output size (2x that when upsample2x)
 *   k in {1,3}; stride in {1,2}; pad = (k-1)/2
 *   leaky_slope: negative-side slope in [0, 1]; 1.0f = linear (no activation)
 *   dtype    element type of x, w_packed, residual (YOLO_F32 / YOLO_BF16); accumulation is f32
 *   y_dtype  element type of y
 */
int yolo_conv2d_fwd(const yolo_view* x, const void* w_packed, const float* bias,
                    const yolo_view* residual, const yolo_view* y,
                    int N, int Cin, int Cout, int k, int stride,
                    float leaky_slope, int upsample2x, int dtype, int y_dtype, void* stream);

/* Rows of the packed weight buffer for Cout output channels (Cout rounded up to the channel tile). */
int yolo_conv_cout_pad(int Cout);

/* Introspection for benchmarks and profilers: the workgroup tile (output channels x output pixels) that
 * yolo_conv2d_fwd uses for M = N*Ho*Wo output pixels and Cout channels, i.e. which kernel instance runs. */
int yolo_conv_tile_shape(int M, int Cin, int Cout, int k, int dtype, int* tile_channels, int* tile_pixels);

/*
 * yolo_conv2d_fwd for layers with too few output pixels to occupy the chip (single images: 16-58 workgroup tiles for
 * 256 CUs): the k loop of every tile is split over several workgroups that write f32 partial sums into `workspace`,
 * and a second launch adds them in a fixed order, applies bias / LeakyReLU / residual and stores the output view.
 * Same arguments and results as yolo_conv2d_fwd (sums are associated differently: equal to rounding).  bf16 inputs;
 * shapes for which splitting does not pay (yolo_conv2d_splitk_plan says splits == 1) run the plain launch.
 *   yolo_conv2d_splitk_plan: splits and workspace bytes for a convolution with M = N*Ho*Wo output pixels.
 */
int yolo_conv2d_splitk_plan(int M, int Cin, int Cout, int k, int stride, int dtype, int* splits, size_t* workspace_bytes);
int yolo_conv2d_fwd_splitk(const yolo_view* x, const void* w_packed, const float* bias,
                           const yolo_view* residual, const yolo_view* y,
                           int N, int Cin, int Cout, int k, int stride, float leaky_slope, int upsample2x,
                           int dtype, int y_dtype, void* workspace, size_t ws_bytes, void* stream);

/* Introspection for benchmarks and profilers: a short name of the kernel instance yolo_conv2d_fwd launches for this
 * convolution, "conv_igemm_kernel<dtype,tile channels,tile pixels,wave layout,form>" with form = plain | tap2 | strip
 * (strip: the 3x3 / stride-1 form whose three kw taps share one staged strip of pixel rows). */
int yolo_conv_kernel_name(const yolo_view* x, const yolo_view* y, int N, int Cin, int Cout, int k, int stride,
                          int upsample2x, int dtype, char* out, int out_len);

/*
 * First layer: pixel normalisation + zero padding + 3x3/s1 stem conv + bias + LeakyReLU.
 * Replaces DenseDetector.preprocess_image (driven by configs/yolov3.yaml:5-6 and fpn.py:161-163)
 * and BasicStem.forward (darknet.py:57-60).
 *   img      N images, planar CHW (3,Hs,Ws) each, contiguous: float32 (img_dtype 0) or uint8 (1),
 *            values 0..255, channel order already the model's (tools/inference.py:38-39)
 *   Hs, Ws   stored image size; pixels outside it (up to y.H, y.W) are zero AFTER normalisation
 *   mean_host, std_host  host arrays [3]: (v - mean[c]) / std[c], as the reference computes it
 *   w        [3][3][3][Cout] (kh,kw,cin,cout) f32, BN folded; bias [Cout] f32; Cout == 32
 *   y        output view (H, W multiples of 32 by the reference's size_divisibility)
 */
int yolo_stem_fwd(const void* img, int img_dtype, int N, int Hs, int Ws,
                  const float* mean_host, const float* std_host,
                  const float* w, const float* bias, const yolo_view* y, int Cout,
                  float leaky_slope, int y_dtype, void* stream);

/*
 * yolo_stem_fwd followed by the first down-sampling conv, in one launch, bf16.  Replaces the same call
 * sites as yolo_stem_fwd plus ResidualBlock.downsample of `res1` (darknet.py:113-116: Conv2d 3x3/s2 +
 * BN + leaky_relu); the stem's 32-channel full-resolution output stays in LDS and is never written.
 *   img .. b_stem   as for yolo_stem_fwd (stem Cout == 32)
 *   w_down_packed   the bf16 packing yolo_conv2d_fwd takes for Cin == 32, k == 3:
 *                   [Cout_pad][10][32] (9 taps + one zero tap), Cout == 64; b_down [>= 64] f32
 *   H, W            padded network input size (even; stem output size); pixels beyond Hs, Ws are zero
 *   y               bf16 output view, H/2 x W/2, 64 channels
 */
int yolo_stem_down_fwd(const void* img, int img_dtype, int N, int Hs, int Ws,
                       const float* mean_host, const float* std_host,
                       const float* w_stem, const float* b_stem,
                       const void* w_down_packed, const float* b_down,
                       const yolo_view* y, int H, int W, float leaky_slope, void* stream);

/*
 * One whole 64-channel residual block in one launch, bf16:
 *   y = x + leaky(conv3x3(leaky(conv1x1(x, w1) + b1), w2) + b2)
 * Replaces ResidualBlock.forward after its down-sampling conv (darknet.py:117-126: residual = x.clone(),
 * conv1 1x1 64 -> 32 + BN + leaky_relu, conv2 3x3 32 -> 64 + BN + leaky_relu, out += residual); the
 * 32-channel intermediate stays in LDS.
 *   x          bf16 view, 64 channels, halo >= 1
 *   w1_packed  yolo_conv2d_fwd's bf16 packing of the 1x1 conv: [Cout_pad][64]; b1 [>= 32] f32
 *   w2_packed  yolo_conv2d_fwd's bf16 packing of the 3x3 conv: [Cout_pad][10][32]; b2 [>= 64] f32
 *   y          bf16 view of x's spatial size, 64 channels, in a buffer that does not overlap x's
 */
int yolo_resblock_c64_fwd(const yolo_view* x, const void* w1_packed, const float* b1,
                          const void* w2_packed, const float* b2, const yolo_view* y,
                          int N, float leaky_slope, void* stream);

/* Nearest-neighbour 2x upsample of an NHWC view into another (channel slice of a) view.
 * Replaces nn.Upsample(scale_factor=2, mode="nearest") + torch.cat at fpn.py:31,36,179 for
 * callers that do not use yolo_conv2d_fwd's fused form.  C % 8 == 0 (bf16) / % 4 (f32). */
int yolo_upsample2x_nhwc(const yolo_view* x, const yolo_view* y, int N, int C, int dtype, void* stream);

/*
 * Head activation + anchor decode + two-stage score filter + order-preserving compaction.
 * Replaces YoloV3Head.forward (yolo/modeling/meta_arch/yolo.py:392-420), _transpose_dense_predictions
 * (:276-278), the anchor generator call (:279, yolo/modeling/anchor_generator.py:18-40) and
 * _decode_multi_level_predictions / _decode_per_level_predictions (:309-383) with
 * Box2BoxTransform.apply_deltas (yolo/modeling/box_regression.py:44-64), for all images at once.
 *
 *   head[l]      raw head tensor of level l (0 = finest stride), NHWC view with A*(5+C) channels,
 *                channel = a*(5+C) + {tx,ty,tw,th,obj,cls0..}
 *   stride_host  [L] strides; anchors_wh_host [L][A][2] anchor (w,h) in input pixels
 *   thr          strict '>' threshold used for BOTH the objectness and the class*objectness filter
 *   outputs (candidate order = image, level, y, x, anchor, class -- the reference's order):
 *     boxes [cap][4] f32 xyxy, scores [cap] f32, classes [cap] i32,
 *     img_offsets [N+1] i32: candidates of image n are [img_offsets[n], img_offsets[n+1]);
 *                img_offsets[N] is the TRUE total even if it exceeds cap (writes beyond cap are
 *                dropped, so compare it with cap to detect overflow)
 *   workspace    yolo_decode_workspace_bytes(...) bytes
 */
int yolo_decode(const yolo_view* head, int L, const int* stride_host, const float* anchors_wh_host,
                int N, int A, int C, float thr, int head_dtype,
                float* boxes, float* scores, int* classes, int* img_offsets, int cap,
                void* workspace, size_t ws_bytes, void* stream);
size_t yolo_decode_workspace_bytes(const yolo_view* head, int L, int N);

/*
 * yolo_decode without a workspace argument, for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same
 * arguments, launches and results as yolo_decode; the per-cell counts / offsets and per-image totals live in a device
 * array built into the library (4.02 MiB, held by every GPU the library runs on until the process ends).  The call
 * allocates and frees nothing; the first call on a device looks up that device's copy of the array
 * (hipGetSymbolAddress), so make it once outside a graph capture.
 *   limits      N * (grid cells per image, all levels) <= YOLO_DECODE_NOALLOC_MAX_SLOTS and
 *               N <= YOLO_DECODE_NOALLOC_MAX_BATCH (batch 64 at 608 x 608 is 485,184 slots); beyond them the call
 *               fails with YOLO_E_ARG and a message naming yolo_decode
 *   concurrency every call on one GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
 *               threads) race; calls on one stream, or on different GPUs, do not.  Nothing of an earlier call's
 *               contents is read.
 */
#define YOLO_DECODE_NOALLOC_MAX_SLOTS (512 * 1024)
#define YOLO_DECODE_NOALLOC_MAX_BATCH 4096
int yolo_decode_noalloc(const yolo_view* head, int L, const int* stride_host, const float* anchors_wh_host,
                        int N, int A, int C, float thr, int head_dtype,
                        float* boxes, float* scores, int* classes, int* img_offsets, int cap, void* stream);

/*
 * Batched per-class NMS over all images.  Replaces batched_nms(boxes, scores, classes, thr) at
 * yolo/modeling/meta_arch/yolo.py:305 (detectron2.layers.batched_nms -> torchvision batched_nms ->
 * torchvision::nms) and the keep[:max_detections] truncation at :307.
 *
 *   boxes/scores/classes  candidates grouped by image through img_offsets [N+1] (device)
 *   M_cap         an upper bound of the number of candidates that exist in the arrays (host value, e.g. their
 *                 capacity); the counts themselves are read on the device from img_offsets, so the host never has to
 *                 read them back.  Offsets beyond M_cap (a decode that overflowed its capacity) are clamped to it.
 *   iou_thr       suppression when (double)(float iou) > iou_thr, strict
 *   mode          YOLO_NMS_*; AUTO applies torchvision's rule per image with `auto_numel_limit`
 *                 (4000 = what the reference does on CPU, 20000 = on a GPU): coordinate trick when
 *                 4*M_img <= limit
 *   top_k         > 0: return only the first top_k kept per image (yolo.py:307); 0: all
 *   num_classes   class ids are in [0, num_classes)
 *   keep          [N][keep_stride] i32: per image, indices RELATIVE to the image's first candidate,
 *                 in descending score order (ties: ascending index)
 *   n_keep        [N] i32 number of valid entries per image
 */
int yolo_batched_nms(const float* boxes, const float* scores, const int* classes,
                     const int* img_offsets, int N, int M_cap, double iou_thr, int mode,
                     int auto_numel_limit, int top_k, int num_classes, int* keep, int keep_stride,
                     int* n_keep, void* workspace, size_t ws_bytes, void* stream);
size_t yolo_nms_workspace_bytes(int N, int M);

/*
 * Gather kept detections and map them to the output frame.  Replaces pred[keep[:100]] (yolo.py:307)
 * and detectron2's detector_postprocess (scale to the requested size, clip, drop empty boxes).
 *   in_hw  [N][2] i32 (device) network-input image size (h,w) before padding; out_hw [N][2] requested size
 *   M_cap       the capacity given to yolo_batched_nms (image starts beyond it are clamped to it, as there)
 *   det         [N][keep_stride][6] f32: x0,y0,x1,y1,score,class ; n_det [N] i32
 */
int yolo_gather_detections(const float* boxes, const float* scores, const int* classes,
                           const int* img_offsets, int M_cap, const int* keep, int keep_stride, const int* n_keep,
                           int N, const int* in_hw, const int* out_hw,
                           float* det, int* n_det, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training path (reference: tools/train_net.py -> detectron2 DefaultTrainer: losses = model(batch);
 * losses.backward(); optimizer.step()).  Modules in training mode run convolution, batch-statistics
 * BatchNorm and LeakyReLU as separate steps; yolo_conv2d_fwd (leaky_slope 1, zero bias) is the convolution,
 * and also the data gradient: dX of a stride-1 conv is yolo_conv2d_fwd of dZ with the weights transposed
 * (Cin <-> Cout) and rotated by 180 degrees; for stride 2, of yolo_dilate2x_nhwc(dZ).
 * ------------------------------------------------------------------------------------------------ */

/*
 * Batch statistics of a conv output z (the BatchNorm2d that detectron2's Conv2d applies after F.conv2d in
 * training mode; modules built by get_norm("BN") at darknet.py:45-53,74-106, fpn.py:23-30,44-61,78-90).
 *   stats  [5][C] f32 out: mean, biased variance, 1/sqrt(var+eps), scale = gamma/sqrt(var+eps),
 *          shift = beta - mean*scale
 *   running_mean/var  NULL, or updated in place as torch does: (1-momentum)*old + momentum*new, with the
 *          UNBIASED batch variance
 *   C = 8 * power of two, <= 2048;  workspace: yolo_bn_workspace_bytes(N, H, W, C)
 */
int yolo_bn_train_stats(const yolo_view* z, int N, int C, int dtype, const float* gamma, const float* beta,
                        float eps, float momentum, float* running_mean, float* running_var, float* stats,
                        void* workspace, size_t ws_bytes, void* stream);
size_t yolo_bn_workspace_bytes(int N, int H, int W, int C);

/*
 * The same convolution as yolo_conv2d_fwd (linear, no residual, output type = operand type) that also leaves
 * the first pass of the following training-mode BatchNorm (detectron2 Conv2d.forward: F.conv2d, then self.norm):
 *   sums  [rows][2][Cout] f32, rows = yolo_conv2d_bnsums_rows(x, y, N, Cin, Cout, k, stride, dtype): per pixel tile
 *         (of the form and tile the launch picks for these views) and channel, the sum of the output values as stored
 *         and of their squares
 * yolo_bn_train_finalize turns the rows into what yolo_bn_train_stats writes (stats [5][C], running statistics);
 * count = N*Ho*Wo.  Together they replace yolo_conv2d_fwd + yolo_bn_train_stats without re-reading z.
 */
int yolo_conv2d_fwd_bnsums(const yolo_view* x, const void* w_packed, const float* bias, const yolo_view* y,
                           int N, int Cin, int Cout, int k, int stride, int dtype, float* sums,
                           size_t sums_bytes, void* stream);
int yolo_conv2d_bnsums_rows(const yolo_view* x, const yolo_view* y, int N, int Cin, int Cout, int k, int stride,
                            int dtype);
int yolo_bn_train_finalize(const float* sums, int rows, long long count, int C, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean,
                           float* running_var, float* stats, void* stream);

/* y = leaky(z*scale + shift) [+ residual]: BatchNorm2d (training) + F.leaky_relu (darknet.py:59,116,120,123;
 * fpn.py:35,69,98) + `out += residual` (darknet.py:125).  stats from yolo_bn_train_stats. */
int yolo_bn_act_fwd(const yolo_view* z, const float* stats, const yolo_view* residual, const yolo_view* y,
                    int N, int C, float leaky_slope, int dtype, void* stream);

/* Backward of yolo_bn_act_fwd (without the residual, whose gradient is dy itself): given dy and the saved z
 * and stats, writes dgamma [C], dbeta [C] (f32) and dz = scale*(g - mean(g) - zhat*mean(g*zhat)),
 * g = dy*leaky'(z*scale+shift), zhat = (z-mean)/sqrt(var+eps) -- what autograd derives for
 * F.batch_norm(training=True) + F.leaky_relu. */
int yolo_bn_act_bwd(const yolo_view* dy, const yolo_view* z, const float* stats, const yolo_view* dz,
                    int N, int C, float leaky_slope, int dtype, float* dgamma, float* dbeta,
                    void* workspace, size_t ws_bytes, void* stream);

/*
 * The data-gradient convolution of a layer (yolo_conv2d_fwd with the transposed, tap-rotated weights, linear, optional
 * residual = the gradient's other contributions, stride 1, bf16) when its output y IS the gradient dL/da of an
 * activation a = leaky(bn(z)) produced by a training-mode BatchNorm (darknet.py:116,120,123; fpn.py:35,69,98): the
 * epilogue also leaves the reduction pass of that BatchNorm's backward,
 *   sums  [rows][2][Cout] f32, rows = yolo_conv2d_bnsums_rows(x, y, ..., stride 1, ...): per pixel tile and channel,
 *         sum g and sum g*z with
 *         g = y * leaky'(z*bn_scale + bn_shift), y as stored
 *   z, bn_scale, bn_shift, act_slope   the producer's saved pre-normalisation output and rows 3 and 4 of its stats
 * yolo_bn_act_bwd_presummed then finishes what yolo_bn_act_bwd does (dgamma, dbeta, dz) in two launches instead of
 * three, reading dy and z once instead of twice.  What autograd runs as separate kernels in the reference
 * (losses.backward(), tools/train_net.py -> DefaultTrainer.run_step).
 */
int yolo_conv2d_fwd_bnbwd_sums(const yolo_view* x, const void* w_packed, const float* bias,
                               const yolo_view* residual, const yolo_view* y, int N, int Cin, int Cout, int k,
                               int dtype, const yolo_view* z, const float* bn_scale, const float* bn_shift,
                               float act_slope, float* sums, size_t sums_bytes, void* stream);
int yolo_bn_act_bwd_presummed(const yolo_view* dy, const yolo_view* z, const float* stats, const yolo_view* dz,
                              int N, int C, float leaky_slope, int dtype, float* dgamma, float* dbeta,
                              const float* sums, int rows, void* stream);

/*
 * Weight gradient of F.conv2d (bf16 or f32 operands per `dtype`, f32 result):
 *   dw[co][kh][kw][ci] (+)= sum_{n,ho,wo} dz[n][ho][wo][co] * x[n][ho*stride+kh-pad][wo*stride+kw-pad][ci]
 *   x   input view of the forward conv (halo >= pad);  dz  gradient view of its output (zero halo >= 1 unless
 *   N*Ho*Wo % 64 == 0);  accumulate != 0 adds to dw;  workspace: yolo_conv2d_wgrad_workspace_bytes(N*Ho*Wo, ...)
 */
int yolo_conv2d_wgrad(const yolo_view* x, const yolo_view* dz, int N, int Cin, int Cout, int k, int stride,
                      int dtype, float* dw, int accumulate, void* workspace, size_t ws_bytes, void* stream);
size_t yolo_conv2d_wgrad_workspace_bytes(int M, int Cin, int Cout, int k);

/*
 * Weight gradient of the stem convolution (3 -> 32 channels, 3x3, stride 1), for training with
 * MODEL.BACKBONE.FREEZE_AT = 0, where DarkNet.freeze (darknet.py:236-258) leaves BasicStem (darknet.py:40-60)
 * trainable and the reference obtains this gradient from autograd:
 *   dw[co][kh][kw][ci] (+)= sum_{n,y,x} dz[n][y][x][co] * xn[n][y+kh-1][x+kw-1][ci]
 *   img / img_dtype / Hs / Ws / mean_host / std_host   as yolo_stem_fwd: xn = (img - mean) / std, zero outside the image
 *   dz                 gradient view of the stem's convolution output (32 channels, dz_dtype f32 or bf16)
 *   round_input_bf16   != 0: xn rounded to bf16, as the bf16 forward kernel multiplies it
 *   workspace          yolo_stem_wgrad_workspace_bytes()
 */
int yolo_stem_wgrad(const void* img, int img_dtype, int N, int Hs, int Ws, const float* mean_host,
                    const float* std_host, const yolo_view* dz, int dz_dtype, int round_input_bf16, float* dw,
                    int accumulate, void* workspace, size_t ws_bytes, void* stream);
size_t yolo_stem_wgrad_workspace_bytes(void);

/*
 * Per-step repacking of a trainable conv's f32 master weights w [Cout][Cin][k][k] (torch layout) into
 *   w_fwd    rows [co][kh][kw][ci] of `fwd_row` elements (what yolo_conv2d_fwd reads), and/or
 *   w_dgrad  rows [ci][kh][kw][co] of `dgrad_row` elements, taps rotated by 180 degrees, co padded to Cout_pad
 *            (what yolo_conv2d_fwd reads when it computes the data gradient);
 * either may be NULL.  Padding rows/columns are not written: zero the buffers once.  Replaces the
 * detach/flip/permute/cast chain a torch implementation would run per layer and step.
 */
int yolo_pack_conv_train(const float* w, int Cout, int Cin, int k, int dtype, void* w_fwd, int fwd_row,
                         void* w_dgrad, int dgrad_row, int Cout_pad, void* stream);

/*
 * yolo_pack_conv_train for every trainable convolution of a model in one launch, for master weights stored
 * pixel-major: w [Cout][k][k][Cin] fp32 (the storage of a torch channels_last [Cout,Cin,k,k] tensor, which is
 * also the layout yolo_conv2d_wgrad writes, so parameters, gradients and optimizer state share one layout).
 *   descs_device  table of n_layers entries in DEVICE memory; first_tile = running sum of
 *                 ceil(Cout/64) * k*k * ceil(Cin/64) over the preceding entries; total_tiles = the full sum
 *   w_fwd / w_dgrad, fwd_row, dgrad_row, Cout_pad: as for yolo_pack_conv_train (w_dgrad may be NULL)
 */
typedef struct yolo_pack_desc {
    const void* w;
    void* w_fwd;
    void* w_dgrad;
    int Cout, Cin, k, fwd_row, dgrad_row, Cout_pad, first_tile, reserved;
} yolo_pack_desc;
int yolo_pack_conv_train_batch(const yolo_pack_desc* descs_device, int n_layers, int total_tiles, int dtype,
                               void* stream);

/*
 * One SGD-with-momentum step over flat fp32 arenas (what torch.optim.SGD does for the parameters the reference
 * trains, tools/train_net.py via detectron2's build_optimizer: d = grad + wd * w; m = momentum * m + d; w -= lr * m,
 * dampening 0, no Nesterov).  block_table (device): [n_blocks][3] i32 = first element, count (<= 4096), group (0 / 1)
 * of each workgroup's run; the two groups carry their own learning rate and weight decay (weights / normalisation
 * parameters).  A zero momentum buffer makes the first step m = d, as torch's.
 */
int yolo_sgd_step(float* param, const float* grad, float* momentum_buf, const int* block_table, int n_blocks, float lr0,
                  float weight_decay0, float lr1, float weight_decay1, float momentum, void* stream);

/*
 * Data gradient of a 3x3 / stride-2 / pad-1 convolution (ResidualBlock.downsample, darknet.py:79-82,113-116, under
 * autograd) in one launch, bf16: dx = yolo_conv2d_fwd(yolo_dilate2x_nhwc(dz), w_dgrad_packed) without the dilated
 * buffer and without multiplying its zeros -- every output pixel visits only the 1, 2 or 4 taps that can reach it.
 *   dz        gradient view [Hi x Wi], Cin channels (the forward conv's output channels, padded to 64), zero halo >= 1
 *   w_dgrad_packed, bias   the data-gradient packing yolo_pack_conv_train(_batch) writes; bias [>= Cout] f32 (zeros)
 *   residual  NULL or a view of dx's geometry that is added (another consumer's gradient)
 *   dx        output view [2Hi x 2Wi], Cout channels (the forward conv's input channels), Cout > 64
 */
int yolo_conv3x3s2_dgrad(const yolo_view* dz, const void* w_dgrad_packed, const float* bias,
                         const yolo_view* residual, const yolo_view* dx, int N, int Cin, int Cout, int dtype,
                         void* stream);

/* y[2i][2j] = x[i][j], zeros elsewhere in each 2x2 cell (input of a stride-2 conv's data gradient). */
int yolo_dilate2x_nhwc(const yolo_view* x, const yolo_view* y, int N, int C, int dtype, void* stream);
/* y[i][j] = sum of the 2x2 cell of x: backward of nn.Upsample(2, "nearest") (fpn.py:31,36). */
int yolo_sum2x2_nhwc(const yolo_view* x, const yolo_view* y, int N, int C, int dtype, void* stream);

/*
 * Label assignment + the three training losses for a whole batch, no host read-back.  Replaces
 * YoloV3Head.forward (yolo/modeling/meta_arch/yolo.py:392-420), _transpose_dense_predictions (:107-109),
 * label_anchors (:214-264) with pairwise_iou and Matcher.__call__ (yolo/modeling/matcher.py:18-55), losses and
 * _compute_*_loss (:131-208) and Box2BoxTransform.get_deltas (yolo/modeling/box_regression.py:18-42).
 *   head[l]     f32 head tensor of level l (0 = finest stride), A*(5+C) channels, as for yolo_decode
 *   anchors     [R][4] f32 xyxy (device), R = sum over levels of H*W*A, index = level offset + (y*W + x)*A + a
 *   gt_boxes [N][K][4] f32 xyxy, gt_classes [N][K] i32, gt_count [N] i32 (device): image n has gt_count[n] <= K boxes
 *   match_thr   anchors whose best IoU exceeds it and that no ground truth picked are ignored (Matcher threshold)
 *   losses      [3] f32 (device): confidence (obj_weight * mean BCE over positives + noobj_weight * mean BCE over
 *               negatives), class (mean BCE over positives x C), box (mean smooth-L1 over positives x 4)
 *   counts_out  NULL or [2] i32 (device): number of positive and of negative anchors in the batch
 *   workspace   yolo_train_loss_workspace_bytes(N, K, R) bytes; it carries the labels to yolo_train_loss_bwd
 */
int yolo_train_loss_fwd(const yolo_view* head, int L, const int* stride_host, const float* anchors, int R, int N,
                        int A, int C, const float* gt_boxes, const int* gt_classes, const int* gt_count, int K,
                        float match_thr, float obj_weight, float noobj_weight, float* losses, int* counts_out,
                        void* workspace, size_t ws_bytes, void* stream);
size_t yolo_train_loss_workspace_bytes(int N, int K, int R);

/*
 * Backward of yolo_train_loss_fwd (what autograd derives for the calls listed there): writes
 * d(sum_i grad_losses[i] * losses[i]) / d(head) into dhead[l] -- every pixel and channel of the views, zeros
 * included (channels beyond A*(5+C) are zeroed).  Same arguments and the same, untouched workspace as the forward
 * call;  dhead[l]: bf16 or f32 (dhead_dtype) views of the head's spatial size with >= A*(5+C) channels, a
 * multiple of 8;  grad_losses: [3] f32 (device).
 */
int yolo_train_loss_bwd(const yolo_view* head, const yolo_view* dhead, int dhead_dtype, int L, const int* stride_host,
                        const float* anchors, int R, int N, int A, int C, const float* gt_boxes,
                        const int* gt_classes, const int* gt_count, int K, float obj_weight, float noobj_weight,
                        const float* grad_losses, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training-image augmentation on the device.  Images are uint8, HWC, 3 channels, rows `stride` bytes apart.
 * The reference runs these steps per image on CPU workers through OpenCV / Pillow.
 * ------------------------------------------------------------------------------------------------ */

/*
 * LetterBoxTransform.apply_image (yolo/data/transforms.py:207-212): cv2.resize(INTER_LINEAR) of the source to
 * resized_height x resized_width (skipped when the size is unchanged), placed at (top, left) of a
 * dst_height x dst_width image whose other pixels take border_value (114 in the reference).
 */
int yolo_aug_letterbox(const void* src, int src_height, int src_width, long long src_stride, void* dst,
                       int dst_height, int dst_width, long long dst_stride, int top, int left, int resized_height,
                       int resized_width, int border_value, void* stream);

/*
 * The resize detectron2's ResizeShortestEdge applies to a uint8 image (yolo/data/detection_utils.py:55,96-104;
 * Predictor, tools/inference.py:36-45): Pillow's Image.resize(BILINEAR) -- a triangle filter widened by the
 * down-scaling factor, 22-bit integer coefficients, horizontal pass into a uint8 intermediate, then vertical.
 *   x_bounds [dst_width][2] i32 (first source column, count), x_coef [dst_width][x_ksize] i32: device tables of
 *   the horizontal pass (NULL when the width does not change); y_* likewise for the vertical pass.
 *   workspace: yolo_aug_resample_workspace_bytes(src_height, dst_width) bytes, needed when both sizes change.
 */
int yolo_aug_resample(const void* src, int src_height, int src_width, long long src_stride, void* dst,
                      int dst_height, int dst_width, long long dst_stride, const int* x_bounds, const int* x_coef,
                      int x_ksize, const int* y_bounds, const int* y_coef, int y_ksize, void* workspace,
                      size_t workspace_bytes, void* stream);
size_t yolo_aug_resample_workspace_bytes(int src_height, int dst_width);
/* The same for many images: 16 per launch, all horizontal passes first, then all vertical ones.  jobs: HOST array.
 * Every job that changes both sizes needs its own workspace (the jobs of a call run concurrently). */
typedef struct yolo_aug_resample_job {
    const void* src;
    void* dst;
    void* workspace;
    long long workspace_bytes, src_stride, dst_stride;
    int src_height, src_width, dst_height, dst_width;
    const int *x_bounds, *x_coef, *y_bounds, *y_coef;
    int x_ksize, y_ksize;
} yolo_aug_resample_job;
int yolo_aug_resample_batch(const yolo_aug_resample_job* jobs, int n_jobs, void* stream);

/*
 * One image's remaining augmentation as ONE pass: AffineTransform.apply_image (transforms.py:43-50: cv2.warpAffine /
 * cv2.warpPerspective, INTER_LINEAR, constant border) -> ColorJitterTransform.apply_image (transforms.py:177-186:
 * to HSV, three 256-entry tables, back) -> RandomFlip -> the crop [crop_y : crop_y + crop_h, crop_x : crop_x + crop_w]
 * of that result written at (dst_y, dst_x) of the destination (YOLOV3DatasetMapper._blend_mosaic,
 * yolo/data/dataset_mapper.py:136-152; crop = whole image and dst = a same-sized image without mosaic).
 * The augmented image has the source's size.
 *   warp         0 none, 1 affine (inverse_map[0..5]), 2 perspective (inverse_map[0..8]); the map takes output
 *                pixels to source pixels, i.e. it is the INVERSE of AffineTransform.M
 *   jitter       0 / 1; hue_gain, saturation_gain, exposure_gain as ColorJitterTransform's arguments
 *   rgb_order    1 if channel 0 is red (INPUT.FORMAT "RGB"), 0 for BGR
 *   reverse_channels  1: write the three channels in reverse order (a BGR source delivered as RGB: the reference
 *                flips the decoded image on the host, yolo/data/detection_utils.py:13-14); rgb_order describes src
 *   dst_plane_stride  0: dst is HWC like the source (dst_stride >= 3 * dst_width); otherwise dst is CHW -- the
 *                layout the model's batched_inputs carry (dataset_mapper.py:132) -- with planes this many bytes
 *                apart and rows dst_stride >= dst_width bytes apart
 */
typedef struct yolo_aug_job {
    const void* src;
    void* dst;
    long long src_stride, dst_stride, dst_plane_stride;
    int src_height, src_width, dst_height, dst_width;
    int dst_x, dst_y, crop_x, crop_y, crop_w, crop_h;
    int warp, jitter, rgb_order, flip_horizontal, flip_vertical, border, reverse_channels, reserved;
    double inverse_map[9];
    double hue_gain, saturation_gain, exposure_gain;
} yolo_aug_job;
/* jobs: HOST array (copied into the launch); any number, 16 images per launch. */
int yolo_aug_apply(const yolo_aug_job* jobs, int n_jobs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HIP_H */

"""Python-side handles of the C-ABI operators of libyolo_hip.so (include/yolo_hip.h).

torch is used for device memory and streams only; every function here forwards raw device pointers
to a hand-written gfx950 kernel.  There is no fallback: without the library or a GPU these raise.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import YoloView, check, lib

F32, BF16 = 0, 1
NMS_PER_CLASS, NMS_COORD_TRICK, NMS_AUTO = 0, 1, 2


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise ValueError(f"unsupported dtype {dt}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Handle of the current stream of the current device (the raw query: a Stream object per launch costs several
    microseconds of host time, which single-image calls with ~130 launches feel)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _require_cuda(t: torch.Tensor, what: str):
    """Every operator launches on the current stream of the CURRENT device.  A tensor that lives on another GPU
    would be computed on by the wrong device's stream (a silent race or a fault), so it is refused here; the engines
    and the model enter ``torch.cuda.device(their device)`` themselves (``on_own_device``)."""
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the HIP path needs a GPU tensor "
                           "(there is no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"{what}: tensor is on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           f"call under `with torch.cuda.device({t.device.index})`")


def on_own_device(method):
    """Decorator for methods of objects with a ``.device``: run under that GPU's device guard, so that
    ``torch.cuda.current_stream()`` and every launch refer to the GPU that holds the object's memory."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        dev = self.device
        if dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
            return method(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return method(self, *args, **kwargs)
    return wrapper


@dataclass
class View:
    """A pixel-major (NHWC) activation buffer with a zero halo, or a channel slice of one.

    ``buf`` has shape [N, H+2*halo, W+2*halo, ch_total]; the view covers channels
    [ch_offset, ch_offset+C).  Mirrors struct yolo_view.
    """
    buf: torch.Tensor
    H: int
    W: int
    halo: int
    C: int
    ch_offset: int = 0

    @property
    def N(self) -> int:
        return self.buf.shape[0]

    @property
    def ch_total(self) -> int:
        return self.buf.shape[3]

    def c(self) -> YoloView:
        return YoloView(self.buf.data_ptr(), self.H, self.W, self.halo, self.ch_total, self.ch_offset)

    def slice(self, ch_offset: int, C: int) -> "View":
        assert self.ch_offset + ch_offset + C <= self.ch_total
        return View(self.buf, self.H, self.W, self.halo, C, self.ch_offset + ch_offset)

    def interior(self) -> torch.Tensor:
        """Logical [N, H, W, C] tensor (a strided view of the buffer, no copy)."""
        h = self.halo
        return self.buf[:, h:h + self.H, h:h + self.W, self.ch_offset:self.ch_offset + self.C]

    def nchw(self) -> torch.Tensor:
        """Logical [N, C, H, W] tensor (strided view, no copy)."""
        return self.interior().permute(0, 3, 1, 2)


def alloc_view(N: int, H: int, W: int, C: int, halo: int, dtype: torch.dtype, device) -> View:
    buf = torch.zeros((N, H + 2 * halo, W + 2 * halo, C), dtype=dtype, device=device)
    return View(buf, H, W, halo, C, 0)


def view_from_nchw(x: torch.Tensor, halo: int, dtype: torch.dtype) -> View:
    """Copy an NCHW tensor into a fresh halo'd NHWC buffer (test / boundary helper)."""
    N, C, H, W = x.shape
    v = alloc_view(N, H, W, C, halo, dtype, x.device)
    v.interior().copy_(x.permute(0, 2, 3, 1))
    return v


# --------------------------------------------------------------------------------------
# weight packing: eval-mode BatchNorm folded into the conv, K-major rows
# --------------------------------------------------------------------------------------
def fold_norm(weight: torch.Tensor, bias: Optional[torch.Tensor], norm: Optional[dict],
              eps: float = 1e-5) -> Tuple[torch.Tensor, torch.Tensor]:
    """w' = w * g / sqrt(var + eps), b' = beta - mean * g / sqrt(var + eps)  (+ conv bias if any).
    ``norm``: dict(weight, bias, running_mean, running_var) or None."""
    w = weight.detach().float()
    cout = w.shape[0]
    b = bias.detach().float() if bias is not None else torch.zeros(cout, device=w.device)
    if norm is not None:
        scale = norm["weight"].float() / torch.sqrt(norm["running_var"].float() + eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = (b - norm["running_mean"].float()) * scale + norm["bias"].float()
    return w, b


def pack_conv(weight: torch.Tensor, bias: torch.Tensor, dtype: torch.dtype, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """[Cout,Cin,k,k] f32 -> packed [Cout_pad, K] rows in ``dtype`` + f32 bias [Cout_pad].

    Order of k inside a row: (kh, kw, cin); with bf16, Cin == 32, k == 3 the row is padded from 9 to 10 taps (zeros)
    because one 128-byte k-step then spans two taps.
    """
    cout, cin, k, _ = weight.shape
    cpad = lib().yolo_conv_cout_pad(cout)
    rows = weight.permute(0, 2, 3, 1).reshape(cout, k * k * cin)
    if dtype == torch.bfloat16 and cin == 32 and k == 3:
        rows = torch.cat([rows, rows.new_zeros(cout, 32)], dim=1)
    wp = torch.zeros((cpad, rows.shape[1]), dtype=dtype, device=device)
    wp[:cout] = rows.to(device=device, dtype=dtype)
    bp = torch.zeros(cpad, dtype=torch.float32, device=device)
    bp[:cout] = bias.to(device)
    return wp.contiguous(), bp


def conv_tile_shape(M: int, cin: int, cout: int, k: int, dtype: torch.dtype = torch.bfloat16) -> Tuple[int, int]:
    """(channels, pixels) of the workgroup tile yolo_conv2d_fwd picks for a cin -> cout, k x k convolution with M
    output pixels."""
    tc, tp = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().yolo_conv_tile_shape(int(M), int(cin), int(cout), int(k), dtype_code(dtype), ctypes.byref(tc),
                                     ctypes.byref(tp)), "yolo_conv_tile_shape")
    return tc.value, tp.value


def conv_kernel_name(x: "View", y: "View", cin: int, cout: int, k: int, stride: int, *, residual: bool = False,
                     upsample2x: bool = False) -> str:
    """Short name of the kernel instance yolo_conv2d_fwd launches for this convolution (tile and form)."""
    buf = ctypes.create_string_buffer(128)
    xv, yv = x.c(), y.c()
    check(lib().yolo_conv_kernel_name(ctypes.byref(xv), ctypes.byref(yv), x.N, cin, cout, k, stride, int(upsample2x),
                                      dtype_code(x.buf.dtype), buf, 128), "yolo_conv_kernel_name")
    return buf.value.decode()


def conv_bnsums_rows(x: "View", y: "View", cin: int, cout: int, k: int, stride: int = 1) -> int:
    """Rows of per-tile sums conv2d_fwd_bnsums / conv2d_fwd_bnbwd_sums write for this convolution (x -> y)."""
    xv, yv = x.c(), y.c()
    return lib().yolo_conv2d_bnsums_rows(ctypes.byref(xv), ctypes.byref(yv), x.N, int(cin), int(cout), int(k), int(stride),
                                         dtype_code(x.buf.dtype))


def pack_stem(weight: torch.Tensor, bias: torch.Tensor, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """[Cout,3,3,3] -> [kh,kw,cin,Cout] f32."""
    return (weight.permute(2, 3, 1, 0).contiguous().float().to(device),
            bias.float().contiguous().to(device))


# --------------------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------------------
def conv2d_fwd(x: View, w_packed: torch.Tensor, bias: torch.Tensor, y: View, *, cin: int, cout: int,
               k: int, stride: int, slope: float = 0.1, residual: Optional[View] = None,
               upsample2x: bool = False) -> None:
    _require_cuda(x.buf, "conv2d_fwd")
    xv, yv = x.c(), y.c()
    rv = residual.c() if residual is not None else None
    code = lib().yolo_conv2d_fwd(ctypes.byref(xv), _ptr(w_packed), _ptr(bias),
                                 ctypes.byref(rv) if rv is not None else None, ctypes.byref(yv),
                                 x.N, cin, cout, k, stride, float(slope), int(upsample2x),
                                 dtype_code(x.buf.dtype), dtype_code(y.buf.dtype), _stream())
    check(code, "yolo_conv2d_fwd")


def conv_splitk_plan(M: int, cin: int, cout: int, k: int, stride: int, dtype: torch.dtype) -> Tuple[int, int]:
    """(splits, workspace bytes) yolo_conv2d_fwd_splitk uses for a convolution with M output pixels; splits == 1 means
    the plain launch is the better one."""
    splits, nbytes = ctypes.c_int(0), ctypes.c_size_t(0)
    check(lib().yolo_conv2d_splitk_plan(int(M), cin, cout, k, stride, dtype_code(dtype), ctypes.byref(splits),
                                        ctypes.byref(nbytes)), "yolo_conv2d_splitk_plan")
    return splits.value, nbytes.value


def conv2d_fwd_splitk(x: View, w_packed: torch.Tensor, bias: torch.Tensor, y: View, workspace: torch.Tensor, *, cin: int,
                      cout: int, k: int, stride: int, slope: float = 0.1, residual: Optional[View] = None,
                      upsample2x: bool = False) -> None:
    """conv2d_fwd with the k loop of every tile split over several workgroups (few-pixel layers, e.g. batch 1)."""
    _require_cuda(x.buf, "conv2d_fwd_splitk")
    xv, yv = x.c(), y.c()
    rv = residual.c() if residual is not None else None
    code = lib().yolo_conv2d_fwd_splitk(ctypes.byref(xv), _ptr(w_packed), _ptr(bias),
                                        ctypes.byref(rv) if rv is not None else None, ctypes.byref(yv),
                                        x.N, cin, cout, k, stride, float(slope), int(upsample2x),
                                        dtype_code(x.buf.dtype), dtype_code(y.buf.dtype), _ptr(workspace),
                                        workspace.numel() * workspace.element_size(), _stream())
    check(code, "yolo_conv2d_fwd_splitk")


def stem_fwd(images: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, y: View, *,
             mean: Sequence[float], std: Sequence[float], slope: float = 0.1) -> None:
    """images: [N,3,Hs,Ws] contiguous, float32 or uint8, values 0..255."""
    _require_cuda(images, "stem_fwd")
    assert images.is_contiguous() and images.dim() == 4 and images.shape[1] == 3
    if images.dtype == torch.float32:
        code_dt = 0
    elif images.dtype == torch.uint8:
        code_dt = 1
    else:
        raise ValueError(f"stem_fwd: image dtype {images.dtype}")
    N, _, Hs, Ws = images.shape
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    yv = y.c()
    code = lib().yolo_stem_fwd(_ptr(images), code_dt, N, Hs, Ws, m, s, _ptr(w), _ptr(bias),
                               ctypes.byref(yv), w.shape[-1], float(slope), dtype_code(y.buf.dtype), _stream())
    check(code, "yolo_stem_fwd")


def stem_down_fwd(images: torch.Tensor, w_stem: torch.Tensor, b_stem: torch.Tensor, w_down: torch.Tensor,
                  b_down: torch.Tensor, y: View, *, padded_hw: Tuple[int, int], mean: Sequence[float],
                  std: Sequence[float], slope: float = 0.1) -> None:
    """Stem + first stride-2 conv in one launch (bf16).  ``w_down``/``b_down`` from pack_conv(bf16), Cin 32."""
    _require_cuda(images, "stem_down_fwd")
    assert images.is_contiguous() and images.dim() == 4 and images.shape[1] == 3
    assert y.buf.dtype == torch.bfloat16 and w_down.dtype == torch.bfloat16 and w_down.shape[1] == 320
    if images.dtype == torch.float32:
        code_dt = 0
    elif images.dtype == torch.uint8:
        code_dt = 1
    else:
        raise ValueError(f"stem_down_fwd: image dtype {images.dtype}")
    N, _, Hs, Ws = images.shape
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    yv = y.c()
    code = lib().yolo_stem_down_fwd(_ptr(images), code_dt, N, Hs, Ws, m, s, _ptr(w_stem), _ptr(b_stem),
                                    _ptr(w_down), _ptr(b_down), ctypes.byref(yv), int(padded_hw[0]),
                                    int(padded_hw[1]), float(slope), _stream())
    check(code, "yolo_stem_down_fwd")


def resblock_c64_fwd(x: View, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, y: View, *,
                     slope: float = 0.1) -> None:
    """One 64-channel residual block (1x1 64->32, 3x3 32->64, + x) in one launch; y must not alias x."""
    _require_cuda(x.buf, "resblock_c64_fwd")
    assert x.buf.dtype == torch.bfloat16 and y.buf.dtype == torch.bfloat16
    assert w1.dtype == torch.bfloat16 and w1.shape[1] == 64 and w2.dtype == torch.bfloat16 and w2.shape[1] == 320
    xv, yv = x.c(), y.c()
    check(lib().yolo_resblock_c64_fwd(ctypes.byref(xv), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), ctypes.byref(yv),
                                      x.N, float(slope), _stream()), "yolo_resblock_c64_fwd")


def upsample2x(x: View, y: View) -> None:
    _require_cuda(x.buf, "upsample2x")
    xv, yv = x.c(), y.c()
    check(lib().yolo_upsample2x_nhwc(ctypes.byref(xv), ctypes.byref(yv), x.N, x.C, dtype_code(x.buf.dtype),
                                     _stream()), "yolo_upsample2x_nhwc")


def _views_array(heads: List[View]):
    arr = (YoloView * len(heads))()
    for i, h in enumerate(heads):
        arr[i] = h.c()
    return arr


def decode_workspace_bytes(heads: List[View]) -> int:
    return int(lib().yolo_decode_workspace_bytes(_views_array(heads), len(heads), heads[0].N))


@dataclass
class Candidates:
    boxes: torch.Tensor        # [cap, 4] f32
    scores: torch.Tensor       # [cap] f32
    classes: torch.Tensor      # [cap] i32
    img_offsets: torch.Tensor  # [N+1] i32


def decode(heads: List[View], strides: Sequence[int], anchors_wh: Sequence[Sequence[Tuple[float, float]]],
           num_classes: int, thr: float, cap: int, *, out: Optional[Candidates] = None,
           workspace: Optional[torch.Tensor] = None) -> Candidates:
    """heads: level views ordered fine -> coarse (p3, p4, p5)."""
    dev = heads[0].buf.device
    _require_cuda(heads[0].buf, "decode")
    N, L, A = heads[0].N, len(heads), len(anchors_wh[0])
    if out is None:
        out = Candidates(torch.empty((cap, 4), dtype=torch.float32, device=dev),
                         torch.empty((cap,), dtype=torch.float32, device=dev),
                         torch.empty((cap,), dtype=torch.int32, device=dev),
                         torch.empty((N + 1,), dtype=torch.int32, device=dev))
    need = decode_workspace_bytes(heads)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    st = (ctypes.c_int * L)(*[int(s) for s in strides])
    flat = [float(v) for lvl in anchors_wh for wh in lvl for v in wh]
    an = (ctypes.c_float * len(flat))(*flat)
    code = lib().yolo_decode(_views_array(heads), L, st, an, N, A, num_classes, float(thr),
                             dtype_code(heads[0].buf.dtype), _ptr(out.boxes), _ptr(out.scores),
                             _ptr(out.classes), _ptr(out.img_offsets), cap, _ptr(workspace),
                             workspace.numel(), _stream())
    check(code, "yolo_decode")
    return out


def decode_noalloc(heads: List[View], strides: Sequence[int], anchors_wh: Sequence[Sequence[Tuple[float, float]]],
                   num_classes: int, thr: float, out: Candidates) -> Candidates:
    """``decode`` through yolo_decode_noalloc: no workspace tensor, nothing allocated by the call (the caller owns
    ``out``); batch x grid cells is limited by the library's built-in scratch array."""
    _require_cuda(heads[0].buf, "decode_noalloc")
    N, L, A = heads[0].N, len(heads), len(anchors_wh[0])
    st = (ctypes.c_int * L)(*[int(s) for s in strides])
    flat = [float(v) for lvl in anchors_wh for wh in lvl for v in wh]
    an = (ctypes.c_float * len(flat))(*flat)
    check(lib().yolo_decode_noalloc(_views_array(heads), L, st, an, N, A, num_classes, float(thr),
                                    dtype_code(heads[0].buf.dtype), _ptr(out.boxes), _ptr(out.scores),
                                    _ptr(out.classes), _ptr(out.img_offsets), out.scores.numel(), _stream()),
          "yolo_decode_noalloc")
    return out


def nms_workspace_bytes(N: int, M: int) -> int:
    return int(lib().yolo_nms_workspace_bytes(N, M))


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, classes: torch.Tensor, img_offsets: torch.Tensor,
                M: int, iou_thr: float, *, num_classes: int, mode: int = NMS_PER_CLASS, auto_limit: int = 20000,
                top_k: int = 0, keep_stride: Optional[int] = None, workspace: Optional[torch.Tensor] = None,
                keep: Optional[torch.Tensor] = None, n_keep: Optional[torch.Tensor] = None):
    """Returns (keep [N, keep_stride] i32 image-relative indices, n_keep [N] i32).  ``M`` is an upper bound of the
    number of candidates in the arrays (their capacity will do): the counts are read on the device from
    ``img_offsets``, nothing is read back."""
    _require_cuda(img_offsets, "batched_nms")
    dev = img_offsets.device
    N = img_offsets.numel() - 1
    if keep_stride is None:
        keep_stride = top_k if top_k > 0 else max(M, 1)
    if keep is None:
        keep = torch.empty((N, keep_stride), dtype=torch.int32, device=dev)
    if n_keep is None:
        n_keep = torch.empty((N,), dtype=torch.int32, device=dev)
    if M > 0:
        need = nms_workspace_bytes(N, M)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    code = lib().yolo_batched_nms(_ptr(boxes), _ptr(scores), _ptr(classes), _ptr(img_offsets), N, int(M),
                                  float(iou_thr), int(mode), int(auto_limit), int(top_k), int(num_classes),
                                  _ptr(keep), keep_stride, _ptr(n_keep),
                                  _ptr(workspace) if workspace is not None else None,
                                  workspace.numel() if workspace is not None else 0, _stream())
    check(code, "yolo_batched_nms")
    return keep, n_keep


def gather_detections(cand: Candidates, keep: torch.Tensor, n_keep: torch.Tensor, in_hw: torch.Tensor,
                      out_hw: torch.Tensor):
    """Returns (det [N, keep_stride, 6] f32 = x0,y0,x1,y1,score,class ; n_det [N] i32)."""
    dev = keep.device
    N, stride = keep.shape
    det = torch.empty((N, stride, 6), dtype=torch.float32, device=dev)
    n_det = torch.empty((N,), dtype=torch.int32, device=dev)
    code = lib().yolo_gather_detections(_ptr(cand.boxes), _ptr(cand.scores), _ptr(cand.classes),
                                        _ptr(cand.img_offsets), cand.scores.numel(), _ptr(keep), stride, _ptr(n_keep), N,
                                        _ptr(in_hw), _ptr(out_hw), _ptr(det), _ptr(n_det), _stream())
    check(code, "yolo_gather_detections")
    return det, n_det


# --------------------------------------------------------------------------------------
# training-path operators (batch-statistics BatchNorm + LeakyReLU, conv weight / data gradients)
# --------------------------------------------------------------------------------------
def bn_workspace_bytes(N: int, H: int, W: int, C: int) -> int:
    return int(lib().yolo_bn_workspace_bytes(N, H, W, C))


def bn_train_stats(z: View, gamma: torch.Tensor, beta: torch.Tensor, stats: torch.Tensor, workspace: torch.Tensor, *,
                   eps: float = 1e-5, momentum: float = 0.1, running_mean: Optional[torch.Tensor] = None,
                   running_var: Optional[torch.Tensor] = None) -> None:
    """stats [5, C] f32 <- mean, biased var, invstd, scale, shift of z over (N, H, W); updates running stats."""
    _require_cuda(z.buf, "bn_train_stats")
    assert stats.dtype == torch.float32 and stats.numel() >= 5 * z.C and stats.is_contiguous()
    zv = z.c()
    check(lib().yolo_bn_train_stats(ctypes.byref(zv), z.N, z.C, dtype_code(z.buf.dtype), _ptr(gamma), _ptr(beta),
                                    float(eps), float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(stats),
                                    _ptr(workspace), workspace.numel() * workspace.element_size(), _stream()),
          "yolo_bn_train_stats")


def bn_act_fwd(z: View, stats: torch.Tensor, y: View, *, slope: float = 0.1, residual: Optional[View] = None) -> None:
    _require_cuda(z.buf, "bn_act_fwd")
    assert z.buf.dtype == y.buf.dtype
    zv, yv = z.c(), y.c()
    rv = residual.c() if residual is not None else None
    check(lib().yolo_bn_act_fwd(ctypes.byref(zv), _ptr(stats), ctypes.byref(rv) if rv is not None else None,
                                ctypes.byref(yv), z.N, z.C, float(slope), dtype_code(z.buf.dtype), _stream()),
          "yolo_bn_act_fwd")


def bn_act_bwd(dy: View, z: View, stats: torch.Tensor, dz: View, dgamma: torch.Tensor, dbeta: torch.Tensor,
               workspace: torch.Tensor, *, slope: float = 0.1) -> None:
    _require_cuda(z.buf, "bn_act_bwd")
    assert dy.buf.dtype == z.buf.dtype == dz.buf.dtype
    assert dgamma.dtype == torch.float32 and dbeta.dtype == torch.float32
    dyv, zv, dzv = dy.c(), z.c(), dz.c()
    check(lib().yolo_bn_act_bwd(ctypes.byref(dyv), ctypes.byref(zv), _ptr(stats), ctypes.byref(dzv), z.N, z.C,
                                float(slope), dtype_code(z.buf.dtype), _ptr(dgamma), _ptr(dbeta), _ptr(workspace),
                                workspace.numel() * workspace.element_size(), _stream()), "yolo_bn_act_bwd")


def wgrad_workspace_bytes(M: int, cin: int, cout: int, k: int) -> int:
    return int(lib().yolo_conv2d_wgrad_workspace_bytes(M, cin, cout, k))


def conv2d_wgrad(x: View, dz: View, dw: torch.Tensor, workspace: torch.Tensor, *, cin: int, cout: int, k: int,
                 stride: int, accumulate: bool = False) -> None:
    """dw: f32 [cout, k, k, cin] (the library's layout; ``dw.permute(0, 3, 1, 2)`` is torch's [cout, cin, k, k])."""
    _require_cuda(x.buf, "conv2d_wgrad")
    assert x.buf.dtype == dz.buf.dtype
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == cout * k * k * cin
    xv, dzv = x.c(), dz.c()
    check(lib().yolo_conv2d_wgrad(ctypes.byref(xv), ctypes.byref(dzv), x.N, cin, cout, k, stride,
                                  dtype_code(x.buf.dtype), _ptr(dw),
                                  int(accumulate), _ptr(workspace), workspace.numel() * workspace.element_size(),
                                  _stream()), "yolo_conv2d_wgrad")


def stem_wgrad_workspace_bytes() -> int:
    return int(lib().yolo_stem_wgrad_workspace_bytes())


def stem_wgrad(images: torch.Tensor, dz: View, dw: torch.Tensor, workspace: torch.Tensor, *, mean: Sequence[float],
               std: Sequence[float], round_input_bf16: bool, accumulate: bool = False) -> None:
    """Weight gradient of the 3 -> 32 stem convolution from the images themselves (normalised on the fly, as
    stem_fwd does) and the gradient of its output; dw: f32 [32, 3, 3, 3] in the library's [cout, kh, kw, cin] layout."""
    _require_cuda(images, "stem_wgrad")
    assert images.is_contiguous() and images.dim() == 4 and images.shape[1] == 3
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == 32 * 27
    code_dt = {torch.float32: 0, torch.uint8: 1}.get(images.dtype)
    if code_dt is None:
        raise ValueError(f"stem_wgrad: image dtype {images.dtype}")
    N, _, Hs, Ws = images.shape
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    dzv = dz.c()
    check(lib().yolo_stem_wgrad(_ptr(images), code_dt, N, Hs, Ws, m, s, ctypes.byref(dzv), dtype_code(dz.buf.dtype),
                                int(round_input_bf16), _ptr(dw), int(accumulate), _ptr(workspace),
                                workspace.numel() * workspace.element_size(), _stream()), "yolo_stem_wgrad")


def pack_conv_dgrad(weight: torch.Tensor, dtype: torch.dtype, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Packed weights that make yolo_conv2d_fwd compute the DATA gradient of a conv with ``weight``
    [Cout, Cin, k, k]: channels swapped and taps rotated by 180 degrees, zero bias."""
    wt = weight.detach().float().flip(2, 3).permute(1, 0, 2, 3).contiguous()  # [Cin, Cout, k, k]
    return pack_conv(wt, torch.zeros(wt.shape[0]), dtype, device)


def dilate2x(x: View, y: View) -> None:
    _require_cuda(x.buf, "dilate2x")
    xv, yv = x.c(), y.c()
    check(lib().yolo_dilate2x_nhwc(ctypes.byref(xv), ctypes.byref(yv), x.N, x.C, dtype_code(x.buf.dtype), _stream()),
          "yolo_dilate2x_nhwc")


def sum2x2(x: View, y: View) -> None:
    _require_cuda(x.buf, "sum2x2")
    xv, yv = x.c(), y.c()
    check(lib().yolo_sum2x2_nhwc(ctypes.byref(xv), ctypes.byref(yv), x.N, x.C, dtype_code(x.buf.dtype), _stream()),
          "yolo_sum2x2_nhwc")


def pack_conv_train(weight: torch.Tensor, w_fwd: Optional[torch.Tensor], w_dgrad: Optional[torch.Tensor], *,
                    cout_pad: int) -> None:
    """One launch: f32 master weights [Cout,Cin,k,k] -> forward and data-gradient operand buffers (2-D, zeroed once
    by the caller; see pack_conv / pack_conv_dgrad for the layouts)."""
    _require_cuda(weight, "pack_conv_train")
    w = weight.detach()
    assert w.dtype == torch.float32 and w.is_contiguous()
    cout, cin, k, _ = w.shape
    ref = w_fwd if w_fwd is not None else w_dgrad
    check(lib().yolo_pack_conv_train(_ptr(w), cout, cin, k, dtype_code(ref.dtype), _ptr(w_fwd),
                                     0 if w_fwd is None else w_fwd.shape[1], _ptr(w_dgrad),
                                     0 if w_dgrad is None else w_dgrad.shape[1], cout_pad, _stream()),
          "yolo_pack_conv_train")


class PackDesc(ctypes.Structure):
    """yolo_pack_desc of include/yolo_hip.h."""
    _fields_ = [("w", ctypes.c_void_p), ("w_fwd", ctypes.c_void_p), ("w_dgrad", ctypes.c_void_p),
                ("Cout", ctypes.c_int), ("Cin", ctypes.c_int), ("k", ctypes.c_int), ("fwd_row", ctypes.c_int),
                ("dgrad_row", ctypes.c_int), ("Cout_pad", ctypes.c_int), ("first_tile", ctypes.c_int),
                ("reserved", ctypes.c_int)]


def pack_table(entries, device):
    """entries: (weight [Cout,Cin,k,k] f32 stored channels_last, w_fwd 2-D, w_dgrad 2-D or None, cout_pad) per layer.
    Returns (device table, number of layers, total tiles) for pack_conv_train_batch; the tensors must stay alive
    and in place for as long as the table is used."""
    arr = (PackDesc * len(entries))()
    tiles = 0
    for d, (w, w_fwd, w_dgrad, cout_pad) in zip(arr, entries):
        cout, cin, k, _ = w.shape
        assert w.dtype == torch.float32 and w.permute(0, 2, 3, 1).is_contiguous(), "master weights must be channels_last"
        d.w, d.w_fwd, d.w_dgrad = w.data_ptr(), w_fwd.data_ptr(), (0 if w_dgrad is None else w_dgrad.data_ptr())
        d.Cout, d.Cin, d.k, d.fwd_row = cout, cin, k, w_fwd.shape[1]
        d.dgrad_row, d.Cout_pad = (0 if w_dgrad is None else w_dgrad.shape[1]), cout_pad
        assert d.fwd_row >= k * k * cin and (w_dgrad is None or d.dgrad_row >= k * k * cout_pad)
        d.first_tile = tiles
        tiles += ((cout + 63) // 64) * k * k * ((cin + 63) // 64)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    return raw, len(entries), tiles


def pack_conv_train_batch(table: torch.Tensor, n_layers: int, total_tiles: int, dtype: torch.dtype) -> None:
    """One launch for every trainable convolution: channels_last f32 master weights -> forward and data-gradient
    operand buffers (see pack_table)."""
    _require_cuda(table, "pack_conv_train_batch")
    check(lib().yolo_pack_conv_train_batch(_ptr(table), n_layers, total_tiles, dtype_code(dtype), _stream()),
          "yolo_pack_conv_train_batch")


def _view_array(views):
    arr = (YoloView * len(views))()
    for i, v in enumerate(views):
        arr[i] = v.c()
    return arr


@dataclass
class LossContext:
    """What train_loss_fwd leaves for train_loss_bwd: the call's arguments and the workspace holding the labels."""
    heads: list
    strides: list
    anchors: torch.Tensor
    gt_boxes: torch.Tensor
    gt_classes: torch.Tensor
    gt_count: torch.Tensor
    A: int
    C: int
    obj_weight: float
    noobj_weight: float
    ws: torch.Tensor
    counts: torch.Tensor


def train_loss_fwd(heads: Sequence[View], strides: Sequence[int], anchors: torch.Tensor, gt_boxes: torch.Tensor,
                   gt_classes: torch.Tensor, gt_count: torch.Tensor, *, num_anchors: int, num_classes: int,
                   match_thr: float, obj_weight: float, noobj_weight: float):
    """Label assignment + the three losses of a batch in a handful of launches, nothing read back.
    heads: f32 head views (finest stride first); anchors [R,4] f32; gt_boxes [N,K,4] f32, gt_classes [N,K] i32,
    gt_count [N] i32.  Returns (losses [3] f32: confidence, class, box; LossContext)."""
    _require_cuda(anchors, "train_loss_fwd")
    N, K = gt_classes.shape
    R = anchors.shape[0]
    assert anchors.dtype == torch.float32 and anchors.is_contiguous() and anchors.shape[1] == 4
    assert gt_boxes.dtype == torch.float32 and gt_boxes.is_contiguous() and gt_boxes.shape == (N, K, 4)
    assert gt_classes.dtype == torch.int32 and gt_classes.is_contiguous()
    assert gt_count.dtype == torch.int32 and gt_count.shape == (N,)
    assert all(h.buf.dtype == torch.float32 and h.N == N for h in heads)
    dev = anchors.device
    ws = torch.empty(lib().yolo_train_loss_workspace_bytes(N, K, R), dtype=torch.uint8, device=dev)
    losses = torch.empty(3, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    st = (ctypes.c_int * len(strides))(*strides)
    check(lib().yolo_train_loss_fwd(_view_array(heads), len(heads), st, _ptr(anchors), R, N, num_anchors, num_classes,
                                    _ptr(gt_boxes), _ptr(gt_classes), _ptr(gt_count), K, match_thr, obj_weight,
                                    noobj_weight, _ptr(losses), _ptr(counts), _ptr(ws), ws.numel(), _stream()),
          "yolo_train_loss_fwd")
    ctx = LossContext(list(heads), list(strides), anchors, gt_boxes, gt_classes, gt_count, num_anchors, num_classes,
                      obj_weight, noobj_weight, ws, counts)
    return losses, ctx


def train_loss_bwd(ctx: LossContext, dheads: Sequence[View], grad_losses: torch.Tensor) -> None:
    """Writes d(sum_i grad_losses[i] * loss_i)/d(head) into the gradient views (every element, zeros included)."""
    assert grad_losses.dtype == torch.float32 and grad_losses.numel() == 3 and grad_losses.is_cuda
    N, K = ctx.gt_classes.shape
    st = (ctypes.c_int * len(ctx.strides))(*ctx.strides)
    check(lib().yolo_train_loss_bwd(_view_array(ctx.heads), _view_array(dheads), dtype_code(dheads[0].buf.dtype),
                                    len(ctx.heads), st, _ptr(ctx.anchors), ctx.anchors.shape[0], N, ctx.A, ctx.C,
                                    _ptr(ctx.gt_boxes), _ptr(ctx.gt_classes), _ptr(ctx.gt_count), K, ctx.obj_weight,
                                    ctx.noobj_weight, _ptr(grad_losses.contiguous()), _ptr(ctx.ws), ctx.ws.numel(),
                                    _stream()), "yolo_train_loss_bwd")


def conv2d_fwd_bnsums(x: View, w_packed: torch.Tensor, bias: torch.Tensor, z: View, sums: torch.Tensor, *, cin: int,
                      cout: int, k: int, stride: int) -> int:
    """z = conv(x) (linear) and, from the same launch, the per-pixel-tile sums of z and z*z that a training-mode
    BatchNorm needs.  Returns the number of rows written to ``sums`` ([rows, 2, cout] f32)."""
    _require_cuda(x.buf, "conv2d_fwd_bnsums")
    assert sums.dtype == torch.float32 and sums.is_contiguous() and z.buf.dtype == x.buf.dtype
    rows = conv_bnsums_rows(x, z, cin, cout, k, stride)
    xv, zv = x.c(), z.c()
    check(lib().yolo_conv2d_fwd_bnsums(ctypes.byref(xv), _ptr(w_packed), _ptr(bias), ctypes.byref(zv), x.N, cin, cout, k,
                                       stride, dtype_code(x.buf.dtype), _ptr(sums), sums.numel() * 4, _stream()),
          "yolo_conv2d_fwd_bnsums")
    return rows


def conv2d_fwd_bnbwd_sums(x: View, w_packed: torch.Tensor, bias: torch.Tensor, y: View, sums: torch.Tensor, *, cin: int,
                          cout: int, k: int, z: View, stats: torch.Tensor, act_slope: float = 0.1,
                          residual: Optional[View] = None) -> int:
    """y = conv(x) (+ residual), linear, stride 1, bf16 -- a data-gradient convolution whose output is the gradient
    of the activation leaky(bn(z)) -- and, from the same launch, the per-pixel-tile sums (g, g*z) of that BatchNorm's
    backward (``stats`` = the producer's [5, cout] statistics).  Returns the number of rows written to ``sums``."""
    _require_cuda(x.buf, "conv2d_fwd_bnbwd_sums")
    assert x.buf.dtype == torch.bfloat16 and y.buf.dtype == torch.bfloat16 and z.buf.dtype == torch.bfloat16
    assert sums.dtype == torch.float32 and sums.is_contiguous()
    assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() >= 5 * cout
    rows = conv_bnsums_rows(x, y, cin, cout, k, 1)
    xv, yv, zv = x.c(), y.c(), z.c()
    rv = residual.c() if residual is not None else None
    flat = stats.view(-1)
    check(lib().yolo_conv2d_fwd_bnbwd_sums(ctypes.byref(xv), _ptr(w_packed), _ptr(bias),
                                           ctypes.byref(rv) if rv is not None else None, ctypes.byref(yv), x.N, cin, cout,
                                           k, dtype_code(x.buf.dtype), ctypes.byref(zv), _ptr(flat[3 * cout:]),
                                           _ptr(flat[4 * cout:]), float(act_slope), _ptr(sums), sums.numel() * 4,
                                           _stream()), "yolo_conv2d_fwd_bnbwd_sums")
    return rows


def bn_act_bwd_presummed(dy: View, z: View, stats: torch.Tensor, dz: View, dgamma: torch.Tensor, dbeta: torch.Tensor,
                         sums: torch.Tensor, rows: int, *, slope: float = 0.1) -> None:
    """bn_act_bwd when the reduction pass came out of the epilogue that wrote ``dy`` (conv2d_fwd_bnbwd_sums)."""
    _require_cuda(z.buf, "bn_act_bwd_presummed")
    assert dy.buf.dtype == z.buf.dtype == dz.buf.dtype
    assert dgamma.dtype == torch.float32 and dbeta.dtype == torch.float32 and sums.dtype == torch.float32
    assert sums.numel() >= rows * 2 * z.C
    dyv, zv, dzv = dy.c(), z.c(), dz.c()
    check(lib().yolo_bn_act_bwd_presummed(ctypes.byref(dyv), ctypes.byref(zv), _ptr(stats), ctypes.byref(dzv), z.N, z.C,
                                          float(slope), dtype_code(z.buf.dtype), _ptr(dgamma), _ptr(dbeta), _ptr(sums),
                                          int(rows), _stream()), "yolo_bn_act_bwd_presummed")


def bn_train_finalize(sums: torch.Tensor, rows: int, count: int, gamma: torch.Tensor, beta: torch.Tensor,
                      stats: torch.Tensor, *, eps: float, momentum: float, running_mean=None, running_var=None) -> None:
    """Per-tile sums -> stats [5, C] (mean, biased var, invstd, scale, shift) + running-statistics update."""
    C = gamma.numel()
    assert stats.dtype == torch.float32 and stats.numel() >= 5 * C and stats.is_contiguous()
    check(lib().yolo_bn_train_finalize(_ptr(sums), rows, count, C, _ptr(gamma), _ptr(beta), eps, momentum,
                                       _ptr(running_mean), _ptr(running_var), _ptr(stats), _stream()),
          "yolo_bn_train_finalize")


def conv3x3s2_dgrad(dz: View, w_dgrad: torch.Tensor, bias: torch.Tensor, dx: View, *, cin: int, cout: int,
                    residual: Optional[View] = None) -> None:
    """Data gradient of a 3x3 / stride-2 convolution straight from the undilated gradient (bf16): dx [2H x 2W, cout]
    from dz [H x W, cin]; ``residual`` is added (another consumer's gradient)."""
    _require_cuda(dz.buf, "conv3x3s2_dgrad")
    assert dz.buf.dtype == torch.bfloat16 and dx.buf.dtype == torch.bfloat16 and w_dgrad.dtype == torch.bfloat16
    zv, xv = dz.c(), dx.c()
    rv = residual.c() if residual is not None else None
    check(lib().yolo_conv3x3s2_dgrad(ctypes.byref(zv), _ptr(w_dgrad), _ptr(bias), ctypes.byref(rv) if rv is not None else None,
                                     ctypes.byref(xv), dz.N, cin, cout, dtype_code(dz.buf.dtype), _stream()),
          "yolo_conv3x3s2_dgrad")


# ---------------------------------------------------------------------------------------------------------
# training-image augmentation (uint8 HWC images on the device)
# ---------------------------------------------------------------------------------------------------------
def _check_image(t: torch.Tensor, what: str) -> None:
    _require_cuda(t, what)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3 or t.stride(2) != 1 or t.stride(1) != 3:
        raise ValueError(f"{what}: expected a uint8 [H, W, 3] tensor with packed pixels, got {t.dtype} "
                         f"{tuple(t.shape)} strides {tuple(t.stride())}")


def aug_letterbox(src: torch.Tensor, out_hw: Tuple[int, int], top: int, left: int, resized_hw: Tuple[int, int],
                  border: int = 114, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cv2-style linear resize of ``src`` to ``resized_hw`` placed at (top, left) of an ``out_hw`` image of ``border``."""
    _check_image(src, "aug_letterbox")
    if out is None:
        out = torch.empty((out_hw[0], out_hw[1], 3), dtype=torch.uint8, device=src.device)
    _check_image(out, "aug_letterbox")
    check(lib().yolo_aug_letterbox(_ptr(src), src.shape[0], src.shape[1], src.stride(0), _ptr(out), out.shape[0],
                                   out.shape[1], out.stride(0), top, left, resized_hw[0], resized_hw[1], border, _stream()),
          "yolo_aug_letterbox")
    return out


_RESAMPLE_TABLES = {}


def resample_tables(in_size: int, out_size: int, device) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """Device tables of one Pillow-style bilinear pass (first source index + count, 22-bit coefficients) for an axis
    going from ``in_size`` to ``out_size``; cached per (sizes, device)."""
    key = (in_size, out_size, str(device))
    hit = _RESAMPLE_TABLES.get(key)
    if hit is None:
        import numpy as np
        scale = in_size / out_size
        filterscale = max(scale, 1.0)
        support = 1.0 * filterscale
        ksize = int(math.ceil(support)) * 2 + 1
        xx = np.arange(out_size, dtype=np.float64)
        center = (xx + 0.5) * scale
        xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
        xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
        cnt = xmax - xmin
        k = np.arange(ksize, dtype=np.float64)[None, :]
        t = np.abs((k + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
        w = np.where((t < 1.0) & (k < cnt[:, None]), 1.0 - t, 0.0)
        # Pillow sums the weights left to right; numpy's pairwise sum may differ in the last bit, so accumulate alike
        ww = np.zeros(out_size)
        for i in range(ksize):
            ww = ww + w[:, i]
        w = np.where(ww[:, None] != 0.0, w / np.where(ww[:, None] != 0.0, ww[:, None], 1.0), w)
        coef = (w * float(1 << 22) + 0.5).astype(np.int64).astype(np.int32)  # weights are >= 0: truncation = Pillow's rounding
        bounds = np.stack([xmin, cnt], 1).astype(np.int32)
        hit = (torch.from_numpy(bounds).to(device), torch.from_numpy(np.ascontiguousarray(coef)).to(device), ksize)
        _RESAMPLE_TABLES[key] = hit
    return hit


def aug_resample(src: torch.Tensor, out_hw: Tuple[int, int], out: Optional[torch.Tensor] = None,
                 workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Pillow ``Image.resize((w, h), BILINEAR)`` of a uint8 HWC device image."""
    _check_image(src, "aug_resample")
    H, W = src.shape[0], src.shape[1]
    nh, nw = out_hw
    if out is None:
        out = torch.empty((nh, nw, 3), dtype=torch.uint8, device=src.device)
    _check_image(out, "aug_resample")
    xb = xc = yb = yc = None
    xk = yk = 0
    if nw != W:
        xb, xc, xk = resample_tables(W, nw, src.device)
    if nh != H:
        yb, yc, yk = resample_tables(H, nh, src.device)
    need = int(lib().yolo_aug_resample_workspace_bytes(H, nw)) if (nw != W and nh != H) else 0
    if need and (workspace is None or workspace.numel() < need):
        workspace = torch.empty(need, dtype=torch.uint8, device=src.device)
    check(lib().yolo_aug_resample(_ptr(src), H, W, src.stride(0), _ptr(out), nh, nw, out.stride(0), _ptr(xb), _ptr(xc), xk,
                                  _ptr(yb), _ptr(yc), yk, _ptr(workspace) if need else None,
                                  workspace.numel() if need else 0, _stream()),
          "yolo_aug_resample")
    return out


def aug_resample_batch(srcs: Sequence[torch.Tensor], out_hws: Sequence[Tuple[int, int]],
                       outs: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
    """``aug_resample`` for many images with two launches per 16 images."""
    n = len(srcs)
    if n == 0:
        return []
    dev = srcs[0].device
    if outs is None:
        outs = [torch.empty((h, w, 3), dtype=torch.uint8, device=dev) for h, w in out_hws]
    need = [int(s.shape[0]) * w * 3 if (s.shape[0] != h and s.shape[1] != w) else 0 for s, (h, w) in zip(srcs, out_hws)]
    ws = torch.empty(max(sum(need), 1), dtype=torch.uint8, device=dev)
    arr = (_lib.AugResampleJob * n)()
    off = 0
    for j, s, o, (nh, nw), nbytes in zip(arr, srcs, outs, out_hws, need):
        _check_image(s, "aug_resample_batch")
        _check_image(o, "aug_resample_batch")
        assert tuple(o.shape[:2]) == (nh, nw), (o.shape, nh, nw)
        H, W = int(s.shape[0]), int(s.shape[1])
        j.src, j.dst = s.data_ptr(), o.data_ptr()
        j.src_stride, j.dst_stride = s.stride(0), o.stride(0)
        j.src_height, j.src_width, j.dst_height, j.dst_width = H, W, nh, nw
        if nw != W:
            xb, xc, j.x_ksize = resample_tables(W, nw, dev)
            j.x_bounds, j.x_coef = xb.data_ptr(), xc.data_ptr()
        if nh != H:
            yb, yc, j.y_ksize = resample_tables(H, nh, dev)
            j.y_bounds, j.y_coef = yb.data_ptr(), yc.data_ptr()
        if nbytes:
            j.workspace, j.workspace_bytes = ws.data_ptr() + off, nbytes
            off += nbytes
    check(lib().yolo_aug_resample_batch(ctypes.cast(arr, ctypes.c_void_p), n, _stream()), "yolo_aug_resample_batch")
    return list(outs)


def aug_job(src: torch.Tensor, dst: torch.Tensor, *, dst_xy: Tuple[int, int] = (0, 0),
            crop: Optional[Tuple[int, int, int, int]] = None, inverse_map=None, perspective: bool = False,
            jitter: Optional[Tuple[float, float, float]] = None, rgb: bool = False, flip_horizontal: bool = False,
            flip_vertical: bool = False, border: int = 114, planar: bool = False,
            reverse_channels: bool = False) -> "_lib.AugJob":
    """One image's fused warp -> colour jitter -> flip -> paste.  ``inverse_map``: 6 (affine) or 9 (perspective)
    doubles taking output pixels to source pixels; ``crop`` = (x, y, w, h) of the augmented image (default: all);
    ``dst_xy``: where the crop lands in ``dst``; ``planar``: ``dst`` is a [3, H, W] tensor (the model's input layout)
    instead of [H, W, 3]; ``reverse_channels``: write channel 2, 1, 0 (``rgb`` describes ``src``).  The job holds raw pointers: keep ``src`` / ``dst`` alive."""
    _check_image(src, "aug_job")
    j = _lib.AugJob()
    j.src, j.dst = src.data_ptr(), dst.data_ptr()
    if planar:
        _require_cuda(dst, "aug_job")
        if dst.dtype != torch.uint8 or dst.dim() != 3 or dst.shape[0] != 3 or dst.stride(2) != 1:
            raise ValueError(f"aug_job: planar destination must be uint8 [3, H, W], got {dst.dtype} {tuple(dst.shape)}")
        j.src_stride, j.dst_stride, j.dst_plane_stride = src.stride(0), dst.stride(1), dst.stride(0)
        j.dst_height, j.dst_width = dst.shape[1], dst.shape[2]
    else:
        _check_image(dst, "aug_job")
        j.src_stride, j.dst_stride = src.stride(0), dst.stride(0)
        j.dst_height, j.dst_width = dst.shape[0], dst.shape[1]
    j.src_height, j.src_width = src.shape[0], src.shape[1]
    cx, cy, cw, ch = crop if crop is not None else (0, 0, src.shape[1], src.shape[0])
    j.dst_x, j.dst_y, j.crop_x, j.crop_y, j.crop_w, j.crop_h = dst_xy[0], dst_xy[1], cx, cy, cw, ch
    if inverse_map is not None:
        m = [float(v) for v in inverse_map]
        if len(m) != (9 if perspective else 6):
            raise ValueError(f"aug_job: inverse_map has {len(m)} values")
        j.warp = 2 if perspective else 1
        for i, v in enumerate(m):
            j.inverse_map[i] = v
    if jitter is not None:
        j.jitter = 1
        j.hue_gain, j.saturation_gain, j.exposure_gain = (float(v) for v in jitter)
    j.rgb_order = int(rgb)
    j.flip_horizontal, j.flip_vertical, j.border = int(flip_horizontal), int(flip_vertical), border
    j.reverse_channels = int(reverse_channels)
    return j


def aug_apply(jobs: Sequence["_lib.AugJob"]) -> None:
    """Run the jobs (16 images per launch)."""
    n = len(jobs)
    if n == 0:
        return
    arr = (_lib.AugJob * n)(*jobs)
    check(lib().yolo_aug_apply(ctypes.cast(arr, ctypes.c_void_p), n, _stream()), "yolo_aug_apply")


def sgd_block_table(segments: Sequence[Tuple[int, int, int]], device) -> Tuple[torch.Tensor, int]:
    """Block table of ``sgd_step`` for ``segments`` = (first element, count, group) runs of the arenas."""
    rows = []
    for first, count, group in segments:
        for o in range(0, count, 4096):
            rows.append((first + o, min(4096, count - o), group))
    return torch.tensor(rows, dtype=torch.int32, device=device).contiguous(), len(rows)


def sgd_step(param: torch.Tensor, grad: torch.Tensor, mom: torch.Tensor, table: torch.Tensor, n_blocks: int, lr0: float,
             wd0: float, lr1: float, wd1: float, momentum: float) -> None:
    """SGD with momentum over the flat fp32 arenas in one launch (two hyper-parameter groups)."""
    _require_cuda(param, "sgd_step")
    assert param.dtype == grad.dtype == mom.dtype == torch.float32 and param.numel() == grad.numel() == mom.numel()
    check(lib().yolo_sgd_step(_ptr(param), _ptr(grad), _ptr(mom), _ptr(table), n_blocks, lr0, wd0, lr1, wd1, momentum,
                              _stream()), "yolo_sgd_step")

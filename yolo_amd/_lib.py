"""ctypes loader for libyolo_hip.so (declared in include/yolo_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing, importing
anything that computes raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YOLO_HIP_LIB: another build of the same library (A/B measurements of kernel changes)
LIB_PATH = os.environ.get("YOLO_HIP_LIB") or os.path.join(_HERE, "libyolo_hip.so")

c_int, c_float, c_double, c_void_p, c_size_t = (ctypes.c_int, ctypes.c_float, ctypes.c_double,
                                                ctypes.c_void_p, ctypes.c_size_t)


class YoloView(ctypes.Structure):
    """struct yolo_view of include/yolo_hip.h."""
    _fields_ = [("ptr", c_void_p), ("H", c_int), ("W", c_int), ("halo", c_int),
                ("ch_total", c_int), ("ch_offset", c_int)]


_VP = ctypes.POINTER(YoloView)
_IP = ctypes.POINTER(c_int)
_FP = ctypes.POINTER(c_float)

class AugResampleJob(ctypes.Structure):
    """``yolo_aug_resample_job`` of include/yolo_hip.h."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
                ("workspace_bytes", ctypes.c_longlong), ("src_stride", ctypes.c_longlong), ("dst_stride", ctypes.c_longlong),
                ("src_height", ctypes.c_int), ("src_width", ctypes.c_int), ("dst_height", ctypes.c_int),
                ("dst_width", ctypes.c_int),
                ("x_bounds", ctypes.c_void_p), ("x_coef", ctypes.c_void_p), ("y_bounds", ctypes.c_void_p),
                ("y_coef", ctypes.c_void_p), ("x_ksize", ctypes.c_int), ("y_ksize", ctypes.c_int)]


class AugJob(ctypes.Structure):
    """``yolo_aug_job`` of include/yolo_hip.h."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p),
                ("src_stride", ctypes.c_longlong), ("dst_stride", ctypes.c_longlong),
                ("dst_plane_stride", ctypes.c_longlong),
                ("src_height", ctypes.c_int), ("src_width", ctypes.c_int), ("dst_height", ctypes.c_int),
                ("dst_width", ctypes.c_int),
                ("dst_x", ctypes.c_int), ("dst_y", ctypes.c_int), ("crop_x", ctypes.c_int), ("crop_y", ctypes.c_int),
                ("crop_w", ctypes.c_int), ("crop_h", ctypes.c_int),
                ("warp", ctypes.c_int), ("jitter", ctypes.c_int), ("rgb_order", ctypes.c_int),
                ("flip_horizontal", ctypes.c_int), ("flip_vertical", ctypes.c_int), ("border", ctypes.c_int),
                ("reverse_channels", ctypes.c_int), ("reserved", ctypes.c_int),
                ("inverse_map", ctypes.c_double * 9),
                ("hue_gain", ctypes.c_double), ("saturation_gain", ctypes.c_double), ("exposure_gain", ctypes.c_double)]


# name -> (restype, argtypes); one entry per declaration in include/yolo_hip.h
SIGNATURES = {
    "yolo_last_error": (ctypes.c_char_p, []),
    "yolo_abi_version": (c_int, []),
    "yolo_conv_cout_pad": (c_int, [c_int]),
    "yolo_conv_tile_shape": (c_int, [c_int, c_int, c_int, c_int, c_int, _IP, _IP]),
    "yolo_conv_kernel_name": (c_int, [_VP, _VP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_char_p, c_int]),
    "yolo_conv2d_splitk_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _IP, ctypes.POINTER(c_size_t)]),
    "yolo_conv2d_fwd_splitk": (c_int, [_VP, c_void_p, c_void_p, _VP, _VP, c_int, c_int, c_int, c_int, c_int,
                                       c_float, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "yolo_conv2d_fwd": (c_int, [_VP, c_void_p, c_void_p, _VP, _VP, c_int, c_int, c_int, c_int, c_int,
                                c_float, c_int, c_int, c_int, c_void_p]),
    "yolo_stem_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, _FP, _FP, c_void_p, c_void_p, _VP,
                              c_int, c_float, c_int, c_void_p]),
    "yolo_stem_down_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, _FP, _FP, c_void_p, c_void_p, c_void_p,
                                   c_void_p, _VP, c_int, c_int, c_float, c_void_p]),
    "yolo_resblock_c64_fwd": (c_int, [_VP, c_void_p, c_void_p, c_void_p, c_void_p, _VP, c_int, c_float, c_void_p]),
    "yolo_bn_train_stats": (c_int, [_VP, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "yolo_bn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "yolo_bn_act_fwd": (c_int, [_VP, c_void_p, _VP, _VP, c_int, c_int, c_float, c_int, c_void_p]),
    "yolo_bn_act_bwd": (c_int, [_VP, _VP, c_void_p, _VP, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                c_size_t, c_void_p]),
    "yolo_conv2d_wgrad": (c_int, [_VP, _VP, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_size_t,
                                  c_void_p]),
    "yolo_conv2d_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "yolo_stem_wgrad": (c_int, [c_void_p, c_int, c_int, c_int, c_int, _FP, _FP, _VP, c_int, c_int, c_void_p, c_int,
                                c_void_p, c_size_t, c_void_p]),
    "yolo_stem_wgrad_workspace_bytes": (c_size_t, []),
    "yolo_train_loss_fwd": (c_int, [_VP, c_int, ctypes.POINTER(c_int), c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                    c_size_t, c_void_p]),
    "yolo_train_loss_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "yolo_train_loss_bwd": (c_int, [_VP, _VP, c_int, c_int, ctypes.POINTER(c_int), c_void_p, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    "yolo_conv2d_fwd_bnsums": (c_int, [_VP, c_void_p, c_void_p, _VP, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                       c_size_t, c_void_p]),
    "yolo_conv2d_bnsums_rows": (c_int, [_VP, _VP, c_int, c_int, c_int, c_int, c_int, c_int]),
    "yolo_conv2d_fwd_bnbwd_sums": (c_int, [_VP, c_void_p, c_void_p, _VP, _VP, c_int, c_int, c_int, c_int, c_int, _VP,
                                           c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "yolo_bn_act_bwd_presummed": (c_int, [_VP, _VP, c_void_p, _VP, c_int, c_int, c_float, c_int, c_void_p, c_void_p,
                                          c_void_p, c_int, c_void_p]),
    "yolo_bn_train_finalize": (c_int, [c_void_p, c_int, ctypes.c_longlong, c_int, c_void_p, c_void_p, c_float, c_float,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "yolo_aug_letterbox": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, c_void_p, c_int, c_int, ctypes.c_longlong,
                                   c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "yolo_aug_resample": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, c_void_p, c_int, c_int, ctypes.c_longlong,
                                  c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "yolo_aug_resample_workspace_bytes": (c_size_t, [c_int, c_int]),
    "yolo_aug_apply": (c_int, [c_void_p, c_int, c_void_p]),
    "yolo_aug_resample_batch": (c_int, [c_void_p, c_int, c_void_p]),
    "yolo_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float,
                              c_void_p]),
    "yolo_conv3x3s2_dgrad": (c_int, [_VP, c_void_p, c_void_p, _VP, _VP, c_int, c_int, c_int, c_int, c_void_p]),
    "yolo_pack_conv_train_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "yolo_pack_conv_train": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                     c_void_p]),
    "yolo_dilate2x_nhwc": (c_int, [_VP, _VP, c_int, c_int, c_int, c_void_p]),
    "yolo_sum2x2_nhwc": (c_int, [_VP, _VP, c_int, c_int, c_int, c_void_p]),
    "yolo_upsample2x_nhwc": (c_int, [_VP, _VP, c_int, c_int, c_int, c_void_p]),
    "yolo_decode": (c_int, [_VP, c_int, _IP, _FP, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p,
                            c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "yolo_decode_workspace_bytes": (c_size_t, [_VP, c_int, c_int]),
    "yolo_decode_noalloc": (c_int, [_VP, c_int, _IP, _FP, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_void_p]),
    "yolo_batched_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_int,
                                 c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
    "yolo_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "yolo_gather_detections": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                       c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def load(path: str) -> ctypes.CDLL:
    """A handle of the library at ``path`` with every entry point's signature set."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `make -C yolo_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    L = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


def use(handle: ctypes.CDLL) -> None:
    """Route every operator through another build of the library (A/B measurements in one process)."""
    global _lib
    _lib = handle


class YoloHipError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().yolo_last_error().decode("utf-8", "replace")
        raise YoloHipError(f"{what} failed ({code}): {msg}")

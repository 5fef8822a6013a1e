"""ctypes binding of oracle/nms_ref.c (test infrastructure only; see oracle/__init__.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    path = os.path.join(_HERE, "liboracle_nms.so")
    src = os.path.join(_HERE, "nms_ref.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_nms.so"])
    return path


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_nms.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.oracle_nms.restype = ctypes.c_int64
        L.oracle_nms.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_double, i64p]
        for name in ("oracle_batched_nms_vanilla", "oracle_batched_nms_trick"):
            fn = getattr(L, name)
            fn.restype = ctypes.c_int64
            fn.argtypes = [f32p, f32p, i64p, ctypes.c_int64, ctypes.c_double, i64p]
        _LIB = L
    return _LIB


def _f32(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i64(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def nms(boxes, scores, iou_thr: float) -> torch.Tensor:
    b, bp = _f32(boxes)
    s, sp = _f32(scores)
    n = s.shape[0]
    keep = np.empty(max(n, 1), np.int64)
    k = _lib().oracle_nms(bp, sp, n, float(iou_thr), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return torch.from_numpy(keep[:k].copy())


def _batched(fn_name, boxes, scores, idxs, iou_thr):
    b, bp = _f32(boxes)
    s, sp = _f32(scores)
    c, cp = _i64(idxs)
    n = s.shape[0]
    keep = np.empty(max(n, 1), np.int64)
    k = getattr(_lib(), fn_name)(bp, sp, cp, n, float(iou_thr), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return torch.from_numpy(keep[:k].copy())


def batched_nms_vanilla(boxes, scores, idxs, iou_thr):
    return _batched("oracle_batched_nms_vanilla", boxes, scores, idxs, iou_thr)


def batched_nms_trick(boxes, scores, idxs, iou_thr):
    return _batched("oracle_batched_nms_trick", boxes, scores, idxs, iou_thr)


def batched_nms(boxes, scores, idxs, iou_thr, device_kind="cpu"):
    """torchvision's branch choice (numel > 4000 on CPU / > 20000 otherwise -> vanilla)."""
    n = int(np.prod(boxes.shape))
    if n > (4000 if device_kind == "cpu" else 20000):
        return batched_nms_vanilla(boxes, scores, idxs, iou_thr)
    return batched_nms_trick(boxes, scores, idxs, iou_thr)

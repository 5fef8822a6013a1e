"""Detection-set comparison used by the parity tests and by __graft_entry__.smoke().

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
from __future__ import annotations

import torch


def box_iou_one(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """IoU of every XYXY box in ``a`` [n, 4] with the single box ``b`` [4]."""
    ix = (torch.min(a[:, 2], b[2]) - torch.max(a[:, 0], b[0])).clamp(min=0)
    iy = (torch.min(a[:, 3], b[3]) - torch.max(a[:, 1], b[1])).clamp(min=0)
    inter = ix * iy
    return inter / ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def matched_fraction(ref: torch.Tensor, got: torch.Tensor, min_score: float, iou_thr: float, score_tol: float):
    """(fraction, count): of the detections of ``ref`` with score > min_score, the fraction that has a twin in
    ``got`` -- same class, IoU > iou_thr, |score difference| < score_tol.  Rows: x0, y0, x1, y1, score, class."""
    ref = ref[ref[:, 4] > min_score]
    if ref.shape[0] == 0:
        return 1.0, 0
    hit = 0
    for r in ref:
        same = (got[:, 5] == r[5]) & ((got[:, 4] - r[4]).abs() < score_tol)
        hit += bool(same.any() and (box_iou_one(got[same, :4], r[:4]) > iou_thr).any())
    return hit / ref.shape[0], int(ref.shape[0])


def rows_of(boxes: torch.Tensor, scores: torch.Tensor, classes: torch.Tensor) -> torch.Tensor:
    return torch.cat([boxes.float(), scores.float()[:, None], classes.float()[:, None]], 1)


def resize_shortest_edge(image, short: int, max_size: int):
    """What the reference's Predictor does to an HWC uint8 image on the host (tools/inference.py:29-31,41:
    detectron2 ``ResizeShortestEdge([short, short], max_size)``): scale so that the shorter side equals ``short``,
    cap the longer side at ``max_size``, round to integers, Pillow bilinear.  Pinned: it IS Pillow."""
    import numpy as np
    from PIL import Image
    h, w = image.shape[:2]
    scale = short * 1.0 / min(h, w)
    newh, neww = (short, scale * w) if h < w else (scale * h, short)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    neww, newh = int(neww + 0.5), int(newh + 0.5)
    return np.asarray(Image.fromarray(np.ascontiguousarray(image)).resize((neww, newh), Image.BILINEAR))

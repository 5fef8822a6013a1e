"""Box <-> delta transform of the YOLO head (reference yolo/modeling/box_regression.py:12-64).

Inference decodes inside the HIP decode kernel (yolo_amd/csrc/decode.hip); ``apply_deltas`` here is the
same arithmetic as torch ops for callers that hold explicit delta / anchor tensors (it runs on whatever
device the tensors are on and is not part of the timed inference path).  ``get_deltas`` builds training
targets.
"""
from typing import Tuple

import torch

from ..structures import Boxes


class Box2BoxTransform:
    def __init__(self, weights: Tuple[float, float, float, float], scale_clamp: float):
        self.weights = weights
        self.scale_clamp = scale_clamp  # stored, never applied: exp() is unclamped (reference :16,:53)

    def get_deltas(self, src_boxes: torch.Tensor, target_boxes, strides: torch.Tensor) -> torch.Tensor:
        """src_boxes: matched ground truth [N, R, 4]; target_boxes: list of per-level anchor Boxes;
        strides: [R].  xy target = frac(gt_centre / stride) - 0.5, wh target = log(gt_wh / anchor_wh + 1e-16)."""
        gt_wh = src_boxes[..., 2:] - src_boxes[..., :2]
        gt_cxy = src_boxes[..., :2] + gt_wh / 2
        anchors = Boxes.cat(target_boxes).tensor
        anchors_wh = anchors[..., 2:] - anchors[..., :2]
        gt_cxy = gt_cxy / strides[..., None]
        d_cxy = gt_cxy - torch.floor(gt_cxy) - 0.5
        d_wh = torch.log(gt_wh / anchors_wh[None, :, :] + 1e-16)
        wx, wy, ww, wh = self.weights
        # the reference multiplies dw by wh and dh by ww (:37-38); identical for unit weights
        return torch.stack((wx * d_cxy[..., 0], wy * d_cxy[..., 1], wh * d_wh[..., 0], ww * d_wh[..., 1]), dim=-1)

    def apply_deltas(self, deltas: torch.Tensor, boxes: torch.Tensor, stride: int) -> torch.Tensor:
        boxes = boxes.to(deltas.dtype)
        boxes_wh = boxes[:, 2:] - boxes[:, :2]
        boxes_cxy = boxes[:, :2] + boxes_wh / 2
        target_cxy = deltas[:, :2] * stride + boxes_cxy
        target_wh = boxes_wh * deltas[..., 2:].exp()
        wx, wy, ww, wh = self.weights
        cx, cy = target_cxy[:, 0] / wx, target_cxy[:, 1] / wy
        w, h = target_wh[:, 0] / ww, target_wh[:, 1] / wh
        return torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], dim=-1)

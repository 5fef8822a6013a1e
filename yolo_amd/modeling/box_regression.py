"""Box <-> delta transform of the YOLO head.

Interface and arithmetic of reference yolo/modeling/box_regression.py:12-64 (``Box2BoxTransform(weights,
scale_clamp)`` with ``get_deltas`` / ``apply_deltas``), written here on (centre, size) pairs.  Inference decodes
inside the HIP decode kernel (yolo_amd/csrc/decode.hip) and the training step's targets are formed inside
``yolo_train_loss_fwd`` (yolo_amd/csrc/loss.hip); this class serves callers that hold explicit delta / anchor tensors
and the tests of those kernels.  Every operation keeps the reference's order, so results are bit-identical to it.
"""
from typing import Sequence, Tuple

import torch

from ..structures import Boxes


def _centre_size(xyxy: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(x0, y0, x1, y1) -> (centre [.., 2], size [.., 2]); the centre is corner + size / 2, as the reference forms it."""
    size = xyxy[..., 2:] - xyxy[..., :2]
    return xyxy[..., :2] + size / 2, size


class Box2BoxTransform:
    def __init__(self, weights: Tuple[float, float, float, float], scale_clamp: float):
        self.weights = weights
        self.scale_clamp = scale_clamp  # kept for the config surface; exp() is not clamped (reference :16, :53)

    def get_deltas(self, src_boxes: torch.Tensor, target_boxes: Sequence[Boxes], strides: torch.Tensor) -> torch.Tensor:
        """Regression targets.  src_boxes: matched ground truth [N, R, 4]; target_boxes: the per-level anchor Boxes
        (R anchors in all); strides: [R].  Offsets: position of the box centre inside its grid cell minus one half;
        sizes: log(box size / anchor size + 1e-16)."""
        centre, size = _centre_size(src_boxes)
        _, anchor_size = _centre_size(Boxes.cat(list(target_boxes)).tensor)
        cell = centre / strides[..., None]
        offset = cell - torch.floor(cell) - 0.5
        log_size = torch.log(size / anchor_size[None, :, :] + 1e-16)
        wx, wy, ww, wh = self.weights
        # the width term is scaled by the height weight and vice versa, as in the reference (:37-38)
        scale = offset.new_tensor([wx, wy, wh, ww])
        return torch.cat((offset, log_size), dim=-1) * scale

    def apply_deltas(self, deltas: torch.Tensor, boxes: torch.Tensor, stride: int) -> torch.Tensor:
        """Boxes of ONE pyramid level from its deltas [R, 4] and anchors [R, 4] (xyxy): centre = delta * stride +
        anchor centre, size = anchor size * exp(delta), both divided by their weights."""
        anchor_centre, anchor_size = _centre_size(boxes.to(deltas.dtype))
        wx, wy, ww, wh = self.weights
        centre = (deltas[:, :2] * stride + anchor_centre) / deltas.new_tensor([wx, wy])
        half = anchor_size * deltas[..., 2:].exp() / deltas.new_tensor([ww, wh]) / 2
        return torch.cat((centre - half, centre + half), dim=-1)

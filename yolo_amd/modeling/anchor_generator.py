"""Anchor description of the three YOLO scales.

Reference: yolo/modeling/anchor_generator.py:15-40 (YoloAnchorGenerator over detectron2's
DefaultAnchorGenerator).  The HIP decode kernel generates anchors analytically from (x, y, anchor,
stride), so this class only carries the per-level (w, h) table and strides; ``grid_anchors`` exists
for callers that want the explicit tensors (training label assignment, tests).
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch

from ..structures import Boxes, ShapeSpec
from .registry import ANCHOR_GENERATOR_REGISTRY


class YoloAnchorGenerator:
    box_dim = 4

    def __init__(self, *, sizes, aspect_ratios, strides: Sequence[int], offset: float = 0.5):
        assert 0.0 <= offset < 1.0, offset
        self.strides = list(strides)
        self.offset = offset
        # (size, ratio) pairs are zipped, not crossed (reference anchor_generator.py:31-40)
        self.cell_anchors = [self.generate_cell_anchors(s, a).float() for s, a in zip(sizes, aspect_ratios)]

    @classmethod
    def from_config(cls, cfg, input_shape: List[ShapeSpec]):
        anchors = cfg.MODEL.ANCHOR_GENERATOR.ANCHORS
        return dict(sizes=[[(w * h) ** (1.0 / 2) for w, h in lvl] for lvl in anchors],
                    aspect_ratios=[[h / w for w, h in lvl] for lvl in anchors],
                    strides=[x.stride for x in input_shape],
                    offset=cfg.MODEL.ANCHOR_GENERATOR.OFFSET)

    @staticmethod
    def generate_cell_anchors(sizes, aspect_ratios) -> torch.Tensor:
        rows = []
        for size, ratio in zip(sizes, aspect_ratios):
            area = size ** 2
            w = math.sqrt(area / ratio)
            h = ratio * w
            rows.append([w / -2, h / -2, w / 2, h / 2])
        return torch.tensor(rows)

    @property
    def num_cell_anchors(self) -> List[int]:
        return [len(c) for c in self.cell_anchors]

    num_anchors = num_cell_anchors

    def anchors_wh(self) -> List[List[tuple]]:
        """Per level [(w, h), ...] as the float32 values of the cell anchors (x1 - x0, y1 - y0)."""
        out = []
        for c in self.cell_anchors:
            out.append([(float(r[2] * 2), float(r[3] * 2)) for r in c])
        return out

    def grid_anchors(self, grid_sizes, device="cpu") -> List[Boxes]:
        res = []
        for (gh, gw), stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            sx = torch.arange(self.offset * stride, gw * stride, step=stride, dtype=torch.float32, device=device)
            sy = torch.arange(self.offset * stride, gh * stride, step=stride, dtype=torch.float32, device=device)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            xx, yy = xx.reshape(-1), yy.reshape(-1)
            shifts = torch.stack((xx, yy, xx, yy), dim=1)
            res.append(Boxes((shifts.view(-1, 1, 4) + base.to(device).view(1, -1, 4)).reshape(-1, 4)))
        return res

    def __call__(self, features) -> List[Boxes]:
        return self.grid_anchors([f.shape[-2:] for f in features], features[0].device)


def _build(cfg, input_shape):
    return YoloAnchorGenerator(**YoloAnchorGenerator.from_config(cfg, input_shape))


ANCHOR_GENERATOR_REGISTRY._add("YoloAnchorGenerator", _build)  # the name configs/yolov3.yaml:12 selects

"""Anchor <-> ground-truth matcher of the YOLOv3 training path (reference yolo/modeling/matcher.py:8-55).

Torch ops on whatever device the boxes live on (label assignment is host-side glue of the reference's
training step; the convolutions around it are the HIP kernels)."""
from typing import List

import torch

from ..structures import Boxes


class Matcher:
    ignore_label_idx = 0
    negative_label_idx = 1
    positive_label_idx = 2

    def __init__(self, threshold: float, labels: List[int], allow_low_quality_matches: bool = False):
        self.threshold = threshold
        self.labels = labels
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix: torch.Tensor, anchors: Boxes, gt_boxes: Boxes):
        """match_quality_matrix: IoU [K ground truths, R anchors over all levels].

        One positive anchor per ground truth: among the anchors that reach the ground truth's best IoU, the
        one whose centre is L1-nearest to the box centre.  Of the rest, anchors whose best IoU exceeds the
        threshold are ignored, the others negative.  Returns (matched gt index [R], label [R] int8)."""
        assert match_quality_matrix.dim() == 2
        dev = match_quality_matrix.device
        matched_val, _ = match_quality_matrix.max(dim=1)
        iou_mask = match_quality_matrix == matched_val[:, None]
        distance = self._calculate_l1_distance(gt_boxes, anchors)
        distance = distance.where(iou_mask, torch.tensor(float("inf"), device=dev))
        _, center_matches = distance.min(dim=1)
        distance_mask = torch.zeros(distance.shape, dtype=torch.bool, device=dev)
        distance_mask[torch.arange(len(gt_boxes), device=dev), center_matches] = True
        gt_idx, anchor_idx = torch.nonzero(iou_mask & distance_mask, as_tuple=True)
        matched_val, matches = match_quality_matrix.max(dim=0)
        match_labels = torch.full(matches.size(), self.labels[self.negative_label_idx], dtype=torch.int8, device=dev)
        match_labels[matched_val > self.threshold] = self.labels[self.ignore_label_idx]
        match_labels[anchor_idx] = self.labels[self.positive_label_idx]
        matches[anchor_idx] = gt_idx
        if self.allow_low_quality_matches:
            raise NotImplementedError
        return matches, match_labels

    @staticmethod
    def _calculate_l1_distance(boxes1: Boxes, boxes2: Boxes) -> torch.Tensor:
        c1, c2 = boxes1.get_centers(), boxes2.get_centers()
        return torch.abs(c1[:, None, :] - c2[None, :, :]).sum(dim=-1)

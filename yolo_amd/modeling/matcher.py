"""Anchor <-> ground-truth matcher of the YOLOv3 training path.

Interface and rules of reference yolo/modeling/matcher.py:8-55 (``Matcher(threshold, labels)(iou, anchors, gt_boxes)
-> (matched gt index per anchor, label per anchor)``); the computation is this build's own: one positive anchor per
ground truth picked by a single arg-max over a two-key score, and anchors claimed by several ground truths resolved
deterministically (highest ground-truth index wins, which is what the reference's sequential index assignment
gives on the CPU).  The training step itself runs the same rules inside ``yolo_train_loss_fwd``
(yolo_amd/csrc/loss.hip); this class serves callers of the operator-level API and the tests of those kernels.
"""
from typing import List, Tuple

import torch

from ..structures import Boxes


class Matcher:
    # positions inside ``labels`` (the reference's MODEL.MATCHER.LABELS = [ignore, negative, positive])
    ignore_label_idx = 0
    negative_label_idx = 1
    positive_label_idx = 2

    def __init__(self, threshold: float, labels: List[int], allow_low_quality_matches: bool = False):
        if allow_low_quality_matches:
            raise NotImplementedError("allow_low_quality_matches is not implemented (nor in the reference)")
        self.threshold = threshold
        self.labels = labels
        self.allow_low_quality_matches = False

    @staticmethod
    def positive_anchor_per_gt(iou: torch.Tensor, anchors: Boxes, gt_boxes: Boxes) -> torch.Tensor:
        """[K] index of each ground truth's positive anchor: of the anchors that reach the ground truth's best IoU,
        the one whose centre is nearest (L1) to the box centre; the first of them on a tie."""
        offset = (gt_boxes.get_centers()[:, None, :] - anchors.get_centers()[None, :, :]).abs().sum(-1)  # [K, R]
        reaches_best = iou == iou.amax(dim=1, keepdim=True)
        return torch.where(reaches_best, offset, offset.new_full((), float("inf"))).argmin(dim=1)

    def __call__(self, match_quality_matrix: torch.Tensor, anchors: Boxes, gt_boxes: Boxes) -> Tuple[torch.Tensor, torch.Tensor]:
        """match_quality_matrix: IoU [K ground truths, R anchors of all levels].  Returns (matches [R] int64: the
        ground truth each anchor is matched to; match_labels [R] int8: positive / ignore / negative label)."""
        iou = match_quality_matrix
        if iou.dim() != 2:
            raise ValueError(f"match_quality_matrix must be [K, R], got {tuple(iou.shape)}")
        K, R = iou.shape
        lab_ignore, lab_negative, lab_positive = (self.labels[i] for i in (self.ignore_label_idx, self.negative_label_idx,
                                                                           self.positive_label_idx))
        best_iou, best_gt = iou.max(dim=0)                       # per anchor
        match_labels = torch.where(best_iou > self.threshold, lab_ignore, lab_negative).to(torch.int8)
        # positives: an anchor claimed by several ground truths goes to the one with the highest index
        claimed_by = torch.full((R,), -1, dtype=torch.int64, device=iou.device)
        claimed_by.scatter_reduce_(0, self.positive_anchor_per_gt(iou, anchors, gt_boxes),
                                   torch.arange(K, device=iou.device), reduce="amax", include_self=True)
        is_positive = claimed_by >= 0
        matches = torch.where(is_positive, claimed_by, best_gt)
        match_labels = torch.where(is_positive, match_labels.new_full((), lab_positive), match_labels)
        return matches, match_labels

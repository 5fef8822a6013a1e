"""YOLOV3 meta-architecture: the reference's model object (yolo/modeling/meta_arch/yolo.py:28-420 on top of
detectron2's DenseDetector) with inference executed as gfx950 kernels.

``model(batched_inputs)`` keeps the reference's call contract: a list of dicts with "image" (CHW, values
0..255 in cfg.INPUT.FORMAT order) and optional "height"/"width" (output resolution); in eval mode it
returns ``[{"instances": Instances(pred_boxes, scores, pred_classes)}]`` with at most 100 detections per
image.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from ... import ops
from ...structures import Boxes, Instances
from ..box_regression import Box2BoxTransform
from ..registry import META_ARCH_REGISTRY, build_anchor_generator, build_backbone

__all__ = ["YOLOV3", "YoloV3Head"]

_NMS_MODES = {"auto": ops.NMS_AUTO, "per_class": ops.NMS_PER_CLASS, "coord_trick": ops.NMS_COORD_TRICK}


class YoloV3Head(nn.Module):
    """Parameter-free head: splits each raw head tensor per anchor into deltas / objectness / class scores
    and applies the sigmoids (reference yolo.py:386-420).  Inference does this inside the decode kernel;
    this module serves callers that want the three activated tensors."""

    def __init__(self, *, num_classes: int, num_anchors: int):
        super().__init__()
        self.num_classes = num_classes
        self.num_anchors = num_anchors

    def forward(self, features: List[torch.Tensor]):
        pred_logits, pred_confs, pred_deltas = [], [], []
        for feature in features:
            N, _, H, W = feature.shape
            f = feature.reshape(N, self.num_anchors, -1, H, W).float()
            d = f[:, :, :4].reshape(N, -1, H, W).clone()
            d[:, 0::4] = torch.sigmoid(d[:, 0::4]) - 0.5
            d[:, 1::4] = torch.sigmoid(d[:, 1::4]) - 0.5
            pred_deltas.append(d)
            pred_confs.append(torch.sigmoid(f[:, :, 4:5].reshape(N, -1, H, W)))
            pred_logits.append(torch.sigmoid(f[:, :, 5:].reshape(N, -1, H, W)))
        return pred_logits, pred_confs, pred_deltas


class YOLOV3(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        shapes = self.backbone.output_shape()
        self.in_features = list(cfg.MODEL.YOLOV3.IN_FEATURES)
        self.head_in_features = self.in_features
        self.anchor_generator = build_anchor_generator(cfg, [shapes[f] for f in self.in_features])
        num_anchors = self.anchor_generator.num_cell_anchors
        assert len(set(num_anchors)) == 1, "Using different number of anchors between levels is not currently supported!"
        self.num_classes = cfg.MODEL.YOLOV3.NUM_CLASSES
        self.head = YoloV3Head(num_anchors=num_anchors[0], num_classes=self.num_classes)
        self.box2box_transform = Box2BoxTransform((1.0, 1.0, 1.0, 1.0), 0.5)
        self.test_score_thresh = cfg.MODEL.YOLOV3.TEST_SCORE_THRESH
        self.test_nms_thresh = cfg.MODEL.YOLOV3.TEST_NMS_THRESH
        self.test_topk_candidates = 1000  # stored, unused (as in the reference, yolo.py:45,61)
        self.max_detections_pre_image = 100
        self.obj_weight, self.noobj_weight = 1, 100
        self.matcher_threshold = cfg.MODEL.MATCHER.IGNORE_THRESHOLD
        self.matcher_labels = list(cfg.MODEL.MATCHER.LABELS)
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), False)
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]
        hip = cfg.MODEL.get("HIP", None)
        self.nms_mode = _NMS_MODES[hip.NMS_MODE] if hip is not None else ops.NMS_AUTO
        self.nms_auto_numel = hip.NMS_AUTO_NUMEL if hip is not None else 20000

    @property
    def device(self):
        return self.pixel_mean.device

    def _get_strides(self) -> List[int]:
        shapes = self.backbone.output_shape()
        return [shapes[f].stride for f in self.head_in_features]

    # ------------------------------------------------------------------------------------------
    def _batch_images(self, batched_inputs: List[Dict]):
        """Stack the images of a batch on the GPU.  Same-size images (the common case) are passed as they
        are -- normalisation and the pad to a multiple of 32 happen inside the stem kernel.  Ragged
        batches are first placed top-left on a canvas filled with the pixel mean, which the kernel's
        normalisation turns into the reference's zero padding."""
        imgs = [x["image"] for x in batched_inputs]
        sizes = [(int(i.shape[-2]), int(i.shape[-1])) for i in imgs]
        dev = self.device
        if all(s == sizes[0] for s in sizes):
            dt = torch.uint8 if all(i.dtype == torch.uint8 for i in imgs) else torch.float32
            batch = torch.stack([i.to(device=dev, dtype=dt, non_blocking=True) for i in imgs])
        else:
            Hm, Wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
            batch = self.pixel_mean.view(1, -1, 1, 1).expand(len(imgs), -1, Hm, Wm).contiguous().float()
            for n, i in enumerate(imgs):
                batch[n, :, :i.shape[-2], :i.shape[-1]] = i.to(device=dev, dtype=torch.float32)
        return batch, sizes

    def forward(self, batched_inputs: List[Dict]):
        if self.training:
            raise NotImplementedError(
                "YOLOV3 training (label assignment, losses, conv backward, DDP) is not built on the HIP path yet")
        return self.inference(batched_inputs)

    @torch.no_grad()
    def inference(self, batched_inputs: List[Dict]):
        if self.device.type != "cuda":
            raise RuntimeError("yolo_amd runs on MI355X only (MODEL.DEVICE must be 'cuda'); there is no CPU fallback")
        batch, sizes = self._batch_images(batched_inputs)
        dev = self.device
        in_hw = torch.tensor(sizes, dtype=torch.int32, device=dev)
        out = [(int(x.get("height", s[0])), int(x.get("width", s[1]))) for x, s in zip(batched_inputs, sizes)]
        out_hw = torch.tensor(out, dtype=torch.int32, device=dev)
        det, n_det = self.backbone.engine().detect(
            batch, score_thr=self.test_score_thresh, nms_thr=self.test_nms_thresh, mean=self._mean, std=self._std,
            max_det=self.max_detections_pre_image, nms_mode=self.nms_mode, auto_limit=self.nms_auto_numel,
            in_hw=in_hw, out_hw=out_hw)
        n_det = n_det.tolist()
        results = []
        for n, (oh, ow) in enumerate(out):
            d = det[n, :n_det[n]]
            inst = Instances((oh, ow), pred_boxes=Boxes(d[:, :4].clone()), scores=d[:, 4].clone(),
                             pred_classes=d[:, 5].to(torch.int64))
            results.append({"instances": inst})
        return results


def _build(cfg):
    return YOLOV3(cfg)


META_ARCH_REGISTRY._add("YOLOV3", _build)  # the name configs/yolov3.yaml:3 selects

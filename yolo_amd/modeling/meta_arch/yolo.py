"""YOLOV3 meta-architecture: the reference's model object (yolo/modeling/meta_arch/yolo.py:28-420 on top of
detectron2's DenseDetector) with inference executed as gfx950 kernels.

``model(batched_inputs)`` keeps the reference's call contract: a list of dicts with "image" (CHW, values
0..255 in cfg.INPUT.FORMAT order) and optional "height"/"width" (output resolution); in eval mode it
returns ``[{"instances": Instances(pred_boxes, scores, pred_classes)}]`` with at most 100 detections per
image.
"""
from __future__ import annotations

import os

from typing import Dict, List

import torch
import torch.nn as nn

from ... import ops
import torch.nn.functional as F

from ...structures import Boxes, Instances, pairwise_iou
from ..box_regression import Box2BoxTransform
from ..matcher import Matcher
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


class _FusedTrainStep(torch.autograd.Function):
    """Backbone + neck + heads + label assignment + losses as ONE autograd node: torch sees the trainable
    parameters going in and three scalar losses coming out (plus the positive-anchor count, not differentiable)."""

    @staticmethod
    def forward(ctx, eng, images, mean, std, named, gt_boxes, gt_classes, gt_count, cfgd, *params):
        from ... import ops
        heads = eng.forward(images, mean, std)
        N, _, Hs, Ws = images.shape
        ctx.eng, ctx.named = eng, named
        ctx.plan = eng.plan(N, (Hs + 31) // 32 * 32, (Ws + 31) // 32 * 32)
        losses, ctx.loss_ctx = ops.train_loss_fwd(heads, cfgd["strides"], cfgd["anchors"], gt_boxes, gt_classes, gt_count,
                                                  num_anchors=cfgd["num_anchors"], num_classes=cfgd["num_classes"],
                                                  match_thr=cfgd["match_thr"], obj_weight=cfgd["obj_weight"],
                                                  noobj_weight=cfgd["noobj_weight"])
        n_pos = ctx.loss_ctx.counts[0]
        ctx.mark_non_differentiable(n_pos)
        return losses[0], losses[1], losses[2], n_pos

    @staticmethod
    def backward(ctx, g_conf, g_cls, g_box, _g_npos):
        from ... import ops
        from ..backbone.fpn import install_param_grads
        dev = ctx.loss_ctx.anchors.device
        gs = torch.stack([g if g is not None else torch.zeros((), device=dev) for g in (g_conf, g_cls, g_box)]).float()

        def run():
            ops.train_loss_bwd(ctx.loss_ctx, ctx.plan.dheads, gs)
            ctx.eng.backward(ctx.plan, None)

        out = install_param_grads(ctx.eng, ctx.named, run)
        return (None,) * 9 + out


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
        self.num_anchors = num_anchors[0]
        self.head = YoloV3Head(num_anchors=num_anchors[0], num_classes=self.num_classes)
        self.box2box_transform = Box2BoxTransform((1.0, 1.0, 1.0, 1.0), 0.5)
        self.test_score_thresh = cfg.MODEL.YOLOV3.TEST_SCORE_THRESH
        self.test_nms_thresh = cfg.MODEL.YOLOV3.TEST_NMS_THRESH
        self.test_topk_candidates = 1000  # stored, unused (as in the reference, yolo.py:45,61)
        self.max_detections_pre_image = 100
        self.obj_weight, self.noobj_weight = 1, 100
        self.fused_loss = True     # training: label assignment + losses as HIP kernels (False: the torch ops below)
        self._anchor_cache: Dict = {}
        self._size_cache: Dict = {}
        self.matcher_threshold = cfg.MODEL.MATCHER.IGNORE_THRESHOLD
        self.matcher_labels = list(cfg.MODEL.MATCHER.LABELS)
        self.anchor_matcher = Matcher(threshold=self.matcher_threshold, labels=self.matcher_labels,
                                      allow_low_quality_matches=False)
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), False)
        self._mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std = [float(v) for v in cfg.MODEL.PIXEL_STD]
        hip = cfg.MODEL.get("HIP", None)
        self.nms_mode = _NMS_MODES[hip.NMS_MODE] if hip is not None else ops.NMS_AUTO
        self.nms_auto_numel = hip.NMS_AUTO_NUMEL if hip is not None else 20000
        # inference calls with at most this many images replay a captured HIP graph (Yolov3Engine.detect_graphed);
        # 0 = never (the default: a single-image call is bound by its kernels, not by the host's launches, and the
        # replay measured within 0-4 % of launching one by one).  YOLO_HIP_GRAPH=1 switches it on for up to
        # YOLO_HIP_GRAPH_MAX_BATCH (8) images.
        self.graph_max_batch = int(os.environ.get("YOLO_HIP_GRAPH_MAX_BATCH", "8")) \
            if os.environ.get("YOLO_HIP_GRAPH", "0") == "1" else 0

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
            ts = [i.to(device=dev, dtype=dt, non_blocking=True) for i in imgs]
            batch = ts[0].unsqueeze(0) if len(ts) == 1 else torch.stack(ts)   # one image: a view, no copy
        else:
            Hm, Wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
            batch = self.pixel_mean.view(1, -1, 1, 1).expand(len(imgs), -1, Hm, Wm).contiguous().float()
            for n, i in enumerate(imgs):
                batch[n, :, :i.shape[-2], :i.shape[-1]] = i.to(device=dev, dtype=torch.float32)
        return batch, sizes

    @ops.on_own_device
    def forward(self, batched_inputs: List[Dict]):
        if self.training:
            return self.training_step(batched_inputs)
        return self.inference(batched_inputs)

    # ------------------------------------------------------------------------------------------
    # training (reference yolo.py:101-264 under detectron2's DenseDetector.forward)
    # ------------------------------------------------------------------------------------------
    def training_step(self, batched_inputs: List[Dict]) -> Dict[str, torch.Tensor]:
        """Returns {"confs_loss", "logits_loss", "regression_loss"}; ``sum(losses.values()).backward()`` runs the
        HIP backward plan of the backbone and neck.  Convolutions, BatchNorm and their gradients are HIP kernels;
        the head activations, label assignment and the three losses are torch ops on the GPU, as in the reference."""
        if self.device.type != "cuda":
            raise RuntimeError("yolo_amd runs on MI355X only (MODEL.DEVICE must be 'cuda'); there is no CPU fallback")
        assert "instances" in batched_inputs[0], "Instance annotations are missing in training!"
        batch, _ = self._batch_images(batched_inputs)
        if self.fused_loss and list(self.matcher_labels) == [-1, 0, 1]:
            return self._fused_training_step(batch, [x["instances"] for x in batched_inputs])
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        feats = self.backbone.forward_train(batch, self._mean, self._std)
        features = [feats[f] for f in self.head_in_features]
        predictions = self.head(features)
        return self.forward_training(batch, features, predictions, gt_instances)

    def _padded_targets(self, gt_instances: List[Instances]):
        """Ground truths of a batch as [N,K,4] boxes, [N,K] classes and [N] counts (K = the batch maximum, >= 1), on
        the device.  Nothing here makes the host wait for the GPU: host-side data goes through pinned staging memory
        and asynchronous copies (a ``torch.tensor(..., device=...)`` from pageable memory returns only when the stream
        has drained, which would stop the host from queueing the next step while this one runs)."""
        dev = self.device
        lens = [len(g) for g in gt_instances]
        N, K = len(lens), max(max(lens), 1)
        total = sum(lens)

        def upload(t: torch.Tensor) -> torch.Tensor:
            if dev.type != "cuda":   # the label-assignment restatement also runs on a CPU model (tests)
                return t
            return t.pin_memory().to(dev, non_blocking=True)

        counts = upload(torch.tensor(lens, dtype=torch.int32))
        if all(g.gt_boxes.tensor.device.type == "cpu" for g in gt_instances):
            # a loader's instances (host tensors): pad on the host, two uploads for the whole batch
            boxes = torch.zeros(N, K, 4)
            classes = torch.zeros(N, K, dtype=torch.int64)
            for n, g in enumerate(gt_instances):
                boxes[n, :lens[n]] = g.gt_boxes.tensor
                classes[n, :lens[n]] = g.gt_classes
            return upload(boxes), upload(classes), counts
        boxes = torch.zeros(N, K, 4, device=dev)
        classes = torch.zeros(N, K, dtype=torch.int64, device=dev)
        if total:  # one scatter for the whole batch instead of slice copies per image
            idx = torch.empty(2, total, dtype=torch.int64)
            o = 0
            for n, k in enumerate(lens):
                idx[0, o:o + k] = n
                idx[1, o:o + k] = torch.arange(k)
                o += k
            idx = upload(idx)
            boxes[idx[0], idx[1]] = torch.cat([g.gt_boxes.tensor.to(dev) for g in gt_instances]).to(boxes.dtype)
            classes[idx[0], idx[1]] = torch.cat([g.gt_classes.to(dev) for g in gt_instances])
        return boxes, classes, counts

    def _fused_training_step(self, batch: torch.Tensor, gt_instances: List[Instances]) -> Dict[str, torch.Tensor]:
        """Forward plan -> one fused label-assignment + loss call on the raw head tensors; backward: the loss
        gradient kernel writes the head gradients where the backward plan reads them.  No torch op touches a head
        tensor and nothing is read back to the host."""
        bb = self.backbone
        eng = bb.train_engine()
        named = bb.trainable_named_parameters()
        boxes, classes, counts = self._padded_targets(gt_instances)
        N, _, Hs, Ws = batch.shape
        H, W = (Hs + 31) // 32 * 32, (Ws + 31) // 32 * 32
        key = (H, W)
        if key not in self._anchor_cache:
            grids = [(H // s, W // s) for s in self._get_strides()]
            self._anchor_cache[key] = Boxes.cat(self.anchor_generator.grid_anchors(grids)).tensor.to(self.device).contiguous()
        cfgd = dict(anchors=self._anchor_cache[key], strides=self._get_strides(), num_anchors=self.num_anchors,
                    num_classes=self.num_classes, match_thr=float(self.matcher_threshold),
                    obj_weight=float(self.obj_weight), noobj_weight=float(self.noobj_weight))
        conf, cls, box, n_pos = _FusedTrainStep.apply(eng, batch, tuple(self._mean), tuple(self._std), named, boxes,
                                                     classes.to(torch.int32), counts, cfgd, *[p for _, p in named])
        self.num_pos_anchors = n_pos
        return {"confs_loss": conf, "logits_loss": cls, "regression_loss": box}

    @staticmethod
    def _transpose_dense_predictions(predictions, dims_per_anchor: List[int]):
        """(N, A*K, H, W) -> (N, H*W*A, K) per level and prediction kind (detectron2 DenseDetector)."""
        res = []
        for pred, K in zip(predictions, dims_per_anchor):
            out = []
            for t in pred:
                N, _, H, W = t.shape
                out.append(t.view(N, -1, K, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, K))
            res.append(out)
        return res

    def forward_training(self, images, features, predictions, gt_instances):
        pred_logits, pred_confs, pred_anchor_deltas = self._transpose_dense_predictions(
            predictions, [self.num_classes, 1, 4])
        anchors = self.anchor_generator(features)
        grid_sizes = [f.shape[-2:] for f in features]
        gt_labels, gt_boxes = self.label_anchors_batched(anchors, gt_instances)
        return self.losses(anchors, pred_logits, pred_confs, pred_anchor_deltas, gt_labels, gt_boxes, self._get_strides())

    def losses(self, anchors: List[Boxes], pred_logits, pred_confs, pred_anchor_deltas, gt_labels, gt_boxes, strides):
        gt_labels = torch.stack(gt_labels)  # [N, R]
        gt_boxes = torch.stack(gt_boxes)
        positive_mask = (gt_labels >= 0) & (gt_labels != self.num_classes)
        negative_mask = gt_labels == self.num_classes
        self.num_pos_anchors = positive_mask.sum()  # the reference logs this per image (yolo.py:143-144)

        pred_confs = torch.cat(pred_confs, dim=1)
        confidence_target = torch.zeros_like(pred_confs)
        confidence_target[positive_mask] = 1
        confidence_loss = self._compute_confidence_loss(pred_confs, confidence_target, positive_mask, negative_mask)

        pred_logits = torch.cat(pred_logits, dim=1)
        target = F.one_hot(gt_labels[positive_mask], self.num_classes + 1)[:, :-1].to(pred_logits.dtype)
        logits_loss = self._compute_logits_loss(pred_logits[positive_mask], target)

        st = torch.tensor(strides, device=self.device)
        st = st.repeat_interleave(torch.tensor([d.shape[1] for d in pred_anchor_deltas], device=self.device))
        pred_anchor_deltas = torch.cat(pred_anchor_deltas, dim=1)
        regression_loss = self._compute_box_regression_loss(pred_anchor_deltas, gt_boxes, anchors, positive_mask, st)
        return {"confs_loss": confidence_loss, "logits_loss": logits_loss, "regression_loss": regression_loss}

    def _compute_box_regression_loss(self, pred_anchor_deltas, gt_boxes, anchors, positive_mask, strides):
        target = self.box2box_transform.get_deltas(gt_boxes, anchors, strides)
        return F.smooth_l1_loss(pred_anchor_deltas[positive_mask], target[positive_mask], reduction="mean")

    def _compute_confidence_loss(self, pred_confs, target, objectness_mask, noobjectness_mask):
        obj = F.binary_cross_entropy(pred_confs[objectness_mask], target[objectness_mask], reduction="mean")
        noobj = F.binary_cross_entropy(pred_confs[noobjectness_mask], target[noobjectness_mask], reduction="mean")
        return obj * self.obj_weight + noobj * self.noobj_weight

    def _compute_logits_loss(self, pred_logits, target):
        return F.binary_cross_entropy(pred_logits, target, reduction="mean")

    @torch.no_grad()
    def label_anchors_batched(self, anchors: List[Boxes], gt_instances: List[Instances]):
        """label_anchors for the whole batch in a fixed number of tensor ops (the per-image loop of the reference,
        yolo.py:223-262, costs ~30 tiny launches per image): ground truths are padded to the batch maximum and
        masked.  Same rules as Matcher.__call__ (matcher.py:18-48); returns per-image lists like label_anchors."""
        a = Boxes.cat(anchors).tensor                                   # [R, 4]
        N, dev = len(gt_instances), a.device
        boxes, classes, counts = self._padded_targets(gt_instances)
        K = boxes.shape[1]
        valid = torch.arange(K, device=dev)[None, :] < counts[:, None]
        area_g = (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])          # [N, K]
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])                                  # [R]
        wh = (torch.min(boxes[:, :, None, 2:], a[None, None, :, 2:]) -
              torch.max(boxes[:, :, None, :2], a[None, None, :, :2])).clamp_(min=0)
        inter = wh[..., 0] * wh[..., 1]                                                     # [N, K, R]
        mq = torch.where(inter > 0, inter / (area_g[..., None] + area_a - inter), inter.new_zeros(()))
        best = mq.max(dim=2, keepdim=True).values
        cg = (boxes[..., :2] + boxes[..., 2:]) / 2
        ca = (a[:, :2] + a[:, 2:]) / 2
        dist = (cg[:, :, None, :] - ca[None, None, :, :]).abs().sum(-1)
        dist = torch.where(mq == best, dist, dist.new_full((), float("inf")))
        pos = dist.argmin(dim=2)                                                            # [N, K] positive anchor per gt
        mqv = torch.where(valid[..., None], mq, mq.new_full((), -1.0))
        best_a, matches = mqv.max(dim=1)                                                    # [N, R]
        labels = torch.full_like(matches, self.matcher_labels[1])
        labels[best_a > self.matcher_threshold] = self.matcher_labels[0]
        n_idx = torch.arange(N, device=dev)[:, None].expand(N, K)[valid]
        k_idx = torch.arange(K, device=dev)[None, :].expand(N, K)[valid]
        labels[n_idx, pos[valid]] = self.matcher_labels[2]
        matches[n_idx, pos[valid]] = k_idx
        gt_labels = classes.gather(1, matches)
        gt_labels[labels == 0] = self.num_classes
        gt_labels[labels == -1] = -1
        gt_boxes = boxes.gather(1, matches[..., None].expand(-1, -1, 4))
        return list(gt_labels.unbind(0)), list(gt_boxes.unbind(0))

    @torch.no_grad()
    def label_anchors(self, anchors: List[Boxes], gt_instances: List[Instances], grid_sizes):
        """Per image: label of every anchor (gt class / num_classes = negative / -1 = ignored) and its matched
        ground-truth box."""
        anchors = Boxes.cat(anchors)
        gt_labels, matched_gt_boxes = [], []
        for gt in gt_instances:
            mq = pairwise_iou(gt.gt_boxes, anchors)
            matched_idx, anchor_labels = self.anchor_matcher(mq, anchors, gt.gt_boxes)
            boxes_i = gt.gt_boxes.tensor[matched_idx]
            labels_i = gt.gt_classes[matched_idx]
            labels_i[anchor_labels == 0] = self.num_classes
            labels_i[anchor_labels == -1] = -1
            gt_labels.append(labels_i)
            matched_gt_boxes.append(boxes_i)
        return gt_labels, matched_gt_boxes

    @torch.no_grad()
    def inference(self, batched_inputs: List[Dict]):
        if self.device.type != "cuda":
            raise RuntimeError("yolo_amd runs on MI355X only (MODEL.DEVICE must be 'cuda'); there is no CPU fallback")
        batch, sizes = self._batch_images(batched_inputs)
        out = [(int(x.get("height", s[0])), int(x.get("width", s[1]))) for x, s in zip(batched_inputs, sizes)]
        in_hw, out_hw = self._size_tensors(tuple(sizes), tuple(out))
        eng = self.backbone.engine()
        N = len(sizes)
        # few images per call: the call's launches are replayed from a captured HIP graph once the same call signature
        # has come a few times (off unless YOLO_HIP_GRAPH=1 / model.graph_max_batch is set)
        detect = eng.detect_graphed if (self.graph_max_batch and N <= self.graph_max_batch) else eng.detect
        for _ in range(2):
            det, n_det = detect(
                batch, score_thr=self.test_score_thresh, nms_thr=self.test_nms_thresh, mean=self._mean, std=self._std,
                max_det=self.max_detections_pre_image, nms_mode=self.nms_mode, auto_limit=self.nms_auto_numel,
                in_hw=in_hw, out_hw=out_hw)
            # ONE read-back per call: the per-image detection counts and, behind them, the number of candidates the
            # decode produced.  Rare (thousands of candidates per image, e.g. a very low threshold): the candidate
            # buffers were too small and the tail of the candidates was dropped -- run the batch again with larger ones
            h = eng.last_in_flight
            counts = torch.cat([n_det, h.cand.img_offsets[N:N + 1]]).tolist()
            if not eng.grow_candidates(h, total=counts[N]):
                break
        n_det = counts[:N]
        boxes = det[..., :4].contiguous()      # three conversions for the whole batch, then views per image
        scores = det[..., 4].contiguous()
        classes = det[..., 5].to(torch.int64)
        results = []
        for n, (oh, ow) in enumerate(out):
            k = n_det[n]
            inst = Instances((oh, ow), pred_boxes=Boxes(boxes[n, :k]), scores=scores[n, :k], pred_classes=classes[n, :k])
            results.append({"instances": inst})
        return results

    def _size_tensors(self, sizes, out):
        """[N, 2] int32 device tensors of the stored and of the requested output sizes; kept for the size
        combinations seen last (a serving loop repeats them), so that a call uploads nothing."""
        key = (sizes, out)
        hit = self._size_cache.get(key)
        if hit is None:
            if len(self._size_cache) >= 64:
                self._size_cache.pop(next(iter(self._size_cache)))
            both = torch.tensor([sizes, out], dtype=torch.int32).pin_memory().to(self.device, non_blocking=True)
            hit = self._size_cache[key] = (both[0], both[1])
        return hit


def _build(cfg):
    return YOLOV3(cfg)


META_ARCH_REGISTRY._add("YOLOV3", _build)  # the name configs/yolov3.yaml:3 selects

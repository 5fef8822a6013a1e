"""Plain-torch CPU restatement of the reference's YOLOv3 TRAINING step (fp32, NCHW, autograd).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference holds no test or
golden vector for this path and cannot be imported here; each function cites the reference lines it
restates (paths relative to the reference checkout).  detectron2 pieces the reference calls
(``pairwise_iou``, ``Boxes.get_centers``, ``FrozenBatchNorm2d``, ``_transpose_dense_predictions``) are
restated from their documented behaviour.

The forward runs on one state dict split into ``params`` (leaf tensors that require grad) and ``buffers``
(batch-norm running statistics); gradients come from torch autograd.  The HIP training path is checked
against these losses and gradients.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import yolo_ref as R

BN_MOMENTUM = 0.1  # torch.nn.BatchNorm2d default, what detectron2 get_norm("BN") builds


# --------------------------------------------------------------------------------------
# backbone + neck in training mode
# --------------------------------------------------------------------------------------
def frozen_names(freeze_at: int, blocks=(1, 2, 8, 8, 4), prefix: str = "backbone.") -> List[str]:
    """Conv names frozen by ``DarkNet.freeze(freeze_at)`` (darknet.py:236-258, called from :201 with
    cfg.MODEL.BACKBONE.FREEZE_AT): 1 freezes the stem, 2 the stem and res1, ...; frozen blocks have
    their parameters' requires_grad cleared and their BatchNorm converted to FrozenBatchNorm2d."""
    names = []
    if freeze_at >= 1:
        names.append(prefix + "bottom_up.stem.conv")
    for si, nb in enumerate(blocks, start=1):
        if freeze_at >= si + 1:
            for b in range(nb):
                p = f"{prefix}bottom_up.res{si}.{b}"
                if b == 0:
                    names.append(p + ".downsample")
                names += [p + ".conv1", p + ".conv2"]
    return names


def conv_unit_train(x, sd, name: str, stride: int, pad: int, act: bool, frozen: bool, stats: dict | None = None,
                    rec: dict | None = None):
    """detectron2 ``Conv2d.forward`` in training mode: F.conv2d, then BatchNorm2d with BATCH statistics
    (biased variance for the normalisation; running stats updated with momentum 0.1 and the unbiased
    variance), or FrozenBatchNorm2d (running statistics, no update) for frozen blocks; then the
    caller's leaky_relu(0.1) (darknet.py:58-59,115-123; fpn.py:34-35,68-69,97-99)."""
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    y = F.conv2d(x, w, b, stride=stride, padding=pad)
    z = y
    if name + ".norm.weight" in sd:
        g, beta = sd[name + ".norm.weight"], sd[name + ".norm.bias"]
        rm, rv = sd[name + ".norm.running_mean"], sd[name + ".norm.running_var"]
        if frozen:
            y = F.batch_norm(y, rm, rv, g, beta, False, BN_MOMENTUM, R.BN_EPS)
        else:
            if stats is not None:  # what the running statistics become after this step
                with torch.no_grad():
                    m = y.mean(dim=(0, 2, 3))
                    n = y.numel() // y.shape[1]
                    v = y.var(dim=(0, 2, 3), unbiased=False) * (n / max(n - 1, 1))
                    stats[name] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * m, (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * v)
            y = F.batch_norm(y, None, None, g, beta, True, BN_MOMENTUM, R.BN_EPS)
    if act:
        y = F.leaky_relu(y, R.LEAKY)
    if rec is not None:  # per-layer record (output before any residual add) for layer-by-layer parity checks
        y.retain_grad()
        if z.requires_grad:
            z.retain_grad()
        rec[name] = dict(x=x, y=y, z=z)
    return y


def fpn_forward_train(x, sd, freeze_at: int = 2, prefix="backbone.", blocks=(1, 2, 8, 8, 4), stats: dict | None = None,
                      rec: dict | None = None):
    """DarkNetFPN.forward (fpn.py:165-185) over DarkNet.forward (darknet.py:203-226) with the modules in
    training mode.  Returns {"p3","p4","p5"} raw head tensors."""
    frozen = set(frozen_names(freeze_at, blocks, prefix))

    def cu(t, name, stride, pad, act=True):
        return conv_unit_train(t, sd, name, stride, pad, act, name in frozen, stats, rec)

    x = cu(x, prefix + "bottom_up.stem.conv", 1, 1)
    bu = {}
    for si, nb in enumerate(blocks, start=1):
        for b in range(nb):
            p = f"{prefix}bottom_up.res{si}.{b}"
            if b == 0:
                x = cu(x, p + ".downsample", 2, 1)  # darknet.py:113-116
            residual = x  # :118 (clone)
            out = cu(x, p + ".conv1", 1, 0)
            out = cu(out, p + ".conv2", 1, 1)
            x = out + residual  # :125
            if rec is not None:
                x.retain_grad()
                rec[p + ".conv2"]["out"] = x
        bu[f"res{si}"] = x
    feats = ["res3", "res4", "res5"]
    result, topdown = [], None
    for idx in range(3):
        f = bu[feats[-idx - 1]]
        if topdown is not None:
            f = torch.cat([topdown, f], dim=1)  # fpn.py:179
        for i in range(5):
            k = 1 if i % 2 == 0 else 3
            f = cu(f, f"{prefix}fpn_lateral{idx}.blocks.{i}", 1, (k - 1) // 2)
        o = cu(f, f"{prefix}fpn_output{idx}.conv", 1, 1)
        o = cu(o, f"{prefix}fpn_output{idx}.head", 1, 0, act=False)
        result.insert(0, o)
        if idx < 2:
            t = cu(f, f"{prefix}fpn_upsample{idx}.conv", 1, 0)
            topdown = F.interpolate(t, scale_factor=2.0, mode="nearest")
    return dict(zip(["p3", "p4", "p5"], result))


# --------------------------------------------------------------------------------------
# label assignment  (yolo/modeling/meta_arch/yolo.py:214-264, yolo/modeling/matcher.py:18-55)
# --------------------------------------------------------------------------------------
def pairwise_iou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """detectron2.structures.pairwise_iou on XYXY tensors: [K,4] x [R,4] -> [K,R]; 0 where the boxes do
    not intersect."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype))


def box_centers(b: torch.Tensor) -> torch.Tensor:
    """Boxes.get_centers."""
    return (b[:, :2] + b[:, 2:]) / 2


def matcher(mq: torch.Tensor, anchors: torch.Tensor, gt_boxes: torch.Tensor, threshold: float = 0.5,
            labels: Sequence[int] = (-1, 0, 1)) -> Tuple[torch.Tensor, torch.Tensor]:
    """Matcher.__call__, matcher.py:18-48.  mq: IoU [K gt, R anchors].  Per gt the positive anchor is,
    among the anchors that reach the gt's best IoU, the one whose centre is L1-nearest to the gt's centre;
    every other anchor is ``ignore`` (labels[0]) when its best IoU over the gts exceeds ``threshold`` and
    ``negative`` (labels[1]) otherwise.  Returns (matched gt index per anchor, label per anchor)."""
    assert mq.dim() == 2
    matched_val, _ = mq.max(dim=1)  # :23
    iou_mask = mq == matched_val[:, None]  # :24
    dist = (box_centers(gt_boxes)[:, None, :] - box_centers(anchors)[None, :, :]).abs().sum(dim=-1)  # :50-55
    dist = dist.where(iou_mask, torch.tensor(float("inf")))  # :27
    _, center_matches = dist.min(dim=1)  # :28
    distance_mask = torch.full(dist.shape, False)
    distance_mask[range(gt_boxes.shape[0]), center_matches] = True  # :30-31
    gt_idx, anchor_idx = torch.nonzero(iou_mask & distance_mask, as_tuple=True)  # :33-35
    matched_val, matches = mq.max(dim=0)  # :37
    match_labels = torch.full(matches.size(), labels[1], dtype=torch.int8)  # :38
    match_labels[matched_val > threshold] = labels[0]  # :40
    match_labels[anchor_idx] = labels[2]  # :41
    matches[anchor_idx] = gt_idx  # :43
    return matches, match_labels


def label_anchors(anchors: torch.Tensor, gt_boxes: List[torch.Tensor], gt_classes: List[torch.Tensor],
                  num_classes: int, threshold: float = 0.5, labels: Sequence[int] = (-1, 0, 1)):
    """YOLOV3.label_anchors, yolo.py:214-264.  anchors: all levels concatenated [R,4].  Per image returns
    labels [R] (gt class for positives, ``num_classes`` for negatives, -1 for ignored) and the matched gt
    box of every anchor [R,4]."""
    out_labels, out_boxes = [], []
    for boxes, classes in zip(gt_boxes, gt_classes):
        mq = pairwise_iou(boxes, anchors)  # :224
        idx, anchor_labels = matcher(mq, anchors, boxes, threshold, labels)  # :227
        mb = boxes[idx]  # :256
        lab = classes[idx].clone()  # :257
        lab[anchor_labels == 0] = num_classes  # :258
        lab[anchor_labels == -1] = -1  # :259
        out_labels.append(lab)
        out_boxes.append(mb)
    return out_labels, out_boxes


# --------------------------------------------------------------------------------------
# losses  (yolo.py:131-212, box_regression.py:18-42)
# --------------------------------------------------------------------------------------
def get_deltas(gt_boxes: torch.Tensor, anchors: torch.Tensor, strides: torch.Tensor) -> torch.Tensor:
    """Box2BoxTransform.get_deltas, box_regression.py:18-42, weights (1,1,1,1).  gt_boxes [N,R,4],
    anchors [R,4], strides [R] -> [N,R,4] = (frac(cx/stride) - 0.5, frac(cy/stride) - 0.5,
    log(w/aw + 1e-16), log(h/ah + 1e-16))."""
    gt_wh = gt_boxes[..., 2:] - gt_boxes[..., :2]  # :23
    gt_cxy = gt_boxes[..., :2] + gt_wh / 2  # :24
    anchors_wh = anchors[..., 2:] - anchors[..., :2]  # :27
    gt_cxy = gt_cxy / strides[..., None]  # :29
    delta_cxy = gt_cxy - torch.floor(gt_cxy) - 0.5  # :30
    delta_wh = torch.log(gt_wh / anchors_wh[None, :, :] + 1e-16)  # :32
    return torch.stack((delta_cxy[..., 0], delta_cxy[..., 1], delta_wh[..., 0], delta_wh[..., 1]), dim=-1)  # :34-40


def losses(anchors: torch.Tensor, pred_logits: List[torch.Tensor], pred_confs: List[torch.Tensor],
           pred_deltas: List[torch.Tensor], gt_labels: List[torch.Tensor], gt_boxes: List[torch.Tensor],
           strides: Sequence[int], num_classes: int, obj_weight: float = 1.0, noobj_weight: float = 100.0):
    """YOLOV3.losses, yolo.py:131-184 with the helpers :186-212.  pred_* are per level [N, HWA, K]."""
    labels = torch.stack(gt_labels)  # :136
    boxes = torch.stack(gt_boxes)  # :137
    positive = (labels >= 0) & (labels != num_classes)  # :140
    negative = labels == num_classes  # :141
    confs = torch.cat(pred_confs, dim=1)  # :148
    target = torch.zeros_like(confs)
    target[positive] = 1  # :150
    obj = F.binary_cross_entropy(confs[positive], target[positive], reduction="mean")  # :197-201
    noobj = F.binary_cross_entropy(confs[negative], target[negative], reduction="mean")  # :202-206
    confs_loss = obj * obj_weight + noobj * noobj_weight  # :207
    logits = torch.cat(pred_logits, dim=1)  # :159
    onehot = F.one_hot(labels[positive], num_classes + 1)[:, :-1].to(logits.dtype)  # :160
    logits_loss = F.binary_cross_entropy(logits[positive], onehot, reduction="mean")  # :211
    st = torch.tensor(list(strides), dtype=torch.float32)
    st = st.repeat_interleave(torch.tensor([d.shape[1] for d in pred_deltas]))  # :167-170
    deltas = torch.cat(pred_deltas, dim=1)  # :171
    tgt = get_deltas(boxes, anchors, st)  # :187
    reg = F.smooth_l1_loss(deltas[positive], tgt[positive], reduction="mean")  # :188-192
    return {"confs_loss": confs_loss, "logits_loss": logits_loss, "regression_loss": reg}


def forward_training(x: torch.Tensor, sd: Dict[str, torch.Tensor], gt_boxes: List[torch.Tensor],
                     gt_classes: List[torch.Tensor], num_classes: int, freeze_at: int = 2,
                     anchors_wh=R.DEFAULT_ANCHORS, strides=R.STRIDES, threshold: float = 0.5,
                     stats: dict | None = None, heads_out: dict | None = None, rec: dict | None = None):
    """DenseDetector.forward in training mode -> YOLOV3.forward_training (yolo.py:101-124) on normalised,
    padded images x [N,3,H,W].  Returns the three losses."""
    heads = fpn_forward_train(x, sd, freeze_at, stats=stats, rec=rec)
    feats = [heads["p3"], heads["p4"], heads["p5"]]
    if heads_out is not None:
        heads_out.update(heads)
    A = len(anchors_wh[0])
    logits, confs, deltas = R.head_forward(feats, A, num_classes)  # YoloV3Head.forward :392-420
    logits = [R.permute_to_N_HWA_K(t, num_classes) for t in logits]  # :107-109
    confs = [R.permute_to_N_HWA_K(t, 1) for t in confs]
    deltas = [R.permute_to_N_HWA_K(t, 4) for t in deltas]
    grids = [(f.shape[2], f.shape[3]) for f in feats]
    anchors = torch.cat(R.grid_anchors(grids, strides, anchors_wh=anchors_wh))  # :110, Boxes.cat :220
    with torch.no_grad():
        labels, boxes = label_anchors(anchors, gt_boxes, gt_classes, num_classes, threshold)  # :121
    return losses(anchors, logits, confs, deltas, labels, boxes, strides, num_classes)  # :123-124


def synthetic_targets(n_images: int, size: int, num_classes: int, seed: int = 7, max_boxes: int = 20):
    """COCO-shaped synthetic targets: per image K ~ U{1..max_boxes} boxes inside the image (XYXY, at least
    8 px wide and high), classes uniform."""
    g = torch.Generator().manual_seed(seed)
    boxes, classes = [], []
    for _ in range(n_images):
        k = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
        wh = 8 + torch.rand(k, 2, generator=g) * (size * 0.6 - 8)
        xy = torch.rand(k, 2, generator=g) * (size - wh)
        boxes.append(torch.cat([xy, xy + wh], dim=1))
        classes.append(torch.randint(0, num_classes, (k,), generator=g))
    return boxes, classes

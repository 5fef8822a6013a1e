"""Plain-torch CPU restatement of the reference's YOLOv3 inference path (fp32, NCHW).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: no reference
test or golden vector exists for this path; each function cites the reference lines
it restates.  Citations are relative to the reference checkout.

Layer naming equals the reference's ``state_dict`` keys so one state dict drives both
this oracle and the HIP engine.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm2d / detectron2 FrozenBatchNorm2d default
LEAKY = 0.1  # yolo/modeling/backbone/darknet.py:59,116,120,123 ; fpn.py:35,69,98

# configs/yolov3.yaml:14-17 ; yolo/config/defaults.py:28-30
DEFAULT_ANCHORS = (
    ((10, 13), (16, 30), (33, 23)),
    ((30, 61), (62, 45), (59, 119)),
    ((116, 90), (156, 198), (373, 326)),
)
STRIDES = (8, 16, 32)  # fpn.py:155-159 via darknet.py:177-179


# --------------------------------------------------------------------------------------
# network description
# --------------------------------------------------------------------------------------
def conv_table(num_classes: int = 80, num_anchors: int = 3, stem_out: int = 32,
               res1_out: int = 64, blocks: Sequence[int] = (1, 2, 8, 8, 4)) -> List[dict]:
    """Every conv of DarkNetFPN in ``darknet_modules()`` order (fpn.py:187-190).

    Module registration order: darknet.py:75-106 (downsample, conv1, conv2), fpn.py:130-141
    (lateral, output, upsample per level, stride-32 level first).  ``stem_out``/``res1_out``
    follow defaults.py:20-21; ``blocks`` follows darknet.py:314-316.  Reduced widths are only
    used by small test fixtures.
    """
    t: List[dict] = []

    def add(name, cin, cout, k, s, bn=True):
        t.append(dict(name=name, cin=cin, cout=cout, k=k, stride=s, pad=(k - 1) // 2, bn=bn))

    add("bottom_up.stem.conv", 3, stem_out, 3, 1)  # darknet.py:45-53
    cin, cout = stem_out, res1_out
    for si, nb in enumerate(blocks, start=1):  # darknet.py:320-331
        for b in range(nb):
            p = f"bottom_up.res{si}.{b}"
            if b == 0:  # darknet.py:282
                add(p + ".downsample", cin, cout, 3, 2)  # darknet.py:76-84
            add(p + ".conv1", cout, cout // 2, 1, 1)  # darknet.py:89-97
            add(p + ".conv2", cout // 2, cout, 3, 1)  # darknet.py:98-106
        cin, cout = cout, cout * 2
    chans = [res1_out * 4, res1_out * 8, res1_out * 16]  # res3, res4, res5 widths
    head_out = (4 + 1 + num_classes) * num_anchors  # fpn.py:206
    for idx, c_in in enumerate(reversed(chans)):  # fpn.py:122
        lat = c_in // 2  # fpn.py:123
        if idx > 0:
            c_in += c_in // 2  # fpn.py:128
        ci, co = c_in, lat
        for i in range(5):  # fpn.py:46-61
            k = 1 if i % 2 == 0 else 3
            if i > 0:
                ci, co = co, (lat if k == 1 else lat * 2)
            add(f"fpn_lateral{idx}.blocks.{i}", ci, co, k, 1)
        add(f"fpn_output{idx}.conv", lat, lat * 2, 3, 1)  # fpn.py:77-85
        add(f"fpn_output{idx}.head", lat * 2, head_out, 1, 1, bn=False)  # fpn.py:86-92
        if idx + 1 < len(chans):
            add(f"fpn_upsample{idx}.conv", lat, lat // 2, 1, 1)  # fpn.py:23-30
    return t


def init_state_dict(seed: int = 0, prefix: str = "backbone.", **kw) -> Dict[str, torch.Tensor]:
    """Seeded random init equal in distribution to the reference's constructors.

    ``c2_msra_fill`` (darknet.py:55,108-111 ; fpn.py:63-64,93-94) = kaiming_normal_(fan_out,
    relu), bias 0.  The two upsample 1x1 convs keep nn.Conv2d's default init (fpn.py:23-31 has
    no fill call).  BatchNorm: weight 1, bias 0, mean 0, var 1.
    """
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for L in conv_table(**kw):
        n = prefix + L["name"]
        shape = (L["cout"], L["cin"], L["k"], L["k"])
        if "fpn_upsample" in L["name"]:
            fan_in = L["cin"] * L["k"] * L["k"]
            bound = math.sqrt(6.0 / ((1 + 5.0) * fan_in))  # kaiming_uniform_(a=sqrt(5))
            w = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:
            std = math.sqrt(2.0 / (L["cout"] * L["k"] * L["k"]))
            w = torch.randn(shape, generator=g) * std
        sd[n + ".weight"] = w
        if L["bn"]:
            sd[n + ".norm.weight"] = torch.ones(L["cout"])
            sd[n + ".norm.bias"] = torch.zeros(L["cout"])
            sd[n + ".norm.running_mean"] = torch.zeros(L["cout"])
            sd[n + ".norm.running_var"] = torch.ones(L["cout"])
        else:
            sd[n + ".bias"] = torch.zeros(L["cout"])
    return sd


def randomize_norm_stats(sd: Dict[str, torch.Tensor], seed: int = 1) -> None:
    """Give every BatchNorm non-trivial affine/statistics (as a trained checkpoint has) so a
    test exercises the folding arithmetic; in place."""
    g = torch.Generator().manual_seed(seed)
    for k in list(sd):
        if k.endswith(".norm.weight"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
        elif k.endswith(".norm.bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith(".norm.running_mean"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith(".norm.running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
        elif k.endswith(".head.bias"):
            sd[k] = 0.5 * torch.randn(sd[k].shape, generator=g)


# --------------------------------------------------------------------------------------
# backbone + neck  (darknet.py, fpn.py)
# --------------------------------------------------------------------------------------
def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.bfloat16().float()


def conv_unit(x: torch.Tensor, sd, name: str, stride: int, pad: int, act: bool = True, quant: bool = False,
              round_out: bool = True, rec: dict | None = None) -> torch.Tensor:
    """detectron2 ``Conv2d.forward`` = F.conv2d then ``self.norm`` (eval BatchNorm), followed by
    the caller's ``F.leaky_relu(., 0.1)`` (e.g. darknet.py:58-59).

    ``quant=True`` restates the same layer as a bf16 run of the reference executes it: the
    batch-norm-folded weights and the layer output are rounded to bfloat16, accumulation in fp32."""
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    has_norm = name + ".norm.weight" in sd
    if not quant:
        y = F.conv2d(x, w, b, stride=stride, padding=pad)
        if has_norm:
            y = F.batch_norm(y, sd[name + ".norm.running_mean"], sd[name + ".norm.running_var"],
                             sd[name + ".norm.weight"], sd[name + ".norm.bias"], False, 0.1, BN_EPS)
    else:
        bias = b if b is not None else torch.zeros(w.shape[0])
        if has_norm:
            scale = sd[name + ".norm.weight"] / torch.sqrt(sd[name + ".norm.running_var"] + BN_EPS)
            w = w * scale.view(-1, 1, 1, 1)
            bias = (bias - sd[name + ".norm.running_mean"]) * scale + sd[name + ".norm.bias"]
        y = F.conv2d(x, _bf16(w), bias, stride=stride, padding=pad)
    if act:
        y = F.leaky_relu(y, LEAKY)
    if quant and round_out:
        y = _bf16(y)
    if rec is not None:  # per-layer record for layer-by-layer ("teacher forced") parity tests
        rec[name] = dict(x=x, y=y, stride=stride, act=act)
    return y


def residual_block(x, sd, p: str, downsample: bool, quant: bool = False, rec: dict | None = None):
    """ResidualBlock.forward, darknet.py:113-126: the add comes AFTER conv2's activation."""
    if downsample:
        x = conv_unit(x, sd, p + ".downsample", 2, 1, quant=quant, rec=rec)  # :114-116
    residual = x.clone()  # :118
    out = conv_unit(x, sd, p + ".conv1", 1, 0, quant=quant, rec=rec)  # :119-120
    out = conv_unit(out, sd, p + ".conv2", 1, 1, quant=quant, round_out=False, rec=rec)  # :122-123
    out = out + residual  # :125 (`out += residual`)
    out = _bf16(out) if quant else out
    if rec is not None:
        rec[p + ".conv2"].update(residual=residual, y=out)
    return out


def darknet_forward(x, sd, prefix="backbone.", blocks=(1, 2, 8, 8, 4), quant: bool = False, rec: dict | None = None):
    """DarkNet.forward darknet.py:203-226 -> {res3,res4,res5}."""
    if quant:
        x = _bf16(x)
    x = conv_unit(x, sd, prefix + "bottom_up.stem.conv", 1, 1, quant=quant, rec=rec)  # BasicStem.forward :57-60
    outs = {}
    for si, nb in enumerate(blocks, start=1):
        for b in range(nb):
            x = residual_block(x, sd, f"{prefix}bottom_up.res{si}.{b}", b == 0, quant, rec)
        outs[f"res{si}"] = x
    return outs


def fpn_forward(x, sd, prefix="backbone.", blocks=(1, 2, 8, 8, 4), taps: dict | None = None, quant: bool = False,
                quant_heads: bool = False, rec: dict | None = None):
    """DarkNetFPN.forward fpn.py:165-185 -> {"p3","p4","p5"} raw head tensors (NCHW).
    ``quant``: bf16 restatement (see conv_unit); ``quant_heads`` also rounds the head tensors."""
    bu = darknet_forward(x, sd, prefix, blocks, quant, rec)
    if taps is not None:
        taps.update(bu)
    feats = ["res3", "res4", "res5"]
    result = []
    topdown = None
    for idx in range(3):
        f = bu[feats[-idx - 1]]  # :173-174
        if topdown is not None:
            f = torch.cat([topdown, f], dim=1)  # :179 -- upsampled first
        for i in range(5):  # LateralBlock.forward :66-71
            k = 1 if i % 2 == 0 else 3
            f = conv_unit(f, sd, f"{prefix}fpn_lateral{idx}.blocks.{i}", 1, (k - 1) // 2, quant=quant, rec=rec)
        o = conv_unit(f, sd, f"{prefix}fpn_output{idx}.conv", 1, 1, quant=quant, rec=rec)  # OutputBlock :96-98
        o = conv_unit(o, sd, f"{prefix}fpn_output{idx}.head", 1, 0, act=False, quant=quant,
                      round_out=quant_heads, rec=rec)  # :99
        result.insert(0, o)  # :177,181
        if idx < 2:  # UpSampleBlock.forward :33-37
            t = conv_unit(f, sd, f"{prefix}fpn_upsample{idx}.conv", 1, 0, quant=quant, rec=rec)
            topdown = F.interpolate(t, scale_factor=2.0, mode="nearest")
    return dict(zip(["p3", "p4", "p5"], result))


# --------------------------------------------------------------------------------------
# pre-processing  (detectron2 DenseDetector.preprocess_image driven by yolov3.yaml:5-6)
# --------------------------------------------------------------------------------------
def preprocess_images(images: List[torch.Tensor], pixel_mean=(0.0, 0.0, 0.0),
                      pixel_std=(255.0, 255.0, 255.0), divisibility: int = 32):
    """(x - mean) / std per image, then zero-pad right/bottom to a common size that is a multiple
    of ``size_divisibility`` (fpn.py:161-163).  Returns (batch NCHW, [(h, w), ...])."""
    mean = torch.tensor(pixel_mean).view(-1, 1, 1)
    std = torch.tensor(pixel_std).view(-1, 1, 1)
    ims = [(im.float() - mean) / std for im in images]
    sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in ims]
    H = max(s[0] for s in sizes)
    W = max(s[1] for s in sizes)
    H = (H + divisibility - 1) // divisibility * divisibility
    W = (W + divisibility - 1) // divisibility * divisibility
    out = torch.zeros(len(ims), ims[0].shape[0], H, W)
    for i, im in enumerate(ims):
        out[i, :, : im.shape[-2], : im.shape[-1]] = im
    return out, sizes


# --------------------------------------------------------------------------------------
# head  (yolo/modeling/meta_arch/yolo.py:392-420)
# --------------------------------------------------------------------------------------
def head_forward(features: List[torch.Tensor], num_anchors: int, num_classes: int):
    """YoloV3Head.forward.  Per anchor the channel layout is [tx,ty,tw,th,obj,cls...]."""
    logits, confs, deltas = [], [], []
    for feat in features:
        N, _, H, W = feat.shape
        f = feat.view(N, num_anchors, -1, H, W)  # :404
        d = f[:, :, :4].reshape(N, -1, H, W)  # :406 (copy)
        c = f[:, :, 4:5].reshape(N, -1, H, W)
        l = f[:, :, 5:].reshape(N, -1, H, W)
        d[:, 0::4] = torch.sigmoid(d[:, 0::4]) - 0.5  # :410
        d[:, 1::4] = torch.sigmoid(d[:, 1::4]) - 0.5  # :411
        confs.append(torch.sigmoid(c))  # :413
        logits.append(torch.sigmoid(l))  # :414
        deltas.append(d)
    return logits, confs, deltas


def permute_to_N_HWA_K(t: torch.Tensor, K: int) -> torch.Tensor:
    """detectron2 ``permute_to_N_HWA_K`` used by ``_transpose_dense_predictions`` (yolo.py:276-278):
    (N, A*K, H, W) -> (N, H*W*A, K); row index = (y*W + x)*A + a."""
    N, _, H, W = t.shape
    return t.view(N, -1, K, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, K)


# --------------------------------------------------------------------------------------
# anchors  (yolo/modeling/anchor_generator.py:18-40 + detectron2 DefaultAnchorGenerator)
# --------------------------------------------------------------------------------------
def cell_anchors(anchors_wh=DEFAULT_ANCHORS) -> List[torch.Tensor]:
    """from_config :20-22 turns (w,h) into size=sqrt(w*h), ratio=h/w (python floats);
    generate_cell_anchors :31-40 turns them back into centred XYXY, cast to float32."""
    out = []
    for level in anchors_wh:
        rows = []
        for (w0, h0) in level:
            size = (w0 * h0) ** (1.0 / 2)
            ratio = h0 / w0
            area = size ** 2
            w = math.sqrt(area / ratio)
            h = ratio * w
            rows.append([w / -2, h / -2, w / 2, h / 2])
        out.append(torch.tensor(rows).float())
    return out


def grid_anchors(grid_sizes: List[Tuple[int, int]], strides=STRIDES, offset: float = 0.5,
                 anchors_wh=DEFAULT_ANCHORS) -> List[torch.Tensor]:
    """DefaultAnchorGenerator._grid_anchors: shifts = arange(offset*stride, size*stride, stride)
    on each axis, row-major over (y, x), anchors innermost -> [H*W*A, 4] per level."""
    res = []
    for (gh, gw), stride, base in zip(grid_sizes, strides, cell_anchors(anchors_wh)):
        sx = torch.arange(offset * stride, gw * stride, step=stride, dtype=torch.float32)
        sy = torch.arange(offset * stride, gh * stride, step=stride, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        xx = xx.reshape(-1)
        yy = yy.reshape(-1)
        shifts = torch.stack((xx, yy, xx, yy), dim=1)
        res.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
    return res


# --------------------------------------------------------------------------------------
# decode  (yolo.py:292-383 + yolo/modeling/box_regression.py:44-64)
# --------------------------------------------------------------------------------------
def apply_deltas(deltas: torch.Tensor, boxes: torch.Tensor, stride: int) -> torch.Tensor:
    """Box2BoxTransform.apply_deltas with weights (1,1,1,1) (yolo.py:88); exp is unclamped."""
    boxes = boxes.to(deltas.dtype)  # :48
    boxes_wh = boxes[:, 2:] - boxes[:, :2]  # :50
    boxes_cxy = boxes[:, :2] + boxes_wh / 2  # :51
    target_cxy = deltas[:, :2] * stride + boxes_cxy  # :52
    target_wh = boxes_wh * deltas[..., 2:].exp()  # :53
    cx = target_cxy[:, 0] / 1.0
    cy = target_cxy[:, 1] / 1.0
    w = target_wh[:, 0] / 1.0
    h = target_wh[:, 1] / 1.0
    x0, y0, x1, y1 = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2  # :61
    return torch.stack([x0, y0, x1, y1], dim=-1)


def decode_per_level(anchors, scores, confs, deltas, stride: int, thresh: float):
    """_decode_per_level_predictions yolo.py:337-383 for num_classes > 1.
    Both filters are strict ``>`` against the same threshold (:356, :367)."""
    mask = (confs > thresh).squeeze(-1)  # :356
    confs = confs[mask]
    deltas = deltas[mask]
    scores = scores[mask]
    anchors = anchors[mask]
    scores = scores * confs  # :363
    i, j = torch.nonzero(scores > thresh, as_tuple=True)  # :367 row-major
    out_scores = scores[i, j]
    boxes = apply_deltas(deltas[i], anchors[i], stride)  # :379
    return boxes, out_scores, j


def decode_image(anchors: List[torch.Tensor], logits, confs, deltas, thresh: float,
                 strides=STRIDES):
    """_decode_multi_level_predictions yolo.py:309-335: levels p3,p4,p5 concatenated in order."""
    bs, ss, cs = [], [], []
    for a, l, c, d, s in zip(anchors, logits, confs, deltas, strides):
        b, sc, cl = decode_per_level(a, l, c, d, s, thresh)
        bs.append(b)
        ss.append(sc)
        cs.append(cl)
    return torch.cat(bs), torch.cat(ss), torch.cat(cs)


def decode_batch(heads: List[torch.Tensor], num_classes: int, thresh: float,
                 anchors_wh=DEFAULT_ANCHORS, strides=STRIDES, num_anchors: int = 3):
    """head + transpose + anchors + per-image decode (yolo.py:266-290 up to, not including, NMS).
    ``heads`` = [p3, p4, p5] raw NCHW tensors.  Returns a list of (boxes, scores, classes)."""
    logits, confs, deltas = head_forward(heads, num_anchors, num_classes)
    logits = [permute_to_N_HWA_K(x, num_classes) for x in logits]
    confs = [permute_to_N_HWA_K(x, 1) for x in confs]
    deltas = [permute_to_N_HWA_K(x, 4) for x in deltas]
    anchors = grid_anchors([tuple(h.shape[-2:]) for h in heads], strides, 0.5, anchors_wh)
    out = []
    for n in range(heads[0].shape[0]):
        out.append(decode_image(anchors, [x[n] for x in logits], [x[n] for x in confs],
                                [x[n] for x in deltas], thresh, strides))
    return out


# --------------------------------------------------------------------------------------
# NMS  (yolo.py:305 -> detectron2 layers/nms.py -> torchvision.ops.boxes.batched_nms)
# --------------------------------------------------------------------------------------
def _smax(a, b):  # std::max(a, b): returns a unless a < b
    return b if a < b else a


def _smin(a, b):  # std::min(a, b): returns a unless b < a
    return b if b < a else a


def nms_greedy(boxes: torch.Tensor, scores: torch.Tensor, iou_thr: float) -> torch.Tensor:
    """torchvision ``nms`` CPU kernel semantics: stable descending sort of the scores, greedy
    sweep, fp32 arithmetic ``inter / (area_i + area_j - inter)``, suppression when the fp32
    quotient compared as double is strictly greater than ``iou_thr``.  Pure Python: small inputs
    only (the C version in oracle/nms_ref.c is the fast one)."""
    import numpy as np

    b = boxes.detach().cpu().numpy().astype(np.float32)
    s = scores.detach().cpu().numpy().astype(np.float32)
    n = b.shape[0]
    order = np.argsort(-s.astype(np.float64), kind="stable") if n else np.zeros(0, np.int64)
    # argsort of the negated values is a stable descending order except for NaN scores, which no
    # caller here produces.
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = np.zeros(n, bool)
    keep = []
    f0 = np.float32(0)
    for _i in range(n):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(int(i))
        for _j in range(_i + 1, n):
            j = order[_j]
            if sup[j]:
                continue
            xx1 = _smax(b[i, 0], b[j, 0])
            yy1 = _smax(b[i, 1], b[j, 1])
            xx2 = _smin(b[i, 2], b[j, 2])
            yy2 = _smin(b[i, 3], b[j, 3])
            with np.errstate(all="ignore"):
                w = _smax(f0, np.float32(xx2 - xx1))
                h = _smax(f0, np.float32(yy2 - yy1))
                inter = np.float32(w * h)
                ovr = np.float32(inter / np.float32(np.float32(areas[i] + areas[j]) - inter))
            if float(ovr) > iou_thr:
                sup[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms_vanilla(boxes, scores, idxs, iou_thr, nms_fn=nms_greedy):
    """torchvision ``_batched_nms_vanilla``: NMS per class on the raw boxes; kept indices returned
    in descending-score order (ties: ascending index, i.e. a stable sort)."""
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for cid in torch.unique(idxs):
        cur = torch.where(idxs == cid)[0]
        k = nms_fn(boxes[cur], scores[cur], iou_thr)
        keep_mask[cur[k]] = True
    keep = torch.where(keep_mask)[0]
    return keep[torch.sort(scores[keep], descending=True, stable=True)[1]]


def batched_nms_coordinate_trick(boxes, scores, idxs, iou_thr, nms_fn=nms_greedy):
    """torchvision ``_batched_nms_coordinate_trick``: one NMS over boxes shifted by
    ``class * (max_coordinate + 1)`` (the shift changes the fp32 rounding of the coordinates)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_c = boxes.max()
    offsets = idxs.to(boxes) * (max_c + torch.tensor(1).to(boxes))
    return nms_fn(boxes + offsets[:, None], scores, iou_thr)


def batched_nms(boxes, scores, idxs, iou_thr, device_kind: str = "cpu", nms_fn=nms_greedy):
    """torchvision ``batched_nms`` branch choice: vanilla when boxes.numel() > 4000 on CPU
    (> 20000 elsewhere), else the coordinate trick.  detectron2's wrapper first does
    ``boxes.float()``."""
    boxes = boxes.float()
    limit = 4000 if device_kind == "cpu" else 20000
    if boxes.numel() > limit:
        return batched_nms_vanilla(boxes, scores, idxs, iou_thr, nms_fn)
    return batched_nms_coordinate_trick(boxes, scores, idxs, iou_thr, nms_fn)


# --------------------------------------------------------------------------------------
# post-processing  (detectron2 detector_postprocess, called after yolo.py:290)
# --------------------------------------------------------------------------------------
def detector_postprocess(boxes, scores, classes, image_size: Tuple[int, int], out_h: int, out_w: int):
    """Scale boxes from the network-input frame to (out_h, out_w), clip, drop empty boxes."""
    sx = out_w / image_size[1]
    sy = out_h / image_size[0]
    b = boxes.clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b[:, 0].clamp_(min=0, max=out_w)
    b[:, 1].clamp_(min=0, max=out_h)
    b[:, 2].clamp_(min=0, max=out_w)
    b[:, 3].clamp_(min=0, max=out_h)
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    return b[keep], scores[keep], classes[keep]


# --------------------------------------------------------------------------------------
# end to end  (DenseDetector.forward, eval branch)
# --------------------------------------------------------------------------------------
def inference(images: List[torch.Tensor], sd, num_classes: int, score_thr: float, nms_thr: float,
              max_det: int = 100, out_sizes=None, nms_mode: str = "auto", nms_fn=nms_greedy,
              anchors_wh=DEFAULT_ANCHORS, blocks=(1, 2, 8, 8, 4), device_kind: str = "cpu"):
    """preprocess -> backbone -> head -> decode -> batched_nms -> top max_det -> postprocess.
    ``images``: list of CHW tensors with values 0..255.  ``nms_mode``: "auto" follows
    torchvision's branch choice, "vanilla"/"trick" force one."""
    x, sizes = preprocess_images(images)
    heads = fpn_forward(x, sd, blocks=blocks)
    dec = decode_batch([heads["p3"], heads["p4"], heads["p5"]], num_classes, score_thr, anchors_wh)
    results = []
    for n, (b, s, c) in enumerate(dec):
        if nms_mode == "vanilla":
            keep = batched_nms_vanilla(b.float(), s, c, nms_thr, nms_fn)
        elif nms_mode == "trick":
            keep = batched_nms_coordinate_trick(b.float(), s, c, nms_thr, nms_fn)
        else:
            keep = batched_nms(b, s, c, nms_thr, device_kind, nms_fn)
        keep = keep[:max_det]  # yolo.py:307
        oh, ow = out_sizes[n] if out_sizes is not None else sizes[n]
        results.append(detector_postprocess(b[keep], s[keep], c[keep], sizes[n], oh, ow))
    return results

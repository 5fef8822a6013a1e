"""YOLOv3 neck + head convolutions on top of Darknet-53: module tree with the reference's parameter
names (yolo/modeling/backbone/fpn.py:20-215).  ``DarkNetFPN.forward`` runs the whole network through
the fused inference plan of ``yolo_amd/engine.py``."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from ..registry import BACKBONE_REGISTRY, build_anchor_generator
from .darknet import LEAKY_SLOPE, Backbone, CNNBlockBase, build_darknet53_backbone
from .layers import Conv2d, c2_msra_fill, get_norm

__all__ = ["DarkNetFPN", "build_darknet53_fpn_backbone"]


class UpSampleBlock(CNNBlockBase):
    """1x1 conv (C -> C/2) + BN + LeakyReLU + nearest 2x (reference fpn.py:20-37; its conv keeps the
    framework-default initialisation)."""

    def __init__(self, in_channels, out_channels, *, norm):
        super().__init__(in_channels, out_channels, stride=1)
        self.conv = Conv2d(in_channels, out_channels, 1, stride=1, padding=0, bias=False,
                           norm=get_norm(norm, out_channels))

    def forward(self, x):
        out = self.conv(x, slope=LEAKY_SLOPE)
        return out.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


class LateralBlock(CNNBlockBase):
    """Five conv+BN+LeakyReLU layers alternating 1x1 (-> c) and 3x3 (-> 2c) (reference fpn.py:40-71)."""

    def __init__(self, in_channels, out_channels, *, norm):
        super().__init__(in_channels, out_channels, stride=1)
        self.blocks = nn.ModuleList()
        use_bias = norm == ""
        cin, cout = in_channels, out_channels
        for i in range(5):
            k = 1 if i % 2 == 0 else 3
            if i > 0:
                cin, cout = cout, (out_channels if k == 1 else out_channels * 2)
            self.blocks.append(Conv2d(cin, cout, k, stride=1, padding=(k - 1) // 2, bias=use_bias,
                                      norm=get_norm(norm, cout)))
        for conv in self.blocks:
            c2_msra_fill(conv)

    def forward(self, x):
        for conv in self.blocks:
            x = conv(x, slope=LEAKY_SLOPE)
        return x


class OutputBlock(CNNBlockBase):
    """3x3 conv (c -> 2c) + BN + LeakyReLU, then the linear 1x1 head with bias (reference fpn.py:74-100)."""

    def __init__(self, in_channels, out_channels, *, norm):
        super().__init__(in_channels, out_channels, stride=1)
        self.conv = Conv2d(in_channels, in_channels * 2, 3, stride=1, padding=1, bias=False,
                           norm=get_norm(norm, in_channels * 2))
        self.head = Conv2d(in_channels * 2, out_channels, 1, stride=1, padding=0, bias=True)
        c2_msra_fill(self.conv)
        c2_msra_fill(self.head)

    def forward(self, x):
        return self.head(self.conv(x, slope=LEAKY_SLOPE), slope=1.0)


class DarkNetFPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", *, num_classes=None, anchors_wh=None):
        super().__init__()
        self.bottom_up = bottom_up
        self.in_features = in_features
        shapes = bottom_up.output_shape()
        chans = [shapes[f].channels for f in in_features]
        self.output_channels = out_channels
        self.lateral_convs, self.output_convs, self.upsample_convs = [], [], []
        for idx, c_in in enumerate(reversed(chans)):  # stride-32 level first
            lat = c_in // 2
            if idx > 0:
                c_in += c_in // 2  # concat with the upsampled coarser level
            lateral = LateralBlock(c_in, lat, norm=norm)
            self.add_module(f"fpn_lateral{idx}", lateral)
            self.lateral_convs.append(lateral)
            output = OutputBlock(lat, out_channels, norm=norm)
            self.add_module(f"fpn_output{idx}", output)
            self.output_convs.append(output)
            if idx + 1 < len(chans):
                up = UpSampleBlock(lat, lat // 2, norm=norm)
                self.add_module(f"fpn_upsample{idx}", up)
                self.upsample_convs.append(up)
        self._out_features = ["p3", "p4", "p5"]
        self._out_feature_channels = {k: shapes[f].channels for f, k in zip(in_features, self._out_features)}
        self._out_feature_strides = {k: shapes[f].stride for f, k in zip(in_features, self._out_features)}
        # engine settings (not part of the state dict)
        self._num_classes = num_classes
        self._anchors_wh = anchors_wh
        self.compute_dtype = torch.bfloat16
        self.head_dtype = torch.float32
        self._engine = None
        self._engine_key = None
        self._flat_tensors = None
        self._train_engine = None
        self.freeze_at = 0

    @property
    def size_divisibility(self):
        return 32

    def darknet_modules(self):
        """(name, conv) in Darknet file order (reference fpn.py:187-190)."""
        for name, m in self.named_modules():
            if isinstance(m, Conv2d):
                yield name, m

    # ---- fused execution -------------------------------------------------------------------
    def engine(self):
        from ...engine import Yolov3Engine
        key = (self.compute_dtype, self.head_dtype) + self._weights_signature()
        if self._engine is None or self._engine_key != key:
            if self._num_classes is None:
                raise RuntimeError("DarkNetFPN was built without num_classes / anchors; use "
                                   "build_darknet53_fpn_backbone(cfg, ...)")
            self._engine = Yolov3Engine(self, num_classes=self._num_classes, anchors_wh=self._anchors_wh,
                                        strides=[self._out_feature_strides[k] for k in self._out_features],
                                        dtype=self.compute_dtype, head_dtype=self.head_dtype,
                                        blocks=self.bottom_up.blocks_per_stage)
            self._engine_key = key
        return self._engine

    def _weights_signature(self):
        """(device, versions and storage addresses of every parameter and buffer): changes whenever a weight is updated
        in place (optimizer step, load_state_dict), moved (module.to: new storage) or replaced (a new Parameter or
        buffer assigned to a module, also where there was none) -- what the packed, BatchNorm-folded copies of the
        engine depend on.  Walking the module tree costs ~0.3 ms, a quarter of a single-image call, so the flat list of
        (owning dict, name, tensor) is kept and only checked: the size of every module's parameter and buffer dicts,
        the identity of every entry, then version counter and storage address of the tensors."""
        kept = self._flat_tensors
        if kept is not None:
            flat, sizes = kept
            if any(len(d) != n for d, n in sizes) or any(d.get(k) is not t for d, k, t in flat):
                kept = None
        if kept is None:
            flat, sizes = [], []
            for m in self.modules():
                for d in (m._parameters, m._buffers):
                    sizes.append((d, len(d)))
                    flat += [(d, k, t) for k, t in d.items()]
            kept = self._flat_tensors = (flat, sizes)
        live = [t for _, _, t in kept[0] if t is not None]
        return (live[0].device, tuple(t._version for t in live), tuple(t.data_ptr() for t in live))

    def train_engine(self):
        from ...train_engine import TrainEngine
        if self._train_engine is not None and self._train_engine.dtype != self.compute_dtype:
            self._train_engine = None
        if self._train_engine is None:
            if self._num_classes is None:
                raise RuntimeError("DarkNetFPN was built without num_classes / anchors; use "
                                   "build_darknet53_fpn_backbone(cfg, ...)")
            self._train_engine = TrainEngine(self, num_classes=self._num_classes, num_anchors=len(self._anchors_wh[0]),
                                             freeze_at=self.freeze_at, blocks=self.bottom_up.blocks_per_stage,
                                             dtype=self.compute_dtype)
        return self._train_engine

    def trainable_named_parameters(self):
        return [(n, p) for n, p in self.named_parameters() if p.requires_grad]

    def forward_train(self, images: torch.Tensor, mean, std) -> Dict[str, torch.Tensor]:
        """Training-mode forward on raw images [N,3,Hs,Ws] (0..255, uint8 or float32, on the GPU): batch-statistics
        BatchNorm in the trainable stages, frozen stages as in inference.  Returns {"p3","p4","p5"} float32 head
        tensors attached to the autograd graph: ``loss.backward()`` runs the HIP backward plan and leaves the
        gradients in the parameters' ``.grad``."""
        eng = self.train_engine()
        named = self.trainable_named_parameters()
        outs = _TrainFunction.apply(eng, images, tuple(mean), tuple(std), named, *[p for _, p in named])
        return dict(zip(self._out_features, outs))

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """x: normalised images [N, 3, H, W] float32 with H, W multiples of 32.  Returns the raw head
        tensors {"p3","p4","p5"} as [N, A*(5+C), H/s, W/s] (strided views of the plan's NHWC buffers; they
        are overwritten by the next call with the same input shape)."""
        if self.training:
            if not x.is_cuda:
                raise RuntimeError("yolo_amd runs on MI355X only: input is on the CPU and there is no CPU fallback")
            return self.forward_train(x.float(), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
        if not x.is_cuda:
            raise RuntimeError("yolo_amd runs on MI355X only: input is on the CPU and there is no CPU fallback")
        assert x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0, "H, W must be multiples of size_divisibility (32)"
        P = self.engine().run_backbone(x.float())
        return {k: v.nchw() for k, v in zip(self._out_features, P.heads)}


def _frozen(param_name: str, freeze_at: int) -> bool:
    """freeze_at 1 = stem, 2 = stem + res1, ... (reference darknet.py:245-247)."""
    if param_name.startswith("bottom_up.stem."):
        return freeze_at >= 1
    for si in range(1, 6):
        if param_name.startswith(f"bottom_up.res{si}."):
            return freeze_at >= si + 1
    return False


class _TrainFunction(torch.autograd.Function):
    """The whole backbone + neck as one autograd node: forward and backward are explicit plans of HIP launches
    (yolo_amd/train_engine.py); torch only sees the trainable parameters going in and the three head tensors
    coming out."""

    @staticmethod
    def forward(ctx, eng, images, mean, std, named, *params):
        heads = eng.forward(images, mean, std)
        N, _, Hs, Ws = images.shape
        ctx.eng, ctx.named = eng, named
        ctx.plan = eng.plan(N, (Hs + 31) // 32 * 32, (Ws + 31) // 32 * 32)
        return tuple(h.nchw() for h in heads)

    @staticmethod
    def backward(ctx, *dheads):
        dheads = [d if d is not None else torch.zeros_like(h.nchw()) for d, h in zip(dheads, ctx.plan.heads)]
        out = install_param_grads(ctx.eng, ctx.named, lambda: ctx.eng.backward(ctx.plan, dheads))
        return (None, None, None, None, None) + out


def install_param_grads(eng, named, run_backward) -> tuple:
    """``run_backward()`` runs the engine's backward plan, which leaves every parameter gradient in the engine's
    gradient arena, laid out like the parameters.  A parameter without a gradient gets its arena slice AS its
    ``.grad`` (no copy; the usual zero_grad() -> backward() -> step() loop always takes this route).  A ``.grad``
    that already is the arena slice keeps accumulation semantics: what it held is added back.  Any other ``.grad``
    tensor is left to autograd: the returned tuple (one entry per parameter of ``named``) carries the gradient
    for exactly those parameters and None for the rest."""
    views = {id(p): g for p, g in eng.grad_views()}
    held = [(p, p.grad.clone()) for _, p in named
            if p.grad is not None and id(p) in views and p.grad.data_ptr() == views[id(p)].data_ptr()]
    run_backward()
    out = []
    for _, p in named:
        g = views.get(id(p))
        if g is None:
            out.append(None)
        elif p.grad is None:
            p.grad = g
            out.append(None)
        elif p.grad.data_ptr() == g.data_ptr():
            out.append(None)
        else:
            out.append(g)
    for p, old in held:
        p.grad.add_(old)
    return tuple(out)


@BACKBONE_REGISTRY.register()
def build_darknet53_fpn_backbone(cfg, input_shape):
    bottom_up = build_darknet53_backbone(cfg, input_shape)
    in_features = cfg.MODEL.DARKNET.OUT_FEATURES
    shapes = bottom_up.output_shape()
    gen = build_anchor_generator(cfg, [shapes[f] for f in in_features])
    num_anchors = gen.num_cell_anchors
    assert len(set(num_anchors)) == 1, "Using different number of anchors between levels is not currently supported!"
    num_classes = cfg.MODEL.YOLOV3.NUM_CLASSES
    out_channels = (4 + 1 + num_classes) * num_anchors[0]
    fpn = DarkNetFPN(bottom_up, in_features, out_channels, norm=cfg.MODEL.DARKNET.NORM, num_classes=num_classes,
                     anchors_wh=gen.anchors_wh())
    # The decode kernel computes anchor centres as (cell + 0.5) * stride; the training path builds its anchors from
    # MODEL.ANCHOR_GENERATOR.OFFSET.  Any other offset would make the two disagree silently, so it is refused.
    if float(gen.offset) != 0.5:
        raise NotImplementedError(f"MODEL.ANCHOR_GENERATOR.OFFSET = {gen.offset}: the decode kernel is built for the "
                                  "reference's 0.5 (configs/yolov3.yaml:13)")
    fpn.freeze_at = cfg.MODEL.BACKBONE.FREEZE_AT
    for name, p in fpn.named_parameters():  # DarkNet.freeze (reference darknet.py:236-258)
        if _frozen(name, fpn.freeze_at):
            p.requires_grad_(False)
    hip = cfg.MODEL.get("HIP", None)
    if hip is not None:
        fpn.compute_dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[hip.DTYPE]
        fpn.head_dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[hip.HEAD_DTYPE]
    return fpn

"""Darknet-53 bottom-up network: module tree with the reference's parameter names
(yolo/modeling/backbone/darknet.py:38-333), every layer executing as a fused gfx950 launch."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from ... import ops
from ...structures import ShapeSpec
from ..registry import BACKBONE_REGISTRY
from .layers import Conv2d, c2_msra_fill, get_norm

LEAKY_SLOPE = 0.1  # darknet.py:59,116,120,123


class Backbone(nn.Module):
    @property
    def size_divisibility(self) -> int:
        return 0

    @property
    def padding_constraints(self) -> Dict[str, int]:
        return {}

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


class CNNBlockBase(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, stride: int):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self


class BasicStem(CNNBlockBase):
    """3x3/s1 conv + BN + LeakyReLU on the 3-channel image (reference darknet.py:38-60)."""

    def __init__(self, in_channels: int, out_channels: int, norm: str = "BN"):
        super().__init__(in_channels, out_channels, 1)
        self.conv = Conv2d(in_channels, out_channels, 3, stride=1, padding=1, bias=False,
                           norm=get_norm(norm, out_channels))
        c2_msra_fill(self.conv)

    def forward(self, x: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("yolo_amd runs on MI355X only: input is on the CPU and there is no CPU fallback")
        if x.device.index != torch.cuda.current_device():  # launch on the GPU that holds the input
            with torch.cuda.device(x.device):
                return self.forward(x, dtype)
        N, _, H, W = x.shape
        w, b = self.conv.packed(dtype)
        y = ops.alloc_view(N, H, W, self.out_channels, 0, dtype, x.device)
        ops.stem_fwd(x.float().contiguous(), w, b, y, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), slope=LEAKY_SLOPE)
        return y.nchw()


class ResidualBlock(CNNBlockBase):
    """[3x3/s2 downsample] -> 1x1 (C -> C/2) -> 3x3 (C/2 -> C) -> + block input, each conv followed by
    BN + LeakyReLU; the add comes after the last activation (reference darknet.py:63-126)."""

    def __init__(self, in_channels: int, out_channels: int, norm: str, downsample: bool = False):
        super().__init__(in_channels, out_channels, 2 if downsample else 1)
        if downsample:
            self.downsample = Conv2d(in_channels, out_channels, 3, stride=2, padding=1, bias=False,
                                     norm=get_norm(norm, out_channels))
        else:
            self.downsample = None
        mid = out_channels // 2
        self.conv1 = Conv2d(out_channels, mid, 1, stride=1, padding=0, bias=False, norm=get_norm(norm, mid))
        self.conv2 = Conv2d(mid, out_channels, 3, stride=1, padding=1, bias=False, norm=get_norm(norm, out_channels))
        c2_msra_fill(self.conv1)
        c2_msra_fill(self.conv2)
        if downsample:
            c2_msra_fill(self.downsample)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.downsample is not None:
            x = self.downsample(x, slope=LEAKY_SLOPE)
        out = self.conv1(x, slope=LEAKY_SLOPE)
        return self.conv2(out, slope=LEAKY_SLOPE, residual=x)


class DarkNet(Backbone):
    def __init__(self, stem, stages: List[List[CNNBlockBase]], out_features=None, freeze_at: int = 0):
        super().__init__()
        self.stem = stem
        stride = stem.stride
        self._out_feature_strides = {"stem": stride}
        self._out_feature_channels = {"stem": stem.out_channels}
        self.stage_names, self.stages = [], []
        if out_features is not None:
            need = max({"res1": 1, "res2": 2, "res3": 3, "res4": 4, "res5": 5}.get(f, 0) for f in out_features)
            stages = stages[:need]
        name = "stem"
        for i, blocks in enumerate(stages):
            name = f"res{i + 1}"
            stage = nn.Sequential(*blocks)
            self.add_module(name, stage)
            self.stage_names.append(name)
            self.stages.append(stage)
            for b in blocks:
                stride *= b.stride
            self._out_feature_strides[name] = stride
            self._out_feature_channels[name] = blocks[-1].out_channels
        self.stage_names = tuple(self.stage_names)
        self._out_features = list(out_features) if out_features is not None else [name]
        self.blocks_per_stage = [len(b) for b in stages]
        self.freeze(freeze_at)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        assert x.dim() == 4, f"DarkNet takes an input of shape (N, C, H, W). Got {x.shape} instead!"
        outputs = {}
        x = self.stem(x)
        if "stem" in self._out_features:
            outputs["stem"] = x
        for name, stage in zip(self.stage_names, self.stages):
            x = stage(x)
            if name in self._out_features:
                outputs[name] = x
        return outputs

    def freeze(self, freeze_at: int = 0):
        """Reference darknet.py:236-258: freeze_at >= 1 freezes the stem, >= k+1 freezes res<k>."""
        if freeze_at >= 1:
            self.stem.freeze()
        for idx, stage in enumerate(self.stages, start=2):
            if freeze_at >= idx:
                for block in stage.children():
                    block.freeze()
        return self

    @staticmethod
    def make_stage(block_class, num_blocks: int, *, in_channels: int, out_channels: int, norm: str):
        blocks = []
        for i in range(num_blocks):
            blocks.append(block_class(in_channels=in_channels, out_channels=out_channels, downsample=(i == 0),
                                      norm=norm))
            in_channels = out_channels
        return blocks


@BACKBONE_REGISTRY.register()
def build_darknet53_backbone(cfg, input_shape):
    norm = cfg.MODEL.DARKNET.NORM
    stem = BasicStem(input_shape.channels, cfg.MODEL.DARKNET.STEM_OUT_CHANNELS, norm=norm)
    in_c = cfg.MODEL.DARKNET.STEM_OUT_CHANNELS
    out_c = cfg.MODEL.DARKNET.RES1_OUT_CHANNELS
    counts = {53: [1, 2, 8, 8, 4]}[cfg.MODEL.DARKNET.DEPTH]
    stages = []
    for n in counts:
        stages.append(DarkNet.make_stage(ResidualBlock, n, in_channels=in_c, out_channels=out_c, norm=norm))
        in_c, out_c = out_c, out_c * 2
    return DarkNet(stem, stages, out_features=cfg.MODEL.DARKNET.OUT_FEATURES, freeze_at=cfg.MODEL.BACKBONE.FREEZE_AT)

"""Importing this package registers the builders under the names the YAML files use
(reference yolo/modeling/__init__.py:6-8)."""
from .anchor_generator import YoloAnchorGenerator
from .backbone import DarkNet, DarkNetFPN, build_darknet53_backbone, build_darknet53_fpn_backbone
from .box_regression import Box2BoxTransform
from .meta_arch import YOLOV3, YoloV3Head
from .registry import (ANCHOR_GENERATOR_REGISTRY, BACKBONE_REGISTRY, META_ARCH_REGISTRY, build_anchor_generator,
                       build_backbone, build_model)

__all__ = [k for k in globals().keys() if not k.startswith("_")]

from .darknet import DarkNet, build_darknet53_backbone
from .fpn import DarkNetFPN, build_darknet53_fpn_backbone

__all__ = ["DarkNet", "DarkNetFPN", "build_darknet53_backbone", "build_darknet53_fpn_backbone"]

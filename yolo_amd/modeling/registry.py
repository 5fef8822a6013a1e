"""Registries under the names the reference's YAML files select builders by
(configs/yolov3.yaml:3,8,12)."""
from ..utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")


def build_backbone(cfg, input_shape=None):
    from ..structures import ShapeSpec
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)


def build_model(cfg):
    """Instantiate cfg.MODEL.META_ARCHITECTURE and move it to cfg.MODEL.DEVICE (what detectron2's
    build_model does for tools/inference.py:25 and tools/train_net.py:120)."""
    import torch
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model

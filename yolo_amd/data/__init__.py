"""Training-image augmentation with the reference's class names (yolo/data/__init__.py), running on the device."""
from .augmentation_impl import LetterBox, RandomAffine, RandomColorJitter, RandomFlip, ResizeShortestEdge
from .dataset_mapper import YOLOV3DatasetMapper
from .detection_utils import build_augmentation, read_image
from .transforms import (AffineTransform, ColorJitterTransform, HFlipTransform, LetterBoxTransform, NoOpTransform,
                         ResizeTransform, TransformList, VFlipTransform)

__all__ = ["YOLOV3DatasetMapper", "build_augmentation", "read_image", "RandomAffine", "RandomColorJitter", "LetterBox",
           "ResizeShortestEdge", "RandomFlip", "AffineTransform", "ColorJitterTransform", "LetterBoxTransform",
           "ResizeTransform", "HFlipTransform", "VFlipTransform", "NoOpTransform", "TransformList"]


def __getattr__(name):
    from . import augmentation_impl, transforms
    if name in augmentation_impl._NOT_PORTED or name in transforms._NOT_PORTED:
        return getattr(augmentation_impl if name in augmentation_impl._NOT_PORTED else transforms, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

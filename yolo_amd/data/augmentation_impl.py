"""Random augmentation policies with the reference's names, arguments and random draws
(yolo/data/augmentation_impl.py; ResizeShortestEdge / RandomFlip / AugInput / AugmentationList as detectron2's,
which yolo/data/detection_utils.py:55,81-86,104,130-135 and dataset_mapper.py:39,178-179 use)."""
from __future__ import annotations

import random
from typing import List, Sequence, Union

import numpy as np

from .transforms import (COLOR_BGR2HSV, COLOR_HSV2BGR, COLOR_HSV2RGB, COLOR_RGB2HSV, AffineTransform,
                         ColorJitterTransform, HFlipTransform, LetterBoxTransform, NoOpTransform, ResizeTransform,
                         Transform, TransformList, VFlipTransform)


class Augmentation:
    def get_transform(self, image) -> Transform:
        raise NotImplementedError

    def __call__(self, aug_input: "AugInput") -> Transform:
        tfm = self.get_transform(aug_input.image)
        aug_input.transform(tfm)
        return tfm


class AugInput:
    """The image being augmented (detectron2 AugInput): ``transform`` applies a transform to it in place."""

    def __init__(self, image):
        self.image = image

    def transform(self, tfm: Transform) -> None:
        self.image = tfm.apply_image(self.image)


class AugmentationList(Augmentation):
    def __init__(self, augs: Sequence[Union[Augmentation, Transform]]):
        self.augs = list(augs)

    def plan(self, image) -> TransformList:
        """Draw every transform without touching pixels (each draw only needs the running image size)."""
        shape = ShapeOnly(image.shape[0], image.shape[1])
        tfms: List[Transform] = []
        for a in self.augs:
            t = a if isinstance(a, Transform) else a.get_transform(shape)
            tfms.append(t)
            shape = shape.after(t)
        return TransformList(tfms)

    def __call__(self, aug_input: AugInput) -> TransformList:
        tfms = []
        for a in self.augs:
            if isinstance(a, Transform):
                aug_input.transform(a)
                tfms.append(a)
            else:
                tfms.append(a(aug_input))
        return TransformList(tfms)


class ShapeOnly:
    """Stands in for an image where only ``.shape[:2]`` is read."""

    def __init__(self, h: int, w: int):
        self.shape = (h, w, 3)

    def after(self, t: Transform) -> "ShapeOnly":
        if isinstance(t, ResizeTransform):
            return ShapeOnly(t.new_h, t.new_w)
        if isinstance(t, LetterBoxTransform):
            return ShapeOnly(t.new_size[0] + t.top + t.bottom, t.new_size[1] + t.left + t.right)
        return self


class RandomAffine(Augmentation):
    """augmentation_impl.py:20-45."""

    def __init__(self, translate: float = 0.1, scale: float = 0.1, degree: int = 0, shear: float = 10.0,
                 perspective: float = 0.0):
        self.translate, self.scale, self.degree, self.shear, self.perspective = translate, scale, degree, shear, perspective

    def get_transform(self, image) -> Transform:
        height, width = image.shape[:2]
        # six numpy draws in the reference's order (angle, scale, translate x / y, shear x / y), then two from
        # ``random`` for the projective terms
        spans = [(-self.degree, self.degree), (1 - self.scale, 1 + self.scale),
                 (0.5 - self.translate, 0.5 + self.translate), (0.5 - self.translate, 0.5 + self.translate),
                 (-self.shear, self.shear), (-self.shear, self.shear)]
        angle, scale, tx, ty, shx, shy = [np.random.uniform(lo, hi) for lo, hi in spans]
        shear = tuple(np.tan(v * np.pi / 180) for v in (shx, shy))
        persp = tuple(random.uniform(-self.perspective, self.perspective) for _ in range(2))
        return AffineTransform(height=height, width=width, angle=angle, scale=scale, translate=(tx, ty), shear=shear,
                               perspective=persp)


class RandomColorJitter(Augmentation):
    """augmentation_impl.py:132-153."""

    def __init__(self, image_format: str, hue_gain: float, saturation_gain: float, exposure_gain: float):
        assert image_format in ["BGR", "RGB"], ValueError(f"Only support BGR, RGB format. {image_format}")
        self.image_format, self.hue_gain, self.saturation_gain, self.exposure_gain = (image_format, hue_gain,
                                                                                      saturation_gain, exposure_gain)

    def get_transform(self, image) -> Transform:
        cvt_format = (COLOR_RGB2HSV, COLOR_HSV2RGB) if self.image_format == "RGB" else (COLOR_BGR2HSV, COLOR_HSV2BGR)
        hue, sat, exp = np.random.uniform(-1, 1, 3) * [self.hue_gain, self.saturation_gain, self.exposure_gain] + 1
        return ColorJitterTransform(hue, sat, exp, cvt_format)


class LetterBox(Augmentation):
    """augmentation_impl.py:156-172."""

    def __init__(self, new_height, new_width):
        self.new_height, self.new_width = new_height, new_width

    def get_transform(self, image) -> Transform:
        img_h, img_w = image.shape[:2]
        ratio = min(self.new_height / img_h, self.new_width / img_w)
        resized_h, resized_w = round(img_h * ratio), round(img_w * ratio)
        dh = (self.new_height - resized_h) // 2
        dw = (self.new_width - resized_w) // 2
        return LetterBoxTransform(dh, dh, dw, dw, ratio, (resized_h, resized_w))


class ResizeShortestEdge(Augmentation):
    """detectron2 ResizeShortestEdge(short_edge_length, max_size, sample_style)."""

    def __init__(self, short_edge_length, max_size=2 ** 31 - 1, sample_style="range"):
        assert sample_style in ["range", "choice"], sample_style
        if isinstance(short_edge_length, int):
            short_edge_length = (short_edge_length, short_edge_length)
        self.is_range = sample_style == "range"
        if self.is_range:
            assert len(short_edge_length) == 2, short_edge_length
        self.short_edge_length, self.max_size = tuple(short_edge_length), max_size

    def get_transform(self, image) -> Transform:
        h, w = image.shape[:2]
        if self.is_range:
            size = np.random.randint(self.short_edge_length[0], self.short_edge_length[1] + 1)
        else:
            size = np.random.choice(self.short_edge_length)
        if size == 0:
            return NoOpTransform()
        newh, neww = self.get_output_shape(h, w, size, self.max_size)
        return ResizeTransform(h, w, newh, neww)

    @staticmethod
    def get_output_shape(oldh: int, oldw: int, short_edge_length: int, max_size: int):
        h, w = oldh, oldw
        size = short_edge_length * 1.0
        scale = size / min(h, w)
        if h < w:
            newh, neww = size, scale * w
        else:
            newh, neww = scale * h, size
        if max(newh, neww) > max_size:
            scale = max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return int(newh + 0.5), int(neww + 0.5)


class RandomFlip(Augmentation):
    """detectron2 RandomFlip(prob=0.5, horizontal, vertical)."""

    def __init__(self, prob=0.5, *, horizontal=True, vertical=False):
        if horizontal and vertical:
            raise ValueError("Cannot do both horiz and vert. Please use two Flip instead.")
        if not horizontal and not vertical:
            raise ValueError("At least one of horiz or vert has to be True!")
        self.prob, self.horizontal, self.vertical = prob, horizontal, vertical

    def get_transform(self, image) -> Transform:
        h, w = image.shape[:2]
        do = np.random.uniform(0, 1.0) < self.prob
        if not do:
            return NoOpTransform()
        return HFlipTransform(w) if self.horizontal else VFlipTransform(h)


_NOT_PORTED = ('RandomBlur', 'RandomGaussianBlur', 'RandomBilateralFilter', 'RandomMedianBlur', 'RandomPixelDropout')


def __getattr__(name):
    """The reference's blur / median / bilateral / pixel-dropout classes (yolo/data/augmentation_impl.py:48-129) are not ported: none
    of its augmentation lists (detection_utils.py build_augmentation) uses them.  Asking for one says so instead of
    failing with a bare ImportError."""
    if name in _NOT_PORTED:
        raise NotImplementedError(f"{name} of the reference's yolo/data is not ported to the device path "
                                  "(no augmentation list of the reference uses it)")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

"""Deterministic image transforms with the reference's names and arguments (yolo/data/transforms.py) plus the three
detectron2 ones its augmentation lists use (ResizeTransform, HFlipTransform / VFlipTransform, NoOpTransform).

``apply_image`` takes a uint8 [H, W, 3] tensor on the GPU (a numpy array is uploaded first) and returns a new GPU
tensor, computed by the HIP kernels of yolo_amd/csrc/augment.hip -- there is no CPU path.  ``apply_coords`` works
on float64 numpy points, as in the reference.  ``fused_job`` folds an image's warp / colour jitter / flip into one
kernel pass.  The blur / median / bilateral / pixel-dropout transforms of the reference file (transforms.py:71-169)
are not built: no augmentation list of the reference uses them (yolo/data/detection_utils.py:5,27-137).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import ops

# the OpenCV conversion codes ColorJitterTransform's ``cvt_format`` carries (transforms.py:173; augmentation_impl.py:147-150)
COLOR_BGR2HSV, COLOR_RGB2HSV, COLOR_HSV2BGR, COLOR_HSV2RGB = 40, 41, 54, 55
BORDER_VALUE = 114  # transforms.py:46,48,211


def as_device_image(img, device=None) -> torch.Tensor:
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(np.ascontiguousarray(img))
        img = img.to(device if device is not None else "cuda")
    if not img.is_cuda:
        raise RuntimeError("augmentation runs on the GPU: the image tensor is on the CPU (there is no CPU path)")
    return img.contiguous()


class Transform:
    def apply_image(self, img):
        raise NotImplementedError

    def apply_coords(self, coords: np.ndarray) -> np.ndarray:
        raise NotImplementedError

    def apply_box(self, box: np.ndarray) -> np.ndarray:
        """Corners through ``apply_coords``, then the axis-aligned hull (detectron2 Transform.apply_box, which
        transform_instance_annotations calls, dataset_mapper.py:191-197)."""
        box = np.asarray(box, dtype=np.float64).reshape(-1, 4)
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = self.apply_coords(box[:, idxs].reshape(-1, 2).copy()).reshape((-1, 4, 2))
        return np.concatenate((coords.min(axis=1), coords.max(axis=1)), axis=1)

    def inverse(self) -> "Transform":
        raise NotImplementedError


class NoOpTransform(Transform):
    def apply_image(self, img):
        return img

    def apply_coords(self, coords):
        return coords

    def inverse(self):
        return self


class TransformList(Transform):
    def __init__(self, transforms: Sequence[Transform]):
        flat: List[Transform] = []
        for t in transforms:
            flat.extend(t.transforms if isinstance(t, TransformList) else [t])
        self.transforms = [t for t in flat if not isinstance(t, NoOpTransform)]

    def apply_image(self, img):
        for t in self.transforms:
            img = t.apply_image(img)
        return img

    def apply_coords(self, coords):
        for t in self.transforms:
            coords = t.apply_coords(coords)
        return coords

    def apply_box(self, box):
        for t in self.transforms:
            box = t.apply_box(box)
        return box

    def __add__(self, other):
        return TransformList(self.transforms + (other.transforms if isinstance(other, TransformList) else [other]))

    def __len__(self):
        return len(self.transforms)


class ResizeTransform(Transform):
    """detectron2 ResizeTransform(h, w, new_h, new_w, BILINEAR): Pillow's resize for uint8 images."""

    def __init__(self, h: int, w: int, new_h: int, new_w: int):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w

    def apply_image(self, img):
        img = as_device_image(img)
        assert img.shape[:2] == (self.h, self.w), (img.shape, self.h, self.w)
        return ops.aug_resample(img, (self.new_h, self.new_w))

    def apply_coords(self, coords):
        coords = np.asarray(coords, dtype=np.float64)
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords

    def inverse(self):
        return ResizeTransform(self.new_h, self.new_w, self.h, self.w)


class HFlipTransform(Transform):
    def __init__(self, width: int):
        self.width = width

    def apply_image(self, img):
        img = as_device_image(img)
        out = torch.empty_like(img)
        ops.aug_apply([ops.aug_job(img, out, flip_horizontal=True)])
        return out

    def apply_coords(self, coords):
        coords = np.asarray(coords, dtype=np.float64)
        coords[:, 0] = self.width - coords[:, 0]
        return coords

    def inverse(self):
        return self


class VFlipTransform(Transform):
    def __init__(self, height: int):
        self.height = height

    def apply_image(self, img):
        img = as_device_image(img)
        out = torch.empty_like(img)
        ops.aug_apply([ops.aug_job(img, out, flip_vertical=True)])
        return out

    def apply_coords(self, coords):
        coords = np.asarray(coords, dtype=np.float64)
        coords[:, 1] = self.height - coords[:, 1]
        return coords

    def inverse(self):
        return self


def rotation_matrix_2d(angle: float, scale: float) -> np.ndarray:
    """cv2.getRotationMatrix2D(center=(0, 0), angle, scale): [[a, b, 0], [-b, a, 0]], a = scale cos, b = scale sin."""
    rad = angle * np.pi / 180.0
    a, b = math.cos(rad) * scale, math.sin(rad) * scale
    return np.array([[a, b, 0.0], [-b, a, 0.0]])


def invert_affine_2x3(M: np.ndarray) -> np.ndarray:
    """The inversion cv2.warpAffine applies to a forward 2x3 map before sampling."""
    m = np.array(M, dtype=np.float64).reshape(-1)[:6].copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def _matrix(**entries) -> np.ndarray:
    """3x3 identity with the given entries replaced (``m02=...`` sets row 0, column 2)."""
    m = np.eye(3)
    for name, value in entries.items():
        m[int(name[1]), int(name[2])] = value
    return m


def _compose_affine(height, width, angle, scale, translate, shear, perspective) -> np.ndarray:
    """transforms.py:24-39: move the image centre to the origin, apply the projective terms, rotate and scale about
    the origin, shear, then move by ``translate`` (fractions of the image size)."""
    rot = rotation_matrix_2d(angle, scale)
    steps = [
        _matrix(m02=translate[0] * width, m12=translate[1] * height),
        _matrix(m01=shear[0], m10=shear[1]),
        _matrix(m00=rot[0, 0], m01=rot[0, 1], m10=rot[1, 0], m11=rot[1, 1]),
        _matrix(m20=perspective[0], m21=perspective[1]),
        _matrix(m02=-width / 2, m12=-height / 2),
    ]
    M = steps[0]
    for step in steps[1:]:
        M = M @ step
    return M


class AffineTransform(Transform):
    """transforms.py:8-68: translate @ shear @ rotate-scale @ perspective @ centre, warped with the 114 border."""

    def __init__(self, height: int, width: int, angle: float = 0, scale: float = 1.0,
                 translate: Tuple[float, float] = (0.0, 0.0), shear: Tuple[float, float] = (0.0, 0.0),
                 perspective: Tuple[float, float] = (0.0, 0.0)):
        self.height, self.width, self.angle, self.scale = height, width, angle, scale
        self.translate, self.shear, self.perspective = translate, shear, perspective
        self.M = _compose_affine(height, width, angle, scale, translate, shear, perspective)

    def warp_arguments(self) -> dict:
        if any(self.perspective):
            return dict(inverse_map=np.linalg.inv(np.asarray(self.M, dtype=np.float64)).reshape(-1), perspective=True)
        return dict(inverse_map=invert_affine_2x3(np.asarray(self.M)[:2]), perspective=False)

    def apply_image(self, img):
        img = as_device_image(img)
        out = torch.empty_like(img)
        ops.aug_apply([ops.aug_job(img, out, border=BORDER_VALUE, **self.warp_arguments())])
        return out

    def apply_coords(self, coords):
        c = np.asarray(coords, dtype=np.float64)
        h = np.concatenate([c, np.ones((c.shape[0], 1))], axis=1) @ np.asarray(self.M, dtype=np.float64).T
        out = (h[:, :2] / h[:, 2:3]) if any(self.perspective) else h[:, :2].copy()
        out[:, 0] = np.clip(out[:, 0], 0, self.width)
        out[:, 1] = np.clip(out[:, 1], 0, self.height)
        return out

    def inverse(self) -> "Transform":
        t = AffineTransform(height=self.height, width=self.width, angle=self.angle, scale=self.scale,
                            translate=self.translate, shear=self.shear, perspective=self.perspective)
        t.M = np.linalg.inv(t.M)
        return t


class ColorJitterTransform(Transform):
    """transforms.py:172-192: HSV, ``x * gain`` tables (hue modulo 180, the others clipped to 255), back."""

    def __init__(self, hue, saturation, exposure, cvt_format: Tuple[int, int] = (COLOR_BGR2HSV, COLOR_HSV2BGR)):
        assert tuple(cvt_format) in ((COLOR_BGR2HSV, COLOR_HSV2BGR), (COLOR_RGB2HSV, COLOR_HSV2RGB)), cvt_format
        self.hue, self.saturation, self.exposure, self.cvt_format = hue, saturation, exposure, tuple(cvt_format)

    @property
    def rgb(self) -> bool:
        return self.cvt_format[0] == COLOR_RGB2HSV

    def apply_image(self, img):
        img = as_device_image(img)
        out = torch.empty_like(img)
        ops.aug_apply([ops.aug_job(img, out, jitter=(self.hue, self.saturation, self.exposure), rgb=self.rgb)])
        return out

    def apply_coords(self, coords):
        return coords

    def inverse(self):
        return NoOpTransform()


class LetterBoxTransform(Transform):
    """transforms.py:195-220.  ``ratio`` is the one scale LetterBox.get_transform passes (augmentation_impl.py:168-172)."""

    def __init__(self, top: int, bottom: int, left: int, right: int, ratio, new_size: Tuple[int, int]):
        self.top, self.bottom, self.left, self.right, self.ratio, self.new_size = top, bottom, left, right, ratio, new_size

    def apply_image(self, img):
        img = as_device_image(img)
        nh, nw = self.new_size
        return ops.aug_letterbox(img, (nh + self.top + self.bottom, nw + self.left + self.right), self.top, self.left,
                                 (nh, nw), BORDER_VALUE)

    def apply_coords(self, coords):
        coords = np.asarray(coords, dtype=np.float64)
        coords[:, 0] = coords[:, 0] * self.ratio + self.left
        coords[:, 1] = coords[:, 1] * self.ratio + self.top
        return coords

    def inverse(self):
        return None


def fused_job(transforms: Sequence[Transform], src: torch.Tensor, dst: torch.Tensor, dst_xy=(0, 0), crop=None,
              planar: bool = False, reversed_source: bool = False):
    """One ``yolo_aug_job`` for [AffineTransform] [ColorJitterTransform] [H/VFlipTransform] (each optional, in the
    order the reference's lists produce them, detection_utils.py:57-86,106-135) applied to ``src``.
    ``reversed_source``: ``src`` holds the channels in the opposite order to the one the transforms were built for
    (a BGR upload for an RGB pipeline); the pass converts while writing."""
    kw: dict = {}
    stage = 0
    for t in transforms:
        if isinstance(t, NoOpTransform):
            continue
        if isinstance(t, AffineTransform) and stage < 1:
            kw.update(t.warp_arguments())
            stage = 1
        elif isinstance(t, ColorJitterTransform) and stage < 2:
            kw.update(jitter=(t.hue, t.saturation, t.exposure), rgb=t.rgb != reversed_source)
            stage = 2
        elif isinstance(t, HFlipTransform) and stage < 3:
            kw.update(flip_horizontal=True)
            stage = 3
        elif isinstance(t, VFlipTransform) and stage < 3:
            kw.update(flip_vertical=True)
            stage = 3
        else:
            raise ValueError(f"fused_job: {type(t).__name__} cannot be folded at this position")
    return ops.aug_job(src, dst, dst_xy=dst_xy, crop=crop, border=BORDER_VALUE, planar=planar,
                       reverse_channels=reversed_source, **kw)


_NOT_PORTED = ('BoxFilterTransform', 'BlurTransform', 'GaussianBlurTransform', 'MedianBlurTransform', 'BilateralFilterTransform', 'PixelDropoutTransform')


def __getattr__(name):
    """The reference's blur / median / bilateral / pixel-dropout classes (yolo/data/transforms.py:71-169) are not ported: none
    of its augmentation lists (detection_utils.py build_augmentation) uses them.  Asking for one says so instead of
    failing with a bare ImportError."""
    if name in _NOT_PORTED:
        raise NotImplementedError(f"{name} of the reference's yolo/data is not ported to the device path "
                                  "(no augmentation list of the reference uses it)")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

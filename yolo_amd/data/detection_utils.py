"""yolo/data/detection_utils.py: image reading and the augmentation lists built from the config."""
from __future__ import annotations

import os

import numpy as np

from .augmentation_impl import LetterBox, RandomAffine, RandomColorJitter, RandomFlip, ResizeShortestEdge


def read_bgr(file_name) -> np.ndarray:
    """The decoded image as cv2.imread returns it: HWC uint8, BGR.  Files are decoded with Pillow (OpenCV is not a
    dependency here); ``.npy`` files hold the BGR array itself."""
    assert os.path.exists(file_name), FileNotFoundError(f"{file_name}")
    if str(file_name).endswith(".npy"):
        return np.ascontiguousarray(np.load(file_name))
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(file_name).convert("RGB"))[:, :, ::-1])


def read_image(file_name, format=None) -> np.ndarray:
    """detection_utils.py:8-15: BGR, or RGB if ``format == "RGB"``."""
    if format:
        assert format in ["BGR", "RGB"], ValueError("Only support `BGR` or `RGB` format")
    image = read_bgr(file_name)
    if format and format == "RGB":
        image = np.ascontiguousarray(image[:, :, ::-1])
    return image


def build_augmentation(cfg, is_train: bool = True):
    """detection_utils.py:18-24."""
    if cfg.INPUT.MOSAIC.ENABLE:
        return build_yolov3_augmentation(cfg, is_train)
    return build_normal_augmentation(cfg, is_train)


def _resize(cfg, is_train):
    if is_train:
        return ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING)
    return ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, "choice")


def _random_tail(cfg, is_train):
    augmentation = []
    if is_train and cfg.INPUT.RANDOM_AFFINE.ENABLE:
        augmentation.append(RandomAffine(translate=cfg.INPUT.RANDOM_AFFINE.TRANSLATE, scale=cfg.INPUT.RANDOM_AFFINE.SCALE,
                                         degree=cfg.INPUT.RANDOM_AFFINE.DEGREE, shear=cfg.INPUT.RANDOM_AFFINE.SHEAR,
                                         perspective=cfg.INPUT.RANDOM_AFFINE.PERSPECTIVE))
    if is_train and cfg.INPUT.COLOR_JITTER.ENABLE:
        augmentation.append(RandomColorJitter(image_format=cfg.INPUT.FORMAT, hue_gain=cfg.INPUT.COLOR_JITTER.HUE,
                                              saturation_gain=cfg.INPUT.COLOR_JITTER.SATURATION,
                                              exposure_gain=cfg.INPUT.COLOR_JITTER.EXPOSURE))
    if is_train and cfg.INPUT.RANDOM_FLIP != "none":
        augmentation.append(RandomFlip(horizontal=cfg.INPUT.RANDOM_FLIP == "horizontal",
                                       vertical=cfg.INPUT.RANDOM_FLIP == "vertical"))
    return augmentation


def build_normal_augmentation(cfg, is_train):
    """detection_utils.py:27-87: letterbox (or shortest-edge resize) - random affine - colour jitter - flip."""
    if is_train and cfg.INPUT.LETTERBOX.ENABLE:
        first = LetterBox(new_height=cfg.INPUT.LETTERBOX.HEIGHT, new_width=cfg.INPUT.LETTERBOX.WIDTH)
    else:
        first = _resize(cfg, is_train)
    return [first] + _random_tail(cfg, is_train)


def build_yolov3_augmentation(cfg, is_train):
    """detection_utils.py:90-137: shortest-edge resize - random affine - colour jitter - flip."""
    return [_resize(cfg, is_train)] + _random_tail(cfg, is_train)

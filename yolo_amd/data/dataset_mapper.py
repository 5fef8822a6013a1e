"""YOLOV3DatasetMapper (yolo/data/dataset_mapper.py) with the pixel work on the GPU.

The random draws -- which transforms, the mosaic partners, the mosaic centre -- happen on the host in the reference's
order with the reference's generators (``numpy.random`` / ``random``), so a seeded run draws the same augmentation
parameters.  The pixels never come back: every image is uploaded once as uint8, resized by one kernel
(`yolo_aug_resample` / `yolo_aug_letterbox`) and finished by ONE fused pass (`yolo_aug_apply`: warp -> colour jitter
-> flip -> paste) that writes straight into the CHW tensor the model takes -- for a mosaic, into its quadrant of the
114-grey canvas.  ``map_batch`` plans a whole batch first and launches the fused pass for all of its images together.

Dataset dicts follow detectron2's format (``file_name`` or, for in-memory data, ``image``: HWC uint8 BGR array;
``annotations``: ``bbox`` + ``bbox_mode`` (0 = XYXY_ABS, 1 = XYWH_ABS) + ``category_id`` [+ ``iscrowd``]).
Instance masks / keypoints are not handled (the reference's mapper only uses masks to recompute boxes).
"""
from __future__ import annotations

import copy
import logging
from collections import deque
from typing import List, Optional, Union

import numpy as np
import torch

from ..structures import Boxes, Instances
from .augmentation_impl import Augmentation, AugmentationList
from .detection_utils import build_augmentation, read_bgr
from .transforms import (BORDER_VALUE, LetterBoxTransform, ResizeTransform, Transform, TransformList, as_device_image,
                         fused_job)
from .. import ops

__all__ = ["YOLOV3DatasetMapper"]

XYXY_ABS, XYWH_ABS = 0, 1


def _copy_dict(d: dict) -> dict:
    """``copy.deepcopy(dataset_dict)`` (dataset_mapper.py:87,99,114) that shares an in-memory image array, which
    nothing here writes to."""
    return {k: (v if k == "image" else copy.deepcopy(v)) for k, v in d.items()}


class _Planned:
    """One source image on its way to a destination: the resized device image, the transforms still to apply, the
    transformed annotations."""

    def __init__(self, resized: torch.Tensor, tail: List[Transform], annos: Optional[List[dict]]):
        self.resized, self.tail, self.annos = resized, tail, annos

    @property
    def shape(self):
        return self.resized.shape


class YOLOV3DatasetMapper:
    def __init__(self, *args, **kwargs):
        """``YOLOV3DatasetMapper(cfg, is_train)`` (the reference's ``@configurable`` form, dataset_mapper.py:23,53-77)
        or the explicit keywords of dataset_mapper.py:24-34 (+ ``device``)."""
        if args and hasattr(args[0], "INPUT"):
            kwargs = dict(self.from_config(*args), **kwargs)
            args = ()
        self._init(*args, **kwargs)

    def _init(self, is_train: bool, *, augmentations: List[Union[Augmentation, Transform]], image_format: str,
              mosaic_transform, use_instance_mask: bool = False, instance_mask_format: str = "polygon",
              recompute_boxes: bool = False, device="cuda", cache_bytes: int = 0):
        if use_instance_mask or recompute_boxes:
            raise NotImplementedError("instance masks are outside this mapper")
        self.is_train = is_train
        self.augmentations = AugmentationList(augmentations)
        self.image_format = image_format
        self.device = torch.device(device)
        mode = "training" if is_train else "inference"
        logging.getLogger(__name__).info(f"[DatasetMapper] Augmentations used in {mode}: {augmentations}")
        self._resizes: list = []
        # decoded source images kept in HBM, keyed by file name / image id, up to cache_bytes (0 = off): a uint8
        # detection data set is ~1 MB per image, so one GPU's 288 GB hold it whole and later epochs skip PCIe
        self.cache_bytes, self._cached_bytes, self._cache = int(cache_bytes), 0, {}
        self.mosaic_transform = mosaic_transform
        if self.mosaic_transform.ENABLE:
            self.mosaic_pool = deque(maxlen=self.mosaic_transform.POOL_CAPACITY)

    @classmethod
    def from_config(cls, cfg, is_tran: bool = True):
        """dataset_mapper.py:53-77 (random crop is not part of any shipped config and is rejected)."""
        crop = getattr(cfg.INPUT, "CROP", None)
        if crop is not None and getattr(crop, "ENABLED", False) and is_tran:
            raise NotImplementedError("INPUT.CROP is outside this mapper")
        return {"is_train": is_tran, "augmentations": build_augmentation(cfg, is_tran), "image_format": cfg.INPUT.FORMAT,
                "mosaic_transform": cfg.INPUT.MOSAIC}

    # ------------------------------------------------------------------------------------------
    def _plan(self, dataset_dict: dict, jobs: list, keep: list) -> dict:
        """Everything of ``__call__`` except the launches: appends this image's fused-pass jobs to ``jobs``, the
        tensors they point into to ``keep`` and its pending resizes to ``self._resizes``; returns the output dict."""
        dataset_dict = _copy_dict(dataset_dict)
        mosaic_samples = None
        if self.mosaic_transform.ENABLE and self.is_train:
            if len(self.mosaic_pool) > self.mosaic_transform.NUM_IMAGES:
                np.random.randint(2)  # the reference draws (and ignores) a flag here, dataset_mapper.py:93
                picks = np.random.choice(len(self.mosaic_pool), self.mosaic_transform.NUM_IMAGES - 1)
                mosaic_samples = [self.mosaic_pool[int(i)] for i in picks]
            self.mosaic_pool.append(_copy_dict(dataset_dict))

        first = self._load_image_with_anns(dataset_dict)
        if self.is_train and self.mosaic_transform.ENABLE and mosaic_samples is not None:
            mw, mh = self.mosaic_transform.MOSAIC_WIDTH, self.mosaic_transform.MOSAIC_HEIGHT
            out = torch.full((3, mh * 2, mw * 2), BORDER_VALUE, dtype=torch.uint8, device=self.device)
            out_annos: List[dict] = []
            mosaic_border = (-mh // 2, -mw // 2)
            yc, xc = [int(np.random.uniform(-x, s * 2 + x)) for x, s in zip(mosaic_border, [mh, mw])]
            for m_idx in range(self.mosaic_transform.NUM_IMAGES):
                item = first if m_idx == 0 else self._load_image_with_anns(_copy_dict(mosaic_samples[m_idx - 1]))
                out_annos.extend(self._blend_mosaic(m_idx, item, out, xc, yc, mw, mh, jobs, keep))
            annos, image_hw = out_annos, (mh * 2, mw * 2)
        else:
            h, w = first.shape[:2]
            out = torch.empty((3, h, w), dtype=torch.uint8, device=self.device)
            jobs.append(fused_job(first.tail, first.resized, out, planar=True, reversed_source=self.image_format == "RGB"))
            keep.append(first.resized)
            annos, image_hw = first.annos, (h, w)

        if not self.is_train:
            dataset_dict.pop("annotations", None)
            dataset_dict.pop("sem_seg_file_name", None)
        if annos:
            boxes = np.stack([np.asarray(a["bbox"], dtype=np.float64) for a in annos])
            inst = Instances(image_hw)
            inst.gt_boxes = Boxes(torch.as_tensor(boxes, dtype=torch.float32))
            inst.gt_classes = torch.tensor([int(a["category_id"]) for a in annos], dtype=torch.int64)
            dataset_dict["instances"] = inst[inst.gt_boxes.nonempty(threshold=1e-5)]  # filter_empty_instances
        dataset_dict["image"] = out
        dataset_dict.pop("annotations", None)
        return dataset_dict

    def _launch(self, jobs: list) -> None:
        if self._resizes:
            ops.aug_resample_batch([r[0] for r in self._resizes], [r[1] for r in self._resizes], [r[2] for r in self._resizes])
            self._resizes = []
        ops.aug_apply(jobs)

    def __call__(self, dataset_dict: dict) -> dict:
        jobs, keep = [], []
        out = self._plan(dataset_dict, jobs, keep)
        self._launch(jobs)
        return out

    def map_batch(self, dataset_dicts: List[dict]) -> List[dict]:
        """``[self(d) for d in dataset_dicts]`` with the resizes and the fused passes of the whole batch launched
        together (16 source images per launch)."""
        jobs, keep = [], []
        outs = [self._plan(d, jobs, keep) for d in dataset_dicts]
        self._launch(jobs)
        return outs

    # ------------------------------------------------------------------------------------------
    def _blend_mosaic(self, idx, item: _Planned, out: torch.Tensor, xc, yc, mosaic_width, mosaic_height, jobs, keep):
        """dataset_mapper.py:136-169: where image ``idx`` lands on the canvas and which part of it is visible."""
        h, w = item.shape[:2]
        canvas_w, canvas_h = 2 * mosaic_width, 2 * mosaic_height
        right, bottom = bool(idx & 1), bool(idx & 2)   # 0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right
        # The image touches the centre (xc, yc) with the corner facing it; the canvas cuts off what sticks out, and
        # the part kept is the one next to the centre.
        if right:
            x1a, x2a = xc, min(xc + w, canvas_w)
            x1b = 0
        else:
            x1a, x2a = max(xc - w, 0), xc
            x1b = w - (x2a - x1a)
        if bottom:
            y1a, y2a = yc, min(yc + h, canvas_h)
            y1b = 0
        else:
            y1a, y2a = max(yc - h, 0), yc
            y1b = h - (y2a - y1a)
        x2b = min(w, x2a - x1a) if right else w
        y2b = min(y2a - y1a, h) if bottom else h
        # numpy slice assignment semantics of `out[y1a:y2a, x1a:x2a] = image[y1b:y2b, x1b:x2b]` for in-range rectangles
        cw, ch = min(x2a - x1a, x2b - x1b), min(y2a - y1a, y2b - y1b)
        if cw > 0 and ch > 0:
            jobs.append(fused_job(item.tail, item.resized, out, dst_xy=(x1a, y1a), crop=(x1b, y1b, cw, ch), planar=True,
                                  reversed_source=self.image_format == "RGB"))
            keep.append(item.resized)
        padw, padh = x1a - x1b, y1a - y1b
        annos = item.annos or []
        for anno in annos:
            box = np.array(anno["bbox"], dtype=np.float64)
            box[0] += padw
            box[1] += padh
            box[2] += padw
            box[3] += padh
            anno["bbox"] = np.clip(box, [0, 0, 0, 0], [mosaic_width * 2, mosaic_height * 2, mosaic_width * 2,
                                                       mosaic_height * 2])
        return annos

    def _load_image_with_anns(self, dataset_dict: dict) -> _Planned:
        """dataset_mapper.py:171-202: read, draw the transforms, resize on the device, transform the annotations."""
        # The image travels as decoded (BGR); for INPUT.FORMAT "RGB" the reference reverses the channels on the host
        # (detection_utils.py:13-14) -- here the fused pass does it while writing, every step before being blind to
        # the channel order except the colour jitter, which is told.
        key = dataset_dict.get("file_name", dataset_dict.get("image_id")) if self.cache_bytes else None
        dev = self._cache.get(key) if key is not None else None
        if dev is not None:
            image = dev  # only its shape is read below
        elif "image" in dataset_dict and not isinstance(dataset_dict["image"], str):
            image = np.asarray(dataset_dict["image"])
        else:
            image = read_bgr(dataset_dict["file_name"])
        if "height" in dataset_dict or "width" in dataset_dict:  # utils.check_image_size
            assert (dataset_dict.get("height", image.shape[0]), dataset_dict.get("width", image.shape[1])) == \
                tuple(image.shape[:2]), "mismatched image shape"
        transforms: TransformList = self.augmentations.plan(image)
        if dev is None:
            dev = as_device_image(image, self.device)
            if key is not None and self._cached_bytes + dev.numel() <= self.cache_bytes:
                self._cache[key] = dev
                self._cached_bytes += dev.numel()
        tail = list(transforms.transforms)
        if tail and isinstance(tail[0], ResizeTransform):  # deferred: all resizes of a batch go out together
            t = tail.pop(0)
            out = torch.empty((t.new_h, t.new_w, 3), dtype=torch.uint8, device=self.device)
            self._resizes.append((dev, (t.new_h, t.new_w), out))
            dev = out
        elif tail and isinstance(tail[0], LetterBoxTransform):
            dev = tail.pop(0).apply_image(dev)
        image_shape = tuple(dev.shape[:2])
        annos = None
        if self.is_train and "annotations" in dataset_dict:
            annos = []
            for obj in dataset_dict.pop("annotations"):
                if obj.get("iscrowd", 0) != 0:
                    continue
                obj = dict(obj)
                bbox = np.asarray(obj["bbox"], dtype=np.float64)
                if obj.get("bbox_mode", XYXY_ABS) == XYWH_ABS:
                    bbox = np.array([bbox[0], bbox[1], bbox[0] + bbox[2], bbox[1] + bbox[3]])
                bbox = transforms.apply_box(np.array([bbox]))[0].clip(min=0)
                obj["bbox"] = np.minimum(bbox, list(image_shape + image_shape)[::-1])
                obj["bbox_mode"] = XYXY_ABS
                annos.append(obj)
        return _Planned(dev, tail, annos)

"""Inference entry point with the reference's flags and ``Predictor`` semantics
(reference tools/inference.py:23-45,92-131), running on MI355X.

    python tools/inference.py --config-file configs/yolov3.yaml --input img1.jpg img2.jpg \
        --score-confs 0.5 --iou-thresh 0.4 --output ./output --opts MODEL.WEIGHTS yolov3.weights

Differences from the reference, all at the edges of the hot path: images are read with PIL instead of
OpenCV, and instead of opening a GUI window (cv2.imshow) the detections of every image are printed and
written to <output>/<image name>.json.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import yolo_amd  # noqa: E402,F401
from yolo_amd.checkpoint import YOLOV3Checkpointer  # noqa: E402
from yolo_amd.config import get_cfg  # noqa: E402
from yolo_amd.modeling import build_model  # noqa: E402


class Predictor:
    """model.eval() + checkpoint load + shortest-edge resize + BGR->RGB flip + CHW float tensor, one image
    per call (reference tools/inference.py:23-45)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.model = build_model(cfg)
        self.model.eval()
        YOLOV3Checkpointer(self.model).load(cfg.MODEL.WEIGHTS)
        self.min_size, self.max_size = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST
        self.input_format = cfg.INPUT.FORMAT
        assert self.input_format in ["RGB", "BGR"], self.input_format

    def preprocess(self, original_image: np.ndarray) -> torch.Tensor:
        """BGR HWC uint8 (host) -> the model's input, CHW uint8 on the GPU: the decoded bytes go up as they are, the
        shortest-edge resize (Pillow's bilinear, bit for bit) and the BGR -> RGB / HWC -> CHW rearrangement are two
        kernel launches.  The reference does all of this on the host and uploads float32 (tools/inference.py:38-43)."""
        from yolo_amd import ops
        from yolo_amd.data.augmentation_impl import ResizeShortestEdge
        dev = torch.from_numpy(np.ascontiguousarray(original_image)).to(self.model.device)
        h, w = original_image.shape[:2]
        newh, neww = ResizeShortestEdge.get_output_shape(h, w, self.min_size, self.max_size)
        resized = ops.aug_resample(dev, (newh, neww))
        out = torch.empty((3, newh, neww), dtype=torch.uint8, device=dev.device)
        ops.aug_apply([ops.aug_job(resized, out, planar=True, reverse_channels=self.input_format == "RGB")])
        return out

    @torch.no_grad()
    def __call__(self, original_image: np.ndarray):
        """original_image: HWC uint8 in BGR order (OpenCV convention)."""
        height, width = original_image.shape[:2]
        return self.model([dict(height=height, width=width, image=self.preprocess(original_image))])[0]


def read_image_bgr(path: str) -> np.ndarray:
    from PIL import Image
    if path.endswith(".npy"):
        return np.load(path)
    return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]


# the reference's flags (tools/inference.py:102-110), same names, defaults and meaning
_FLAGS = (
    ("--config-file", dict(metavar="FILE", default="configs/yolov3.yaml")),
    ("--input", dict(nargs="+", required=True)),
    ("--output", dict(default="./output")),
    ("--score-confs", dict(type=float, default=0.5)),
    ("--iou-thresh", dict(type=float, default=0.4)),
    ("--opts", dict(default=None, nargs=argparse.REMAINDER)),
)


def get_parser():
    ap = argparse.ArgumentParser(description="YOLO V3 Inference")
    for flag, kw in _FLAGS:
        ap.add_argument(flag, **kw)
    return ap


def setup(args):
    """Config from the YAML, then --opts, then the two threshold flags, which always win over the YAML
    (reference tools/inference.py:92-99)."""
    cfg = get_cfg()
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(list(args.opts or []) + ["MODEL.YOLOV3.TEST_SCORE_THRESH", str(args.score_confs),
                                                 "MODEL.YOLOV3.TEST_NMS_THRESH", str(args.iou_thresh)])
    cfg.freeze()
    return cfg


def main():
    args = get_parser().parse_args()
    cfg = setup(args)
    predictor = Predictor(cfg)
    if len(args.input) == 1:
        args.input = glob.glob(os.path.expanduser(args.input[0]))
        assert args.input, "The input path(s) was not found"
    if args.output:
        os.makedirs(args.output, exist_ok=True)
    for path in args.input:
        predictions = predictor(read_image_bgr(path))
        inst = predictions["instances"].to("cpu")
        print(path, inst)
        if args.output:
            rec = dict(image=path, height=inst.image_size[0], width=inst.image_size[1],
                       boxes=inst.pred_boxes.tensor.tolist(), scores=inst.scores.tolist(),
                       classes=inst.pred_classes.tolist())
            with open(os.path.join(args.output, os.path.basename(path) + ".json"), "w") as f:
                json.dump(rec, f)


if __name__ == "__main__":
    main()

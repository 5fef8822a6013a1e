"""Training / evaluation entry point with the reference's flags (tools/train_net.py:117-147 uses
detectron2's default_argument_parser and launch()).

Status: the HIP path implements inference.  ``--eval-only`` loads the weights and runs image-sharded
inference over the given inputs, one process per GPU (start N processes with torch.distributed.run; backend
"nccl" = RCCL).  Training (label assignment, losses, convolution backward, DDP gradient all-reduce) is not
built yet and raises.
"""
from __future__ import annotations

import argparse
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import yolo_amd  # noqa: E402,F401
from yolo_amd.checkpoint import YOLOV3Checkpointer  # noqa: E402
from yolo_amd.config import get_cfg  # noqa: E402
from yolo_amd.modeling import build_model  # noqa: E402
from yolo_amd.utils import dist as D  # noqa: E402


def default_argument_parser():
    p = argparse.ArgumentParser(description="YOLOv3 on MI355X: train / evaluate")
    p.add_argument("--config-file", default="configs/yolov3.yaml", metavar="FILE")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--eval-only", action="store_true")
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    p.add_argument("--dist-url", default="tcp://127.0.0.1:29500")
    p.add_argument("--input", nargs="*", default=[], help="images for --eval-only (no dataset registry offline)")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p


def setup(args):
    cfg = get_cfg()
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    return cfg


def main(args):
    cfg = setup(args)
    world, rank, local = D.init_from_env("nccl")
    if world > 1:
        torch.cuda.set_device(local)
    model = build_model(cfg)
    ck = YOLOV3Checkpointer(model, cfg.OUTPUT_DIR)
    ck.resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume)
    if not args.eval_only:
        raise NotImplementedError("training is not built on the HIP path yet (inference only); use --eval-only")
    from inference import Predictor, read_image_bgr  # same directory
    model.eval()
    paths = sorted(p for pat in args.input for p in glob.glob(os.path.expanduser(pat)))
    b, e = D.shard_range(len(paths), rank, world)
    pred = Predictor.__new__(Predictor)
    pred.cfg, pred.model = cfg, model
    pred.min_size, pred.max_size, pred.input_format = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, cfg.INPUT.FORMAT
    counts = []
    for path in paths[b:e]:
        inst = pred(read_image_bgr(path))["instances"]
        counts.append(len(inst))
        print(f"[rank {rank}] {path}: {len(inst)} detections")
    all_counts = D.gather_counts(torch.tensor(counts, dtype=torch.int32, device=model.device))
    if rank == 0:
        print("detections per image:", [c.tolist() for c in all_counts])
    D.barrier()


if __name__ == "__main__":
    args = default_argument_parser().parse_args()
    print("Command Line Args:", args)
    main(args)

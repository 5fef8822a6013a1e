"""Training / evaluation entry point with the reference's flags (tools/train_net.py:117-147 uses
detectron2's default_argument_parser and launch()).

Training: one process per GPU (start N processes with torch.distributed.run; backend "nccl" = RCCL); every rank
runs the HIP forward/backward plan on its own images and the gradients are all-reduced in buckets from inside the
backward pass.  There is no dataset registry offline, so ``--synthetic`` (the default data source) feeds
COCO-shaped synthetic targets; a real loader plugs in as any iterable of ``list[dict(image, instances)]``.
``--eval-only`` loads the weights and runs image-sharded inference over ``--input`` images.
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
    p.add_argument("--synthetic", action="store_true", default=True, help="train on synthetic COCO-shaped targets")
    p.add_argument("--train-size", type=int, default=416, help="training image side (multiple of 32)")
    p.add_argument("--multi-scale", action="store_true", help="draw the side per step from 320..608 step 32")
    p.add_argument("--augment", action="store_true",
                   help="feed the model through YOLOV3DatasetMapper (GPU augmentation of a synthetic in-memory dataset)")
    p.add_argument("--max-iter", type=int, default=None, help="override SOLVER.MAX_ITER")
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
    if not args.eval_only:
        from yolo_amd.trainer import Trainer, synthetic_batches
        per_gpu = max(cfg.SOLVER.IMS_PER_BATCH // world, 1)
        sizes = list(range(320, 609, 32)) if args.multi_scale else None
        if args.augment:
            from yolo_amd.trainer import augmented_batches
            data = augmented_batches(cfg, per_gpu, seed=7 + rank, device=model.device)
        else:
            data = synthetic_batches(cfg, per_gpu, args.train_size, seed=7 + rank, device=model.device, sizes=sizes)
        trainer = Trainer(cfg, model, data, log=lambda m: print(f"[rank {rank}] {m}") if rank == 0 else None)
        trainer.resume_or_load(resume=args.resume)
        trainer.train(max_iter=args.max_iter)
        return
    ck = YOLOV3Checkpointer(model, cfg.OUTPUT_DIR)
    ck.resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume)
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

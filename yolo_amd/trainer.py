"""Training loop with the reference's shape: ``losses = model(batch); sum(losses).backward(); optimizer.step()``
(detectron2 SimpleTrainer.run_step under DefaultTrainer, which tools/train_net.py:80-134 of the reference
subclasses), periodic checkpoints through YOLOV3Checkpointer, one process per GPU.

Data parallelism: every rank runs the same plan on its own images; the only exchange is the gradient all-reduce,
bucketed over the plan's flat gradient arena and launched from inside the backward pass (yolo_amd/train_engine.py,
yolo_amd/utils/dist.py) -- what DistributedDataParallel's reducer does for the reference.  BatchNorm statistics
stay per-rank (the reference uses plain "BN", defaults.py:19).
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Iterable, Iterator, List, Optional

import torch

from .checkpoint import YOLOV3Checkpointer
from .solver import build_lr_scheduler, build_optimizer
from .structures import Boxes, Instances
from .utils import dist as D


def synthetic_batches(cfg, per_gpu: int, size: int, *, seed: int = 7, max_boxes: int = 20, device="cuda",
                      sizes: Optional[List[int]] = None) -> Iterator[List[Dict]]:
    """Endless COCO-shaped synthetic batches: uint8 images already on the GPU, K ~ U{1..max_boxes} boxes per image
    inside the frame, classes uniform.  ``sizes``: draw the side per step from this list (multi-scale training;
    the draw depends only on the seed and the step, so all ranks agree)."""
    C = cfg.MODEL.YOLOV3.NUM_CLASSES
    g = torch.Generator().manual_seed(seed)
    gs = torch.Generator().manual_seed(seed + 1)
    while True:
        s = size if not sizes else sizes[int(torch.randint(0, len(sizes), (1,), generator=gs))]
        batch = []
        for _ in range(per_gpu):
            img = torch.randint(0, 256, (3, s, s), generator=g, dtype=torch.uint8)
            k = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
            wh = 8 + torch.rand(k, 2, generator=g) * (s * 0.6 - 8)
            xy = torch.rand(k, 2, generator=g) * (s - wh)
            inst = Instances((s, s), gt_boxes=Boxes(torch.cat([xy, xy + wh], 1)),
                             gt_classes=torch.randint(0, C, (k,), generator=g))
            batch.append(dict(image=img.to(device), instances=inst.to(device), height=s, width=s))
        yield batch


def synthetic_dataset(cfg, n_images: int = 64, *, seed: int = 7, max_boxes: int = 20, min_side: int = 320,
                      max_side: int = 640) -> List[Dict]:
    """Dataset dicts in detectron2's format with in-memory images (HWC uint8, BGR as decoded): random sizes, K ~
    U{1..max_boxes} XYXY boxes inside the frame, classes uniform -- a stand-in for a registered dataset."""
    import numpy as np
    rng = np.random.default_rng(seed)
    C = cfg.MODEL.YOLOV3.NUM_CLASSES
    out = []
    for i in range(n_images):
        h, w = int(rng.integers(min_side, max_side + 1)), int(rng.integers(min_side, max_side + 1))
        k = int(rng.integers(1, max_boxes + 1))
        wh = 8 + rng.random((k, 2)) * (np.array([w, h]) * 0.6 - 8)
        xy = rng.random((k, 2)) * (np.array([w, h]) - wh)
        annos = [dict(bbox=[float(xy[j, 0]), float(xy[j, 1]), float(xy[j, 0] + wh[j, 0]), float(xy[j, 1] + wh[j, 1])],
                      bbox_mode=0, category_id=int(rng.integers(0, C))) for j in range(k)]
        out.append(dict(image=rng.integers(0, 256, (h, w, 3), dtype=np.uint8), height=h, width=w, image_id=i,
                        annotations=annos))
    return out


def augmented_batches(cfg, per_gpu: int, *, seed: int = 7, n_images: int = 64, device="cuda",
                      cache_bytes: int = 1 << 32) -> Iterator[List[Dict]]:
    """Endless training batches through the reference's input pipeline (tools/train_net.py:109-111:
    ``build_detection_train_loader(cfg, mapper=YOLOV3DatasetMapper(cfg, True))``) over a synthetic in-memory dataset:
    the augmentation the config asks for (resize / letterbox, random affine, colour jitter, flip, mosaic) runs on
    the GPU, sources stay resident in HBM after their first use."""
    import random

    import numpy as np

    from .data import YOLOV3DatasetMapper
    np.random.seed(seed)
    random.seed(seed)
    data = synthetic_dataset(cfg, n_images, seed=seed)
    mapper = YOLOV3DatasetMapper(cfg, True, device=device, cache_bytes=cache_bytes)
    order = np.random.default_rng(seed)
    while True:
        picks = order.integers(0, len(data), per_gpu)
        batch = mapper.map_batch([data[int(i)] for i in picks])
        for d in batch:
            if "instances" in d:
                d["instances"] = d["instances"].to(device)
            else:
                d["instances"] = Instances(tuple(d["image"].shape[-2:]), gt_boxes=Boxes(torch.zeros((0, 4), device=device)),
                                           gt_classes=torch.zeros((0,), dtype=torch.int64, device=device))
        yield batch


class Trainer:
    def __init__(self, cfg, model, data: Iterable[List[Dict]], *, log: Callable[[str], None] = print):
        self.cfg, self.model, self.data, self.log = cfg, model, iter(data), log
        self.optimizer = build_optimizer(cfg, model)
        self.scheduler = build_lr_scheduler(cfg, self.optimizer)
        self.world = int(torch.distributed.get_world_size()) if torch.distributed.is_initialized() else 1
        self.rank = int(torch.distributed.get_rank()) if torch.distributed.is_initialized() else 0
        self.checkpointer = YOLOV3Checkpointer(model, cfg.OUTPUT_DIR, save_to_disk=self.rank == 0,
                                               optimizer=self.optimizer, scheduler=self.scheduler)
        self.iter = 0
        if self.world > 1:   # identical starting point on every rank
            D.broadcast_tensors([t.data for t in list(model.parameters()) + list(model.buffers())], src=0)

    def resume_or_load(self, resume: bool = True) -> None:
        ck = self.checkpointer.resume_or_load(self.cfg.MODEL.WEIGHTS, resume=resume)
        if resume and self.checkpointer.has_checkpoint():
            self.iter = int(ck.get("iteration", -1)) + 1

    def run_step(self) -> Dict[str, float]:
        self.model.train()
        batch = next(self.data)
        losses = self.model(batch)
        total = sum(losses.values())
        self.optimizer.zero_grad(set_to_none=True)
        total.backward()           # HIP backward plan + bucketed gradient all-reduce
        self.optimizer.step()
        self.scheduler.step()
        return {k: v.detach() for k, v in losses.items()}

    def train(self, max_iter: Optional[int] = None, log_period: int = 20) -> None:
        max_iter = self.cfg.SOLVER.MAX_ITER if max_iter is None else max_iter
        t0, n0 = time.perf_counter(), self.iter
        while self.iter < max_iter:
            losses = self.run_step()
            self.iter += 1
            if self.iter % log_period == 0 or self.iter == max_iter:
                vals = {k: float(v) for k, v in losses.items()}   # the only host sync of the loop
                dt = time.perf_counter() - t0
                self.log(f"iter {self.iter}  " + "  ".join(f"{k} {v:.4f}" for k, v in vals.items()) +
                         f"  lr {self.scheduler.get_last_lr()[0]:.6f}  {(self.iter - n0) / dt:.2f} it/s")
            if self.cfg.SOLVER.CHECKPOINT_PERIOD and self.iter % self.cfg.SOLVER.CHECKPOINT_PERIOD == 0:
                self.checkpointer.save(f"model_{self.iter - 1:07d}", iteration=self.iter - 1)
        self.checkpointer.save("model_final", iteration=self.iter - 1)
        D.barrier()

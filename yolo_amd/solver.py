"""Optimizer and learning-rate schedule of the training entry point, with detectron2's defaults for the keys the
reference's configs set (tools/train_net.py builds both through detectron2's DefaultTrainer: SGD with momentum,
no weight decay on normalisation parameters, WarmupMultiStepLR).  Plain torch: one fused update per step over
parameters whose gradients the HIP backward plan produced."""
from __future__ import annotations

from bisect import bisect_right
from typing import List

import torch


def build_optimizer(cfg, model: torch.nn.Module) -> torch.optim.Optimizer:
    norm_params, other = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (norm_params if ".norm." in name else other).append(p)
    groups = [dict(params=other, weight_decay=cfg.SOLVER.WEIGHT_DECAY),
              dict(params=norm_params, weight_decay=cfg.SOLVER.WEIGHT_DECAY_NORM)]
    return torch.optim.SGD(groups, lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM, nesterov=cfg.SOLVER.NESTEROV)


class WarmupMultiStepLR(torch.optim.lr_scheduler.LRScheduler):
    """lr = base * gamma^(number of milestones passed), ramped linearly from warmup_factor over warmup_iters."""

    def __init__(self, optimizer, milestones: List[int], gamma: float = 0.1, warmup_factor: float = 0.001,
                 warmup_iters: int = 1000, warmup_method: str = "linear", last_epoch: int = -1):
        assert list(milestones) == sorted(milestones), milestones
        self.milestones, self.gamma = list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def _warmup(self) -> float:
        it = self.last_epoch
        if it >= self.warmup_iters:
            return 1.0
        if self.warmup_method == "constant":
            return self.warmup_factor
        alpha = it / self.warmup_iters
        return self.warmup_factor * (1 - alpha) + alpha

    def get_lr(self) -> List[float]:
        w = self._warmup()
        return [b * w * self.gamma ** bisect_right(self.milestones, self.last_epoch) for b in self.base_lrs]


def build_lr_scheduler(cfg, optimizer):
    if cfg.SOLVER.LR_SCHEDULER_NAME != "WarmupMultiStepLR":
        raise ValueError(f"unsupported LR scheduler {cfg.SOLVER.LR_SCHEDULER_NAME}")
    steps = [int(s) for s in cfg.SOLVER.STEPS]
    return WarmupMultiStepLR(optimizer, steps, cfg.SOLVER.GAMMA, cfg.SOLVER.WARMUP_FACTOR, cfg.SOLVER.WARMUP_ITERS,
                             cfg.SOLVER.WARMUP_METHOD)

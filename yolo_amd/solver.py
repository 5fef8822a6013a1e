"""Optimizer and learning-rate schedule of the training entry point, with detectron2's defaults for the keys the
reference's configs set (tools/train_net.py builds both through detectron2's DefaultTrainer: SGD with momentum,
no weight decay on normalisation parameters, WarmupMultiStepLR).  The update is one kernel over the training
engine's flat arenas when every parameter lives there, torch's own step otherwise."""
from __future__ import annotations

from bisect import bisect_right
from typing import List

import torch


class ArenaSGD(torch.optim.SGD):
    """torch.optim.SGD whose step is ONE kernel launch when every parameter it updates lives in the training
    engine's flat arenas (parameters, gradients; the momentum buffers then live in a third arena and the per-parameter
    ``momentum_buffer`` entries of the state are views of it, so ``state_dict()`` / ``load_state_dict()`` are torch's).
    Anything else -- parameters outside the arenas, Nesterov, dampening, a closure -- takes torch's own step."""

    def __init__(self, params, model=None, **kw):
        super().__init__(params, **kw)
        self._model = model
        self._fused = None  # (engine, mom arena, block table, n blocks, [(param, mom view)], the two groups)
        self._fused_off = None  # (engine, reason): the one-launch step is not possible with this engine; not retried

    def _engine(self):
        bb = getattr(self._model, "backbone", None)
        return getattr(bb, "_train_engine", None)

    def _prepare(self, eng):
        """The one-launch step's state, or a string saying why it cannot be used (empty hyper-parameter groups --
        e.g. the norm group of a model built with MODEL.DARKNET.NORM "" -- do not count)."""
        groups = [g for g in self.param_groups if g["params"]]
        if not 1 <= len(groups) <= 2:
            return f"{len(groups)} non-empty parameter groups (the kernel takes one or two)"
        if any(g["nesterov"] or g["dampening"] != 0 or g["maximize"] for g in groups):
            return "nesterov / dampening / maximize"
        if groups[0]["momentum"] != groups[-1]["momentum"] or groups[0]["momentum"] == 0:
            return "momentum is zero or differs between the groups"
        arena, garena = eng.param_arena, eng.grad_arena
        base, gbase, n = arena.data_ptr(), garena.data_ptr(), arena.numel()
        segments, views, covered = [], [], 0
        mom = torch.zeros_like(arena)
        for gi, g in enumerate(groups):
            for prm in g["params"]:
                off = (prm.data_ptr() - base) // 4
                if prm.dtype != torch.float32 or off < 0 or off + prm.numel() > n or prm.grad is None or \
                        prm.grad.data_ptr() != gbase + 4 * off:
                    return "a parameter (or its .grad) lives outside the training engine's arenas"
                # a parameter is one contiguous run of the arena, possibly permuted (conv weights are channels_last)
                mview = torch.as_strided(mom, prm.shape, prm.stride(), off)
                st = self.state[prm]
                if st.get("momentum_buffer") is not None:
                    mview.copy_(st["momentum_buffer"])
                st["momentum_buffer"] = mview
                segments.append((off, prm.numel(), gi))
                views.append((prm, mview))
                covered += prm.numel()
        if covered != n:
            return "the optimizer's parameters do not cover the arena"
        segments.sort()
        from . import ops
        table, nb = ops.sgd_block_table(segments, arena.device)
        return eng, mom, table, nb, views, (groups[0], groups[-1])

    @torch.no_grad()
    def step(self, closure=None):
        import os
        eng = self._engine()
        usable = closure is None and eng is not None and eng.param_arena is not None and \
            os.environ.get("YOLO_FUSED_SGD", "1") != "0"
        if usable and self._fused_off is not None and self._fused_off[0] is eng:
            usable = False  # already found impossible with this engine: torch's step, without preparing again
        if usable and (self._fused is None or self._fused[0] is not eng or
                       any(self.state[p].get("momentum_buffer") is not m for p, m in self._fused[4])):
            res = self._prepare(eng)
            if isinstance(res, str):
                import logging
                logging.getLogger(__name__).warning("ArenaSGD: one-launch step disabled (%s); using torch's step", res)
                self._fused, self._fused_off = None, (eng, res)
            else:
                self._fused = res
        if not usable or self._fused is None:
            self._fused = None
            return super().step(closure)
        from . import ops
        eng, mom, table, nb, views, (g0, g1) = self._fused
        if any(p.grad is None for p, _ in views):
            self._fused = None
            return super().step(closure)
        ops.sgd_step(eng.param_arena, eng.grad_arena, mom, table, nb, g0["lr"], g0["weight_decay"], g1["lr"],
                     g1["weight_decay"], g0["momentum"])
        # in-place update outside autograd's sight: move the version counters ourselves (the engines key their
        # packed / folded weights on them)
        ps = [p for p, _ in views]
        torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])
        return None


def build_optimizer(cfg, model: torch.nn.Module) -> torch.optim.Optimizer:
    norm_params, other = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (norm_params if ".norm." in name else other).append(p)
    groups = [dict(params=other, weight_decay=cfg.SOLVER.WEIGHT_DECAY),
              dict(params=norm_params, weight_decay=cfg.SOLVER.WEIGHT_DECAY_NORM)]
    return ArenaSGD(groups, model=model, lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM, nesterov=cfg.SOLVER.NESTEROV)


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

"""Least-recently-used cache of per-batch-shape launch plans with a byte budget.

A plan owns every activation buffer of one (batch, height, width): ~3.4 GB for batch-64 inference at 416 x 416, ~8 GB
for training, ~17 GB for training at 608 x 608.  Multi-scale training (the reference's ``MIN_SIZE_TRAIN`` list,
configs/yolov3.yaml:36) and single-image inference on arbitrary image sizes would otherwise grow the set without
bound; 288 GB of HBM hold all ten training scales at batch 64 (~103 GB), so eviction is the rare case, not the rule.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Hashable, Optional

import torch


class PlanCache:
    def __init__(self, device, budget_bytes: Optional[int] = None, max_plans: int = 32):
        self.device = device
        if budget_bytes is None:
            budget_bytes = int(0.6 * torch.cuda.get_device_properties(device).total_memory)
        self.budget_bytes = budget_bytes
        self.max_plans = max_plans
        self._plans: "OrderedDict[Hashable, tuple]" = OrderedDict()  # key -> (plan, bytes)
        self.evictions = 0

    def __len__(self) -> int:
        return len(self._plans)

    def __contains__(self, key) -> bool:
        return key in self._plans

    def keys(self):
        return list(self._plans.keys())

    def values(self):
        """The cached plans, least recently used first."""
        return [plan for plan, _ in self._plans.values()]

    @property
    def nbytes(self) -> int:
        return sum(b for _, b in self._plans.values())

    def get(self, key: Hashable, build: Callable[[], object]):
        hit = self._plans.get(key)
        if hit is not None:
            self._plans.move_to_end(key)
            return hit[0]
        before = torch.cuda.memory_allocated(self.device)
        plan = build()
        size = max(torch.cuda.memory_allocated(self.device) - before, 0)
        self._plans[key] = (plan, size)
        self._evict(keep=key)
        return plan

    def _evict(self, keep: Hashable) -> None:
        while len(self._plans) > 1 and (self.nbytes > self.budget_bytes or len(self._plans) > self.max_plans):
            victim = next(k for k in self._plans if k != keep)
            # the victim's buffers may still be read by kernels queued on side streams: wait before they are freed
            torch.cuda.synchronize(self.device)
            del self._plans[victim]
            self.evictions += 1

    def clear(self) -> None:
        if self._plans:
            torch.cuda.synchronize(self.device)
        self._plans.clear()

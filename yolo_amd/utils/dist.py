"""One-process-per-GPU helpers for image-sharded inference.

Inference shards by image with no data-path collective (the reference is single-process at inference,
tools/inference.py:113-131, and detectron2's evaluation shards the dataset by rank); the only exchanges
are the barrier and the MAX over ranks that bracket a timed region, and an optional gather of the per-image
detection counts.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl") -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    torch.distributed.run).  Returns (world, rank, local_rank); a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return world, rank, local


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [begin, end) of n_items for this rank (sizes differ by at most 1)."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def max_over_ranks(value: float, device="cpu") -> float:
    """MAX all-reduce of a scalar (the timed region of the slowest rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local_counts: torch.Tensor) -> List[torch.Tensor]:
    """All-gather of each rank's per-image detection counts (ragged: sizes are exchanged first)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local_counts]
    world = dist.get_world_size()
    n = torch.tensor([local_counts.numel()], dtype=torch.int64, device=local_counts.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(m, dtype=local_counts.dtype, device=local_counts.device)
    pad[:local_counts.numel()] = local_counts
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [o[:int(s.item())] for o, s in zip(out, sizes)]


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class GradBucketer:
    """Bucketed gradient all-reduce over one flat fp32 arena, overlapped with the backward pass.

    The training plan lays every parameter gradient out in one arena in the order backward produces them
    (reference: detectron2's DefaultTrainer wraps the model in DistributedDataParallel, tools/train_net.py:127,
    whose reducer buckets gradients in reverse execution order the same way).  ``ready(end)`` says that
    arena[:end] is final; whenever at least ``bucket_bytes`` are pending, that range goes out as one asynchronous
    all-reduce on the process group's own stream (RCCL over xGMI on the GPU; gloo in the CPU tests), so the
    collective runs under the remaining backward kernels.  ``finish()`` flushes the tail, waits and averages.
    xGMI rings are bound by the per-link rate, so buckets are large (default 64 MiB: 4 buckets for Darknet-53).
    """

    def __init__(self, arena: torch.Tensor, bucket_bytes: int = 64 << 20, group=None):
        assert arena.dim() == 1 and arena.dtype == torch.float32
        self.arena, self.bucket = arena, max(int(bucket_bytes) // 4, 1)
        self.group = group
        self.sent = 0
        self.works = []
        self.launched = []  # (begin, end) of every bucket, for tests / logs

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def start(self) -> None:
        self.sent, self.works, self.launched = 0, [], []

    def _launch(self, end: int) -> None:
        if end <= self.sent:
            return
        if self.world > 1:
            self.works.append(dist.all_reduce(self.arena[self.sent:end], op=dist.ReduceOp.SUM, group=self.group,
                                              async_op=True))
        self.launched.append((self.sent, end))
        self.sent = end

    def ready(self, end: int) -> None:
        if end - self.sent >= self.bucket:
            self._launch(end)

    def finish(self) -> None:
        self._launch(self.arena.numel())
        for w in self.works:
            w.wait()
        if self.world > 1:
            self.arena.mul_(1.0 / self.world)
        self.works = []

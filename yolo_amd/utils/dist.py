"""One-process-per-GPU helpers for image-sharded inference.

Inference shards by image with no data-path collective (the reference is single-process at inference,
tools/inference.py:113-131, and detectron2's evaluation shards the dataset by rank); the only exchanges
are the barrier and the MAX over ranks that bracket a timed region, and an optional gather of the per-image
detection counts.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl") -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    torch.distributed.run).  Returns (world, rank, local_rank); a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # YOLO_DIST_BACKEND=gloo rehearses the multi-rank paths on a box with fewer GPUs than ranks (ranks then share
    # cards and collectives go through the host); production runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("YOLO_DIST_BACKEND", backend)
    if backend != "nccl" and torch.cuda.is_available():
        local %= max(torch.cuda.device_count(), 1)
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
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
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


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """gloo moves GPU tensors through the host itself, slowly and (on this stack) not reliably; when a GPU tensor
    meets the gloo backend -- the rehearsal set-up, never production -- the collective is done on a host copy."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def broadcast_tensors(tensors: List[torch.Tensor], src: int = 0, group=None) -> None:
    """Overwrite ``tensors`` on every rank with rank ``src``'s values: one collective per dtype over a flat copy
    instead of one per tensor."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for ts in by_dtype.values():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            if _host_staged(flat, group):
                host = flat.cpu()
                dist.broadcast(host, src=src, group=group)
                flat = host.to(flat.device)
            else:
                dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view(t.shape))
                off += t.numel()


class GradBucketer:
    """Bucketed gradient exchange (sum, then mean) over one flat fp32 arena, overlapped with the backward pass.

    The training plan lays every parameter gradient out in one arena in the order backward produces them
    (reference: detectron2's DefaultTrainer wraps the model in DistributedDataParallel, tools/train_net.py:127,
    whose reducer buckets gradients in reverse execution order the same way).  ``ready(end)`` says that
    arena[:end] is final; whenever at least ``bucket_bytes`` are pending, that range goes out asynchronously, so the
    exchange runs under the remaining backward kernels.  ``finish()`` flushes the tail, waits and averages.
    Buckets are large (default 64 MiB: 4 buckets for Darknet-53): xGMI transfers are bound by the per-link rate, not
    by a switch.

    Two ways to exchange a bucket (``mode``, or YOLO_GRAD_EXCHANGE = allreduce | direct | auto):

    * ``allreduce``: one asynchronous all-reduce on the process group's own stream (RCCL picks the schedule).
    * ``direct``: written for a fully connected xGMI node (every GPU has its own link to each of the 7 others): the
      bucket is cut into ``world`` equal shards; every rank SENDS shard j to rank j and RECEIVES its own shard from
      every peer at the same time -- world-1 point-to-point transfers per rank, one per link, grouped into one batch
      -- sums the ``world`` copies in rank order, and sends the reduced shard back to every peer the same way.  Each
      link carries 2/world of the bucket, against 2(world-1)/world of it through one link pair for a single ring.
      Every element is summed by exactly one rank, so all ranks end with bit-identical gradients.  The exchange runs
      on a stream of its own; a remainder of fewer than ``world`` elements at the arena's end goes out as a small
      all-reduce.
    * ``auto`` (default): all-reduce.  The direct exchange has only run over gloo host copies so far; it stays an
      opt-in (YOLO_GRAD_EXCHANGE=direct) until one multi-GPU run on RCCL has shown bit-identical gradients and no
      stall, then ``auto`` may pick it from 3 ranks on.

    A GPU arena on the gloo backend (several ranks sharing one card: a rehearsal of the control flow, never
    production) goes through host copies.  Those copies block the host, so they are NOT made from inside the backward
    loop -- there the caller is inside its weight-gradient stream's context, and a host thread parked in a device-to-host
    copy behind that stream cannot queue the rest of the backward pass while its peers wait for it in the host-side
    collective.  ``ready()`` only notes the range; ``finish()`` exchanges all of them behind the backward pass.
    """

    def __init__(self, arena: torch.Tensor, bucket_bytes: int = 64 << 20, group=None, mode: Optional[str] = None):
        assert arena.dim() == 1 and arena.dtype == torch.float32
        self.arena, self.bucket = arena, max(int(bucket_bytes) // 4, 1)
        self.group = group
        self.mode = (mode or os.environ.get("YOLO_GRAD_EXCHANGE", "auto")).lower()
        if self.mode not in ("allreduce", "direct", "auto"):
            raise ValueError(f"gradient exchange mode {self.mode!r}: allreduce, direct or auto")
        self.sent = 0
        self.works = []
        self.launched = []  # (begin, end) of every bucket, for tests / logs
        self._recv = {}     # bucket begin -> receive buffer [world, shard] of the direct exchange (kept across steps)
        self._comm = None   # the direct exchange's stream

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def direct(self) -> bool:
        return self.world >= 2 and self.mode == "direct"

    @property
    def exchange(self) -> str:
        """The exchange this bucketer runs: "allreduce" or "direct"."""
        return "direct" if self.direct else "allreduce"

    def start(self) -> None:
        self.sent, self.works, self.launched = 0, [], []

    def _all_reduce(self, piece: torch.Tensor) -> None:
        if _host_staged(piece, self.group):
            host = piece.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            piece.copy_(host)
        else:
            self.works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _peer(self, j: int) -> int:
        return j if self.group is None else dist.get_global_rank(self.group, j)

    def _exchange_direct(self, begin: int, piece: torch.Tensor) -> None:
        """Direct reduce-scatter + all-gather of ``piece`` (numel divisible by world) by grouped point-to-point sends."""
        W, r = self.world, dist.get_rank(self.group)
        staged = _host_staged(piece, self.group)
        buf = piece.cpu() if staged else piece
        shards = buf.view(W, -1)
        key = (begin, buf.numel(), buf.device)
        recv = self._recv.get(key)
        if recv is None:
            recv = self._recv[key] = torch.empty_like(shards)
        peers = [(r + d) % W for d in range(1, W)]

        def run():
            first = []
            for j in peers:
                first.append(dist.P2POp(dist.isend, shards[j], self._peer(j), self.group))
                first.append(dist.P2POp(dist.irecv, recv[j], self._peer(j), self.group))
            for w in dist.batch_isend_irecv(first):
                w.wait()   # on a GPU: the current stream waits, not the host
            recv[r].copy_(shards[r])
            torch.sum(recv, dim=0, out=shards[r])   # rank order, whatever the arrival order
            second = []
            for j in peers:
                second.append(dist.P2POp(dist.isend, shards[r], self._peer(j), self.group))
                second.append(dist.P2POp(dist.irecv, shards[j], self._peer(j), self.group))
            return dist.batch_isend_irecv(second)

        if buf.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(buf.device)
            self._comm.wait_stream(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(self._comm):
                self.works.extend(run())
        else:
            for w in run():
                w.wait()
            if staged:
                piece.copy_(buf)

    def _launch(self, end: int, flush: bool = False) -> None:
        if end <= self.sent:
            return
        if self.world > 1:
            if self.direct:
                n = (end - self.sent) // self.world * self.world   # whole shards; the rest waits for the next bucket
                if n:
                    self._exchange_direct(self.sent, self.arena[self.sent:self.sent + n])
                    self.launched.append((self.sent, self.sent + n))
                    self.sent += n
                if flush and self.sent < end:                        # fewer than `world` elements at the very end
                    self._all_reduce(self.arena[self.sent:end])
                    self.launched.append((self.sent, end))
                    self.sent = end
                return
            self._all_reduce(self.arena[self.sent:end])
        self.launched.append((self.sent, end))
        self.sent = end

    def ready(self, end: int) -> None:
        if self.world > 1 and _host_staged(self.arena, self.group):
            return   # host-staged rehearsal: every range goes out from finish(), see the class comment
        if end - self.sent >= self.bucket:
            self._launch(end)

    def finish(self) -> None:
        if self.world > 1 and _host_staged(self.arena, self.group):
            # the same bucket boundaries as the overlapped form, one after the other, on the caller's stream
            n = self.arena.numel()
            while n - self.sent > self.bucket:
                self._launch(self.sent + self.bucket)
        self._launch(self.arena.numel(), flush=True)
        for w in self.works:
            w.wait()
        if self._comm is not None and self.arena.is_cuda:
            torch.cuda.current_stream(self.arena.device).wait_stream(self._comm)
        if self.world > 1:
            self.arena.mul_(1.0 / self.world)
        self.works = []

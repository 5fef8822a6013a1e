"""A stand-in workload for bench.py's launcher tests: no GPU, a step is a short sleep whose length depends on the
rank, so that the MAX over ranks and the per-rank values can be told apart."""
import time


class StubWorkload:
    metric = "stub steps/sec"
    unit = "steps/sec"
    dtype = "none"
    data = "stub"
    units_per_step = 1

    def __init__(self, rank, world, local):
        self.rank, self.world = rank, world
        self.steps_run = 0

    def prepare(self):
        pass

    def run(self, k):
        time.sleep(0.002 * k * (1 + self.rank))
        self.steps_run += k

    def sync(self):
        pass

    def reduce_device(self):
        return "cpu"

    def device_label(self):
        return f"stub device {self.rank}"

    def config(self, world, steps_total, elapsed):
        return {"workload": "stub", "steps_run_on_rank0": self.steps_run}

    def roofline(self):
        return {"bound": "hbm", "achieved": 0.0, "peak": 1.0, "unit": "GB/s", "frac": 0.0, "traffic": None}

    def train(self):
        """Collective like the real workload's: every rank takes part (an all-reduce over the job's process group)."""
        import torch
        import torch.distributed as dist
        self.train_calls = getattr(self, "train_calls", 0) + 1
        n = 1
        if dist.is_available() and dist.is_initialized():
            t = torch.ones(1)
            dist.all_reduce(t)
            n = int(t.item())
        return {"metric": "stub train", "value": 1.0, "n_gpus": n, "params_in_sync": True, "exchange": "allreduce"}

    def cpu_baseline(self):
        return {"value": 0.0, "unit": "steps/sec", "cores": 1, "kind": "port", "sample": "stub"}


def make(rank, world, local):
    return StubWorkload(rank, world, local)


class HangingTrainStub(StubWorkload):
    """Rank 1 never reaches the training block's collective: rank 0 waits in it for ever."""

    def train(self):
        if self.rank == 1:
            time.sleep(3600)
        return super().train()


def make_hanging(rank, world, local):
    return HangingTrainStub(rank, world, local)


class SameDeviceStub(StubWorkload):
    """Every rank reports the same device, as ranks that mis-bound their GPU would."""

    def device_label(self):
        return "stub device 0"


def make_same_device(rank, world, local):
    return SameDeviceStub(rank, world, local)

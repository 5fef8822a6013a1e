"""Training throughput (images/sec: forward + backward + gradient all-reduce + SGD step) of the HIP training path
on synthetic COCO-shaped targets, one process per GPU.

    python tools/train_bench.py --batch 64 --size 416 --steps 10 [--config yolov3.yaml]
    python -m torch.distributed.run --nproc-per-node N tools/train_bench.py ...
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workload  # noqa: E402
from yolo_amd.trainer import Trainer, synthetic_batches  # noqa: E402
from yolo_amd.utils import dist as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="yolov3.yaml")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    if os.environ.get("YOLO_DEBUG_HANG"):   # dump every thread's stack and exit if the run takes this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["YOLO_DEBUG_HANG"]), exit=True)
    world, rank, local = D.init_from_env("nccl")
    torch.cuda.set_device(local if world > 1 else 0)
    model, cfg = workload.build_model(config=args.config, dtype="bf16", device=f"cuda:{torch.cuda.current_device()}",
                                      opts=["OUTPUT_DIR", "", "SOLVER.CHECKPOINT_PERIOD", 0, "SOLVER.BASE_LR", 1e-4])
    # one resident batch, reused: the loop measures the step, not the synthetic generator
    batch = next(synthetic_batches(cfg, args.batch, args.size, seed=7 + rank, device=model.device))
    tr = Trainer(cfg, model, iter(lambda: batch, None), log=lambda m: None)
    for _ in range(args.warmup):
        tr.run_step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = tr.run_step()
    host = time.perf_counter() - t0   # the host's part: everything queued, nothing awaited
    D.barrier()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, model.device)
    # data-parallel invariant: after identical initial states and averaged gradients every rank holds the same weights
    chk = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
    in_sync = D.max_over_ranks(chk) == -D.max_over_ranks(-chk)
    if rank == 0:
        eng = model.backbone.train_engine()
        fwd = 65.428e9 if cfg.MODEL.YOLOV3.NUM_CLASSES == 20 else 65.864e9
        train_flop = (187.357e9 if cfg.MODEL.YOLOV3.NUM_CLASSES == 20 else 188.665e9) * args.size ** 2 / 416 ** 2
        print(json.dumps({
            "metric": "train images/sec (fwd + bwd + all-reduce + SGD)", "value": round(world * args.batch * args.steps / dt, 1),
            "n_gpus": world, "batch_per_gpu": args.batch, "size": args.size, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "host_ms_per_step": round(host / args.steps * 1e3, 2),
            "conv_tflops": round(train_flop * world * args.batch * args.steps / dt / 1e12, 1),
            "losses": {k: round(float(v), 4) for k, v in losses.items()}, "dtype": "bf16", "freeze_at": eng.freeze_at, "params_in_sync": bool(in_sync),
            "config": args.config}))
    D.barrier()


if __name__ == "__main__":
    main()

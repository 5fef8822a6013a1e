"""The BatchNorm passes of the training step on the network's own tensor shapes (batch 64, 416x416, bf16), one kernel
sequence at a time: statistics (read z), apply forward (read z [+ residual], write y), backward (reduce: read dy, z;
apply: read dy, z, write dz).  Prints time and algorithmic bytes per second per shape.
usage: python tools/bn_bench.py [--batch 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import ops  # noqa: E402

SHAPES = [(104, 64, 2), (104, 128, 3), (52, 128, 13), (52, 256, 12), (26, 256, 13), (26, 512, 12), (13, 512, 8),
          (13, 1024, 7)]  # (side, channels, layers of that shape in the network, roughly)


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    dev, dt, N = "cuda", torch.bfloat16, args.batch
    tot = {"stats": 0.0, "fwd": 0.0, "fwd+res": 0.0, "bwd": 0.0}
    print("side  C     stats us  TB/s |  fwd us  TB/s | fwd+res us  TB/s |  bwd us  TB/s(5 passes)")
    for side, C, count in SHAPES:
        z = ops.alloc_view(N, side, side, C, 0, dt, dev)
        z.buf.normal_()
        y = ops.alloc_view(N, side, side, C, 1, dt, dev)
        r = ops.alloc_view(N, side, side, C, 1, dt, dev)
        r.buf.normal_()
        dy = ops.alloc_view(N, side, side, C, 1, dt, dev)
        dy.buf.normal_()
        dz = ops.alloc_view(N, side, side, C, 1, dt, dev)
        stats = torch.zeros(5, C, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ws = torch.empty(ops.bn_workspace_bytes(N, side, side, C), dtype=torch.uint8, device=dev)
        nb = N * side * side * C * 2
        t_s = timed(lambda: ops.bn_train_stats(z, gamma, beta, stats, ws))
        t_f = timed(lambda: ops.bn_act_fwd(z, stats, y))
        t_fr = timed(lambda: ops.bn_act_fwd(z, stats, y, residual=r))
        t_b = timed(lambda: ops.bn_act_bwd(dy, z, stats, dz, dg, db, ws))
        print(f"{side:4d} {C:5d}  {t_s * 1e6:8.1f} {nb / t_s / 1e12:5.2f} | {t_f * 1e6:7.1f} {2 * nb / t_f / 1e12:5.2f} | "
              f"{t_fr * 1e6:9.1f} {3 * nb / t_fr / 1e12:5.2f} | {t_b * 1e6:7.1f} {5 * nb / t_b / 1e12:5.2f}")
        for k, t in (("stats", t_s), ("fwd", t_f), ("fwd+res", t_fr), ("bwd", t_b)):
            tot[k] += t * count
    print("weighted by layer count (ms): " + "  ".join(f"{k} {v * 1e3:.3f}" for k, v in tot.items()))


if __name__ == "__main__":
    main()

"""Phase timing of the convolution kernel's workgroups (needs a library built with -DYOLO_CONV_TRACE -- `make -C
yolo_amd/csrc trace` -- selected with YOLO_HIP_LIB): where a launch's time goes between kernel entry, first operands
landed, loop end and epilogue end.  YOLO_CONV_LONE=1: one workgroup per CU.
usage: YOLO_HIP_LIB=yolo_amd/libyolo_hip_trace.so python tools/conv_trace.py CIN COUT K STRIDE H [N=64] [RES=0]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from yolo_amd import ops  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    cin, cout, k, s, H = a[:5]
    N = a[5] if len(a) > 5 else 64
    res = a[6] if len(a) > 6 else 0
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    x = ops.alloc_view(N, H, H, cin, 1, torch.bfloat16, dev)
    x.interior().copy_(torch.randn(N, H, H, cin, generator=g).to(dev))
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp = ops.pack_conv(w, torch.zeros(cout), torch.bfloat16, dev)
    Ho = (H + 2 * (k // 2) - k) // s + 1
    y = ops.alloc_view(N, Ho, Ho, cout, 1, torch.bfloat16, dev)
    r = ops.alloc_view(N, Ho, Ho, cout, 1, torch.bfloat16, dev) if res else None
    run = lambda: ops.conv2d_fwd(x, wp, bp, y, cin=cin, cout=cout, k=k, stride=s, residual=r)
    import time
    t_end = time.time() + float(os.environ.get("YOLO_TRACE_WARM_SECONDS", "2"))
    while time.time() < t_end:   # the clock the chip settles at under this load, not the one of a cold start
        for _ in range(50):
            run()
        torch.cuda.synchronize()
    for _ in range(200):         # the traced launch follows a full queue directly
        run()
    path = "/tmp/conv_trace.txt"
    os.environ["YOLO_CONV_TRACE_DUMP"] = path
    run()
    torch.cuda.synchronize()
    del os.environ["YOLO_CONV_TRACE_DUMP"]
    head = open(path).readline().strip()
    rows = [l.split() for l in open(path) if not l.startswith("#")]
    t = np.array([[int(v) for v in r[1:5]] for r in rows], dtype=np.int64)
    hw = np.array([int(r[5], 16) for r in rows], dtype=np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0   # 100 MHz constant clock
    print(head)
    print(f"launch span (first entry -> last epilogue end): {us[:, 3].max():.1f} us, {len(rows)} workgroups")
    ph = np.stack([us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2], us[:, 3] - us[:, 0]], 1)
    for name, col in zip(("prologue (entry -> first operands)", "loop", "epilogue (+ store drain)", "whole tile"), ph.T):
        q = np.percentile(col, [5, 50, 95])
        print(f"  {name:36s} mean {col.mean():7.2f}  p5 {q[0]:7.2f}  p50 {q[1]:7.2f}  p95 {q[2]:7.2f} us")
    if len(rows[0]) >= 12:   # shader cycles over the loop / constant-clock ticks over the loop
        cyc = np.array([int(r[11]) for r in rows], dtype=np.float64)
        ticks = (t[:, 2] - t[:, 1]).astype(np.float64)
        ok = ticks > 0
        ghz = cyc[ok] / ticks[ok] * 0.1
        print(f"  shader clock inside the loop: median {np.median(ghz):.2f} GHz (p5 {np.percentile(ghz, 5):.2f}, "
              f"p95 {np.percentile(ghz, 95):.2f})")
    if len(rows[0]) >= 11:   # set-up part of the prologue
        d = (np.array([int(r[10]) for r in rows], dtype=np.int64) - t0) / 100.0 - us[:, 0]
        print(f"    prologue: entry -> addresses ready {d.mean():5.2f} us (p5 {np.percentile(d, 5):5.2f}, p95 {np.percentile(d, 95):5.2f})")
    if len(rows[0]) >= 10:   # epilogue sub-phases (per half: tile staged in LDS, stores issued)
        e = (np.array([[int(v) for v in r[6:10]] for r in rows], dtype=np.int64) - t0) / 100.0
        if (e[:, 2] > 0).all():
            marks = [us[:, 2], e[:, 0], e[:, 1], e[:, 2], e[:, 3], us[:, 3]]
            names = ["half 0 staged", "half 0 stores issued", "half 1 staged", "half 1 stores issued", "stores drained"]
        else:
            marks = [us[:, 2], e[:, 0], e[:, 1], us[:, 3]]
            names = ["tile staged", "stores issued", "stores drained"]
        for i, nm in enumerate(names):
            d = marks[i + 1] - marks[i]
            print(f"    epilogue: {nm:22s} +{d.mean():5.2f} us (p5 {np.percentile(d, 5):5.2f}, p95 {np.percentile(d, 95):5.2f})")
    # per hardware slot: gaps between one tile's end and the next tile's start on the same CU
    xcc, cu, se = hw >> 32, (hw >> 8) & 0xf, (hw >> 13) & 0x7
    key = (xcc * 8 + se) * 16 + cu
    order = np.argsort(us[:, 0])
    print(f"  distinct CUs seen: {len(np.unique(key))}")
    starts = np.sort(us[:, 0])
    print("  workgroup entry times, deciles:", " ".join(f"{v:.1f}" for v in np.percentile(starts, range(0, 101, 10))))
    ends = np.sort(us[:, 3])
    print("  workgroup end times, deciles:  ", " ".join(f"{v:.1f}" for v in np.percentile(ends, range(0, 101, 10))))
    # busy fraction: sum of loop time over the 512 slots' span
    span = us[:, 3].max()
    print(f"  sum(loop) / (512 slots x span) = {ph[:, 1].sum() / (512 * span):.3f}; "
          f"sum(whole) / (512 x span) = {ph[:, 3].sum() / (512 * span):.3f}")
    # tiles per CU
    cnt = np.bincount(key)
    cnt = cnt[cnt > 0]
    print(f"  tiles per CU: min {cnt.min()} max {cnt.max()} mean {cnt.mean():.2f}")
    _ = order


if __name__ == "__main__":
    main()

"""Times every distinct convolution shape of YOLOv3-416 (batch 64, bf16) with the pixel tile pinned (trace build:
YOLO_HIP_LIB=yolo_amd/libyolo_hip_trace.so YOLO_TP=64|96|128|192) or chosen by the library (YOLO_TP unset), to check
the tile choice of pick_tp against measurements.  usage: python tools/tp_sweep.py  -> one JSON line"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from yolo_amd import ops  # noqa: E402

# (cin, cout, k, stride, H_in, residual)
SHAPES = [(64, 128, 3, 2, 208, 0), (128, 64, 1, 1, 104, 0), (128, 256, 3, 2, 104, 0),
          (256, 128, 1, 1, 52, 0), (128, 256, 3, 1, 52, 1), (256, 512, 3, 2, 52, 0), (384, 128, 1, 1, 52, 0),
          (256, 255, 1, 1, 52, 0), (512, 256, 1, 1, 26, 0), (256, 512, 3, 1, 26, 1), (512, 1024, 3, 2, 26, 0),
          (768, 256, 1, 1, 26, 0), (512, 255, 1, 1, 26, 0), (256, 128, 1, 1, 26, 0), (1024, 512, 1, 1, 13, 0),
          (512, 1024, 3, 1, 13, 1), (1024, 255, 1, 1, 13, 0), (512, 256, 1, 1, 13, 0)]


def main():
    N = 64
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    out = {"tp": os.environ.get("YOLO_TP", "auto"), "us": {}}
    for cin, cout, k, s, H, res in SHAPES:
        x = ops.alloc_view(N, H, H, cin, 1, torch.bfloat16, dev)
        x.interior().copy_(torch.randn(N, H, H, cin, generator=g).to(dev))
        w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        wp, bp = ops.pack_conv(w, torch.zeros(cout), torch.bfloat16, dev)
        Ho = (H + 2 * (k // 2) - k) // s + 1
        y = ops.alloc_view(N, Ho, Ho, (cout + 7) // 8 * 8, 1, torch.bfloat16, dev)
        r = ops.alloc_view(N, Ho, Ho, cout, 1, torch.bfloat16, dev) if res else None
        run = lambda: ops.conv2d_fwd(x, wp, bp, y, cin=cin, cout=cout, k=k, stride=s, residual=r)
        for _ in range(5):
            run()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                run()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 100.0)
        out["us"][f"{cin}->{cout} k{k} s{s} {H}"] = round(sorted(ts)[2], 1)
        del x, y, r
    print(json.dumps(out))


if __name__ == "__main__":
    main()

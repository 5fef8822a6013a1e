"""A/B timing of two builds of the library on single convolution layers, interleaved in ONE process.
usage: conv_ab.py LIB_A LIB_B [N=64]     (paths of two libyolo_hip*.so; prints per layer the median time of each)"""
import statistics
import sys

import torch

sys.path.insert(0, ".")
from yolo_amd import _lib, ops  # noqa: E402

# (Cin, Cout, k, stride, H, residual): the 3x3 / stride-1 layers of Darknet-53 + neck at 416 x 416 and a few others;
# Cout 255 = the head convolutions (f32 output, linear)
LAYERS = [(64, 128, 3, 1, 104, 1), (128, 256, 3, 1, 52, 1), (128, 256, 3, 1, 52, 0), (256, 512, 3, 1, 26, 1),
          (256, 512, 3, 1, 26, 0), (512, 1024, 3, 1, 13, 1), (512, 1024, 3, 1, 13, 0),
          (256, 128, 1, 1, 52, 0), (512, 256, 1, 1, 26, 0), (256, 512, 3, 2, 52, 0), (1024, 512, 1, 1, 13, 0),
          (128, 64, 1, 1, 104, 0), (512, 1024, 3, 2, 26, 0), (256, 255, 1, 1, 52, 0), (512, 255, 1, 1, 26, 0),
          (1024, 255, 1, 1, 13, 0), (64, 128, 3, 2, 208, 0), (128, 256, 3, 2, 104, 0), (384, 128, 1, 1, 52, 0)]


def main():
    A, B = _lib.load(sys.argv[1]), _lib.load(sys.argv[2])
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    print(f"{'layer':34s} {'A us':>9s} {'B us':>9s} {'B/A':>6s}  {'A TF/s':>8s} {'B TF/s':>8s}  kernels")
    for cin, cout, k, s, H, res in LAYERS:
        x = ops.alloc_view(N, H, H, cin, 1, torch.bfloat16, dev)
        x.interior().copy_(torch.randn(N, H, H, cin, generator=g).to(dev))
        w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        wp, bp = ops.pack_conv(w, torch.zeros(cout), torch.bfloat16, dev)
        Ho = (H + 2 * (k // 2) - k) // s + 1
        head = cout == 255
        y = ops.alloc_view(N, Ho, Ho, (cout + 7) // 8 * 8, 0 if head else 1, torch.float32 if head else torch.bfloat16, dev)
        r = ops.alloc_view(N, Ho, Ho, cout, 1, torch.bfloat16, dev) if res else None
        if r is not None:
            r.interior().copy_(torch.randn(N, Ho, Ho, cout, generator=g).to(dev))
        run = lambda: ops.conv2d_fwd(x, wp, bp, y, cin=cin, cout=cout, k=k, stride=s, residual=r,
                                     slope=1.0 if head else 0.1)
        times = {0: [], 1: []}
        names = {}
        for rnd in range(7):
            for which, h in enumerate((A, B)):
                _lib.use(h)
                names[which] = ops.conv_kernel_name(x, y, cin, cout, k, s)
                run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[which].append(e0.elapsed_time(e1) * 100.0)
        a, b = statistics.median(times[0]), statistics.median(times[1])
        fl = 2.0 * N * Ho * Ho * cout * cin * k * k
        tag = f"{cin}->{cout} k{k} s{s} {H}x{H}{' +res' if res else ''}"
        print(f"{tag:34s} {a:9.1f} {b:9.1f} {b / a:6.3f}  {fl / a / 1e6:8.1f} {fl / b / 1e6:8.1f}  "
              f"{names[0].split('<')[1]} | {names[1].split('<')[1]}", flush=True)


if __name__ == "__main__":
    main()

"""Single-layer convolution micro-benchmark (for rocprofv3 counter passes and A/B timing).
usage: conv_bench.py CIN COUT K STRIDE H [N=64] [REPS=20] [RES=0]"""
import sys

import torch

sys.path.insert(0, ".")
from yolo_amd import ops  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    cin, cout, k, s, H = a[:5]
    N = a[5] if len(a) > 5 else 64
    reps = a[6] if len(a) > 6 else 20
    res = a[7] if len(a) > 7 else 0
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
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * N * Ho * Ho * cout * cin * k * k
    print(f"conv {cin}->{cout} k{k} s{s} H{H} N{N}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()

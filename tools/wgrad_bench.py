"""Weight-gradient micro-benchmark over the distinct trainable conv shapes of YOLOv3-416 (batch 64)."""
import sys
import torch
sys.path.insert(0, ".")
from yolo_amd import ops

SHAPES = [(64, 128, 3, 2, 208, 1), (128, 64, 1, 1, 104, 2), (64, 128, 3, 1, 104, 2), (128, 256, 3, 2, 104, 1),
          (256, 128, 1, 1, 52, 10), (128, 256, 3, 1, 52, 11), (384, 128, 1, 1, 52, 1), (256, 255, 1, 1, 52, 1),
          (256, 512, 3, 2, 52, 1), (512, 256, 1, 1, 26, 10), (256, 512, 3, 1, 26, 11), (768, 256, 1, 1, 26, 1),
          (512, 1024, 3, 2, 26, 1), (1024, 512, 1, 1, 13, 7), (512, 1024, 3, 1, 13, 7), (1024, 255, 1, 1, 13, 1)]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda")
    tot = 0.0
    for cin, cout, k, s, H, cnt in SHAPES:
        Ho = (H + 2 * (k // 2) - k) // s + 1
        x = ops.alloc_view(N, H, H, cin, 1, torch.bfloat16, dev)
        x.interior().normal_()
        cp = (cout + 7) // 8 * 8
        dz = ops.alloc_view(N, Ho, Ho, cp, 1, torch.bfloat16, dev)
        dz.interior().normal_()
        dw = torch.zeros(cout, k, k, cin, device=dev)
        ws = torch.empty(ops.wgrad_workspace_bytes(N * Ho * Ho, cin, cout, k), dtype=torch.uint8, device=dev)
        run = lambda: ops.conv2d_wgrad(x, dz, dw, ws, cin=cin, cout=cout, k=k, stride=s)
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 2.0 * N * Ho * Ho * cout * cin * k * k
        tot += ms * cnt
        print(f"wgrad {cin:4d}->{cout:4d} k{k} s{s} H{H:3d} x{cnt:2d}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF/s  ws {ws.numel()/1e6:6.1f} MB")
    print(f"total per step {tot:.2f} ms")


if __name__ == "__main__":
    main()

"""Per-layer timing of the inference plan on one MI355X (HIP events around repeated launches of each
step): time, effective TFLOP/s and the HBM GB/s a single pass over the layer's tensors would need."""
import sys

import torch

sys.path.insert(0, ".")
import workload  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
    reps = 10
    model, cfg = workload.build_model(config="yolov3-aug.yaml", dtype="bf16")
    eng = model.backbone.engine()
    P = eng.plan(N, size, size)
    imgs = workload.synthetic_images(N, size, size, seed=1, device="cuda")
    eng.run_backbone(imgs, cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
    torch.cuda.synchronize()
    rows = []
    convs = dict(model.backbone.darknet_modules())

    def timeit(fn):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    stem = eng._packed["bottom_up.stem.conv"]
    from yolo_amd import ops
    if P.stem_fused_y is not None:
        d = eng._packed["bottom_up.res1.0.downsample"]
        t = timeit(lambda: ops.stem_down_fwd(imgs, stem.w, stem.b, d.w, d.b, P.stem_fused_y, padded_hw=(size, size),
                                             mean=cfg.MODEL.PIXEL_MEAN, std=cfg.MODEL.PIXEL_STD))
        fl = 2 * N * size * size * 32 * 27 + 2 * N * (size // 2) ** 2 * 64 * 32 * 9
        by = N * size * size * 3 + N * (size // 2) ** 2 * 64 * 2
        rows.append(("stem+res1.0.downsample (fused)", 3, 64, 3, 2, size, t, fl, by))
    else:
        t = timeit(lambda: ops.stem_fwd(imgs, stem.w, stem.b, P.stem_out, mean=cfg.MODEL.PIXEL_MEAN, std=cfg.MODEL.PIXEL_STD))
        fl = 2 * N * size * size * 32 * 27
        by = N * size * size * (3 + 32 * 2)
        rows.append(("stem", 3, 32, 3, 1, size, t, fl, by))
    for i, step in enumerate(P.pre_steps):  # fused residual blocks of the first stage (64 channels)
        h = size // 2
        fl = 2 * N * h * h * (32 * 64 + 64 * 32 * 9)
        by = N * h * h * 64 * 2 * 2
        rows.append((f"bottom_up.res1.{i} (fused block)", 64, 64, 3, 1, h, timeit(step), fl, by))
    for meta, step in zip(P.meta, P.steps):
        c = convs[meta["name"]]
        y = meta["y"]
        up = 2 if meta["up"] else 1
        ho, wo = y.H // up, y.W // up
        fl = 2 * N * ho * wo * c.out_channels * c.in_channels * c.kernel_size ** 2
        esz = 2
        by = N * (meta["x"].H * meta["x"].W * c.in_channels * esz + y.H * y.W * c.out_channels * y.buf.element_size()
                  + (ho * wo * c.out_channels * esz if meta["residual"] is not None else 0)) \
            + c.out_channels * c.in_channels * c.kernel_size ** 2 * esz
        rows.append((meta["name"], c.in_channels, c.out_channels, c.kernel_size, c.stride, meta["x"].H, timeit(step), fl, by))
    tot = sum(r[6] for r in rows)
    print(f"{'layer':34s} {'cin':>5s} {'cout':>5s} k s {'H':>4s} {'ms':>8s} {'TF/s':>7s} {'GB/s':>7s} {'%':>5s}")
    for r in rows:
        print(f"{r[0]:34s} {r[1]:5d} {r[2]:5d} {r[3]} {r[4]} {r[5]:4d} {r[6]:8.4f} {r[7]/r[6]/1e9:7.1f} {r[8]/r[6]/1e6:7.0f} {100*r[6]/tot:5.1f}")
    print(f"total {tot:.3f} ms  -> {sum(r[7] for r in rows)/tot/1e9:.1f} TFLOP/s, {N/tot*1e3:.0f} img/s (sum of isolated layers)")
    # group by distinct shape
    agg = {}
    for r in rows:
        k = r[1:6]
        a = agg.setdefault(k, [0, 0.0, 0, 0])
        a[0] += 1; a[1] += r[6]; a[2] += r[7]; a[3] += r[8]
    print("\nby shape (cin,cout,k,s,H): count ms TF/s GB/s %")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{str(k):28s} x{a[0]:2d} {a[1]:8.3f} {a[2]/a[1]/1e9:7.1f} {a[3]/a[1]/1e6:7.0f} {100*a[1]/tot:5.1f}")


if __name__ == "__main__":
    main()

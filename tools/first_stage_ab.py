"""A/B timing of the two fused first-stage kernels (stem + first down-sampling conv; the 64-channel residual block) of
several builds of the library, interleaved in ONE process, with a bit-for-bit comparison of their outputs.
usage: first_stage_ab.py LIB_A LIB_B [...]     (paths of libyolo_hip*.so; batch 64, 416 x 416, bf16)"""
import statistics
import sys

import torch

sys.path.insert(0, ".")
import workload  # noqa: E402
from yolo_amd import _lib, ops  # noqa: E402


def main():
    libs = [_lib.load(p) for p in sys.argv[1:]]
    model, cfg = workload.build_model(config="yolov3-aug.yaml", dtype="bf16")
    eng = model.backbone.engine()
    P = eng.plan(64, 416, 416)
    imgs = workload.synthetic_images(64, 416, 416, seed=1, device="cuda")
    stem, d = eng._packed["bottom_up.stem.conv"], eng._packed["bottom_up.res1.0.downsample"]
    c1, c2 = eng._packed["bottom_up.res1.0.conv1"], eng._packed["bottom_up.res1.0.conv2"]
    X = P.stem_fused_y
    Y = ops.alloc_view(64, 208, 208, 64, 1, torch.bfloat16, torch.device("cuda"))
    runs = {"stem_down": (lambda: ops.stem_down_fwd(imgs, stem.w, stem.b, d.w, d.b, X, padded_hw=(416, 416),
                                                    mean=cfg.MODEL.PIXEL_MEAN, std=cfg.MODEL.PIXEL_STD), X),
            "resblock_c64": (lambda: ops.resblock_c64_fwd(X, c1.w, c1.b, c2.w, c2.b, Y), Y)}
    for name, (run, out) in runs.items():
        times = {i: [] for i in range(len(libs))}
        outs = []
        for rnd in range(6):
            for i, h in enumerate(libs):
                _lib.use(h)
                run()
                torch.cuda.synchronize()
                if rnd == 0:
                    outs.append(out.buf.clone())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[i].append(e0.elapsed_time(e1) / 20 * 1e3)
        _lib.use(libs[0])
        run()   # leave the first library's output in place for the next kernel's input
        print(name, "  ".join(f"{p.split('/')[-1]}: {statistics.median(times[i]):.1f} us"
                              f"{'' if torch.equal(outs[0], outs[i]) else ' (OUTPUT DIFFERS)'}" for i, p in enumerate(sys.argv[1:])))


if __name__ == "__main__":
    main()

"""Single-image latency of the whole hot path (the reference's own use: tools/inference.py runs one image per call):
batch 1 (and a few small batches), 416 x 416, bf16, with and without split-K convolutions, interleaved in one process.
(Replaying the pass from a captured HIP graph was measured too: 1.14 ms against 1.12 ms eager at batch 1 -- the pass is
bound by its ~125 dependent kernels, not by the host -- so no graph path is kept.)
usage: python tools/latency_bench.py [reps] [batches, e.g. 1,4,16]"""
import json
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
import workload  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3-aug.yaml", dtype="bf16", head_dtype="f32")
    eng = model.backbone.engine()
    kw = dict(score_thr=cfg.MODEL.YOLOV3.TEST_SCORE_THRESH, nms_thr=cfg.MODEL.YOLOV3.TEST_NMS_THRESH,
              mean=cfg.MODEL.PIXEL_MEAN, std=cfg.MODEL.PIXEL_STD)
    batches = tuple(int(b) for b in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 4, 16)
    for batch in batches:
        imgs = workload.synthetic_images(batch, 416, 416, seed=1, device=model.device)
        row = dict(batch=batch)
        dets = {}
        for rnd in range(3):
            for split in (True, False):
                eng.split_k = split
                eng._plans.clear()
                for _ in range(5):
                    det, n = eng.detect(imgs, **kw)
                torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    det, n = eng.detect(imgs, **kw)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                dets[split] = (det.clone(), n.clone())
                row.setdefault(f"ms_splitk_{'on' if split else 'off'}", []).append(round(statistics.median(ts) * 1e3, 3))
        a, b = dets[True], dets[False]
        row["same_detection_counts"] = bool(torch.equal(a[1], b[1]))
        k = int(a[1][0])
        row["max_score_difference_image0"] = round(float((a[0][0, :k, 4] - b[0][0, :k, 4]).abs().max()) if k and torch.equal(a[1], b[1]) else -1.0, 5)
        for key in ("ms_splitk_on", "ms_splitk_off"):
            row[key] = min(row[key])
        row["images_per_s_splitk_on"] = round(batch / row["ms_splitk_on"] * 1e3, 1)
        print(json.dumps(row), flush=True)
    eng.split_k = True


if __name__ == "__main__":
    main()

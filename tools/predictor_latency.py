"""Latency of the reference's own use of the path: ``Predictor(cfg)(image)`` of tools/inference.py, one decoded image
per call (host uint8 HWC in, Instances out), configs/yolov3.yaml, random weights with calibrated BatchNorm statistics.
Wall time per call as the caller sees it (upload, resize, model, read-back of the detections), and the same call
split into its parts.   usage: python tools/predictor_latency.py [reps]"""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import workload  # noqa: E402
from inference import Predictor  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3.yaml", dtype="bf16", head_dtype="f32")
    pred = Predictor.__new__(Predictor)   # the calibrated model instead of a checkpoint file
    pred.cfg, pred.model = cfg, model.eval()
    pred.min_size, pred.max_size, pred.input_format = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, cfg.INPUT.FORMAT
    rng = np.random.default_rng(0)
    for name, (h, w) in (("416x416", (416, 416)), ("480x640", (480, 640))):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for _ in range(10):
            out = pred(img)
        torch.cuda.synchronize()
        ts, tp, tm = [], [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = pred(img)
            n = len(out["instances"])
            ts.append(time.perf_counter() - t0)
        for _ in range(reps):
            t0 = time.perf_counter()
            x = pred.preprocess(img)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            with torch.no_grad():
                pred.model([dict(height=h, width=w, image=x)])
            t2 = time.perf_counter()
            tp.append(t1 - t0)
            tm.append(t2 - t1)
        med = lambda v: round(statistics.median(v) * 1e3, 3)  # noqa: E731
        print(json.dumps(dict(image=name, detections=n, ms_per_call=med(ts), ms_preprocess=med(tp), ms_model=med(tm),
                              calls_per_s=round(1.0 / statistics.median(ts), 1))), flush=True)


if __name__ == "__main__":
    main()

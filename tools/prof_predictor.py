"""Host-side profile (cProfile) of ``Predictor(cfg)(image)``, one 416 x 416 image per call: which Python functions a
single-image call spends its time in (the GPU's pass takes 1.0 ms; everything above that is host work).
usage: python tools/prof_predictor.py [calls]"""
import cProfile
import io
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import workload  # noqa: E402
from inference import Predictor  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3.yaml", dtype="bf16", head_dtype="f32")
    pred = Predictor.__new__(Predictor)   # the calibrated model instead of a checkpoint file
    pred.cfg, pred.model = cfg, model.eval()
    pred.min_size, pred.max_size, pred.input_format = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, cfg.INPUT.FORMAT
    img = np.random.default_rng(0).integers(0, 256, (416, 416, 3), dtype=np.uint8)
    for _ in range(20):
        pred(img)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(calls):
        pred(img)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()

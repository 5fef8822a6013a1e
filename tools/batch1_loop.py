"""Batch-1 inference loop for profiling (rocprofv3 --kernel-trace --stats -- python3 tools/batch1_loop.py)."""
import sys

import torch

sys.path.insert(0, ".")
import workload  # noqa: E402

model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3-aug.yaml", dtype="bf16", head_dtype="f32")
eng = model.backbone.engine()
kw = dict(score_thr=cfg.MODEL.YOLOV3.TEST_SCORE_THRESH, nms_thr=cfg.MODEL.YOLOV3.TEST_NMS_THRESH,
          mean=cfg.MODEL.PIXEL_MEAN, std=cfg.MODEL.PIXEL_STD)
imgs = workload.synthetic_images(int(sys.argv[1]) if len(sys.argv) > 1 else 1, 416, 416, seed=1, device=model.device)
eng._plans.clear()
for _ in range(100):
    eng.detect(imgs, **kw)
torch.cuda.synchronize()

"""Throughput of the whole inference path (pipelined steps, inputs resident in HBM, bf16, fp32 heads) over batch sizes
and input sides: images per second and the convolutions' end-to-end TFLOP/s.
usage: python tools/throughput_table.py [batch:side ...]   (default: a table from 1:416 to 64:608)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workload  # noqa: E402


def main():
    combos = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or \
        [(1, 416), (4, 416), (16, 416), (32, 416), (64, 416), (128, 416), (16, 608), (32, 608), (64, 608)]
    model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3-aug.yaml", dtype="bf16", head_dtype="f32")
    eng = model.backbone.engine()
    kw = dict(score_thr=cfg.MODEL.YOLOV3.TEST_SCORE_THRESH, nms_thr=cfg.MODEL.YOLOV3.TEST_NMS_THRESH,
              mean=cfg.MODEL.PIXEL_MEAN, std=cfg.MODEL.PIXEL_STD, max_det=100)
    print("batch side   ms/step  images/s  conv TFLOP/s")
    for batch, side in combos:
        imgs = workload.synthetic_images(batch, side, side, seed=1, device=model.device)
        eng._plans.clear()
        torch.cuda.empty_cache()

        def run(k):
            for _ in eng.detect_pipelined((imgs for _ in range(k)), **kw):
                pass

        run(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(10)
        torch.cuda.synchronize()
        k = max(10, int(1.0 / ((time.perf_counter() - t0) / 10)))
        t0 = time.perf_counter()
        run(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        flops = eng.plan(batch, side, side).flops
        print(f"{batch:5d} {side:4d}  {dt * 1e3:8.3f}  {batch / dt:8.0f}  {flops / dt / 1e12:8.1f}", flush=True)


if __name__ == "__main__":
    main()

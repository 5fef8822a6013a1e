"""Diagnostic: bf16-vs-fp32 distance of the ORACLE itself at the heads as a function of the residual-branch gain."""
import sys

import torch

sys.path.insert(0, ".")
import workload
from oracle import nms_c, yolo_ref as R
from oracle.compare import matched_fraction, rows_of

x8 = workload.synthetic_images(2, 416, 416, seed=31)
x = x8.float() / 255.0
for gain in (None, 0.5, 0.25, 0.1):
    model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3-aug.yaml", dtype="f32", residual_gain=gain)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        a = R.fpn_forward(x, sd)
        q = R.fpn_forward(x, sd, quant=True)
        row = [f"{k} {float((q[k] - a[k]).abs().max() / a[k].abs().max()):.3f}/{float((q[k] - a[k]).norm() / a[k].norm()):.3f}" for k in ("p3", "p4", "p5")]
        dets = []
        for h in (a, q):
            dec = R.decode_batch([h["p3"], h["p4"], h["p5"]], 80, 0.25, R.DEFAULT_ANCHORS)
            out = []
            for b, s, c in dec:
                keep = R.batched_nms(b, s, c, 0.5, "gpu", nms_c.nms)[:100]
                out.append(rows_of(b[keep], s[keep], c[keep]))
            dets.append(out)
        m = [matched_fraction(r, g, 0.27, 0.9, 0.02) for r, g in zip(*dets)]
        m5 = [matched_fraction(r, g, 0.27, 0.9, 0.05) for r, g in zip(*dets)]
    print("gain", gain, "| oracle bf16 vs fp32 (max/l2):", "  ".join(row), "| matched@0.02", m, "@0.05", m5,
          "| dets", [len(d) for d in dets[0]], flush=True)

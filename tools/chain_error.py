"""Diagnostic: error of the bf16 / f32 plans against the CPU oracle along the network."""
import sys
import torch
sys.path.insert(0, ".")
import workload
from oracle import yolo_ref as R

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model, cfg = workload.calibrated_model(size=size, calib_batch=4, config="yolov3-aug.yaml", dtype="f32")
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
imgs = workload.synthetic_images(2, 128, 160, seed=5)
x = imgs.float() / 255.0
t32, tq = {}, {}
with torch.no_grad():
    r32 = R.fpn_forward(x, sd, taps=t32)
    rq = R.fpn_forward(x, sd, taps=tq, quant=True)
t32.update(r32); tq.update(rq)
for dt in (torch.float32, torch.bfloat16):
    model.backbone.compute_dtype = dt
    out = model.backbone(x.cuda())
    torch.cuda.synchronize()
    P = model.backbone.engine().plan(2, 128, 160)
    for name in ("res1", "res2", "res3", "res4", "res5", "p5", "p4", "p3"):
        got = P.taps[name].nchw().float().cpu()
        def e(ref):
            d = (got.double() - ref.double())
            return float(d.abs().max() / ref.double().abs().max()), float(d.norm() / ref.double().norm())
        a = e(t32[name]) if name in t32 else (0, 0)
        b = e(tq[name]) if name in tq else (0, 0)
        print(f"{str(dt):16s} {name:5s} vs f32 oracle max {a[0]:.2e} l2 {a[1]:.2e} | vs bf16 oracle max {b[0]:.2e} l2 {b[1]:.2e} | std {float(got.std()):.3f}")

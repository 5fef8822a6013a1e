"""bf16 error of the whole chained network on TRAINED weights: fit the structured synthetic set (tools/overfit_check.py),
then compare the bf16 plan's stage outputs and head tensors with the fp32 oracle (CPU) on the trained state dict --
max |err| / max |ref| per tensor, next to what the oracle's own bf16 restatement costs.
usage: python tools/trained_bf16_error.py [steps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import overfit_check  # noqa: E402
from oracle import yolo_ref as R  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    res, model, data = overfit_check.run(steps=steps, n_images=32, side=256, dtype="bf16", log=lambda m: None,
                                         return_model=True)
    print(json.dumps(res))
    model.eval()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    imgs = torch.stack([d["image"] for d in data[:2]]).cpu()
    x = imgs.float() / 255.0
    taps32, tapsq = {}, {}
    with torch.no_grad():
        ref32 = R.fpn_forward(x, sd, taps=taps32)
        refq = R.fpn_forward(x, sd, taps=tapsq, quant=True)
    model.backbone.compute_dtype = torch.bfloat16
    out = model.backbone(x.to(model.device))
    torch.cuda.synchronize()
    plan = model.backbone.engine().plan(2, 256, 256)
    rows = {}
    for name in ("res3", "res4", "res5"):
        got = plan.taps[name].nchw().float().cpu()
        rows[name] = dict(hip_vs_fp32=rel(got, taps32[name]), oracle_bf16_vs_fp32=rel(tapsq[name], taps32[name]))
    for name in ("p3", "p4", "p5"):
        got = out[name].float().cpu()
        rows[name] = dict(hip_vs_fp32=rel(got, ref32[name]), oracle_bf16_vs_fp32=rel(refq[name], ref32[name]))
    for k, v in rows.items():
        print(k, {a: round(b, 5) for a, b in v.items()})


if __name__ == "__main__":
    main()

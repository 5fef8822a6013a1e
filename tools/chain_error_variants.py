"""Diagnostic (CPU arithmetic; the GPU only calibrates the state dict): how far a bf16 run of the network is from
the fp32 oracle at the stage outputs and heads, when the residual stream of some stages is kept in fp32 and only
the convolution operands are rounded to bf16.  Decides whether an fp32 residual stream is worth its bytes."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import workload
from oracle import yolo_ref as R

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model, cfg = workload.calibrated_model(size=size, calib_batch=4, config="yolov3-aug.yaml", dtype="f32")
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
x = workload.synthetic_images(2, 128, 160, seed=5).float() / 255.0
bf = R._bf16


def forward(fp32_stages):
    p = "backbone."
    t = R.conv_unit(bf(x), sd, p + "bottom_up.stem.conv", 1, 1, quant=True)
    outs = {}
    for si, nb in enumerate((1, 2, 8, 8, 4), start=1):
        keep = si in fp32_stages
        for b in range(nb):
            q = f"{p}bottom_up.res{si}.{b}"
            if b == 0:
                t = R.conv_unit(bf(t), sd, q + ".downsample", 2, 1, quant=True, round_out=not keep)
            res = t
            o = R.conv_unit(bf(t), sd, q + ".conv1", 1, 0, quant=True)
            o = R.conv_unit(o, sd, q + ".conv2", 1, 1, quant=True, round_out=False) + res
            t = o if keep else bf(o)
        outs[f"res{si}"] = t
    feats = ["res3", "res4", "res5"]
    result, topdown = [], None
    for idx in range(3):
        f = bf(outs[feats[-idx - 1]])
        if topdown is not None:
            f = torch.cat([topdown, f], dim=1)
        for i in range(5):
            k = 1 if i % 2 == 0 else 3
            f = R.conv_unit(f, sd, f"{p}fpn_lateral{idx}.blocks.{i}", 1, (k - 1) // 2, quant=True)
        o = R.conv_unit(f, sd, f"{p}fpn_output{idx}.conv", 1, 1, quant=True)
        o = R.conv_unit(o, sd, f"{p}fpn_output{idx}.head", 1, 0, act=False, quant=True, round_out=False)
        result.insert(0, o)
        if idx < 2:
            u = R.conv_unit(f, sd, f"{p}fpn_upsample{idx}.conv", 1, 0, quant=True)
            topdown = F.interpolate(u, scale_factor=2.0, mode="nearest")
    outs.update(zip(["p3", "p4", "p5"], result))
    return outs


with torch.no_grad():
    t32 = {}
    r32 = R.fpn_forward(x, sd, taps=t32)
    t32.update(r32)
    for stages in ((), (4, 5), (3, 4, 5), (1, 2, 3, 4, 5)):
        o = forward(set(stages))
        row = []
        for name in ("res3", "res4", "res5", "p5", "p4", "p3"):
            d = o[name].double() - t32[name].double()
            row.append(f"{name} {float(d.abs().max() / t32[name].double().abs().max()):.3f}")
        print("fp32 residual stream in stages", stages, "| max-norm error vs fp32 oracle:", "  ".join(row), flush=True)

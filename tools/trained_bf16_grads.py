"""Gradients of the bf16 training plan on TRAINED weights against the fp32 oracle's autograd (CPU): fit the structured
synthetic set, then one forward/backward on fresh images of the same kind; cosine and relative L2 per parameter tensor
and over all of them.   usage: python tools/trained_bf16_grads.py [steps]"""
import os
import statistics
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import overfit_check  # noqa: E402
from oracle import yolo_train_ref as T  # noqa: E402


def compare(steps=400, n_new=4, log=print):
    res, model, data = overfit_check.run(steps=steps, n_images=32, side=256, dtype="bf16", log=lambda m: None,
                                         return_model=True)
    C = model.num_classes
    fresh = overfit_check.make_set(n_new, 256, C, 77, model.device)
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    frozen = set(T.frozen_names(2))
    layer_of = lambda k: k.rsplit(".norm.", 1)[0].rsplit(".weight", 1)[0].rsplit(".bias", 1)[0]  # noqa: E731
    names = {k for k, _ in model.named_parameters()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k in names and layer_of(k) not in frozen}
    full = dict(sd)
    full.update(params)
    x = torch.stack([d["image"] for d in fresh]).cpu().float() / 255.0
    gb = [d["instances"].gt_boxes.tensor.cpu() for d in fresh]
    gc = [d["instances"].gt_classes.cpu() for d in fresh]
    want = T.forward_training(x, full, gb, gc, C, freeze_at=2)
    sum(want.values()).backward()
    model.train()
    for p in model.parameters():
        p.grad = None
    got = model(fresh)
    sum(got.values()).backward()
    torch.cuda.synchronize()
    live = dict(model.named_parameters())
    cos, l2 = [], []
    for k, p in params.items():
        g, r = live[k].grad.float().cpu().flatten().double(), p.grad.flatten().double()
        cos.append(float(F.cosine_similarity(g, r, dim=0)))
        l2.append(float((g - r).norm() / r.norm().clamp_min(1e-30)))
    ga = torch.cat([live[k].grad.float().cpu().flatten() for k in params]).double()
    ra = torch.cat([p.grad.flatten() for p in params.values()]).double()
    out = dict(losses_hip={k: round(float(v), 5) for k, v in got.items()},
               losses_oracle={k: round(float(v), 5) for k, v in want.items()},
               tensors=len(cos), cos_min=round(min(cos), 4), cos_median=round(statistics.median(cos), 4),
               l2_median=round(statistics.median(l2), 4), l2_max=round(max(l2), 4),
               cos_all=round(float(F.cosine_similarity(ga, ra, dim=0)), 4), norm_ratio_all=round(float(ga.norm() / ra.norm()), 4))
    log(out)
    return out


if __name__ == "__main__":
    compare(int(sys.argv[1]) if len(sys.argv) > 1 else 400)

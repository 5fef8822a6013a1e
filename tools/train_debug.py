"""Debug aid: one training step on the HIP path vs the CPU oracle, printing per-tensor errors."""
import sys
import torch
sys.path.insert(0, ".")
import workload
from oracle import yolo_train_ref as T
from yolo_amd.structures import Boxes, Instances

DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model, cfg = workload.build_model(config="yolov3.yaml", dtype=DT)
C = cfg.MODEL.YOLOV3.NUM_CLASSES
N, S = 4, 128
imgs = workload.synthetic_images(N, S, S, seed=3)
gb, gc = T.synthetic_targets(N, S, C, seed=11, max_boxes=6)
inputs = [dict(image=im, instances=Instances((S, S), gt_boxes=Boxes(b), gt_classes=c)) for im, b, c in zip(imgs, gb, gc)]
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
full = dict(sd); full.update(params)
heads_ref = {}
rec = {}
want = T.forward_training(imgs.float() / 255.0, full, gb, gc, C, freeze_at=2, heads_out=heads_ref, rec=rec)
for h in heads_ref.values():
    h.retain_grad()
sum(want.values()).backward()
model.train()
got = model(inputs)
sum(got.values()).backward()
torch.cuda.synchronize()
print({k: (float(got[k]), float(want[k])) for k in want})
P = model.backbone.train_engine().plan(N, S, S)
for name, hv in zip(["p3", "p4", "p5"], P.heads):
    g = hv.nchw().float().cpu(); r = heads_ref[name].detach()
    print(name, "head rel max", float((g - r).abs().max() / r.abs().max()), "rel l2", float((g - r).norm() / r.norm()), "std", float(r.std()))
rows = []
for name, p in model.named_parameters():
    if p.grad is None:
        continue
    ref = params[name].grad
    g = p.grad.float().cpu()
    cos = float(torch.nn.functional.cosine_similarity(g.flatten().double(), ref.flatten().double(), dim=0))
    l2 = float((g.double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-30))
    rows.append((l2, cos, name, float(ref.norm())))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f"{r[2]:55s} l2 {r[0]:.4f} cos {r[1]:.5f} |ref| {r[3]:.3e}")
print("median l2", sorted(r[0] for r in rows)[len(rows) // 2], "n", len(rows))

print("---- per layer: forward y and gradient dy vs oracle (rel l2)")
for L in P.layers:
    r = rec["backbone." + L.name]
    ref = r.get("out", r["y"]).detach()
    if L.up_dst is not None or L.conv.norm is None:
        got_y = L.y.nchw()[:, :ref.shape[1]].float().cpu()
    else:
        got_y = L.y.nchw().float().cpu()
    fy = float((got_y - ref).norm() / ref.norm())
    gy = float("nan")
    refg = r.get("out", r["y"]).grad
    if L.dy is not None and refg is not None and not L.frozen:
        gg = L.dy.nchw().float().cpu()
        gy = float((gg - refg).norm() / refg.norm().clamp_min(1e-30))
    gz = float("nan")
    if L.dz is not None and r["z"].grad is not None and L.conv.norm is not None:
        zz = L.dz.nchw().float().cpu()
        gz = float((zz - r["z"].grad).norm() / r["z"].grad.norm().clamp_min(1e-30))
    print(f"{L.name:40s} fwd {fy:.4f}  dy {gy:.4f}  dz {gz:.4f}")
print("---- one layer in detail")
for L in P.layers:
    if L.name not in ("fpn_output1.conv", "fpn_upsample1.conv"):
        continue
    r = rec["backbone." + L.name]
    zo = r["z"].detach()
    zm = L.z.nchw().float().cpu()
    print(L.name, "z rel l2", float((zm - zo).norm() / zo.norm()))
    mean = zo.mean((0, 2, 3)); var = zo.var((0, 2, 3), unbiased=False)
    st = L.stats.cpu()
    print("  mean err", float((st[0] - mean).abs().max()), "var rel err", float(((st[1] - var) / var).abs().max()),
          "min var", float(var.min()), "max |mean|/std", float((mean.abs() / var.sqrt()).max()))
    dzo = r["z"].grad
    dzm = L.dz.nchw().float().cpu()
    per_c = ((dzm - dzo).pow(2).sum((0, 2, 3)) / dzo.pow(2).sum((0, 2, 3))).sqrt()
    print("  per-channel dz rel err: max", float(per_c.max()), "median", float(per_c.median()), "argmax", int(per_c.argmax()))
    c = int(per_c.argmax())
    print("  worst channel: var", float(var[c]), "mean", float(mean[c]), "gamma", float(L.conv.norm.weight[c]))
    # recompute dz on the CPU from the engine's own dy, z
    dym = L.dy.nchw().float().cpu().requires_grad_(False)
    zz = zm.clone().requires_grad_(True)
    yy = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(zz, None, None, L.conv.norm.weight.detach().cpu(),
                                        L.conv.norm.bias.detach().cpu(), True, 0.1, 1e-5), 0.1)
    yy.backward(dym)
    print("  engine dz vs torch-on-engine-inputs", float((dzm - zz.grad).norm() / zz.grad.norm()),
          " oracle dz vs torch-on-engine-inputs", float((dzo - zz.grad).norm() / zz.grad.norm()))

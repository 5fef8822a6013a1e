"""End-to-end check that the training path trains a detector: a small synthetic set with real structure (filled
rectangles on a noisy background, class = colour) is fitted with the reference's loop shape
(losses = model(batch); sum(losses).backward(); optimizer.step()), then the trained model's own inference path
(decode + NMS) has to find the rectangles of the training images.  Prints the loss curve and the recall
(ground-truth boxes with a detection of the same class at IoU > 0.5).
usage: python tools/overfit_check.py [steps] [images] [side] [dtype]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import workload  # noqa: E402
from yolo_amd.structures import Boxes, Instances  # noqa: E402
from yolo_amd.trainer import Trainer  # noqa: E402


def make_set(n, side, classes, seed, device):
    g = torch.Generator().manual_seed(seed)
    palette = torch.randint(40, 256, (classes, 3), generator=g)
    data = []
    for _ in range(n):
        img = torch.randint(0, 48, (3, side, side), generator=g)
        k = int(torch.randint(1, 4, (1,), generator=g))
        boxes, cls = [], []
        for _ in range(k):
            w, h = (int(v) for v in torch.randint(side // 6, side // 2, (2,), generator=g))
            x0, y0 = int(torch.randint(0, side - w, (1,), generator=g)), int(torch.randint(0, side - h, (1,), generator=g))
            c = int(torch.randint(0, classes, (1,), generator=g))
            img[:, y0:y0 + h, x0:x0 + w] = palette[c].view(3, 1, 1)
            boxes.append([x0, y0, x0 + w, y0 + h])
            cls.append(c)
        inst = Instances((side, side), gt_boxes=Boxes(torch.tensor(boxes, dtype=torch.float32)),
                         gt_classes=torch.tensor(cls, dtype=torch.int64))
        data.append(dict(image=img.to(torch.uint8).to(device), instances=inst.to(device), height=side, width=side))
    return data


def iou(a, b):
    ix = (torch.min(a[:, None, 2], b[None, :, 2]) - torch.max(a[:, None, 0], b[None, :, 0])).clamp(min=0)
    iy = (torch.min(a[:, None, 3], b[None, :, 3]) - torch.max(a[:, None, 1], b[None, :, 1])).clamp(min=0)
    inter = ix * iy
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])  # noqa: E731
    return inter / (area(a)[:, None] + area(b)[None, :] - inter)


def recall(model, data, batch):
    model.eval()
    hit = total = dets = 0
    with torch.no_grad():
        for i in range(0, len(data), batch):
            chunk = data[i:i + batch]
            out = model([dict(image=d["image"], height=d["height"], width=d["width"]) for d in chunk])
            for d, o in zip(chunk, out):
                gt, pr = d["instances"], o["instances"]
                total += len(gt)
                dets += len(pr)
                if len(pr) == 0:
                    continue
                m = iou(gt.gt_boxes.tensor, pr.pred_boxes.tensor) > 0.5
                same = gt.gt_classes[:, None] == pr.pred_classes[None, :]
                hit += int((m & same).any(dim=1).sum())
    return hit / max(total, 1), dets / len(data)


def run(steps=400, n_images=32, side=256, dtype="bf16", batch=16, lr=0.01, log=print, return_model=False):
    model, cfg = workload.build_model(config="yolov3.yaml", dtype=dtype, device="cuda:0",
                                      opts=["OUTPUT_DIR", "", "SOLVER.CHECKPOINT_PERIOD", 0, "SOLVER.BASE_LR", lr,
                                            "SOLVER.WARMUP_ITERS", 50, "SOLVER.MAX_ITER", steps, "SOLVER.STEPS", (steps,)])
    data = make_set(n_images, side, cfg.MODEL.YOLOV3.NUM_CLASSES, 5, model.device)
    order = torch.Generator().manual_seed(9)

    def batches():
        while True:
            idx = torch.randperm(len(data), generator=order)[:batch].tolist()
            yield [data[i] for i in idx]

    tr = Trainer(cfg, model, batches(), log=lambda m: None)
    r0, _ = recall(model, data, batch)
    curve = []
    t0 = time.perf_counter()
    for it in range(steps):
        losses = tr.run_step()
        if it % max(steps // 20, 1) == 0 or it == steps - 1:
            vals = {k: round(float(v), 4) for k, v in losses.items()}
            curve.append((it, vals))
            log(f"iter {it:5d}  " + "  ".join(f"{k} {v:.4f}" for k, v in vals.items()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r1, per_image = recall(model, data, batch)
    first, last = sum(curve[0][1].values()), sum(curve[-1][1].values())
    res = dict(steps=steps, images=n_images, side=side, dtype=dtype, batch=batch, seconds=round(dt, 1),
               loss_first=round(first, 4), loss_last=round(last, 4), recall_before=round(r0, 3), recall_after=round(r1, 3),
               detections_per_image_after=round(per_image, 2))
    log(json.dumps(res))
    return (res, model, data) if return_model else res


if __name__ == "__main__":
    a = sys.argv[1:]
    run(steps=int(a[0]) if len(a) > 0 else 400, n_images=int(a[1]) if len(a) > 1 else 32,
        side=int(a[2]) if len(a) > 2 else 256, dtype=a[3] if len(a) > 3 else "bf16")

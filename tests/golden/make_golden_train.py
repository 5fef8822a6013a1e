"""Regenerates tests/golden/oracle_train_golden.npz from the CPU training oracle (oracle/yolo_train_ref.py):
seeded raw head tensors + synthetic ground truths -> per-anchor labels, the three losses and the gradient of
1.0 * confs + 0.5 * logits + 2.0 * regression w.r.t. the heads.

Like oracle_golden.npz these vectors come from the oracle's restatement (the reference cannot be imported in the
build container), so they pin the oracle against drift and give the HIP loss kernels a fixed target.  Run from
the repository root:
    python tests/golden/make_golden_train.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import yolo_ref as R, yolo_train_ref as T  # noqa: E402

C, A, S, N = 20, 3, 96, 3
WEIGHTS = (1.0, 0.5, 2.0)


def inputs():
    g = torch.Generator().manual_seed(41)
    heads = [torch.randn(N, A * (5 + C), S // s, S // s, generator=g) * 1.5 for s in R.STRIDES]
    gb, gc = T.synthetic_targets(N, S, C, seed=43, max_boxes=5)
    # (an image without ground truth is not representable here: Matcher.__call__, matcher.py:37, takes a max over
    # the empty gt axis and raises, in the reference as in the oracle)
    gb[0], gc[0] = torch.cat([gb[0], gb[0][:1]]), torch.cat([gc[0], (gc[0][:1] + 1) % C])  # two gts, one anchor
    return heads, gb, gc


def evaluate(heads, gb, gc):
    feats = [h.clone().requires_grad_(True) for h in heads]
    logits, confs, deltas = R.head_forward(feats, A, C)
    logits = [R.permute_to_N_HWA_K(t, C) for t in logits]
    confs = [R.permute_to_N_HWA_K(t, 1) for t in confs]
    deltas = [R.permute_to_N_HWA_K(t, 4) for t in deltas]
    anchors = torch.cat(R.grid_anchors([(f.shape[2], f.shape[3]) for f in feats], R.STRIDES))
    with torch.no_grad():
        labels, boxes = T.label_anchors(anchors, gb, gc, C, 0.5)
    losses = T.losses(anchors, logits, confs, deltas, labels, boxes, R.STRIDES, C)
    sum(w * v for w, v in zip(WEIGHTS, losses.values())).backward()
    return labels, losses, [f.grad for f in feats]


def main():
    heads, gb, gc = inputs()
    labels, losses, grads = evaluate(heads, gb, gc)
    out = {f"head{i}": h.numpy() for i, h in enumerate(heads)}
    for n in range(N):
        out[f"gt_boxes{n}"], out[f"gt_classes{n}"] = gb[n].numpy(), gc[n].numpy()
    out["labels"] = torch.stack(labels).numpy()
    out["losses"] = np.array([float(v.detach()) for v in losses.values()], dtype=np.float64)
    for i, g in enumerate(grads):
        out[f"dhead{i}"] = g.numpy()
    path = os.path.join(ROOT, "tests", "golden", "oracle_train_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes", out["losses"], int((out["labels"] >= 0).sum()))


if __name__ == "__main__":
    main()

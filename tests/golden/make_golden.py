"""Regenerates tests/golden/oracle_golden.npz from the CPU oracle (oracle/yolo_ref.py, oracle/nms_ref.c).

The reference cannot be imported in the build container (detectron2, torchvision, fvcore, yacs and cv2 are
absent), so these vectors are produced by the oracle's restatement and serve as a regression pin of the
oracle, not as vectors of the reference itself.  Run from the repository root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nms_c, yolo_ref as R  # noqa: E402
from tests import util  # noqa: E402


def main():
    out = {}
    heads = util.random_heads(1, 80, [(13, 13), (7, 7), (4, 4)], seed=31, obj_bias=-1.0)
    for i, h in enumerate(heads):
        out[f"dec_head{i}"] = h.numpy()
    for thr, tag in ((0.25, "a"), (0.001, "b")):
        (b, s, c), = R.decode_batch(heads, 80, thr)
        out[f"dec_{tag}_boxes"], out[f"dec_{tag}_scores"], out[f"dec_{tag}_classes"] = b.numpy(), s.numpy(), c.numpy()
    b, s, c = util.stress_boxes(2000, 77, num_classes=20)
    b[5] = b[4]  # exact duplicate
    s[9] = s[8]  # equal scores
    out["nms_boxes"], out["nms_scores"], out["nms_classes"] = b.numpy(), s.numpy(), c.numpy()
    out["nms_keep_vanilla"] = nms_c.batched_nms_vanilla(b, s, c, 0.5).numpy()
    out["nms_keep_trick"] = nms_c.batched_nms_trick(b, s, c, 0.5).numpy()
    sd = R.init_state_dict(7, num_classes=20, blocks=(1, 1, 1, 1, 1))
    R.randomize_norm_stats(sd, 8)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        o = R.fpn_forward(x, sd, blocks=(1, 1, 1, 1, 1))
    out["net_input"] = x.numpy()
    for k in ("p3", "p4", "p5"):
        out[f"net_{k}"] = o[k].numpy()
    path = os.path.join(ROOT, "tests", "golden", "oracle_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

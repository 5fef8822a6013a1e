"""Writes tests/golden/aug_golden.npz: small input / output pairs for the augmentation oracle (oracle/aug_ref.py).

  * ``pil_*``: outputs of Pillow itself (``Image.resize(..., BILINEAR)``, the resize detectron2's ResizeShortestEdge
    performs for the reference, yolo/data/detection_utils.py:55,104) -- these PIN the oracle's restatement.
    Generated with the Pillow version recorded in the file.
  * ``cv_*``, ``warp_*``, ``jitter_*``: outputs of the oracle's restatement of OpenCV's algorithms.  OpenCV is not
    installed here and the reference holds no image fixtures, so these only guard against regressions (parity
    unpinned for them).

usage: python tests/golden/make_golden_aug.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import aug_ref as A  # noqa: E402


def image(h, w, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[: h // 3, : w // 2] = rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)
    a[h // 2:, w // 2:, :] = a[h // 2:, w // 2:, :1]
    return a


def main():
    import PIL
    from PIL import Image
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, (h, w, nh, nw) in enumerate([(48, 64, 30, 40), (37, 53, 37, 20), (20, 31, 45, 77), (60, 40, 20, 40), (7, 9, 3, 2),
                                        (50, 70, 51, 69)]):
        img = image(h, w, i)
        out[f"pil_in_{i}"] = img
        out[f"pil_out_{i}"] = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    img = image(40, 56, 20)
    out["cv_in"] = img
    out["cv_letterbox_64"] = A.letterbox_apply(img, *[A.letterbox_params(40, 56, 64, 64)[k] for k in (0, 1, 3)])
    out["cv_resize_up"] = A.cv_resize_linear(img, (90, 101))
    out["cv_resize_down"] = A.cv_resize_linear(img, (17, 23))
    M = A.affine_matrix(40, 56, angle=7.0, scale=1.15, translate=(0.46, 0.55), shear=(0.05, -0.08))
    out["warp_M"] = M
    out["warp_affine"] = A.warp_affine(img, M)
    Mp = A.affine_matrix(40, 56, angle=-4.0, scale=0.9, translate=(0.5, 0.5), shear=(0.02, 0.03), perspective=(0.001, -0.0007))
    out["warp_Mp"] = Mp
    out["warp_perspective"] = A.warp_perspective(img, Mp)
    out["jitter_bgr"] = A.color_jitter(img, 0.93, 1.41, 0.62, False)
    out["jitter_rgb"] = A.color_jitter(img, 1.08, 0.7, 1.3, True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aug_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

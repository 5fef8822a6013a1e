"""CPU checks of the augmentation oracle (oracle/aug_ref.py) and of the host side of yolo_amd.data (no GPU):
the Pillow restatement against Pillow's own outputs (committed fixtures and, when Pillow is importable, live),
hand-computed known answers for the OpenCV restatements, the committed regression vectors, the random draws and the
box arithmetic of the transform classes against the oracle."""
import os
import random

import numpy as np
import pytest

from oracle import aug_ref as A

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aug_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def test_pillow_restatement_matches_pillow_fixtures(golden):
    n = 0
    while f"pil_in_{n}" in golden:
        want = golden[f"pil_out_{n}"]
        assert np.array_equal(A.pil_resize_bilinear(golden[f"pil_in_{n}"], want.shape[:2]), want)
        n += 1
    assert n == 6


def test_pillow_restatement_matches_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for h, w, nh, nw in [(33, 47, 20, 31), (31, 20, 64, 41), (90, 120, 30, 40), (16, 16, 16, 5)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(A.pil_resize_bilinear(img, (nh, nw)),
                              np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR)))


def test_device_tables_equal_the_oracle_coefficients():
    from yolo_amd import ops
    for a, b in [(64, 40), (53, 20), (31, 77), (100, 33), (640, 427), (480, 640), (500, 375), (1, 4), (5, 1)]:
        bounds, coef = A.pil_coeffs(a, b)
        tb, tk, ks = ops.resample_tables(a, b, "cpu")
        assert ks == coef.shape[1] and np.array_equal(tb.numpy(), bounds) and np.array_equal(tk.numpy(), coef)


def test_cv_linear_resize_known_answers():
    # 2 -> 4 columns: source offsets -0.25 (clamped), 0.25, 0.75, 1.25 (clamped) -> weights 0, 1/4, 3/4, 1
    img = np.zeros((1, 2, 3), np.uint8)
    img[0, 1] = 100
    assert A.cv_resize_linear(img, (1, 4))[0, :, 0].tolist() == [0, 25, 75, 100]
    # a constant image stays constant, at any ratio
    c = np.full((13, 7, 3), 201, np.uint8)
    assert (A.cv_resize_linear(c, (40, 3)) == 201).all()
    # 2x decimation samples midway between pixel pairs
    ramp = np.repeat(np.arange(0, 64, 4, dtype=np.uint8)[None, :, None], 3, 2)
    assert A.cv_resize_linear(ramp, (1, 8))[0, :, 0].tolist() == [2, 10, 18, 26, 34, 42, 50, 58]


def test_letterbox_geometry():
    # augmentation_impl.py:166-172 on a 480 x 640 image into 640 x 640: ratio 1, 80-pixel bars of 114 above and below
    img = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    top, left, ratio, new_size = A.letterbox_params(480, 640, 640, 640)
    assert (top, left, ratio, new_size) == (80, 0, 1.0, (480, 640))
    out = A.letterbox_apply(img, top, left, new_size)
    assert out.shape == (640, 640, 3) and (out[:80] == 114).all() and (out[560:] == 114).all()
    assert np.array_equal(out[80:560], img)
    pts = A.letterbox_coords(np.array([[10.0, 20.0]]), 0.5, 7, 3)
    assert pts.tolist() == [[8.0, 17.0]]


def test_hsv_known_answers_and_round_trip():
    px = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [255, 255, 255], [0, 0, 0], [128, 128, 128], [0, 255, 255]]],
                  np.uint8)  # BGR: red, green, blue, white, black, grey, yellow
    hsv = A.to_hsv_u8(px, rgb=False)[0].tolist()
    assert hsv == [[0, 255, 255], [60, 255, 255], [120, 255, 255], [0, 0, 255], [0, 0, 0], [0, 0, 128], [30, 255, 255]]
    assert np.array_equal(A.from_hsv_u8(A.to_hsv_u8(px, False), False), px)
    # RGB order is the same conversion on swapped channels
    img = np.random.default_rng(1).integers(0, 256, (32, 32, 3), dtype=np.uint8)
    assert np.array_equal(A.to_hsv_u8(img, True), A.to_hsv_u8(img[..., ::-1], False))
    # unit gains change a pixel by at most the quantisation of the 8-bit HSV round trip
    back = A.color_jitter(img, 1.0, 1.0, 1.0, False).astype(int)
    assert np.abs(back - img).max() <= 4
    # transforms.py:181-183: hue table wraps at 180, the others saturate at 255
    lh, ls, lv = A.jitter_luts(1.1, 1.5, 0.5)
    assert lh[170] == int(170 * 1.1 % 180) == 7 and ls[200] == 255 and lv[200] == 100


def test_affine_known_answers():
    img = np.random.default_rng(2).integers(0, 256, (40, 50, 3), dtype=np.uint8)
    # translate (0.5, 0.5) undoes the centring (transforms.py:24-39): identity
    assert np.array_equal(A.affine_matrix(40, 50, translate=(0.5, 0.5)), np.eye(3))
    assert np.array_equal(A.warp_affine(img, np.eye(3)), img)
    # a whole-pixel shift moves the picture and fills with 114
    M = np.eye(3)
    M[0, 2], M[1, 2] = 3, -2
    out = A.warp_affine(img, M)
    assert np.array_equal(out[:38, 3:], img[2:, :47]) and (out[38:] == 114).all() and (out[:, :3] == 114).all()
    # a half-pixel shift averages neighbours (rounded to nearest)
    M[0, 2], M[1, 2] = 0.5, 0
    out = A.warp_affine(img, M)
    want = (img[:, :-1].astype(int) + img[:, 1:] + 1) >> 1
    assert np.array_equal(out[:, 1:], want)
    # perspective with a zero projective row is the same map; the two paths round the coordinates differently
    # (10-bit row / column terms vs. one rounding of the 5-bit coordinate), so pixels agree only closely
    Ma = A.affine_matrix(40, 50, angle=5, scale=1.1, translate=(0.45, 0.52), shear=(0.03, -0.02))
    d = np.abs(A.warp_perspective(img, Ma).astype(int) - A.warp_affine(img, Ma))
    assert d.mean() < 1.0 and (d == 0).mean() > 0.5
    # cv2.warpAffine's inversion
    inv = A.invert_affine(Ma[:2]).reshape(2, 3)
    assert np.allclose(np.vstack([inv, [0, 0, 1]]) @ Ma, np.eye(3), atol=1e-12)


def test_committed_regression_vectors(golden):
    img = golden["cv_in"]
    assert np.array_equal(A.letterbox_apply(img, *[A.letterbox_params(40, 56, 64, 64)[k] for k in (0, 1, 3)]),
                          golden["cv_letterbox_64"])
    assert np.array_equal(A.cv_resize_linear(img, (90, 101)), golden["cv_resize_up"])
    assert np.array_equal(A.cv_resize_linear(img, (17, 23)), golden["cv_resize_down"])
    assert np.array_equal(A.warp_affine(img, golden["warp_M"]), golden["warp_affine"])
    assert np.array_equal(A.warp_perspective(img, golden["warp_Mp"]), golden["warp_perspective"])
    assert np.array_equal(A.color_jitter(img, 0.93, 1.41, 0.62, False), golden["jitter_bgr"])
    assert np.array_equal(A.color_jitter(img, 1.08, 0.7, 1.3, True), golden["jitter_rgb"])


def test_mosaic_rectangles_tile_the_canvas():
    # dataset_mapper.py:136-152 with four 300 x 400 images around (xc, yc) = (350, 280) on a 640 x 640 canvas
    imgs = [np.full((300, 400, 3), 10 * (i + 1), np.uint8) for i in range(4)]
    boxes = [np.array([[10.0, 20.0, 390.0, 290.0]]) for _ in range(4)]
    out, bx = A.mosaic_blend(imgs, boxes, 350, 280, 320, 320)
    assert out.shape == (640, 640, 3)
    assert (out[:280, :350] == 10).all() and (out[:280, 350:] == 20).all()
    assert (out[280:580, :350] == 30).all() and (out[280:580, 350:] == 40).all() and (out[580:] == 114).all()
    assert bx.tolist() == [[0.0, 0.0, 340.0, 270.0], [360.0, 0.0, 640.0, 270.0], [0.0, 300.0, 340.0, 570.0],
                           [360.0, 300.0, 640.0, 570.0]]


def test_random_draws_and_box_arithmetic_follow_the_reference_order():
    """The transform classes draw from numpy.random / random exactly as augmentation_impl.py:31-45,146-153 and the
    box path equals the oracle's (hull per transform, clipped)."""
    from yolo_amd.data import (AffineTransform, HFlipTransform, LetterBox, RandomAffine, RandomColorJitter, RandomFlip,
                               ResizeShortestEdge, TransformList)
    np.random.seed(3)
    random.seed(3)
    shape = np.empty((300, 400, 3), np.uint8)
    t = RandomAffine(translate=0.1, scale=0.1, degree=8, shear=10.0, perspective=0.0005).get_transform(shape)
    np.random.seed(3)
    random.seed(3)
    angle = np.random.uniform(-8, 8)
    scale = np.random.uniform(0.9, 1.1)
    tx, ty = np.random.uniform(0.4, 0.6), np.random.uniform(0.4, 0.6)
    sx = np.tan(np.random.uniform(-10, 10) * np.pi / 180)
    sy = np.tan(np.random.uniform(-10, 10) * np.pi / 180)
    px, py = random.uniform(-0.0005, 0.0005), random.uniform(-0.0005, 0.0005)
    assert np.array_equal(t.M, A.affine_matrix(300, 400, angle, scale, (tx, ty), (sx, sy), (px, py)))

    np.random.seed(4)
    j = RandomColorJitter("RGB", 0.1, 1.5, 1.5).get_transform(shape)
    np.random.seed(4)
    h, s, e = np.random.uniform(-1, 1, 3) * [0.1, 1.5, 1.5] + 1
    assert (j.hue, j.saturation, j.exposure, j.rgb) == (h, s, e, True)

    assert ResizeShortestEdge.get_output_shape(480, 640, 416, 416) == A.shortest_edge_size(480, 640, 416, 416) == (312, 416)
    assert ResizeShortestEdge.get_output_shape(375, 500, 608, 1333) == A.shortest_edge_size(375, 500, 608, 1333) == (608, 811)

    np.random.seed(6)
    lb = LetterBox(320, 320).get_transform(shape)
    aff = AffineTransform(320, 320, angle=6, scale=1.05, translate=(0.52, 0.47), shear=(0.04, 0.02))
    flip = HFlipTransform(320)
    boxes = np.array([[10.0, 20.0, 200.0, 250.0], [300.0, 100.0, 400.0, 300.0], [0.0, 0.0, 5.0, 5.0]])
    got = TransformList([lb, aff, flip]).apply_box(boxes.copy())
    want = A.transform_boxes(boxes, [lambda c: A.letterbox_coords(c, lb.ratio, lb.top, lb.left),
                                     lambda c: A.affine_coords(c, aff.M, 320, 320, False),
                                     lambda c: A.flip_coords(c, 0, 320, True)], (10 ** 6, 10 ** 6))
    assert np.array_equal(got, want)
    assert isinstance(RandomFlip(prob=0.0).get_transform(shape).apply_coords(boxes[:, :2]), np.ndarray)


def test_pixel_paths_refuse_cpu_tensors():
    import torch
    from yolo_amd.data import HFlipTransform
    with pytest.raises(RuntimeError, match="GPU"):
        HFlipTransform(8).apply_image(torch.zeros((8, 8, 3), dtype=torch.uint8))

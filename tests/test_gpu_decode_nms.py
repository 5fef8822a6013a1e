"""GPU parity of yolo_decode / yolo_batched_nms / yolo_gather_detections against the CPU oracle
(oracle/yolo_ref.py restating yolo/modeling/meta_arch/yolo.py:266-383 and box_regression.py:44-64;
oracle/nms_ref.c restating torchvision's batched_nms as called at yolo.py:305).

Decode is floating point: candidates must agree except where a score sits within 1e-5 of the
threshold; boxes/scores within 1e-4 relative (torch CPU vs GPU exp differ by ulps).
NMS is integer output: kept indices bit-exact given identical fp32 boxes and scores.
"""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _heads_to_views(heads, dtype, cuda, ops, ch_pad=256):
    views = []
    for h in heads:
        N, AK, gh, gw = h.shape
        v = ops.alloc_view(N, gh, gw, max(ch_pad, (AK + 7) // 8 * 8), 0, dtype, cuda)
        v.interior()[..., :AK].copy_(h.permute(0, 2, 3, 1).to(cuda))
        views.append(ops.View(v.buf, gh, gw, 0, AK, 0))
    return views


def _check_decode(cuda, N, C, grids, thr, dtype, seed, obj_bias=-2.0, ch_pad=256):
    from oracle import yolo_ref as R
    from yolo_amd import ops
    heads = util.random_heads(N, C, grids, seed, obj_bias)
    if dtype == torch.bfloat16:
        heads = [h.bfloat16().float() for h in heads]
    ref = R.decode_batch(heads, C, thr)
    views = _heads_to_views(heads, dtype, cuda, ops, ch_pad)
    cap = sum(r[0].shape[0] for r in ref) + 4096
    anchors = [[tuple(map(float, R.cell_anchors()[l][a, 2:] * 2)) for a in range(3)] for l in range(3)]
    cand = ops.decode(views, R.STRIDES, anchors, C, thr, cap)
    torch.cuda.synchronize()
    offs = cand.img_offsets.cpu().tolist()
    assert offs[0] == 0
    total_mismatch = 0
    for n in range(N):
        b, s, c = ref[n]
        gb = cand.boxes[offs[n]:offs[n + 1]].cpu()
        gs = cand.scores[offs[n]:offs[n + 1]].cpu()
        gc = cand.classes[offs[n]:offs[n + 1]].cpu().long()
        if gb.shape[0] == b.shape[0]:
            assert torch.equal(gc, c), "candidate order / classes differ"
            assert torch.allclose(gs, s, rtol=1e-4, atol=1e-6)
            fin = torch.isfinite(b).all(1)
            assert torch.allclose(gb[fin], b[fin], rtol=1e-4, atol=1e-3)
        else:
            # only scores within 1e-5 of the threshold may flip
            near = int(((torch.cat([s, gs]) - thr).abs() < 1e-5).sum())
            assert abs(gb.shape[0] - b.shape[0]) <= near
            total_mismatch += 1
    assert total_mismatch <= max(1, N // 4)
    return cand, ref


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("thr", [0.25, 0.001, 0.5])
def test_decode_416(cuda, dtype, thr):
    _check_decode(cuda, 2, 80, [(52, 52), (26, 26), (13, 13)], thr, dtype, seed=11)


def test_decode_voc20_rect(cuda):
    """configs/yolov3.yaml: 20 classes (75 head channels), non-square input."""
    _check_decode(cuda, 3, 20, [(40, 52), (20, 26), (10, 13)], 0.25, torch.float32, seed=12, ch_pad=80)


def test_decode_empty_and_overflow(cuda):
    from oracle import yolo_ref as R
    from yolo_amd import ops
    heads = util.random_heads(2, 80, [(8, 8), (4, 4), (2, 2)], 13, obj_bias=-50.0)
    views = _heads_to_views(heads, torch.float32, cuda, ops)
    anchors = [[tuple(map(float, R.cell_anchors()[l][a, 2:] * 2)) for a in range(3)] for l in range(3)]
    cand = ops.decode(views, R.STRIDES, anchors, 80, 0.25, 16)
    assert cand.img_offsets.cpu().tolist() == [0, 0, 0]
    # overflow: true total reported, nothing written past the capacity
    heads = util.random_heads(2, 80, [(8, 8), (4, 4), (2, 2)], 13, obj_bias=3.0)
    ref = R.decode_batch(heads, 80, 0.25)
    views = _heads_to_views(heads, torch.float32, cuda, ops)
    cap = 100
    out = ops.Candidates(torch.full((cap + 8, 4), -1.0, device=cuda), torch.full((cap + 8,), -1.0, device=cuda),
                         torch.full((cap + 8,), -1, dtype=torch.int32, device=cuda),
                         torch.zeros(3, dtype=torch.int32, device=cuda))
    ops.decode(views, R.STRIDES, anchors, 80, 0.25, cap, out=out)
    total = int(out.img_offsets[2])
    assert total == ref[0][0].shape[0] + ref[1][0].shape[0] and total > cap
    assert bool((out.scores[cap:] == -1.0).all()) and bool((out.scores[:cap] > 0.25).all())


def _run_nms(cuda, per_image, iou_thr, mode, top_k=0, num_classes=80, auto_limit=20000):
    from yolo_amd import ops
    boxes = torch.cat([p[0] for p in per_image]).to(cuda)
    scores = torch.cat([p[1] for p in per_image]).to(cuda)
    classes = torch.cat([p[2] for p in per_image]).to(torch.int32).to(cuda)
    counts = [p[0].shape[0] for p in per_image]
    offs = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=cuda)
    M = sum(counts)
    keep, n_keep = ops.batched_nms(boxes, scores, classes, offs, M, iou_thr, num_classes=num_classes, mode=mode,
                                   top_k=top_k, auto_limit=auto_limit, keep_stride=max(counts + [1]))
    torch.cuda.synchronize()
    return [keep[i, :int(n_keep[i])].cpu().long() for i in range(len(per_image))]


def _oracle(p, thr, mode, auto_limit=20000):
    from oracle import nms_c
    from yolo_amd import ops
    b, s, c = p
    if b.shape[0] == 0:
        return torch.empty(0, dtype=torch.int64)
    trick = mode == ops.NMS_COORD_TRICK or (mode == ops.NMS_AUTO and b.numel() <= auto_limit)
    return nms_c.batched_nms_trick(b, s, c, thr) if trick else nms_c.batched_nms_vanilla(b, s, c, thr)


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["per_class", "coord_trick", "auto"])
def test_nms_random_bit_exact(cuda, mode):
    per_image = [util.stress_boxes(n, 100 + i, num_classes=20) for i, n in enumerate([2000, 1, 0, 777, 5003, 64, 65])]
    got = _run_nms(cuda, per_image, 0.5, mode, num_classes=20)
    for p, g in zip(per_image, got):
        assert torch.equal(g, _oracle(p, 0.5, mode))


def test_nms_stress_608(cuda):
    """NMS stress configuration: 22743 candidates / image, 80 classes, IoU 0.5 (4 images here)."""
    per_image = [util.stress_boxes(22743, 100 + i) for i in range(4)]
    got = _run_nms(cuda, per_image, 0.5, 0)
    for p, g in zip(per_image, got):
        assert torch.equal(g, _oracle(p, 0.5, 0))
    top = _run_nms(cuda, per_image, 0.5, 0, top_k=100)
    for g, t in zip(got, top):
        assert torch.equal(g[:100], t)


def test_nms_stress_608_batch_32(cuda):
    """The NMS stress configuration at its full size: 32 images x 22743 candidates, 80 classes, IoU 0.5, every image
    against the oracle (all kept indices), and the top-100 form the model uses."""
    per_image = [util.stress_boxes(22743, 100 + i) for i in range(32)]
    got = _run_nms(cuda, per_image, 0.5, 0)
    top = _run_nms(cuda, per_image, 0.5, 0, top_k=100)
    for p, g, t in zip(per_image, got, top):
        assert torch.equal(g, _oracle(p, 0.5, 0))
        assert torch.equal(g[:100], t)


def test_nms_counts_come_from_the_device(cuda):
    """The host passes only an upper bound of the candidate count: arrays larger than the candidates they hold (the
    tail uninitialised), and offsets beyond the capacity (a decode that overflowed) clamped to it."""
    from yolo_amd import ops
    per_image = [util.stress_boxes(n, 300 + i, num_classes=20) for i, n in enumerate([700, 0, 1300, 9000])]
    want = [_oracle(p, 0.5, 0) for p in per_image]
    boxes = torch.cat([p[0] for p in per_image])
    scores = torch.cat([p[1] for p in per_image])
    classes = torch.cat([p[2] for p in per_image]).to(torch.int32)
    M = boxes.shape[0]
    cap = M + 5000
    pad = lambda t, v: torch.cat([t, torch.full((cap - M,) + tuple(t.shape[1:]), v, dtype=t.dtype)]).to(cuda)
    offs = torch.tensor([0, 700, 700, 2000, 11000], dtype=torch.int32, device=cuda)
    keep, n_keep = ops.batched_nms(pad(boxes, float("nan")), pad(scores, 9.0), pad(classes, 3), offs, cap, 0.5,
                                   num_classes=20, keep_stride=9000)
    torch.cuda.synchronize()
    for i, w in enumerate(want):
        assert torch.equal(keep[i, :int(n_keep[i])].cpu().long(), w)
    # overflow: the last image claims 9000 candidates but only the first 4000 of them exist
    cap2 = 2000 + 4000
    keep, n_keep = ops.batched_nms(boxes[:cap2].to(cuda), scores[:cap2].to(cuda), classes[:cap2].to(cuda), offs, cap2, 0.5,
                                   num_classes=20, keep_stride=9000)
    torch.cuda.synchronize()
    b, s, c = per_image[3]
    assert torch.equal(keep[3, :int(n_keep[3])].cpu().long(), _oracle((b[:4000], s[:4000], c[:4000]), 0.5, 0))
    assert torch.equal(keep[2, :int(n_keep[2])].cpu().long(), want[2])


def test_nms_coordinate_trick_with_negative_corners(cuda):
    """torchvision's coordinate trick shifts every class by max_coordinate + 1; a box with negative corners then
    reaches into the square of the class before it and CAN be suppressed by (or suppress) a box of that class when
    the IoU threshold is low.  Reproduced exactly: images where that is possible are swept class-agnostically."""
    from yolo_amd import ops
    g = torch.Generator().manual_seed(77)
    n = 400
    # filler inside [0, 390]^2, three classes
    c = torch.rand(n, 2, generator=g) * 300 + 40
    wh = 10 + torch.rand(n, 2, generator=g) * 60
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 390)
    scores = torch.rand(n, generator=g) * 0.8
    classes = torch.randint(0, 3, (n,), generator=g)
    # the largest coordinate is 400 (class 0, high score); class-1 boxes hanging over the top-left corner land, after the
    # shift by 401, on top of it: [341, 461]^2 against [300, 400]^2 has IoU 0.166
    boxes = torch.cat([boxes, torch.tensor([[300., 300, 400, 400], [-60., -60, 60, 60], [-50., -70, 65, 55]])])
    scores = torch.cat([scores, torch.tensor([0.99, 0.9, 0.85])])
    classes = torch.cat([classes, torch.tensor([0, 1, 1])])
    case = (boxes, scores, classes)
    differs = 0
    for thr in (0.05, 0.1, 0.3):
        got = _run_nms(cuda, [case], thr, ops.NMS_COORD_TRICK, num_classes=3)[0]
        ref = _oracle(case, thr, ops.NMS_COORD_TRICK)
        assert torch.equal(got, ref), thr
        differs += int(not torch.equal(ref, _oracle(case, thr, ops.NMS_PER_CLASS)))
    assert differs > 0, "the case does not exercise cross-class suppression"
    # boxes whose centres are inside the frame and a threshold of 0.25 or more: per-class segments are exact
    inside = boxes + 70.0
    shifted = torch.cat([inside[:, :2] - 45.0, inside[:, 2:]], 1)  # negative corners, centres still positive
    assert float(shifted.min()) < 0 and float((shifted[:, :2] + shifted[:, 2:]).min()) > 0
    got = _run_nms(cuda, [(shifted, scores, classes)], 0.3, ops.NMS_COORD_TRICK, num_classes=3)[0]
    assert torch.equal(got, _oracle((shifted, scores, classes), 0.3, ops.NMS_COORD_TRICK))


def test_nms_clustered_deep_chains(cuda):
    per_image = [util.clustered_boxes(300, 76, 7 + i) for i in range(2)]
    for mode in (0, 1):
        got = _run_nms(cuda, per_image, 0.5, mode)
        for p, g in zip(per_image, got):
            assert torch.equal(g, _oracle(p, 0.5, mode))


def test_nms_top_k_with_short_class_segments(cuda):
    """top_k > 0 with every class segment of an image short enough for one wave to sort it (the form the model's
    inference uses): images are sorted by class only, every segment is put into score order by its sweep wave, and the
    image's first top_k detections come from a merge of the segments' kept lists.  Bit-exact against the oracle's first
    top_k, with: scores quantised to 33 levels (equal keys inside and across segments; torch's order on equal scores is
    the candidate index), boxes spread out (long kept lists) and clustered (short ones: segments run out during the
    merge), top_k of 1 / a few / more than survive, images that do NOT qualify (a class with more than 512
    candidates, at most 512 candidates in all, none) in the same batch, and the coordinate-trick mode."""
    from yolo_amd import ops
    g = torch.Generator().manual_seed(5)

    def quantised(p, levels=32):
        b, s, c = p
        return b, torch.round(s * levels) / levels, c

    def spread(p, f):
        return p[0] * f, p[1], p[2]

    tied = quantised(spread(util.stress_boxes(6000, 11, num_classes=20), 4.0))
    tied_few = quantised(util.stress_boxes(3000, 12, num_classes=80), levels=4)
    plain = util.stress_boxes(9000, 13, num_classes=80)
    long_class = util.stress_boxes(3000, 14, num_classes=2)       # 1500 per class: not sorted by a wave
    small = util.stress_boxes(300, 15, num_classes=20)            # the counting path
    empty = (torch.empty(0, 4), torch.empty(0), torch.empty(0, dtype=torch.int64))
    clustered = util.clustered_boxes(300, 76, 9)
    clustered_small = util.clustered_boxes(150, 40, 10)
    one_per_class = (torch.rand(600, 4, generator=g).cumsum(1) * 50, torch.rand(600, generator=g),
                     torch.arange(600) % 80)
    uneven = util.stress_boxes(5000, 16, num_classes=80)
    uneven = (uneven[0], uneven[1], torch.where(torch.arange(5000) < 500, torch.zeros(5000, dtype=uneven[2].dtype),
                                                uneven[2]))        # class 0 right at the limit and beyond
    batch = [tied, long_class, plain, small, empty, clustered, tied_few, one_per_class, uneven, clustered_small]
    for mode in (ops.NMS_PER_CLASS, ops.NMS_COORD_TRICK):
        want = [_oracle(p, 0.5, mode) for p in batch]
        for top_k in (1, 7, 100, 1000, 20000):
            got = _run_nms(cuda, batch, 0.5, mode, top_k=top_k)
            for i, (w, gk) in enumerate(zip(want, got)):
                assert torch.equal(gk, w[:top_k]), (mode, top_k, i)
    # a lower threshold (more suppression, shorter lists) and a higher one
    for thr in (0.2, 0.8):
        want = [_oracle(p, thr, 0) for p in batch]
        got = _run_nms(cuda, batch, thr, 0, top_k=100)
        for i, (w, gk) in enumerate(zip(want, got)):
            assert torch.equal(gk, w[:100]), (thr, i)


def test_nms_single_class_long_segment(cuda):
    """One class only: a 6000-box segment (kept list spills past the LDS list)."""
    b, s, c = util.stress_boxes(6000, 5, num_classes=1)
    b = b * 8  # spread out so that many boxes survive
    got = _run_nms(cuda, [(b, s, c)], 0.5, 0, num_classes=1)
    ref = _oracle((b, s, c), 0.5, 0)
    assert ref.numel() > 1000
    assert torch.equal(got[0], ref)


def test_nms_edge_cases(cuda):
    f = torch.tensor
    # exact duplicates, equal scores (stable order), IoU exactly at the threshold, zero-area boxes
    dup = (f([[0., 0, 10, 10]] * 4), f([0.9, 0.9, 0.9, 0.9]), f([1, 1, 2, 1]))
    thr_edge = (f([[0., 0, 4, 4], [0, 0, 4, 2], [0, 0, 2, 2]]), f([0.9, 0.8, 0.7]), f([0, 0, 0]))  # IoU 0.5, 0.25
    zero = (f([[5., 5, 5, 5], [5, 5, 5, 5], [0, 0, 1, 1]]), f([0.5, 0.6, 0.7]), f([0, 0, 0]))
    inf = (f([[-float("inf"), 0, float("inf"), 10], [0., 0, 10, 10], [1, 1, 9, 9]]), f([0.9, 0.8, 0.7]), f([3, 3, 3]))
    for case in (dup, thr_edge, zero, inf):
        case = (case[0].float(), case[1].float(), case[2])
        got = _run_nms(cuda, [case], 0.5, 0, num_classes=4)[0]
        assert torch.equal(got, _oracle(case, 0.5, 0)), case
    # hand-derived: IoU exactly 0.5 is NOT suppressed by strict '>'
    got = _run_nms(cuda, [(thr_edge[0].float(), thr_edge[1].float(), thr_edge[2])], 0.5, 0, num_classes=4)[0]
    assert got.tolist() == [0, 1, 2]
    got = _run_nms(cuda, [(dup[0].float(), dup[1].float(), dup[2])], 0.5, 0, num_classes=4)[0]
    assert got.tolist() == [0, 2]


def test_gather_postprocess(cuda):
    from oracle import yolo_ref as R
    from yolo_amd import ops
    per_image = [util.stress_boxes(500, 40 + i, side=416.0) for i in range(3)]
    boxes = torch.cat([p[0] for p in per_image]).to(cuda)
    scores = torch.cat([p[1] for p in per_image]).to(cuda)
    classes = torch.cat([p[2] for p in per_image]).to(torch.int32).to(cuda)
    offs = torch.tensor([0, 500, 1000, 1500], dtype=torch.int32, device=cuda)
    keep, n_keep = ops.batched_nms(boxes, scores, classes, offs, 1500, 0.5, num_classes=80, top_k=100)
    cand = ops.Candidates(boxes, scores, classes, offs)
    in_hw = torch.tensor([[416, 416], [384, 416], [416, 320]], dtype=torch.int32, device=cuda)
    out_hw = torch.tensor([[480, 640], [768, 832], [100, 77]], dtype=torch.int32, device=cuda)
    det, n_det = ops.gather_detections(cand, keep, n_keep, in_hw, out_hw)
    torch.cuda.synchronize()
    for i, p in enumerate(per_image):
        k = keep[i, :int(n_keep[i])].cpu().long()
        rb, rs, rc = R.detector_postprocess(p[0][k], p[1][k], p[2][k], tuple(in_hw[i].tolist()), *out_hw[i].tolist())
        d = det[i, :int(n_det[i])].cpu()
        assert d.shape[0] == rb.shape[0]
        assert torch.allclose(d[:, :4], rb, rtol=1e-6, atol=1e-4)
        assert torch.equal(d[:, 4], rs) and torch.equal(d[:, 5].long(), rc)

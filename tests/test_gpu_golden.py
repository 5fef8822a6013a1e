"""The HIP path against the COMMITTED golden vectors (tests/golden/oracle_golden.npz), not against a live run of the
oracle: decode (reference yolo/modeling/meta_arch/yolo.py:337-383 + box_regression.py:44-64), batched NMS
(yolo.py:305) and the reduced-depth backbone + neck + heads (darknet.py:113-126, fpn.py:165-185).  The other GPU
tests regenerate their expectation from the oracle each time; here a drift of the oracle and of a kernel in the same
direction would show, because the expectation is a file in the repository.  (The file itself was produced by the oracle,
tests/golden/make_golden.py: the reference cannot be imported in the build container.)
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _anchors(R):
    return [[tuple(map(float, R.cell_anchors()[l][a, 2:] * 2)) for a in range(3)] for l in range(3)]


@pytest.mark.parametrize("thr,tag", [(0.25, "a"), (0.001, "b")])
def test_decode_on_the_golden_heads(cuda, gold, thr, tag):
    from oracle import yolo_ref as R   # anchors and strides only: the expectation is the committed file
    from yolo_amd import ops
    heads = [torch.from_numpy(gold[f"dec_head{i}"]) for i in range(3)]
    views = []
    for h in heads:
        N, AK, gh, gw = h.shape
        v = ops.alloc_view(N, gh, gw, 256, 0, torch.float32, cuda)
        v.interior()[..., :AK].copy_(h.permute(0, 2, 3, 1).to(cuda))
        views.append(ops.View(v.buf, gh, gw, 0, AK, 0))
    want_b, want_s, want_c = (gold[f"dec_{tag}_{k}"] for k in ("boxes", "scores", "classes"))
    cand = ops.decode(views, R.STRIDES, _anchors(R), 80, thr, want_s.shape[0] + 1024)
    torch.cuda.synchronize()
    offs = cand.img_offsets.cpu().tolist()
    n = offs[1]
    got_s = cand.scores[:n].cpu().numpy()
    if n != want_s.shape[0]:   # only scores within 1e-5 of the threshold may flip
        near = int((np.abs(np.concatenate([want_s, got_s]) - thr) < 1e-5).sum())
        assert abs(n - want_s.shape[0]) <= near
        return
    assert np.array_equal(cand.classes[:n].cpu().numpy().astype(np.int64), want_c)   # order: level, y, x, anchor, class
    assert np.allclose(got_s, want_s, rtol=1e-4, atol=1e-6)
    fin = np.isfinite(want_b).all(1)
    assert np.allclose(cand.boxes[:n].cpu().numpy()[fin], want_b[fin], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("mode,key", [(0, "nms_keep_vanilla"), (1, "nms_keep_trick")], ids=["per_class", "coord_trick"])
def test_nms_on_the_golden_boxes_is_bit_exact(cuda, gold, mode, key):
    from yolo_amd import ops
    b = torch.from_numpy(gold["nms_boxes"]).to(cuda)
    s = torch.from_numpy(gold["nms_scores"]).to(cuda)
    c = torch.from_numpy(gold["nms_classes"]).to(torch.int32).to(cuda)
    M = b.shape[0]
    offs = torch.tensor([0, M], dtype=torch.int32, device=cuda)
    keep, n_keep = ops.batched_nms(b, s, c, offs, M, 0.5, num_classes=20, mode=mode, top_k=0, keep_stride=M)
    torch.cuda.synchronize()
    got = keep[0, :int(n_keep[0])].cpu().long().numpy()
    assert np.array_equal(got, gold[key])


def test_reduced_network_heads_match_the_golden_tensors(cuda, gold):
    """Backbone (one block per stage) + neck + heads, fp32 plan, on the committed input: p3 / p4 / p5 within 1e-3 of
    the committed tensors.  The weights are the oracle's seeded initialisation (seed 7, as make_golden.py)."""
    from oracle import yolo_ref as R
    from yolo_amd.modeling.backbone.darknet import BasicStem, DarkNet, ResidualBlock
    from yolo_amd.modeling.backbone.fpn import DarkNetFPN
    sd = R.init_state_dict(7, num_classes=20, blocks=(1, 1, 1, 1, 1))
    R.randomize_norm_stats(sd, 8)
    stem = BasicStem(3, 32, norm="BN")
    stages, cin, cout = [], 32, 64
    for _ in range(5):
        stages.append(DarkNet.make_stage(ResidualBlock, 1, in_channels=cin, out_channels=cout, norm="BN"))
        cin, cout = cout, cout * 2
    bottom_up = DarkNet(stem, stages, out_features=["res3", "res4", "res5"])
    fpn = DarkNetFPN(bottom_up, ["res3", "res4", "res5"], 75, norm="BN", num_classes=20, anchors_wh=_anchors(R))
    missing, unexpected = fpn.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")},
                                              strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    fpn.to(cuda).eval()
    fpn.compute_dtype = torch.float32
    x = torch.from_numpy(gold["net_input"]).to(cuda)
    out = fpn(x)
    torch.cuda.synchronize()
    for k in ("p3", "p4", "p5"):
        ref = gold[f"net_{k}"]
        got = out[k].float().cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), k

"""detect_graphed(): a captured HIP graph of one detect() call returns exactly what detect() returns."""
import pytest
import torch

import workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def calibrated(cuda):
    model, cfg = workload.calibrated_model(size=256, calib_batch=4, config="yolov3.yaml", dtype="bf16", head_dtype="f32")
    return model.eval(), cfg


def _same(a, b):
    (d0, n0), (d1, n1) = a, b
    assert torch.equal(n0, n1)
    total = 0
    for i in range(d0.shape[0]):
        k = int(n0[i])
        total += k
        assert torch.equal(d0[i, :k], d1[i, :k])
    return total


@pytest.mark.parametrize("batch", [1, 4])
def test_graphed_detect_equals_detect(cuda, calibrated, batch):
    """The first calls of a signature run detect() itself; from the third on the call is one graph launch.  Replays
    with new inputs return what detect() returns for those inputs (bit for bit: same kernels, same buffers), a second
    signature gets a graph of its own, and the first one still replays afterwards."""
    model, cfg = calibrated
    eng = model.backbone.engine()
    kw = dict(score_thr=0.02, nms_thr=0.5, mean=cfg.MODEL.PIXEL_MEAN, std=cfg.MODEL.PIXEL_STD)
    imgs = [workload.synthetic_images(batch, 160, 192, seed=70 + i, device="cuda") for i in range(6)]
    want = []
    for x in imgs:
        d, n = eng.detect(x, **kw)
        want.append((d.clone(), n.clone()))
    eng._graphs.clear()
    eng._graph_seen.clear()
    total = 0
    for i, x in enumerate(imgs):
        d, n = eng.detect_graphed(x, **kw)
        assert (len(eng._graphs) == 1) == (i + 1 >= eng.GRAPH_AFTER)
        total += _same(want[i], (d.clone(), n.clone()))
    assert total > 0
    other = [workload.synthetic_images(batch, 128, 128, seed=90 + i, device="cuda") for i in range(4)]
    for x in other:
        got = eng.detect_graphed(x, **kw)
        got = (got[0].clone(), got[1].clone())
        d, n = eng.detect(x, **kw)
        _same((d, n), got)
    assert len(eng._graphs) == 2
    d, n = eng.detect_graphed(imgs[1], **kw)
    _same(want[1], (d, n))


def test_model_calls_replay_a_graph_and_follow_the_weights(cuda, calibrated):
    """model(batched_inputs) with one image: same Instances with and without the graph; after a weight change the
    engine (and with it every captured graph) is rebuilt and the results follow the new weights."""
    model, cfg = calibrated
    img = workload.synthetic_images(1, 200, 256, seed=5)[0]
    inp = [dict(image=img, height=200, width=256)]

    def run():
        with torch.no_grad():
            r = model(inp)[0]["instances"]
        return r.pred_boxes.tensor.clone(), r.scores.clone(), r.pred_classes.clone()

    old_thr = model.test_score_thresh
    model.test_score_thresh = 0.02
    try:
        model.graph_max_batch = 0
        ref = run()
        model.graph_max_batch = 8
        for _ in range(5):
            got = run()
            assert all(torch.equal(a, b) for a, b in zip(ref, got))
        eng = model.backbone.engine()
        assert sum(1 for k in eng._graphs if k[0] == (1, 3, 200, 256)) == 1
        head = [m for n_, m in model.backbone.named_modules() if n_.endswith("fpn_output0.head")][0]
        with torch.no_grad():
            head.bias.add_(0.5)
        model.graph_max_batch = 0
        ref2 = run()
        model.graph_max_batch = 8
        assert model.backbone.engine() is not eng
        for _ in range(4):
            got = run()
            assert all(torch.equal(a, b) for a, b in zip(ref2, got))
        assert ref2[0].shape != ref[0].shape or not torch.equal(ref2[1], ref[1])
        with torch.no_grad():
            head.bias.sub_(0.5)
    finally:
        model.test_score_thresh = old_thr

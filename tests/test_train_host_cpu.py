"""CPU checks of the training path's host logic (torch ops, no kernels): Matcher, label_anchors, get_deltas and the
three losses of yolo_amd's YOLOV3 against the oracle's restatement of the reference
(yolo/modeling/matcher.py:18-55, meta_arch/yolo.py:131-264, box_regression.py:18-42), and known answers."""
import pytest
import torch

import workload
from oracle import yolo_ref as R, yolo_train_ref as T
from yolo_amd.structures import Boxes, Instances


@pytest.fixture(scope="module")
def model_cpu():
    model, cfg = workload.build_model(config="yolov3.yaml", device="cpu", dtype="f32")
    return model, cfg


def _targets(n, size, C, seed):
    gb, gc = T.synthetic_targets(n, size, C, seed=seed, max_boxes=6)
    inst = [Instances((size, size), gt_boxes=Boxes(b), gt_classes=c) for b, c in zip(gb, gc)]
    return gb, gc, inst


def test_matcher_known_answer():
    from yolo_amd.modeling.matcher import Matcher
    anchors = Boxes(torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 10.], [20., 20., 40., 40.], [100., 100., 110., 110.]]))
    anchors.tensor[1] += 1.0  # same shape, centre 1 px further
    gt = Boxes(torch.tensor([[0., 0., 10., 10.], [22., 22., 38., 38.]]))
    from yolo_amd.structures import pairwise_iou
    mq = pairwise_iou(gt, anchors)
    m, lab = Matcher(0.5, [-1, 0, 1])(mq, anchors, gt)
    assert lab.tolist() == [1, -1, 1, 0]      # exact match positive; shifted twin above 0.5 IoU ignored; far one negative
    assert m.tolist()[:3] == [0, 0, 1]
    om, ol = T.matcher(T.pairwise_iou(gt.tensor, anchors.tensor), anchors.tensor, gt.tensor)
    assert torch.equal(om, m) and torch.equal(ol, lab)


def test_label_anchors_and_losses_match_oracle(model_cpu):
    model, cfg = model_cpu
    C = cfg.MODEL.YOLOV3.NUM_CLASSES
    N, S = 3, 96
    gb, gc, inst = _targets(N, S, C, seed=3)
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(N, 3 * (5 + C), S // s, S // s, generator=g) for s in (8, 16, 32)]
    feats_o = [f.clone().requires_grad_(True) for f in feats]
    feats_m = [f.clone().requires_grad_(True) for f in feats]
    # oracle
    lo, co, do = R.head_forward(feats_o, 3, C)
    lo = [R.permute_to_N_HWA_K(t, C) for t in lo]
    co = [R.permute_to_N_HWA_K(t, 1) for t in co]
    do = [R.permute_to_N_HWA_K(t, 4) for t in do]
    anchors = torch.cat(R.grid_anchors([(f.shape[2], f.shape[3]) for f in feats]))
    labels, boxes = T.label_anchors(anchors, gb, gc, C)
    want = T.losses(anchors, lo, co, do, labels, boxes, R.STRIDES, C)
    # product host code
    pred = model.head(feats_m)
    got = model.forward_training(None, feats_m, pred, inst)
    ml, mb = model.label_anchors(model.anchor_generator(feats_m), inst, None)
    for a, b in zip(ml, labels):
        assert torch.equal(a, b)
    for a, b in zip(mb, boxes):
        assert torch.equal(a, b)
    bl, bb = model.label_anchors_batched(model.anchor_generator(feats_m), inst)   # what training_step uses
    for a, b, c, d in zip(bl, labels, bb, boxes):
        assert torch.equal(a, b) and torch.equal(c, d)
    for k in want:
        assert torch.allclose(got[k], want[k], rtol=1e-6, atol=1e-7), k
    sum(want.values()).backward()
    sum(got.values()).backward()
    for a, b in zip(feats_m, feats_o):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-8)
    # every image contributes one positive anchor per ground-truth box (distinct boxes)
    assert int(sum((l >= 0).logical_and(l != C).sum() for l in labels)) <= sum(len(b) for b in gb)


def test_get_deltas_known_answer(model_cpu):
    model, _ = model_cpu
    anchors = [Boxes(torch.tensor([[0., 0., 16., 32.]]))]
    gt = torch.tensor([[[4., 8., 20., 24.]]])  # centre (12,16), wh (16,16)
    d = model.box2box_transform.get_deltas(gt, anchors, torch.tensor([8.]))
    # cx/stride = 1.5 -> frac 0.5 - 0.5 = 0 ; cy/stride = 2.0 -> 0 - 0.5 ; log(16/16), log(16/32)
    assert torch.allclose(d[0, 0], torch.tensor([0.0, -0.5, 0.0, -0.6931472]), atol=1e-6)
    assert torch.equal(d, T.get_deltas(gt, anchors[0].tensor, torch.tensor([8.])))


def test_frozen_parameters_follow_freeze_at(model_cpu):
    model, cfg = model_cpu
    assert cfg.MODEL.BACKBONE.FREEZE_AT == 2
    frozen = {n for n, p in model.backbone.named_parameters() if not p.requires_grad}
    assert all(n.startswith(("bottom_up.stem.", "bottom_up.res1.")) for n in frozen) and frozen
    want = {"backbone." + n for n in frozen if n.endswith("conv.weight") or n.endswith(".weight") and ".norm." not in n}
    names = set(T.frozen_names(2))
    assert {w.rsplit(".weight", 1)[0] for w in want} == names


def test_solver_schedule_and_groups(model_cpu):
    """detectron2 defaults the reference trains with: no weight decay on norm parameters, frozen parameters left
    out, linear warm-up into a multi-step schedule."""
    from yolo_amd.solver import build_lr_scheduler, build_optimizer
    model, cfg = model_cpu
    opt = build_optimizer(cfg, model)
    g_other, g_norm = opt.param_groups
    assert g_other["weight_decay"] == cfg.SOLVER.WEIGHT_DECAY and g_norm["weight_decay"] == 0.0
    n_train = sum(p.requires_grad for p in model.parameters())
    assert len(g_other["params"]) + len(g_norm["params"]) == n_train == 210
    sch = build_lr_scheduler(cfg, opt)
    lrs = []
    for _ in range(1002):
        lrs.append(sch.get_last_lr()[0])
        opt.step()
        sch.step()
    base = cfg.SOLVER.BASE_LR
    assert abs(lrs[0] - base * cfg.SOLVER.WARMUP_FACTOR) < 1e-12
    assert abs(lrs[500] - base * (0.001 * 0.5 + 0.5)) < 1e-9 and abs(lrs[1001] - base) < 1e-12
    sch.last_epoch = cfg.SOLVER.STEPS[0]
    assert abs(sch.get_lr()[0] - base * cfg.SOLVER.GAMMA) < 1e-12


def test_training_raises_on_cpu(model_cpu):
    model, cfg = model_cpu
    model.train()
    img = torch.zeros(3, 64, 64, dtype=torch.uint8)
    inst = Instances((64, 64), gt_boxes=Boxes(torch.tensor([[4., 4., 40., 40.]])), gt_classes=torch.tensor([1]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model([dict(image=img, instances=inst)])
    model.eval()

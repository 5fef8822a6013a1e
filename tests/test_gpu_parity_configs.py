"""Detection-level parity of the benchmarked precision and of the reference's own use case.

* Config 1 (tools/inference.py:23-45 of the reference): ``Predictor(cfg)(image)`` on configs/yolov3.yaml, one image
  per call, weights read from a Darknet ``.weights`` FILE through ``YOLOV3Checkpointer`` (checkpoint.py:13-22,
  load_darknet.py:77-125), against ``oracle.inference`` on the state dict the oracle's own table-driven reader
  returns for the same file, fed the oracle's own preprocessing (Pillow resize, BGR -> RGB).
* bf16 (the benchmarked precision) end to end on configs/yolov3-aug.yaml at 416 x 416 against the fp32 oracle.

Tolerances.  fp32: heads 1e-3 (max-norm relative); the detection lists agree one for one (same count, same class,
IoU > 0.98, score within 2e-3).  bf16: a bf16 run of this randomly initialised network is 0.10-0.18 (max-norm
relative, 0.13-0.25 in relative L2) from the fp32 run at the heads WHOEVER executes it -- the oracle's own bf16
restatement is, and neither an fp32 residual stream nor a small residual-branch gain changes that
(tools/chain_error_variants.py, tools/chain_error_gain.py): box sizes move by ~20 %, so a one-for-one comparison of
bf16 detections with fp32 ones measures the network's sensitivity, not the kernels.  The bf16 gate is therefore split:
  (1) every layer on its own meets 5e-2 in bf16 (tests/test_gpu_e2e.py::test_every_layer_bf16_teacher_forced,
      worst ~4e-3);
  (2) everything after the convolutions is exact at the benchmarked precision: the oracle's decode + NMS + top-100 +
      postprocess applied to the HIP plan's OWN bf16-computed head tensors gives the HIP detections one for one
      (same count, IoU > 0.98, score within 2e-3);
  (3) the chained bf16 run is no further from the fp32 oracle than the oracle's own bf16 restatement is: with a loose
      match (same class, IoU > 0.5, score within 0.1) the HIP bf16 detections recover at least (own - 0.1) of the
      fp32 oracle's detections, ``own`` being what the bf16 restatement recovers.
"""
import numpy as np
import pytest
import torch

import workload

pytestmark = pytest.mark.gpu

MARGIN = 0.02


from oracle.compare import matched_fraction, rows_of  # noqa: E402


def _as_rows(inst):
    return torch.cat([inst.pred_boxes.tensor, inst.scores[:, None], inst.pred_classes[:, None].float()], 1).cpu()


def _ref_rows(r):
    b, s, c = r
    return torch.cat([b, s[:, None], c[:, None].float()], 1)


def _oracle_preprocess(bgr: np.ndarray, short: int, max_size: int, fmt: str) -> torch.Tensor:
    """What Predictor.__call__ does on the host in the reference (tools/inference.py:36-43): optional BGR -> RGB,
    detectron2 ResizeShortestEdge (Pillow bilinear), HWC -> CHW float32."""
    from oracle.compare import resize_shortest_edge
    img = bgr[:, :, ::-1] if fmt == "RGB" else bgr
    out = resize_shortest_edge(img, short, max_size)
    return torch.as_tensor(out.astype("float32").transpose(2, 0, 1))


def _oracle_postprocess(heads, num_classes, thr, nms, sizes, out_sizes):
    """decode -> batched_nms -> top 100 -> detector_postprocess of the oracle on given head tensors [N, A*K, G, G]."""
    from oracle import nms_c, yolo_ref as R
    out = []
    with torch.no_grad():
        for n, (b, s, c) in enumerate(R.decode_batch(list(heads), num_classes, thr, R.DEFAULT_ANCHORS)):
            keep = R.batched_nms(b, s, c, nms, "gpu", nms_c.nms)[:100]
            out.append(_ref_rows(R.detector_postprocess(b[keep], s[keep], c[keep], sizes[n], *out_sizes[n])))
    return out


@pytest.fixture(scope="module")
def voc_weights_file(cuda, tmp_path_factory):
    """A calibrated 20-class model's parameters as a Darknet file written by the oracle's table-driven writer."""
    from oracle import darknet_weights_ref as W
    model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3.yaml", dtype="f32")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    path = str(tmp_path_factory.mktemp("w") / "voc.weights")
    W.write_weights(path, sd, num_classes=20)
    return path


def _predictor(path, dtype, thr, nms):
    import os
    from tools.inference import Predictor
    from yolo_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(workload.CONFIGS, "yolov3.yaml"))
    cfg.merge_from_list(["MODEL.WEIGHTS", path, "INPUT.MIN_SIZE_TEST", 416, "MODEL.DEVICE", "cuda",
                         "MODEL.HIP.DTYPE", dtype, "MODEL.HIP.HEAD_DTYPE", "f32",
                         "MODEL.YOLOV3.TEST_SCORE_THRESH", thr, "MODEL.YOLOV3.TEST_NMS_THRESH", nms])
    cfg.freeze()
    torch.manual_seed(123)  # the constructor's random init must be overwritten by the file
    return Predictor(cfg), cfg


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_config1_predictor_on_a_weights_file(cuda, voc_weights_file, dtype):
    """configs/yolov3.yaml, 416, batch 1, tools/inference.py::Predictor.__call__ end to end."""
    from oracle import darknet_weights_ref as W, nms_c, yolo_ref as R
    thr, nms = 0.25, 0.5
    pred, cfg = _predictor(voc_weights_file, dtype, thr, nms)
    sd = W.read_weights(voc_weights_file, num_classes=20)
    assert int(sd.pop("__trailing_floats__")) == 0
    own = pred.model.state_dict()
    for k, v in sd.items():  # the file reached the model unchanged
        assert torch.equal(own[k].cpu(), v), k
    # image 1: the baseline's 416 x 416 synthetic image (the resize is the identity); image 2: 480 x 640, resized to
    # 416 x 555 and padded to 416 x 576 inside the stem kernel, boxes scaled back to 480 x 640
    g = torch.Generator().manual_seed(0)
    images = [workload.synthetic_images(1, 416, 416, seed=77)[0].permute(1, 2, 0).numpy()[:, :, ::-1].copy(),
              workload.synthetic_images(1, 480, 640, seed=78)[0].permute(1, 2, 0).numpy()[:, :, ::-1].copy()]
    total = 0
    for bgr in images:
        h, w = bgr.shape[:2]
        out = pred(bgr)
        inst = out["instances"]
        assert inst.image_size == (h, w)
        got = _as_rows(inst)
        x = _oracle_preprocess(bgr, 416, cfg.INPUT.MAX_SIZE_TEST, cfg.INPUT.FORMAT)
        with torch.no_grad():
            ref = _ref_rows(R.inference([x], sd, 20, thr, nms, out_sizes=[(h, w)], nms_fn=nms_c.nms, device_kind="gpu")[0])
        total += ref.shape[0]
        if dtype == "f32":
            assert got.shape[0] == ref.shape[0], (got.shape, ref.shape)
            f, _ = matched_fraction(ref, got, 0.0, 0.98, 2e-3)
            assert f == 1.0, f
        else:
            # (2): the oracle's post-processing on the plan's own head tensors
            hw = tuple(int(v) for v in x.shape[-2:])
            H, Wd = (hw[0] + 31) // 32 * 32, (hw[1] + 31) // 32 * 32
            plan = pred.model.backbone.engine().plan(1, H, Wd)
            own_ref = _oracle_postprocess([h.nchw().float().cpu() for h in plan.heads], 20, thr, nms, [hw], [(h, w)])[0]
            assert got.shape[0] == own_ref.shape[0] > 0, (got.shape, own_ref.shape)
            assert matched_fraction(own_ref, got, 0.0, 0.98, 2e-3)[0] == 1.0
    assert total >= 20, "too few detections for the comparison to mean anything"


def test_config1_heads_fp32_from_the_weights_file(cuda, voc_weights_file):
    """fp32 plan on the file's parameters: head tensors within 1e-3 of the oracle run on the oracle-read state dict."""
    from oracle import darknet_weights_ref as W, yolo_ref as R
    pred, cfg = _predictor(voc_weights_file, "f32", 0.25, 0.5)
    sd = W.read_weights(voc_weights_file, num_classes=20)
    sd.pop("__trailing_floats__")
    x = workload.synthetic_images(1, 416, 416, seed=77).float() / 255.0
    with torch.no_grad():
        ref = R.fpn_forward(x, sd)
    out = pred.model.backbone(x.to(cuda))
    for k in ("p3", "p4", "p5"):
        assert out[k].shape == ref[k].shape == (1, 75, ref[k].shape[2], ref[k].shape[3])
        err = float((out[k].float().cpu() - ref[k]).abs().max() / ref[k].abs().max())
        assert err < 1e-3, (k, err)


def test_detector_end_to_end_bf16(cuda):
    """configs/yolov3-aug.yaml (80 classes), 416 x 416, bf16 with fp32 heads -- the benchmarked configuration --
    through model(batched_inputs): gates (2) and (3) of this file's header."""
    from oracle import nms_c, yolo_ref as R
    model, cfg = workload.calibrated_model(size=416, calib_batch=4, config="yolov3-aug.yaml", dtype="bf16", head_dtype="f32")
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    thr, nms = cfg.MODEL.YOLOV3.TEST_SCORE_THRESH, cfg.MODEL.YOLOV3.TEST_NMS_THRESH
    imgs = workload.synthetic_images(4, 416, 416, seed=31)
    res = model([dict(image=i) for i in imgs])
    got = [_as_rows(r["instances"]) for r in res]
    plan = model.backbone.engine().plan(4, 416, 416)
    sizes = [(416, 416)] * 4
    own_heads = _oracle_postprocess([h.nchw().float().cpu() for h in plan.heads], 80, thr, nms, sizes, sizes)
    for g, o in zip(got, own_heads):  # (2)
        assert g.shape[0] == o.shape[0] > 0
        assert matched_fraction(o, g, 0.0, 0.98, 2e-3)[0] == 1.0
    x, _ = R.preprocess_images([i.float() for i in imgs])
    with torch.no_grad():
        h32 = R.fpn_forward(x, sd)
        hq = R.fpn_forward(x, sd, quant=True)
    ref32 = _oracle_postprocess([h32[k] for k in ("p3", "p4", "p5")], 80, thr, nms, sizes, sizes)
    refq = _oracle_postprocess([hq[k] for k in ("p3", "p4", "p5")], 80, thr, nms, sizes, sizes)
    mean = lambda v: sum(v) / len(v)
    ours = mean([matched_fraction(a, g, thr + MARGIN, 0.5, 0.1)[0] for a, g in zip(ref32, got)])
    own = mean([matched_fraction(a, q, thr + MARGIN, 0.5, 0.1)[0] for a, q in zip(ref32, refq)])
    n_ref = sum(int((a[:, 4] > thr + MARGIN).sum()) for a in ref32)
    assert n_ref >= 100, n_ref
    print(f"bf16 detections recovering the fp32 oracle's (IoU > 0.5, score within 0.1): HIP {ours:.3f}, "
          f"the oracle's own bf16 restatement {own:.3f} ({n_ref} detections)")
    assert ours >= own - 0.1, (ours, own)  # (3)


def test_chained_bf16_meets_5e_2_on_trained_weights(cuda):
    """north_star's bf16 tolerance (conv / head outputs within 5e-2 of the fp32 reference) on the WHOLE chained network,
    in the regime the tolerance is meant for: trained weights.  The network is first fitted to the structured synthetic
    set (tools/overfit_check.py: 400 bf16 SGD steps through the HIP training path), then its bf16 inference plan is
    compared with the fp32 oracle on the trained state dict: every stage output and head tensor within 5e-2 max-norm
    (measured 0.3e-2 .. 1.7e-2, the same as the oracle's own bf16 restatement -- against 0.10 .. 0.18 on the randomly
    initialised network, where each stage doubles a perturbation).  And at detection level: at
    least 95 % of the fp32 oracle's detections above thr + 0.02 have a bf16 twin of the same class at
    IoU > 0.9 with a score within 0.02."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import overfit_check
    from oracle import nms_c, yolo_ref as R
    res, model, data = overfit_check.run(steps=400, n_images=32, side=256, dtype="bf16", log=lambda m: None,
                                         return_model=True)
    assert res["recall_after"] >= 0.8, res
    model.eval()
    cfg_thr, cfg_nms = model.test_score_thresh, model.test_nms_thresh
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    imgs = torch.stack([d["image"] for d in data[:4]]).cpu()
    x = imgs.float() / 255.0
    taps32 = {}
    with torch.no_grad():
        ref32 = R.fpn_forward(x, sd, taps=taps32)
    model.backbone.compute_dtype = torch.bfloat16
    out = model.backbone(x.to(cuda))
    torch.cuda.synchronize()
    plan = model.backbone.engine().plan(4, 256, 256)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())  # noqa: E731
    worst = {}
    for name in ("res3", "res4", "res5"):
        worst[name] = rel(plan.taps[name].nchw().float().cpu(), taps32[name])
    for name in ("p3", "p4", "p5"):
        worst[name] = rel(out[name].float().cpu(), ref32[name])
    print("bf16 plan against the fp32 oracle on trained weights:", {k: round(v, 4) for k, v in worst.items()})
    assert max(worst.values()) < 5e-2, worst
    # the same at the benchmark's input size (the network is fully convolutional): fresh 416 x 416 images of the same
    # kind, heads within 5e-2 of the fp32 oracle
    x416 = torch.stack([d["image"] for d in overfit_check.make_set(2, 416, model.num_classes, 6, "cpu")]).float() / 255.0
    with torch.no_grad():
        ref416 = R.fpn_forward(x416, sd)
    out416 = model.backbone(x416.to(cuda))
    torch.cuda.synchronize()
    worst416 = {name: rel(out416[name].float().cpu(), ref416[name]) for name in ("p3", "p4", "p5")}
    print("the same at 416 x 416:", {k: round(v, 4) for k, v in worst416.items()})
    assert max(worst416.values()) < 5e-2, worst416
    # detections: bf16 HIP path against the fp32 oracle's inference(), on 16 images -- some 55 reference detections, so
    # that the 95 % is a rate and not "none of eleven" (a detection whose score moves by more than 0.02 between bf16
    # and fp32 turns up about once in fifty on a network trained this briefly, whichever way it was trained)
    imgs = torch.stack([d["image"] for d in data[:16]]).cpu()
    got = [_as_rows(r["instances"]) for r in model([dict(image=i) for i in imgs])]
    with torch.no_grad():
        want = R.inference([i.float() for i in imgs], sd, model.num_classes, cfg_thr, cfg_nms, nms_fn=nms_c.nms,
                           device_kind="gpu")
    n_ref, hit = 0, 0.0
    for g, (rb, rs, rc) in zip(got, want):
        w = rows_of(rb, rs, rc)
        f, n = matched_fraction(w, g, cfg_thr + 0.02, 0.9, 0.02)
        hit += f * n
        n_ref += n
    assert n_ref >= 30, n_ref
    assert hit / n_ref >= 0.95, (hit, n_ref)

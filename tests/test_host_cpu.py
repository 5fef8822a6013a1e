"""CPU tests of the host side: configuration surface, registries, state-dict schema, checkpoint formats,
containers, the C-ABI library's exports and argument validation (no compute without a GPU)."""
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

import yolo_amd  # noqa: F401
from yolo_amd import _lib
from yolo_amd.config import get_cfg
from yolo_amd.modeling import (ANCHOR_GENERATOR_REGISTRY, BACKBONE_REGISTRY, META_ARCH_REGISTRY, build_model)
from yolo_amd.structures import Boxes, Instances

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(name="yolov3.yaml", opts=()):
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", name))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.WEIGHTS", ""] + list(opts))
    return cfg


# ---------------------------------------------------------------- configuration
def test_yaml_configs_merge_and_types():
    voc = _cfg("yolov3.yaml")
    assert voc.MODEL.YOLOV3.NUM_CLASSES == 20 and voc.MODEL.META_ARCHITECTURE == "YOLOV3"
    assert voc.DATASETS.TRAIN == ("voc_2012_train",) and voc.SOLVER.STEPS == (15000, 18000)
    assert voc.INPUT.MIN_SIZE_TRAIN == (320, 352, 384, 416, 448, 480, 512, 544, 576, 608)
    assert voc.MODEL.PIXEL_STD == [255.0, 255.0, 255.0] and voc.INPUT.FORMAT == "RGB"
    coco = _cfg("yolov3-aug.yaml")
    assert coco.MODEL.YOLOV3.NUM_CLASSES == 80 and coco.INPUT.MOSAIC.ENABLE is True
    assert coco.INPUT.LETTERBOX.WIDTH == 640 and coco.SOLVER.IMS_PER_BATCH == 16
    assert tuple(map(tuple, coco.MODEL.ANCHOR_GENERATOR.ANCHORS[2])) == ((116, 90), (156, 198), (373, 326))


def test_opts_freeze_and_errors():
    cfg = _cfg(opts=["INPUT.MIN_SIZE_TEST", "416", "MODEL.YOLOV3.TEST_SCORE_THRESH", "0.5"])
    assert cfg.INPUT.MIN_SIZE_TEST == 416 and cfg.MODEL.YOLOV3.TEST_SCORE_THRESH == 0.5
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.MODEL.DEVICE = "cuda"
    c2 = cfg.clone()
    c2.defrost()
    c2.MODEL.DEVICE = "cuda"
    assert cfg.MODEL.DEVICE == "cpu"
    with pytest.raises(KeyError):
        get_cfg().merge_from_list(["MODEL.YOLO.NUM_CLASSES", "3"])  # the stale key of yolov3.back.yaml
    with pytest.raises(ValueError):
        get_cfg().merge_from_list(["MODEL.YOLOV3.NUM_CLASSES", "many"])
    # dead keys the reference defines are kept
    assert get_cfg().MODEL.WEIGHT == "./config/yolov3.weights" and get_cfg().MODEL.FREEZE_AT == 2


# ---------------------------------------------------------------- model tree / registries
def test_registry_names_match_the_yaml():
    assert "YOLOV3" in META_ARCH_REGISTRY
    assert "build_darknet53_fpn_backbone" in BACKBONE_REGISTRY and "build_darknet53_backbone" in BACKBONE_REGISTRY
    assert "YoloAnchorGenerator" in ANCHOR_GENERATOR_REGISTRY


@pytest.fixture(scope="module")
def model20():
    torch.manual_seed(0)
    m = build_model(_cfg())
    m.eval()
    return m


def test_state_dict_schema_and_module_order(model20):
    from oracle import yolo_ref as R
    sd = model20.state_dict()
    ref = R.init_state_dict(0, num_classes=20)
    assert set(k for k in sd if not k.endswith("num_batches_tracked")) == set(ref)
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    names = [n for n, _ in model20.backbone.darknet_modules()]
    assert names == [L["name"] for L in R.conv_table(20)]
    assert model20.backbone.size_divisibility == 32
    shapes = model20.backbone.output_shape()
    assert [shapes[k].stride for k in ("p3", "p4", "p5")] == [8, 16, 32]
    # FREEZE_AT = 2 (detectron2 default): stem and res1 frozen
    assert not model20.backbone.bottom_up.stem.conv.weight.requires_grad
    assert not model20.backbone.bottom_up.res1[0].conv1.weight.requires_grad
    assert model20.backbone.bottom_up.res2[0].conv1.weight.requires_grad


def test_no_cpu_fallback(model20):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model20([{"image": torch.zeros(3, 64, 64)}])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model20.backbone(torch.zeros(1, 3, 64, 64))
    model20.train()   # training runs on the GPU only, too
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model20([{"image": torch.zeros(3, 64, 64)}])
    model20.eval()


def test_anchor_generator(model20):
    from oracle import yolo_ref as R
    gen = model20.anchor_generator
    assert gen.num_cell_anchors == [3, 3, 3] and gen.strides == [8, 16, 32]
    got = gen.grid_anchors([(52, 52), (26, 26), (13, 13)])
    for a, b in zip(got, R.grid_anchors([(52, 52), (26, 26), (13, 13)])):
        assert torch.equal(a.tensor, b)
    assert gen.anchors_wh()[1] == [(30.0, 61.0), (62.0, 45.0), (59.0, 119.0)]


# ---------------------------------------------------------------- checkpoints
def test_darknet_weights_roundtrip(tmp_path, model20):
    from yolo_amd.checkpoint import YOLOV3Checkpointer, save_darknet_weights
    path = str(tmp_path / "net.weights")
    with open(path, "wb") as f:
        save_darknet_weights(f, model20.backbone.darknet_modules(), seen=12345)
    n_floats = sum(c.weight.numel() + (4 if c.norm is not None else 1) * c.out_channels
                   for _, c in model20.backbone.darknet_modules())
    assert os.path.getsize(path) == 20 + 4 * n_floats
    torch.manual_seed(1)
    other = build_model(_cfg())
    YOLOV3Checkpointer(other).load(path)
    for (k, a), (_, b) in zip(model20.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k
    # layout check against the format description: header, then beta, gamma, mean, var, weights of the stem
    raw = np.fromfile(path, dtype=np.float32, offset=20)
    stem = model20.backbone.bottom_up.stem.conv
    assert np.array_equal(raw[:32], stem.norm.bias.detach().numpy())
    assert np.array_equal(raw[32:64], stem.norm.weight.detach().numpy())
    assert np.array_equal(raw[128:128 + 864], stem.weight.detach().numpy().ravel())
    # a truncated file is an error, not silent garbage
    with open(path, "rb") as f:
        blob = f.read()
    bad = str(tmp_path / "short.weights")
    with open(bad, "wb") as f:
        f.write(blob[:-8])
    with pytest.raises(ValueError, match="ends early"):
        YOLOV3Checkpointer(other).load(bad)


def test_pth_roundtrip_and_resume(tmp_path, model20):
    from yolo_amd.checkpoint import YOLOV3Checkpointer
    ck = YOLOV3Checkpointer(model20, str(tmp_path))
    assert ck.load("") == {}  # empty path keeps the random init, as the reference's base class does
    ck.save("model_0000009", iteration=9)
    assert ck.has_checkpoint() and ck.get_checkpoint_file().endswith("model_0000009.pth")
    torch.manual_seed(3)
    other = build_model(_cfg())
    extra = YOLOV3Checkpointer(other, str(tmp_path)).resume_or_load("", resume=True)
    assert extra["iteration"] == 9
    assert torch.equal(other.state_dict()["backbone.fpn_output2.head.bias"],
                       model20.state_dict()["backbone.fpn_output2.head.bias"])
    with pytest.raises(FileNotFoundError):
        ck.load(str(tmp_path / "missing.pth"))


# ---------------------------------------------------------------- containers
def test_boxes_and_instances():
    b = Boxes(torch.tensor([[0.0, 0, 10, 10], [5, 5, 5, 9], [-3, 2, 30, 40]]))
    assert b.nonempty().tolist() == [True, False, True]
    b.scale(2.0, 0.5)
    b.clip((10, 16))
    assert b.tensor.tolist() == [[0.0, 0.0, 16.0, 5.0], [10.0, 2.5, 10.0, 4.5], [0.0, 1.0, 16.0, 10.0]]
    inst = Instances((10, 16), pred_boxes=b, scores=torch.tensor([0.9, 0.8, 0.7]))
    sub = inst[torch.tensor([True, False, True])]
    assert len(sub) == 2 and sub.image_size == (10, 16) and sub.scores.tolist() == [pytest.approx(0.9), pytest.approx(0.7)]
    cat = Instances.cat([inst, sub])
    assert len(cat) == 5 and isinstance(cat.pred_boxes, Boxes)
    with pytest.raises(AssertionError):
        inst.set("bad", torch.zeros(2))


# ---------------------------------------------------------------- C ABI
def _declared_functions():
    src = open(os.path.join(ROOT, "include", "yolo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(yolo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build yolo_amd/libyolo_hip.so first (make -C yolo_amd/csrc)"
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 11
    for n in names:
        assert hasattr(L, n), n
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert L.yolo_abi_version() == 6
    assert L.yolo_conv_cout_pad(255) == 256 and L.yolo_conv_cout_pad(32) == 128


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected before any HIP call, with a message."""
    L = _lib.lib()
    v = _lib.YoloView(1, 8, 8, 0, 64, 0)
    code = L.yolo_conv2d_fwd(ctypes.byref(v), 1, 1, None, ctypes.byref(v), 1, 64, 64, 3, 1, 0.1, 0, 1, 1, None)
    assert code == -1 and b"halo" in L.yolo_last_error()
    code = L.yolo_conv2d_fwd(ctypes.byref(v), 1, 1, None, ctypes.byref(v), 1, 64, 64, 5, 1, 0.1, 0, 1, 1, None)
    assert code == -1 and b"kernel size" in L.yolo_last_error()
    code = L.yolo_conv2d_fwd(None, None, None, None, None, 1, 64, 64, 1, 1, 0.1, 0, 1, 1, None)
    assert code == -1 and b"null" in L.yolo_last_error()
    code = L.yolo_batched_nms(None, None, None, 1, 0, 5, 0.5, 0, 20000, 0, 80, 1, 1, 1, None, 0, None)
    assert code == -1 and b"batch" in L.yolo_last_error()
    code = L.yolo_decode(None, 3, None, None, 1, 3, 80, 0.25, 0, None, None, None, None, 0, None, 0, None)
    assert code == -1


def test_darknet_weights_reader_equals_the_oracle_reader(tmp_path, model20):
    """A file written from a state dict with randomised BatchNorm statistics by the oracle's table-driven writer:
    the product reader (through YOLOV3Checkpointer, checkpoint.py:13-22) and the oracle's statement-by-statement
    restatement of load_darknet.py:103-124 return the same tensors under the same keys."""
    from oracle import darknet_weights_ref as W, yolo_ref as R
    from yolo_amd.checkpoint import YOLOV3Checkpointer
    sd = R.init_state_dict(5, num_classes=20)
    R.randomize_norm_stats(sd, seed=6)
    path = str(tmp_path / "rand.weights")
    n = W.write_weights(path, sd, num_classes=20)
    assert os.path.getsize(path) == 20 + 4 * n
    ref = W.read_weights(path, num_classes=20)
    assert int(ref.pop("__trailing_floats__")) == 0
    assert set(ref) == set(sd) and all(torch.equal(ref[k], sd[k]) for k in sd)
    torch.manual_seed(9)
    m = build_model(_cfg())
    YOLOV3Checkpointer(m).load(path)
    got = m.state_dict()
    for k, v in ref.items():
        assert torch.equal(got[k], v), k
    # the 80-class file does not fit the 20-class model: reported, not loaded misaligned
    sd80 = R.init_state_dict(5, num_classes=80)
    p80 = str(tmp_path / "coco.weights")
    assert W.write_weights(p80, sd80, num_classes=80) == 62001757  # the official yolov3.weights' float count
    with pytest.raises(ValueError, match="different network"):
        YOLOV3Checkpointer(m).load(p80)
    # ... and the 20-class file is too short for an 80-class reader (the oracle's reader agrees)
    with pytest.raises(ValueError, match="ends early"):
        W.read_weights(path, num_classes=80)


def test_pth_with_other_head_shapes_loads_the_rest(tmp_path, model20):
    """fvcore's Checkpointer skips parameters of incompatible shape with a warning (an 80-class checkpoint
    fine-tuned on 20 classes re-initialises only the three head convs)."""
    from oracle import yolo_ref as R
    from yolo_amd.checkpoint import YOLOV3Checkpointer
    sd80 = R.init_state_dict(11, num_classes=80)
    path = str(tmp_path / "coco.pth")
    torch.save({"model": sd80}, path)
    torch.manual_seed(4)
    m = build_model(_cfg())
    before = {k: v.clone() for k, v in m.state_dict().items()}
    ck = YOLOV3Checkpointer(m)
    ck.load(path)
    after = m.state_dict()
    heads = [f"backbone.fpn_output{i}.head.{p}" for i in range(3) for p in ("weight", "bias")]
    assert sorted(ck.skipped_keys) == sorted(heads)
    for k in heads:
        assert torch.equal(after[k], before[k])
    assert torch.equal(after["backbone.bottom_up.res3.0.conv2.weight"], sd80["backbone.bottom_up.res3.0.conv2.weight"])


def test_anchor_offset_other_than_half_is_refused():
    cfg = _cfg(opts=["MODEL.ANCHOR_GENERATOR.OFFSET", 0.25])
    with pytest.raises(NotImplementedError, match="OFFSET"):
        build_model(cfg)


def test_unported_augmentations_say_so():
    with pytest.raises(NotImplementedError, match="not ported"):
        from yolo_amd.data import RandomMedianBlur  # noqa: F401
    with pytest.raises(NotImplementedError, match="not ported"):
        from yolo_amd.data.transforms import PixelDropoutTransform  # noqa: F401
    with pytest.raises(ImportError):
        from yolo_amd.data import NoSuchThing  # noqa: F401


def test_reference_package_name_resolves_to_the_same_modules():
    """Code written against the reference imports ``yolo.*`` (tools/inference.py:12-13, tools/train_net.py:19-21 of the
    reference): the alias package hands out the very module objects of yolo_amd, so registries are filled once."""
    import importlib
    import yolo
    import yolo_amd
    for name in ("config", "checkpoint", "modeling", "data", "modeling.backbone.fpn", "modeling.backbone.darknet",
                 "modeling.meta_arch.yolo", "modeling.anchor_generator", "modeling.box_regression", "modeling.matcher",
                 "data.dataset_mapper", "data.transforms", "checkpoint.load_darknet"):
        assert importlib.import_module("yolo." + name) is importlib.import_module("yolo_amd." + name), name
    from yolo.config import get_cfg
    from yolo.checkpoint import YOLOV3Checkpointer
    assert get_cfg is yolo_amd.config.get_cfg and YOLOV3Checkpointer is yolo_amd.checkpoint.YOLOV3Checkpointer
    assert yolo.modeling is yolo_amd.modeling
    with pytest.raises(ModuleNotFoundError):
        importlib.import_module("yolo.no_such_module")
    # the aliased modules keep their own identity: importing them as ``yolo.X`` must not re-stamp the spec of
    # ``yolo_amd.X`` (reload and relative imports inside functions depend on it)
    for name in ("checkpoint", "modeling.backbone.fpn", "data.transforms"):
        m = importlib.import_module("yolo_amd." + name)
        assert m.__spec__.name == "yolo_amd." + name and m.__name__ == "yolo_amd." + name, (name, m.__spec__.name)
        assert m.__package__ == m.__spec__.parent
    importlib.reload(importlib.import_module("yolo_amd.modeling.box_regression"))

"""Default configuration tree.

Two layers, as in the reference (yolo/config/defaults.py:6-73 mutates detectron2's ``_C``):
  1. the subset of detectron2's base keys that the reference's YAML files and entry points touch;
  2. the YOLO keys the reference adds (MODEL.DARKNET.*, MODEL.YOLOV3.*, MODEL.MATCHER.*,
     MODEL.ANCHOR_GENERATOR.ANCHORS, INPUT.{MOSAIC,RANDOM_AFFINE,COLOR_JITTER,LETTERBOX}.*).
"""
from .node import CfgNode as CN

_C = CN()
_C.VERSION = 2

# ---- detectron2 base keys -----------------------------------------------------------------
_C.MODEL = CN()
_C.MODEL.DEVICE = "cuda"
_C.MODEL.META_ARCHITECTURE = "YOLOV3"            # reference defaults.py:10
_C.MODEL.WEIGHTS = ""
_C.MODEL.WEIGHT = "./config/yolov3.weights"      # reference defaults.py:11 (dead key, kept)
_C.MODEL.FREEZE_AT = 2                           # reference defaults.py:15 (dead key, kept)
_C.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
_C.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
_C.MODEL.BACKBONE = CN()
_C.MODEL.BACKBONE.NAME = "build_darknet53_fpn_backbone"  # reference defaults.py:14
_C.MODEL.BACKBONE.FREEZE_AT = 2
_C.MODEL.ANCHOR_GENERATOR = CN()
_C.MODEL.ANCHOR_GENERATOR.NAME = "YoloAnchorGenerator"   # reference defaults.py:26
_C.MODEL.ANCHOR_GENERATOR.SIZES = [[32, 64, 128, 256, 512]]
_C.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.5, 1.0, 2.0]]
_C.MODEL.ANCHOR_GENERATOR.ANGLES = [[-90, 0, 90]]
_C.MODEL.ANCHOR_GENERATOR.OFFSET = 0.5                   # reference defaults.py:27
_C.MODEL.ANCHOR_GENERATOR.ANCHORS = ([[10, 13], [16, 30], [33, 23]],      # reference defaults.py:28-30
                                     [[30, 61], [62, 45], [59, 119]],
                                     [[116, 90], [156, 198], [373, 326]])

_C.INPUT = CN()
_C.INPUT.MIN_SIZE_TRAIN = (800,)
_C.INPUT.MIN_SIZE_TRAIN_SAMPLING = "choice"
_C.INPUT.MAX_SIZE_TRAIN = 1333
_C.INPUT.MIN_SIZE_TEST = 800
_C.INPUT.MAX_SIZE_TEST = 1333
_C.INPUT.RANDOM_FLIP = "horizontal"
_C.INPUT.FORMAT = "BGR"

_C.DATASETS = CN()
_C.DATASETS.TRAIN = ()
_C.DATASETS.TEST = ()

_C.DATALOADER = CN()
_C.DATALOADER.NUM_WORKERS = 4
_C.DATALOADER.ASPECT_RATIO_GROUPING = True
_C.DATALOADER.SAMPLER_TRAIN = "TrainingSampler"
_C.DATALOADER.FILTER_EMPTY_ANNOTATIONS = True

_C.SOLVER = CN()
_C.SOLVER.LR_SCHEDULER_NAME = "WarmupMultiStepLR"
_C.SOLVER.MAX_ITER = 40000
_C.SOLVER.BASE_LR = 0.001
_C.SOLVER.MOMENTUM = 0.9
_C.SOLVER.NESTEROV = False
_C.SOLVER.WEIGHT_DECAY = 0.0001
_C.SOLVER.WEIGHT_DECAY_NORM = 0.0
_C.SOLVER.GAMMA = 0.1
_C.SOLVER.STEPS = (30000,)
_C.SOLVER.WARMUP_FACTOR = 1.0 / 1000
_C.SOLVER.WARMUP_ITERS = 1000
_C.SOLVER.WARMUP_METHOD = "linear"
_C.SOLVER.CHECKPOINT_PERIOD = 5000
_C.SOLVER.IMS_PER_BATCH = 16
_C.SOLVER.REFERENCE_WORLD_SIZE = 0

_C.TEST = CN()
_C.TEST.EVAL_PERIOD = 0
_C.TEST.DETECTIONS_PER_IMAGE = 100

_C.OUTPUT_DIR = "./output"
_C.SEED = -1
_C.CUDNN_BENCHMARK = False
_C.VIS_PERIOD = 0

# ---- keys added by the reference (yolo/config/defaults.py:17-73) ------------------------------
_C.MODEL.DARKNET = CN()
_C.MODEL.DARKNET.NORM = "BN"
_C.MODEL.DARKNET.STEM_OUT_CHANNELS = 32
_C.MODEL.DARKNET.RES1_OUT_CHANNELS = 64
_C.MODEL.DARKNET.DEPTH = 53
_C.MODEL.DARKNET.OUT_FEATURES = ["res3", "res4", "res5"]

_C.MODEL.YOLOV3 = CN()
_C.MODEL.YOLOV3.NORM = "BN"
_C.MODEL.YOLOV3.IN_FEATURES = ["p3", "p4", "p5"]
_C.MODEL.YOLOV3.NUM_CLASSES = 80
_C.MODEL.YOLOV3.TEST_SCORE_THRESH = 0.25
_C.MODEL.YOLOV3.TEST_NMS_THRESH = 0.5

_C.MODEL.MATCHER = CN()
_C.MODEL.MATCHER.IGNORE_THRESHOLD = 0.5
_C.MODEL.MATCHER.LABELS = [-1, 0, 1]

_C.INPUT.MOSAIC = CN()
_C.INPUT.MOSAIC.ENABLE = False
_C.INPUT.MOSAIC.NUM_IMAGES = 4
_C.INPUT.MOSAIC.MOSAIC_WIDTH = 640
_C.INPUT.MOSAIC.MOSAIC_HEIGHT = 640
_C.INPUT.MOSAIC.POOL_CAPACITY = 10

_C.INPUT.RANDOM_AFFINE = CN()
_C.INPUT.RANDOM_AFFINE.ENABLE = False
_C.INPUT.RANDOM_AFFINE.TRANSLATE = 0.1
_C.INPUT.RANDOM_AFFINE.SCALE = 0.1
_C.INPUT.RANDOM_AFFINE.DEGREE = 0
_C.INPUT.RANDOM_AFFINE.SHEAR = 10.0
_C.INPUT.RANDOM_AFFINE.PERSPECTIVE = 0.0

_C.INPUT.COLOR_JITTER = CN()
_C.INPUT.COLOR_JITTER.ENABLE = False
_C.INPUT.COLOR_JITTER.HUE = 0.1
_C.INPUT.COLOR_JITTER.SATURATION = 1.5
_C.INPUT.COLOR_JITTER.EXPOSURE = 1.5

_C.INPUT.LETTERBOX = CN()
_C.INPUT.LETTERBOX.ENABLE = False
_C.INPUT.LETTERBOX.WIDTH = 640
_C.INPUT.LETTERBOX.HEIGHT = 640

# ---- keys of this implementation (additive; absent from the reference) -----------------------
_C.MODEL.HIP = CN()
_C.MODEL.HIP.DTYPE = "bf16"        # conv compute/storage type: "bf16" or "f32"
_C.MODEL.HIP.HEAD_DTYPE = "f32"    # element type of the three head tensors fed to decode
_C.MODEL.HIP.NMS_MODE = "auto"     # "auto" (torchvision's branch rule) | "per_class" | "coord_trick"
_C.MODEL.HIP.NMS_AUTO_NUMEL = 20000  # torchvision's switch: 20000 on a GPU, 4000 on the CPU

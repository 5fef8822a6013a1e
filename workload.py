"""Synthetic workload shared by bench.py, the GPU tests and __graft_entry__.smoke():
seeded uint8 images and a randomly initialised YOLOv3 whose BatchNorm statistics are calibrated on them
(see yolo_amd/utils/calibrate.py for why).  No datasets or checkpoints are available offline."""
from __future__ import annotations

import os
from typing import Optional

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
CONFIGS = os.path.join(ROOT, "configs")


def synthetic_images(n: int, h: int, w: int, seed: int = 1, device="cpu") -> torch.Tensor:
    """[n,3,h,w] uint8, i.i.d. uniform 0..255 blended with a smooth gradient so feature maps vary."""
    g = torch.Generator().manual_seed(seed)
    noise = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.int32)
    yy = torch.linspace(0, 1, h).view(1, 1, h, 1)
    xx = torch.linspace(0, 1, w).view(1, 1, 1, w)
    phase = torch.rand(n, 3, 1, 1, generator=g)
    smooth = 127.5 * (1 + torch.sin(6.28318 * (2 * xx + 3 * yy + phase)))
    img = (0.5 * noise + 0.5 * smooth).clamp(0, 255).to(torch.uint8)
    return img.to(device)


def build_model(config: str = "yolov3-aug.yaml", device: str = "cuda", dtype: str = "bf16", head_dtype: str = "f32",
                opts: Optional[list] = None, seed: int = 0):
    """Model from configs/<config> with seeded random-init weights (the reference's initialisers)."""
    import yolo_amd  # noqa: F401
    from yolo_amd.config import get_cfg
    from yolo_amd.modeling import build_model as _build
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(CONFIGS, config))
    cfg.merge_from_list(["MODEL.WEIGHTS", "", "MODEL.DEVICE", device, "MODEL.HIP.DTYPE", dtype,
                         "MODEL.HIP.HEAD_DTYPE", head_dtype] + list(opts or []))
    cfg.freeze()
    torch.manual_seed(seed)
    model = _build(cfg)
    model.eval()
    return model, cfg


def calibrated_model(size: int = 416, calib_batch: int = 4, obj_bias: float = -4.5, residual_gain=None, **kw):
    """build_model + BatchNorm statistics / head scale calibrated on ``calib_batch`` synthetic images."""
    from yolo_amd.utils.calibrate import calibrate_norm_stats
    model, cfg = build_model(**kw)
    imgs = synthetic_images(calib_batch, size, size, seed=12345, device=model.device).float()
    calibrate_norm_stats(model.backbone, imgs, mean=cfg.MODEL.PIXEL_MEAN, std=cfg.MODEL.PIXEL_STD, obj_bias=obj_bias,
                         residual_gain=residual_gain)
    return model, cfg

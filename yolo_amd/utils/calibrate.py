"""Re-estimate BatchNorm running statistics (and optionally rescale the linear heads) from a batch of
images, executing every convolution through the HIP kernels.

Purpose: a randomly initialised Darknet-53 with identity BatchNorm statistics (what the reference's
constructors produce, yolo/modeling/backbone/darknet.py:55,108-111) lets activations grow by orders of
magnitude per stage, so its head saturates and every anchor "detects" every class.  Calibrating the
statistics on the synthetic input gives the network the unit-scale activations and the sparse detections a
trained checkpoint has, which is what benchmarks and parity tests want to exercise.  The result is an
ordinary state dict: the CPU oracle and the HIP engine both consume it.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops


@torch.no_grad()
def calibrate_norm_stats(backbone, images: torch.Tensor, mean=(0.0, 0.0, 0.0), std=(255.0, 255.0, 255.0),
                         head_logit_std: Optional[float] = 1.0, obj_bias: Optional[float] = -4.5,
                         residual_gain: Optional[float] = None) -> None:
    """backbone: DarkNetFPN on the GPU; images [N,3,H,W] (0..255), H and W multiples of 32.  In place.

    ``residual_gain``: BatchNorm weight given to the last conv of every residual block (None keeps the
    constructor's 1).  With gain 1 every block doubles the variance of the stream it adds to and the network
    amplifies relative perturbations about twofold per stage; a small gain (the "zero-gamma" initialisation trained
    residual networks start from) gives a well-conditioned network, which is what a test needs when it wants to see
    the arithmetic error of the kernels rather than the network's own sensitivity."""
    from ..engine import Yolov3Engine
    dev = next(backbone.parameters()).device
    if dev.type == "cuda" and dev.index != torch.cuda.current_device():  # launch on the GPU that holds the model
        with torch.cuda.device(dev):
            return calibrate_norm_stats(backbone, images, mean, std, head_logit_std, obj_bias, residual_gain)
    images = images.to(dev)
    eng = Yolov3Engine(backbone, num_classes=backbone._num_classes, anchors_wh=backbone._anchors_wh,
                       dtype=torch.float32, head_dtype=torch.float32, blocks=backbone.bottom_up.blocks_per_stage)
    N, _, H, W = images.shape
    P = eng.plan(N, H, W)
    convs = dict(backbone.darknet_modules())

    def set_stats(conv, raw: torch.Tensor):  # raw: [N,h,w,C] pre-norm conv output
        flat = raw.reshape(-1, raw.shape[-1]).double()
        conv.norm.running_mean.copy_(flat.mean(0).float())
        conv.norm.running_var.copy_(flat.var(0, unbiased=False).float().clamp_min(1e-6))

    # stem: run with identity norm, measure, then with the calibrated norm
    stem = convs["bottom_up.stem.conv"]
    w_raw, b0 = ops.pack_stem(stem.weight.detach().float(), torch.zeros(stem.out_channels), dev)
    ops.stem_fwd(images.contiguous(), w_raw, b0, P.stem_out, mean=mean, std=std, slope=1.0)
    set_stats(stem, P.stem_out.interior())
    w, b = ops.pack_stem(*stem.folded(), dev)
    ops.stem_fwd(images.contiguous(), w, b, P.stem_out, mean=mean, std=std)

    ak = (5 + backbone._num_classes)
    for meta in P.meta:
        conv = convs[meta["name"]]
        x, y = meta["x"], meta["y"]
        up = 2 if meta["up"] else 1
        tmp = ops.alloc_view(N, y.H // up, y.W // up, (conv.out_channels + 7) // 8 * 8, 0, torch.float32, dev)
        if conv.norm is not None:
            wr, br = ops.pack_conv(conv.weight.detach().float(), torch.zeros(conv.out_channels), torch.float32, dev)
            ops.conv2d_fwd(x, wr, br, tmp, cin=conv.in_channels, cout=conv.out_channels, k=conv.kernel_size,
                           stride=conv.stride, slope=1.0)
            set_stats(conv, tmp.interior()[..., :conv.out_channels])
            if residual_gain is not None and meta["residual"] is not None:
                conv.norm.weight.fill_(residual_gain)
        elif head_logit_std is not None:
            wr, br = ops.pack_conv(conv.weight.detach().float(), torch.zeros(conv.out_channels), torch.float32, dev)
            ops.conv2d_fwd(x, wr, br, tmp, cin=conv.in_channels, cout=conv.out_channels, k=conv.kernel_size,
                           stride=conv.stride, slope=1.0)
            s = float(tmp.interior()[..., :conv.out_channels].std())
            conv.weight.mul_(head_logit_std / max(s, 1e-12))
            conv.bias.zero_()
            if obj_bias is not None:
                conv.bias[4::ak] = obj_bias
        wf, bf = ops.pack_conv(*conv.folded(), torch.float32, dev)
        ops.conv2d_fwd(x, wf, bf, y, cin=conv.in_channels, cout=conv.out_channels, k=conv.kernel_size,
                       stride=conv.stride, slope=meta["slope"], residual=meta["residual"], upsample2x=meta["up"])
    torch.cuda.synchronize()

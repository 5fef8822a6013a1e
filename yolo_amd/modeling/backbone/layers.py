"""Parameter containers with the reference's state-dict schema and a per-module HIP forward.

A ``Conv2d`` here owns ``weight`` (+ ``bias``) and an optional ``norm`` child with
``weight/bias/running_mean/running_var`` -- the attribute layout detectron2's ``Conv2d`` + ``get_norm("BN")``
give the reference (yolo/modeling/backbone/darknet.py:45-53; yolo/checkpoint/load_darknet.py:113-124),
so reference ``.pth`` / darknet ``.weights`` files load unchanged.

Calling a module runs the fused gfx950 kernel for that layer on NCHW tensors (layout converted at the
module boundary).  The whole-network fast path does not go through these per-module calls: see
``yolo_amd/engine.py``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ... import ops


class BatchNorm2d(nn.Module):
    """Eval-mode batch-norm parameters (always folded into the convolution before launch)."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        # darknet .weights files and FrozenBatchNorm2d checkpoints carry no num_batches_tracked
        key = prefix + "num_batches_tracked"
        if key not in state_dict:
            state_dict[key] = self.num_batches_tracked.clone()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def as_dict(self):
        return dict(weight=self.weight, bias=self.bias, running_mean=self.running_mean, running_var=self.running_var)


def get_norm(norm: str, out_channels: int) -> Optional[nn.Module]:
    if norm in (None, ""):
        return None
    if norm in ("BN", "FrozenBN"):
        return BatchNorm2d(out_channels)
    raise ValueError(f"norm '{norm}' is not supported on this path (the reference uses 'BN', defaults.py:19)")


class Conv2d(nn.Module):
    """conv (+ folded norm) (+ LeakyReLU) (+ residual) as one HIP launch."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 bias: bool = True, norm: Optional[nn.Module] = None):
        super().__init__()
        assert padding == (kernel_size - 1) // 2, "only 'same' padding occurs in the reference's network"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.norm = norm
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)  # nn.Conv2d default (fpn.py:23-31 keeps it)
        self._packed = None  # (key, w_packed, bias_packed)

    def folded(self):
        return ops.fold_norm(self.weight, self.bias, self.norm.as_dict() if self.norm is not None else None,
                             self.norm.eps if self.norm is not None else 1e-5)

    def packed(self, dtype: torch.dtype):
        key = (dtype, self.weight._version, self.weight.device,
               None if self.norm is None else tuple(t._version for t in self.norm.as_dict().values()))
        if self._packed is None or self._packed[0] != key:
            w, b = self.folded()
            if self.in_channels == 3:
                self._packed = (key,) + ops.pack_stem(w, b, self.weight.device)
            else:
                self._packed = (key,) + ops.pack_conv(w, b, dtype, self.weight.device)
        return self._packed[1], self._packed[2]

    def forward(self, x: torch.Tensor, slope: float = 1.0, residual: Optional[torch.Tensor] = None,
                dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """x: [N, Cin, H, W] on the GPU; returns [N, Cout, Ho, Wo] (a strided view of an NHWC buffer)."""
        if self.training:
            raise NotImplementedError("training (batch-statistics BatchNorm, conv backward) is not built on the HIP "
                                      "path yet; call .eval()")
        if not x.is_cuda:
            raise RuntimeError("yolo_amd runs on MI355X only: input is on the CPU and there is no CPU fallback")
        if x.device.index != torch.cuda.current_device():  # launch on the GPU that holds the input
            with torch.cuda.device(x.device):
                return self.forward(x, slope, residual, dtype)
        dtype = dtype or (x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        N, _, H, W = x.shape
        k, s = self.kernel_size, self.stride
        Ho, Wo = (H + 2 * self.padding - k) // s + 1, (W + 2 * self.padding - k) // s + 1
        cpad = (self.out_channels + 7) // 8 * 8
        y = ops.alloc_view(N, Ho, Wo, cpad, 0, dtype, x.device)
        w, b = self.packed(dtype)
        if self.in_channels == 3:
            raise RuntimeError("the 3-channel stem runs fused with pixel normalisation: call BasicStem")
        xv = ops.view_from_nchw(x, self.padding, dtype)
        rv = ops.view_from_nchw(residual, 0, dtype) if residual is not None else None
        ops.conv2d_fwd(xv, w, b, y, cin=self.in_channels, cout=self.out_channels, k=k, stride=s, slope=slope,
                       residual=rv)
        return y.nchw()[:, :self.out_channels]


def c2_msra_fill(module: Conv2d) -> None:
    """kaiming_normal_(fan_out, relu) + zero bias: what fvcore's c2_msra_fill does for the reference
    (darknet.py:55,108-111; fpn.py:63-64,93-94)."""
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)

"""Reader for the official Darknet ``.weights`` binary format (reference
yolo/checkpoint/load_darknet.py:77-125).

Layout: 5 x int32 header (major, minor, revision, images-seen as 2 words), then for every conv in network
file order: with BatchNorm -> beta, gamma, running_mean, running_var (Cout f32 each); without -> conv bias;
then the conv weight [Cout, Cin, k, k] f32.
"""
from __future__ import annotations

import io
import logging
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

HEADER_WORDS = 5


def _take(buf: np.ndarray, pos: int, like: torch.Tensor) -> Tuple[torch.Tensor, int]:
    n = like.numel()
    if pos + n > buf.size:
        raise ValueError(f"darknet weights file ends early: need {pos + n} floats, file has {buf.size}")
    return torch.from_numpy(buf[pos:pos + n].copy()).view_as(like), pos + n


def load_darknet_weights_to_dict(f, module_list: Iterable, module_prefix: str = "backbone") -> Dict[str, torch.Tensor]:
    """f: binary file object; module_list: (name, conv) pairs in Darknet order (DarkNetFPN.darknet_modules())."""
    assert isinstance(f, io.IOBase)
    raw = f.read()
    header = np.frombuffer(raw, dtype=np.int32, count=HEADER_WORDS)
    logging.getLogger(__name__).info(f"load official weights (v{header[0]}.{header[1]}.{header[2]})")
    buf = np.frombuffer(raw, dtype=np.float32, offset=4 * HEADER_WORDS)
    pos = 0
    sd: Dict[str, torch.Tensor] = {}
    for field, m in module_list:
        name = f"{module_prefix}.{field}" if module_prefix else field
        if m.norm is not None:
            for key, t in (("norm.bias", m.norm.bias), ("norm.weight", m.norm.weight),
                           ("norm.running_mean", m.norm.running_mean), ("norm.running_var", m.norm.running_var)):
                sd[f"{name}.{key}"], pos = _take(buf, pos, t)
        else:
            sd[f"{name}.bias"], pos = _take(buf, pos, m.bias)
        sd[f"{name}.weight"], pos = _take(buf, pos, m.weight)
    if pos != buf.size:
        # The reference stops reading here and keeps whatever the sequential read produced.  A file with floats left
        # over was written for another network (typically the 80-class yolov3.weights read into a 20-class model:
        # every tensor after the first head conv is then misaligned), so it is reported instead of loaded.
        raise ValueError(f"darknet weights file holds {buf.size} floats but this model reads {pos}: "
                         "the file belongs to a different network (number of classes?)")
    return sd


def save_darknet_weights(f, module_list: Iterable, version=(0, 2, 5), seen: int = 0) -> None:
    """Inverse of the reader (used to produce fixtures and to export)."""
    np.array([version[0], version[1], version[2], seen & 0xFFFFFFFF, seen >> 32], dtype=np.int32).tofile(f)
    for _, m in module_list:
        if m.norm is not None:
            for t in (m.norm.bias, m.norm.weight, m.norm.running_mean, m.norm.running_var):
                t.detach().cpu().float().numpy().tofile(f)
        else:
            m.bias.detach().cpu().float().numpy().tofile(f)
        m.weight.detach().cpu().float().numpy().tofile(f)

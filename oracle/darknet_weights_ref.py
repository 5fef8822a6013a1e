"""Darknet ``.weights`` files read and written from the layer TABLE alone (no model object).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference ships no weight file
and no test of its reader; this restates reference yolo/checkpoint/load_darknet.py:77-125 statement by
statement, with oracle.yolo_ref.conv_table() standing in for ``DarkNetFPN.darknet_modules()``
(yolo/modeling/backbone/fpn.py:187-190).  One known answer pins the table: the official 80-class
``yolov3.weights`` holds 62,001,757 floats after its header, and so does a file written from this table.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import yolo_ref as R


def _read_tensor(fp, shape) -> torch.Tensor:
    """load_darknet.py:18-23: ``np.fromfile(fp, float32, count=numel)`` then ``view_as``; a short read
    makes ``view_as`` raise there, a ValueError here."""
    count = int(np.prod(shape))
    w = np.fromfile(fp, dtype=np.float32, count=count)
    if w.size != count:
        raise ValueError(f"file ends early: wanted {count} floats, got {w.size}")
    return torch.from_numpy(w).view(*shape)


def read_weights(path: str, num_classes: int = 80, prefix: str = "backbone.") -> Dict[str, torch.Tensor]:
    """-> state dict with the reference's keys.  load_darknet.py:103-125."""
    sd: Dict[str, torch.Tensor] = {}
    with open(path, "rb") as fp:
        header = np.fromfile(fp, dtype=np.int32, count=5)  # :103
        assert header.size == 5
        for L in R.conv_table(num_classes=num_classes):  # :107, module order of fpn.py:187-190
            field = prefix + L["name"]  # :110-111
            c = L["cout"]
            if L["bn"]:  # :113-118: bias, weight, running_mean, running_var
                sd[field + ".norm.bias"] = _read_tensor(fp, (c,))
                sd[field + ".norm.weight"] = _read_tensor(fp, (c,))
                sd[field + ".norm.running_mean"] = _read_tensor(fp, (c,))
                sd[field + ".norm.running_var"] = _read_tensor(fp, (c,))
            else:  # :119-121: conv bias
                sd[field + ".bias"] = _read_tensor(fp, (c,))
            sd[field + ".weight"] = _read_tensor(fp, (c, L["cin"], L["k"], L["k"]))  # :124
        sd["__trailing_floats__"] = torch.tensor(np.fromfile(fp, dtype=np.float32).size)
    return sd


def write_weights(path: str, sd: Dict[str, torch.Tensor], num_classes: int = 80, prefix: str = "backbone.",
                  version=(0, 2, 5), seen: int = 32013312) -> int:
    """The file a Darknet run would save for this state dict (the inverse of read_weights).  Returns the number of
    floats after the header."""
    n = 0
    with open(path, "wb") as fp:
        np.array([version[0], version[1], version[2], seen & 0xFFFFFFFF, seen >> 32], dtype=np.int32).tofile(fp)
        for L in R.conv_table(num_classes=num_classes):
            field = prefix + L["name"]
            keys = ([".norm.bias", ".norm.weight", ".norm.running_mean", ".norm.running_var"] if L["bn"] else [".bias"])
            for k in keys + [".weight"]:
                a = sd[field + k].detach().cpu().float().contiguous().numpy()
                a.tofile(fp)
                n += a.size
    return n

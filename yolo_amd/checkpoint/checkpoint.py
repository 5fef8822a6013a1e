"""Checkpoint loading / saving with the reference's entry point
(yolo/checkpoint/checkpoint.py:8-22: a DetectionCheckpointer whose ``_load_file`` also reads Darknet
``.weights``).  ``.pth`` files are read with ``weights_only=True``."""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, Optional

import torch

from .load_darknet import load_darknet_weights_to_dict

__all__ = ["YOLOV3Checkpointer"]


class YOLOV3Checkpointer:
    def __init__(self, model, save_dir: str = "", *, save_to_disk: bool = True, **checkpointables: Any):
        self.model = model
        self.save_dir = save_dir
        self.save_to_disk = save_to_disk
        self.checkpointables = dict(checkpointables)
        self.logger = logging.getLogger(__name__)

    # ---- loading -------------------------------------------------------------------------
    def _load_file(self, filename: str) -> Dict[str, Any]:
        if filename.endswith(".weights"):
            with open(filename, "rb") as f:
                loaded = load_darknet_weights_to_dict(f, self.model.backbone.darknet_modules())
        else:
            loaded = torch.load(filename, map_location="cpu", weights_only=True)
        if "model" not in loaded:
            loaded = {"model": loaded}
        return loaded

    def load(self, path: str, checkpointables: Optional[list] = None) -> Dict[str, Any]:
        """Empty path = keep the current (randomly initialised) parameters, as fvcore's Checkpointer does."""
        if not path:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        if not os.path.isfile(path):
            raise FileNotFoundError(f"Checkpoint {path} not found!")
        ckpt = self._load_file(path)
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ckpt.pop("model").items()}
        own = self.model.state_dict()
        # fvcore's Checkpointer._load_model (the reference's base class): parameters whose shape differs are
        # skipped with a warning and keep their initialisation -- e.g. the three head convs when an 80-class
        # checkpoint is fine-tuned on 20 classes -- and everything else is loaded.
        bad = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
        for k in bad:
            self.logger.warning("Skip loading parameter '%s' to the model due to incompatible shapes: %s in the "
                                "checkpoint but %s in the model! You might want to double check if this is expected.",
                                k, tuple(sd[k].shape), tuple(own[k].shape))
            del sd[k]
        self.skipped_keys = bad
        res = self.model.load_state_dict(sd, strict=False)
        missing = [k for k in res.missing_keys if not k.endswith("num_batches_tracked") and k not in bad]
        if missing:
            self.logger.warning("keys missing from the checkpoint: %s", missing[:8])
        if res.unexpected_keys:
            self.logger.warning("keys in the checkpoint not used by the model: %s", res.unexpected_keys[:8])
        for key in (checkpointables if checkpointables is not None else self.checkpointables):
            if key in ckpt and key in self.checkpointables:
                self.checkpointables[key].load_state_dict(ckpt.pop(key))
        return ckpt

    def resume_or_load(self, path: str, *, resume: bool = True) -> Dict[str, Any]:
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])

    # ---- saving --------------------------------------------------------------------------
    def save(self, name: str, **kwargs: Any) -> None:
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": self.model.state_dict()}
        for k, obj in self.checkpointables.items():
            data[k] = obj.state_dict()
        data.update(kwargs)
        os.makedirs(self.save_dir, exist_ok=True)
        fname = os.path.join(self.save_dir, f"{name}.pth")
        torch.save(data, fname)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(os.path.basename(fname))

    def has_checkpoint(self) -> bool:
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self) -> str:
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
                return os.path.join(self.save_dir, f.read().strip())
        except IOError:
            return ""

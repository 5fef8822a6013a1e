"""yolo_amd: the YOLOv3 inference hot path of shiboyang/YOLO on MI355X (gfx950).

Importing the package registers the model pieces under the reference's registry names
(reference: yolo/__init__.py:6-7, yolo/modeling/__init__.py:6-8).
"""
from . import modeling  # noqa: F401  (registration side effects)
from .config import get_cfg  # noqa: F401

__all__ = ["get_cfg", "modeling"]

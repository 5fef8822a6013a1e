from .node import CfgNode


def get_cfg() -> CfgNode:
    """A fresh copy of the default configuration (reference: yolo/config/config.py:9-16)."""
    from .defaults import _C
    return _C.clone()


__all__ = ["CfgNode", "get_cfg"]

"""Name -> object registry with the lookup semantics the reference relies on (detectron2 registries
BACKBONE_REGISTRY / META_ARCH_REGISTRY / ANCHOR_GENERATOR_REGISTRY, used at
yolo/modeling/meta_arch/yolo.py:28, yolo/modeling/backbone/fpn.py:193, yolo/modeling/anchor_generator.py:15)."""
from __future__ import annotations

from typing import Any, Dict


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._map: Dict[str, Any] = {}

    def register(self, obj: Any = None):
        if obj is None:  # used as @REG.register()
            def deco(f):
                self._add(f.__name__, f)
                return f
            return deco
        self._add(obj.__name__, obj)
        return obj

    def _add(self, name: str, obj: Any):
        if name in self._map:
            raise KeyError(f"'{name}' is already registered in '{self._name}'")
        self._map[name] = obj

    def get(self, name: str) -> Any:
        if name not in self._map:
            raise KeyError(f"no object named '{name}' in the '{self._name}' registry; known: {sorted(self._map)}")
        return self._map[name]

    def __contains__(self, name: str) -> bool:
        return name in self._map

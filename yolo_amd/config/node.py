"""A small attribute-dict configuration node with the yacs / detectron2 CfgNode behaviour the
reference's entry points use: clone, merge_from_file (YAML), merge_from_list (--opts KEY VALUE ...),
freeze (tools/inference.py:92-99, tools/train_net.py:70-76)."""
from __future__ import annotations

import copy
from ast import literal_eval
from typing import Any, List

import yaml


class CfgNode(dict):
    _IMMUTABLE = "__immutable__"

    def __init__(self, init: dict | None = None):
        super().__init__()
        self.__dict__[CfgNode._IMMUTABLE] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute access ---------------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        if self.is_frozen():
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    # freezing -----------------------------------------------------------------------
    def is_frozen(self) -> bool:
        return self.__dict__[CfgNode._IMMUTABLE]

    def _set_immutable(self, flag: bool) -> None:
        self.__dict__[CfgNode._IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self) -> None:
        self._set_immutable(True)

    def defrost(self) -> None:
        self._set_immutable(False)

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        new = CfgNode()
        for k, v in self.items():
            dict.__setitem__(new, k, copy.deepcopy(v, memo))
        new.__dict__[CfgNode._IMMUTABLE] = self.is_frozen()
        return new

    # merging ------------------------------------------------------------------------
    @staticmethod
    def _decode(value: Any) -> Any:
        """YAML strings such as '("voc_2012_train",)' or '(15000, 18000)' become Python literals."""
        if isinstance(value, dict):
            return CfgNode(value)
        if not isinstance(value, str):
            return value
        try:
            return literal_eval(value)
        except (ValueError, SyntaxError):
            return value

    @staticmethod
    def _coerce(new: Any, old: Any, key: str) -> Any:
        if old is None or type(new) is type(old):
            return new
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) for config key: {key}")

    def _merge(self, other: dict, path: List[str]) -> None:
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            v = self._decode(copy.deepcopy(v))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"config key {full} is a section")
                self[k]._merge(v, path + [k])
            else:
                self[k] = self._coerce(v, self[k], full)

    def merge_from_other_cfg(self, other: "CfgNode") -> None:
        self._merge(other, [])

    def merge_from_file(self, filename: str) -> None:
        with open(filename, "r") as f:
            loaded = yaml.safe_load(f) or {}
        self._merge(loaded, [])

    def merge_from_list(self, opts: List[str] | None) -> None:
        opts = opts or []
        if len(opts) % 2 != 0:
            raise ValueError(f"Override list has odd length: {opts}; it must be a list of pairs")
        for full, v in zip(opts[0::2], opts[1::2]):
            node = self
            keys = full.split(".")
            for k in keys[:-1]:
                if k not in node:
                    raise KeyError(f"Non-existent config key: {full}")
                node = node[k]
            if keys[-1] not in node:
                raise KeyError(f"Non-existent config key: {full}")
            node[keys[-1]] = self._coerce(self._decode(v), node[keys[-1]], full)

    def dump(self) -> str:
        def plain(n):
            if isinstance(n, CfgNode):
                return {k: plain(v) for k, v in n.items()}
            if isinstance(n, tuple):
                return list(n)
            return n
        return yaml.safe_dump(plain(self), default_flow_style=None)

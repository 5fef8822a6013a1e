"""``import yolo`` for code written against the reference package (shiboyang/YOLO's ``yolo``): every ``yolo.X.Y`` module
is the SAME module object as ``yolo_amd.X.Y`` (no second copy is executed, so registries are filled once).

    from yolo.config import get_cfg
    from yolo.checkpoint import YOLOV3Checkpointer
    from yolo.modeling import build_model            # registers "YOLOV3", "build_darknet53_fpn_backbone", ...
"""
import importlib
import importlib.abc
import importlib.util
import sys

import yolo_amd as _impl


class _AliasLoader(importlib.abc.Loader):
    """The import machinery stamps the spec it was given (name ``yolo.X``, this loader) on whatever module
    ``create_module`` returns.  The module returned here is the live ``yolo_amd.X``, which must keep its own identity
    (``__spec__``, ``__loader__``, ``__package__``: ``importlib.reload`` and relative imports inside functions depend
    on them), so they are saved before the machinery touches the module and put back in ``exec_module``."""

    _KEEP = ("__spec__", "__loader__", "__package__", "__name__", "__file__", "__path__", "__cached__")

    def __init__(self, target: str):
        self.target = target
        self._own = {}

    def create_module(self, spec):
        mod = importlib.import_module(self.target)    # the existing module object is the module
        self._own = {k: getattr(mod, k) for k in self._KEEP if hasattr(mod, k)}
        return mod

    def exec_module(self, module):                    # already executed under its own name
        for k, v in self._own.items():
            setattr(module, k, v)


class _AliasFinder(importlib.abc.MetaPathFinder):
    """Resolves ``yolo.<anything>`` to ``yolo_amd.<anything>``."""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("yolo."):
            return None
        real = "yolo_amd." + fullname[len("yolo."):]
        try:
            found = importlib.util.find_spec(real)
        except ModuleNotFoundError:
            return None
        if found is None:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real), is_package=found.submodule_search_locations is not None)


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

from yolo import checkpoint, config, data, modeling  # noqa: E402,F401  (what the reference's __init__ imports, and more)

__all__ = ["checkpoint", "config", "data", "modeling"]
__version__ = getattr(_impl, "__version__", "0")

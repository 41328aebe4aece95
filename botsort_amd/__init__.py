"""Importable alias of the package directory `bot-sort-onnx-tensorrt_amd/` (hyphens are not valid
in a Python identifier): `import botsort_amd as bts`."""
import importlib as _importlib
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_pkg = _importlib.import_module("bot-sort-onnx-tensorrt_amd")
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
__all__ = list(_pkg.__all__)

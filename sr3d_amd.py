"""Importable alias of the engine package.

The package directory is called ``3d-sr-micrometeorology_amd`` (mandated repo
layout), which is not a Python identifier; ``import sr3d_amd`` gives the same
module object."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_pkg = importlib.import_module("3d-sr-micrometeorology_amd")
sys.modules[__name__] = _pkg

"""Importable alias of the engine package.

The package directory is called ``3d-sr-micrometeorology_amd`` (mandated repo
layout), which is not a Python identifier; ``import sr3d_amd`` gives the same
module object, and ``sr3d_amd.src.loss_maker`` etc. are the SAME module objects
as ``3d-sr-micrometeorology_amd.src.loss_maker`` (a second copy of a submodule
would have its own classes, and ``isinstance`` checks across the two would fail)."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_REAL = "3d-sr-micrometeorology_amd"
_ALIAS = __name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """resolves ``sr3d_amd.x.y`` to the already imported (or importable) ``3d-sr-micrometeorology_amd.x.y``"""

    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _pkg

"""Importable alias of the package directory ``pharmacophore-diffusion_amd`` (hyphen in the name)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("pharmacophore-diffusion_amd")
sys.modules[__name__] = _pkg

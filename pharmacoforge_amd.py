"""Importable alias of the package directory ``pharmacophore-diffusion_amd`` (hyphen in the name)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("pharmacophore-diffusion_amd")
# one module object per submodule: 'pharmacoforge_amd.x' must be THE module 'pharmacophore-diffusion_amd.x' (classes are
# compared by identity, e.g. isinstance(g, PocketGraph)), not a second copy loaded under the alias name
for _name in ("pocket_io", "dataset", "sharding"):
    importlib.import_module("pharmacophore-diffusion_amd." + _name)
for _k, _m in list(sys.modules.items()):
    if _k.startswith("pharmacophore-diffusion_amd."):
        sys.modules[__name__ + _k[len("pharmacophore-diffusion_amd"):]] = _m
sys.modules[__name__] = _pkg

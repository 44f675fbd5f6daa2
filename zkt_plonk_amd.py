"""Import shim: the package directory is named ``zkt-plonk_amd`` (not a valid Python identifier),
so ``import zkt_plonk_amd`` loads it from there and replaces this module with the package."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "zkt-plonk_amd")
_spec = _u.spec_from_file_location("zkt_plonk_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["zkt_plonk_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Import shim: the product package lives in ``quadraticprogramnetworks.jl_amd/`` (named after
the reference); the dot in that name is not importable, so ``import qpn_amd`` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quadraticprogramnetworks.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "qpn_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["qpn_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Import shim: the package directory is named `cannoles.jl_amd/` (not a valid
Python identifier), so `import cannoles_jl_amd` loads it under this name."""
import importlib.util
import os
import sys

_pkgdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cannoles.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "cannoles_jl_amd", os.path.join(_pkgdir, "__init__.py"), submodule_search_locations=[_pkgdir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cannoles_jl_amd"] = _mod
_spec.loader.exec_module(_mod)

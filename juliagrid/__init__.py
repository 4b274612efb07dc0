"""Import shim: `import juliagrid.jl_amd` resolves to the repo directory `juliagrid.jl_amd/`.

The product package directory carries the repository's name (with a dot), which Python's import
system cannot spell; this namespace makes it importable under its dotted name.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "juliagrid.jl_amd")
_name = __name__ + ".jl_amd"
if _name not in sys.modules:
    _spec = importlib.util.spec_from_file_location(_name, os.path.join(_dir, "__init__.py"),
                                                   submodule_search_locations=[_dir])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_name] = _mod
    _spec.loader.exec_module(_mod)
jl_amd = sys.modules[_name]

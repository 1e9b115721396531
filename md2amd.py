"""Import shim: the package directory is `monodepth2.jl_amd/` (name fixed by the project layout) and a dotted
directory name is not importable with a plain `import`; this registers it as module `monodepth2_jl_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(_ROOT, "monodepth2.jl_amd")
_NAME = "monodepth2_jl_amd"

if _NAME not in sys.modules:
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_PKG, "__init__.py"), submodule_search_locations=[_PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)

pkg = sys.modules[_NAME]
synthetic = pkg.synthetic
_lib = pkg._lib


def __getattr__(name):
    return getattr(pkg, name)

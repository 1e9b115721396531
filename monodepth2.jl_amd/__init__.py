"""monodepth2.jl_amd — MI355X-native implementation of Monodepth2.jl's self-supervised training step.

Import through the repo-root shim `md2amd` (the directory name contains a dot):  `import md2amd as md2`.
`synthetic` is importable without the HIP library; everything in `monodepth` needs libmd2hip.so and a GPU.
"""
from . import synthetic  # noqa: F401
from . import _lib  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must work on the CPU-only build box; using the product API must not silently fall back
    if name in ("monodepth", "parallel", "data", "train"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)

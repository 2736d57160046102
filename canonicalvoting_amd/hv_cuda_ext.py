"""Loader of the COMPILED `hv_cuda` module (csrc/hv_cuda_ext.cpp, built by csrc/build.py:build_ext into
canonicalvoting_amd/_C/hv_cuda.<abi>.so): the pybind / torch extension a maintainer of the reference drops next to
houghvoting/ in place of the CUDA one (houghvoting/src/hv_cuda.cpp:74-77, houghvoting/setup.py:5-10).

    from canonicalvoting_amd import hv_cuda_ext
    hv_cuda = hv_cuda_ext.load()          # module object named "hv_cuda": forward / backward as hv_cuda.cpp:30-71
    hv_cuda_ext.install()                 # or: make `import hv_cuda` resolve to it for the rest of the process

The ctypes module canonicalvoting_amd/hv_cuda.py presents the same two functions (plus the pipeline's extras: the
prefetched grid geometry, the grid origin kept on the returned tensor) and stays the module the in-tree pipeline uses;
tests/test_cabi.py checks the two against each other."""
import importlib.util
import os
import sys

_mod = None


def path():
    from .csrc import build
    return build.ext_path()


def load():
    """the compiled extension module (raises with build instructions when it is missing - there is no fallback here)"""
    global _mod
    if _mod is None:
        p = path()
        if not os.path.exists(p):
            raise ImportError("canonicalvoting_amd: compiled hv_cuda extension %s not found. Build it with "
                              "`python -m canonicalvoting_amd.csrc.build`." % p)
        import torch  # noqa: F401  (libtorch has to be resident before the extension resolves it)
        spec = importlib.util.spec_from_file_location("hv_cuda", p)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        _mod = m
    return _mod


def install():
    """`import hv_cuda` (eval_joint.py:10) resolves to the compiled module from now on"""
    sys.modules["hv_cuda"] = load()
    return sys.modules["hv_cuda"]

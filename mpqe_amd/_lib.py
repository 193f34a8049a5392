"""Loads the gfx950 C-ABI library. There is no CPU or PyTorch fallback: if the
library is missing or lacks a symbol the import fails loudly."""
import ctypes
import os

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libmpqe_amd.so')
_lib = None


def load():
    """MPQE_AMD_LIB (development aid: same-box A/B of builds, tools/ab.sh) names another build of the SAME library to load
    instead of the installed one; it must exist and bind every symbol like the installed one -- there is no fallback."""
    global _lib, LIB_PATH
    if _lib is None:
        if os.environ.get('MPQE_AMD_LIB'):
            LIB_PATH = os.environ['MPQE_AMD_LIB']
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'mpqe_amd: %s not found. Build it with `python -m mpqe_amd.build` '
                '(hipcc --offload-arch=gfx950). There is no fallback path.' % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so). The kernels
        # are launched on torch's streams with torch's allocations, so they must live in
        # THAT runtime: import torch first so the loader resolves our libamdhip64.so.7
        # dependency to the copy torch already mapped (two runtimes in one process make
        # every launch fail with an invalid-handle error).
        import torch  # noqa: F401
        _lib = _capi.bind(ctypes.CDLL(LIB_PATH))
    return _lib

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


_pyhost = None


def load_pyhost():
    """The CPython extension of the drop-in entry points' host path (csrc/host/pyhost.c, built by mpqe_amd.build next to
    the library), bound to this library's mpqe_host_random_choice. Missing = ImportError, like the library itself."""
    global _pyhost
    if _pyhost is None:
        import importlib.util
        from .build import pyhost_path
        path = pyhost_path()
        if not os.path.exists(path):
            raise ImportError('mpqe_amd: %s not found. Build it with `python -m mpqe_amd.build`.' % path)
        spec = importlib.util.spec_from_file_location('mpqe_amd._pyhost', path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        lib = load()
        mod.bind(ctypes.cast(lib.mpqe_host_random_choice, ctypes.c_void_p).value)
        mod.step_fn = ctypes.cast(lib.mpqe_step_forward_backward_ex, ctypes.c_void_p).value
        mod.mt_ok = _mt_selftest(mod)
        _pyhost = mod
    return _pyhost


def _mt_selftest(mod):
    """May _pyhost.choice_mt read this interpreter's Mersenne-Twister state in place? Only if its outputs AND the state it
    leaves behind equal getrandbits' on a copy of the same generator, across regeneration boundaries; otherwise the drop-in
    keeps to _pyhost.choice (raw outputs through getrandbits, a few us more per call)."""
    import random
    import sys
    try:
        import _random
        if sys.implementation.name != 'cpython' or not issubclass(random.Random, _random.Random):
            return False
        mod.mt_bind(_random.Random, 0)
        a = random.Random(20240229)
        b = random.Random()
        b.setstate(a.getstate())
        for n in (1, 7, 616, 1, 700, 1300):
            if a.getrandbits(32 * n).to_bytes(4 * n, 'little') != mod.mt_words(b, n) or a.getstate() != b.getstate():
                return False
        mod.mt_bind(_random.Random, 1)
        return True
    except Exception:          # noqa: BLE001 -- any surprise: the slower path
        return False


_autograd_node = None


def load_autograd_node():
    """The C++ autograd node of the drop-in margin_loss call (csrc/host/autograd_node.cpp, built by mpqe_amd.build), or None
    when it has not been built: mpqe_amd/dropin.py then uses its torch.autograd.Function form (same semantics, ~20 us more
    interpreter time per call and step)."""
    global _autograd_node
    if _autograd_node is None:
        import importlib.util
        from .build import autograd_node_path
        path = autograd_node_path()
        if not os.path.exists(path):
            _autograd_node = False
            return None
        import torch  # noqa: F401
        spec = importlib.util.spec_from_file_location('_autograd_node', path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _autograd_node = mod
    return _autograd_node or None

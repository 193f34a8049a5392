"""Two ways to drive the C ABI from the tests:
  emu -- tests/emu/_build/libmpqe_emu.so, the kernels compiled for the host and run on the
         fiber emulator; arrays are numpy. Checks kernel logic in the GPU-less container.
  hip -- the product library mpqe_amd/lib/libmpqe_amd.so on cuda:0; arrays are torch tensors.
Both go through the same prototypes (mpqe_amd/_capi.py)."""
import ctypes

import numpy as np

from mpqe_amd import _capi


class _Base(object):
    def check(self, st, what='call'):
        _capi.check(self.lib, st, what)


class EmuBackend(_Base):
    name = 'emu'

    def __init__(self):
        from tests.emu.build_emu import build_emu
        self.lib = _capi.bind(ctypes.CDLL(build_emu()))
        self.stream = None

    def put(self, a):
        return np.ascontiguousarray(a).copy()

    def empty(self, shape, dtype=np.float32, fill=None):
        a = np.empty(shape, dtype=dtype)
        if fill is None:
            fill = np.nan if np.issubdtype(np.dtype(dtype), np.floating) else -7
        a.fill(fill)
        return a

    def zeros(self, shape, dtype=np.float32):
        return np.zeros(shape, dtype=dtype)

    def get(self, a):
        return np.array(a)

    def ptr(self, a):
        return None if a is None else a.ctypes.data

    def nbytes(self, n):
        return np.zeros(max(int(n), 1) // 4 + 64, dtype=np.float32)


class HipBackend(_Base):
    name = 'hip'

    def __init__(self):
        import torch
        from mpqe_amd import _lib
        self.torch = torch
        self.lib = _lib.load()
        self.dev = torch.device('cuda:0')

    @property
    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def put(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def empty(self, shape, dtype=np.float32, fill=None):
        if fill is None:
            fill = float('nan') if np.issubdtype(np.dtype(dtype), np.floating) else -7
        t = self.torch.empty(shape, dtype=getattr(self.torch, np.dtype(dtype).name), device=self.dev)
        return t.fill_(fill)

    def zeros(self, shape, dtype=np.float32):
        return self.torch.zeros(shape, dtype=getattr(self.torch, np.dtype(dtype).name), device=self.dev)

    def get(self, t):
        self.torch.cuda.synchronize()
        return t.cpu().numpy()

    def ptr(self, t):
        return None if t is None else t.data_ptr()

    def nbytes(self, n):
        return self.torch.zeros(max(int(n), 1) // 4 + 64, dtype=self.torch.float32, device=self.dev)

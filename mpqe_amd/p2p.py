"""One-hop gradient exchange over peer-mapped buffers (csrc/p2p.hip; SURVEY.md 5 / 8e: "one-hop reduce-scatter + all-gather
using all 7 links concurrently ... over IPC-mapped buffers"). The reference has no distributed code.

    ex = PeerExchange(capacity_floats)            # collective: every rank of the node, same capacity
    ex.bucket[:n].copy_(...)                      # the bucket lives in the communication buffer
    ex.all_reduce(n)                              # stream-ordered: push -> wait for my slots -> reduce my shard -> wait for all shards
    ... = ex.bucket[:n]

`PeerExchange.ok` is False when the buffers could not be set up or the self-test against torch.distributed's all-reduce
failed: callers (mpqe_amd.parallel.StepExchange(transport='p2p')) then keep to RCCL. Every device-side wait is bounded
(~25 s) and reports through the error word, never a hang -- but an exchange in which a wait ran out is INCOMPLETE: its
result must not be used. StepExchange.check() (collective, before the optimiser step) is what notices, agrees between the
ranks, falls back to RCCL for good and raises. EXPERIMENTAL until it has run across two or more GPUs.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _capi, ops


class _Blob(object):
    """A raw device allocation as a CUDA-array-interface object (torch.as_tensor wraps it without copying)."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {'shape': (int(nfloats),), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2}


class PeerExchange(object):
    def __init__(self, capacity, group=None, device=None, self_test=True, err=None):
        """Collective. err: the error word the exchange's bounded waits report into (default: one of its own) -- hand in the
        fused step's, and FusedTrainStep.check() / StepExchange.check() see a peer that never arrived.
        Set-up is three stages with an agreement after each (every rank reaches every collective whatever failed on it
        locally): allocate + export my buffer | exchange handles, map the peers' | self test. A rank that fails a stage
        releases what it holds; `ok` is False everywhere and the caller keeps to RCCL."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.capacity = int(capacity)
        self.epoch = 0
        self.ok = False
        self.reason = None
        self._own = None
        self._mapped = []
        self._bufs = None
        self.bucket = None
        self.err = ops.new_error_word(self.device) if err is None else err
        L = ops.lib()
        # ---- stage 1 (local): my buffer and its IPC handle; every peer pair must be able to map each other
        handle = None
        try:
            so, fo = ctypes.c_int64(), ctypes.c_int64()
            nbytes = L.mpqe_p2p_buffer_bytes(self.capacity, self.world, ctypes.byref(so), ctypes.byref(fo))
            if nbytes == 0:
                raise RuntimeError('mpqe_p2p_buffer_bytes rejected capacity %d / world %d' % (self.capacity, self.world))
            handle = ctypes.create_string_buffer(L.mpqe_p2p_handle_bytes())
            ptr = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                _capi.check(L, L.mpqe_p2p_alloc(nbytes, ctypes.byref(ptr), handle), 'mpqe_p2p_alloc')
            self._own = ptr.value
            stage = True
        except Exception as e:            # noqa: BLE001 -- any failure: the caller keeps to RCCL
            self.reason = '%s: %s' % (type(e).__name__, e)
            stage = False
        if not self._all_agree(stage):
            self.reason = self.reason or 'a peer could not allocate / export its buffer'
            self._release()
            return
        # ---- stage 2: handles (+ each rank's device) to everyone; peer access between every pair of devices; mappings
        try:
            mine = (bytes(handle.raw), self.device.index)
            everyone = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(everyone, mine, group=self.group)
            else:
                everyone = [mine]
            bufs = (ctypes.c_void_p * self.world)()
            for p in range(self.world):
                if p == self.rank:
                    bufs[p] = self._own
                    continue
                pdev = everyone[p][1]
                if pdev != self.device.index and not torch.cuda.can_device_access_peer(self.device.index, pdev):
                    raise RuntimeError('device %d cannot access its peer device %d (hipDeviceCanAccessPeer)'
                                       % (self.device.index, pdev))
                m = ctypes.c_void_p()
                with torch.cuda.device(self.device):
                    _capi.check(L, L.mpqe_p2p_open(everyone[p][0], ctypes.byref(m)), 'mpqe_p2p_open (rank %d)' % p)
                self._mapped.append(m.value)
                bufs[p] = m.value
            self._bufs = bufs
            self.bucket = torch.as_tensor(_Blob(self._own, self.capacity), device=self.device)
            stage = True
        except Exception as e:            # noqa: BLE001
            self.reason = '%s: %s' % (type(e).__name__, e)
            stage = False
        if not self._all_agree(stage):
            self.reason = self.reason or 'a peer could not map the buffers'
            self._release()
            return
        self.ok = True
        # ---- stage 3: one exchange against torch.distributed's all-reduce
        if self_test and self.world > 1:
            self.ok = self._self_test()
            if not self.ok:
                self._release(barrier=True)

    def _release(self, barrier=False):
        """Unmap the peers' buffers and free my own (no kernel of the exchange is in flight: set-up failed, or the self test
        has been synchronised and -- barrier -- every rank is past it)."""
        L = ops.lib()
        if barrier:
            torch.cuda.synchronize(self.device)
            if self.world > 1 and dist.is_initialized():
                dist.barrier(group=self.group)
        for m in self._mapped:
            L.mpqe_p2p_close(m)
        self._mapped = []
        self.bucket = None
        if self._own is not None:
            L.mpqe_p2p_free(self._own)
            self._own = None
        self.ok = False

    def _all_agree(self, flag):
        if self.world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if dist.get_backend(self.group) != 'gloo':
            t = t.to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def _self_test(self):
        """One exchange of a rank-dependent pattern against torch.distributed's all-reduce (ragged size: the last shard is
        short). Collective."""
        n = max(1, min(self.capacity, 100003))
        g = torch.Generator(device='cpu').manual_seed(1234 + self.rank)
        mine = torch.randn(n, generator=g)
        self.bucket[:n].copy_(mine.to(self.device))
        try:
            self.all_reduce(n)
            torch.cuda.synchronize(self.device)
            got = self.bucket[:n].cpu()
            flags = int(self.err.item())
        except Exception as e:            # noqa: BLE001
            self.reason = 'self test: %s: %s' % (type(e).__name__, e)
            return self._all_agree(False)
        ref = mine.clone()
        if dist.get_backend(self.group) == 'gloo':
            dist.all_reduce(ref, group=self.group)
        else:
            r = ref.to(self.device)
            dist.all_reduce(r, group=self.group)
            ref = r.cpu()
        good = flags == 0 and bool(torch.allclose(got, ref, rtol=1e-6, atol=1e-6))
        if not good:
            self.err.zero_()
            self.reason = 'self test: flags %d, max abs diff %.3g' % (flags, float((got - ref).abs().max()))
        return self._all_agree(good)

    def all_reduce(self, n=None):
        """bucket[:n] <- sum over ranks, on the current stream (no host synchronisation)."""
        n = self.capacity if n is None else int(n)
        self.epoch = self.epoch % 0x7fffffff + 1
        L = ops.lib()
        with torch.cuda.device(self.device):
            st = L.mpqe_p2p_allreduce(self._bufs, self.rank, self.world, self.capacity, n, self.epoch, 7, self.err.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
        _capi.check(L, st, 'mpqe_p2p_allreduce')

    def check(self):
        """RuntimeError if a peer did not arrive within the bound in an earlier exchange (one D2H read)."""
        ops.raise_on_flags(self.err)

    def close(self):
        """Collective: nobody unmaps a buffer a peer's kernel may still write."""
        self._release(barrier=True)

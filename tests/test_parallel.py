"""Data-parallel pieces on CPU with the gloo backend, world size 2 (the GPU path uses the same code
over RCCL): the gradient bucket all-reduce equals the single-process gradient of the whole batch,
ranks that touched different parameters still agree on the bucket layout, and graph sharding covers
every query exactly once."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpqe_amd.parallel import GradReducer, shard_slice


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Toy(torch.nn.Module):
    """Stands in for the encoder's parameter set: two 'relation' matrices of which a rank may use
    only one (different formulas touch different relations), plus a shared table."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.w = torch.nn.Parameter(torch.randn(2, 4, 4))
        self.table = torch.nn.Parameter(torch.randn(6, 4))

    def loss(self, ids, rel):
        return (self.table[ids] @ self.w[rel]).pow(2).mean()


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        model = _Toy()
        ids = torch.arange(6)
        lo, hi = shard_slice(6, rank, world)
        red = GradReducer(model)
        # rank 0 only uses relation 0, rank 1 only relation 1 -> w.grad rows differ in support
        model.loss(ids[lo:hi], rank).backward()
        red.all_reduce()
        out[rank] = {k: p.grad.clone() for k, p in model.named_parameters()}
        # second step without the table: its grad must come back as exact zeros on both ranks
        model.zero_grad(set_to_none=True)
        (model.w[rank].sum() * 2.0).backward()
        red.all_reduce()
        out[rank + world] = {k: p.grad.clone() for k, p in model.named_parameters()}
        # third step under zero_grad(set_to_none=False): every p.grad is now a view of the bucket and autograd
        # accumulates into it in place -- the reducer must reduce those values, not clear them first
        model.zero_grad(set_to_none=False)
        model.loss(ids[lo:hi], rank).backward()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(red.params, red.views))
        red.all_reduce()
        out[rank + 2 * world] = {k: p.grad.clone() for k, p in model.named_parameters()}
        # and gradient accumulation over two micro-batches without any zero_grad in between
        model.zero_grad(set_to_none=False)
        model.loss(ids[lo:hi], rank).backward()
        model.loss(ids[lo:hi], rank).backward()
        red.all_reduce()
        out[rank + 3 * world] = {k: p.grad.clone() for k, p in model.named_parameters()}
    finally:
        dist.destroy_process_group()


def test_grad_bucket_allreduce_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    # single-process reference: mean over ranks of each rank's loss
    model = _Toy()
    ids = torch.arange(6)
    total = 0
    for r in range(world):
        lo, hi = shard_slice(6, r, world)
        total = total + model.loss(ids[lo:hi], r) / world
    total.backward()
    for k, p in model.named_parameters():
        np.testing.assert_allclose(out[0][k].numpy(), p.grad.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(out[0][k].numpy(), out[1][k].numpy())
    assert float(out[2]['table'].abs().max()) == 0.0 and float(out[3]['table'].abs().max()) == 0.0
    np.testing.assert_allclose(out[2]['w'].numpy(), np.ones((2, 4, 4), np.float32))   # 2.0 * (1/2) each
    for k, p in model.named_parameters():          # set_to_none=False: same gradient as the first step
        np.testing.assert_allclose(out[4][k].numpy(), p.grad.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(out[4][k].numpy(), out[5][k].numpy())
        np.testing.assert_allclose(out[6][k].numpy(), 2 * p.grad.numpy(), rtol=1e-6, atol=1e-7)   # accumulated twice


@pytest.mark.parametrize('n,world', [(512, 8), (513, 8), (5, 8), (0, 2), (7, 3)])
def test_shard_slice_partitions_exactly(n, world):
    covered = []
    for r in range(world):
        lo, hi = shard_slice(n, r, world)
        assert 0 <= lo <= hi <= n
        covered.extend(range(lo, hi))
    assert covered == list(range(n))


def _sparse_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mpqe_amd.parallel import RowSparseExchange
        g = torch.Generator().manual_seed(100 + rank)
        tabs = [torch.zeros(50, 8), torch.zeros(7, 8)]
        touched = [torch.randint(0, 50, (12,), generator=g), torch.randint(0, 7, (3 + rank,), generator=g)]
        for t, rows in zip(tabs, touched):
            t.index_add_(0, rows, torch.randn(rows.numel(), 8, generator=g))      # this rank's dense local gradient
        dense = [t.clone() for t in tabs]
        for t in dense:
            dist.all_reduce(t)                                                      # what a literal port would do
        ex = RowSparseExchange(tabs)
        ex.exchange(touched)
        out[rank] = ([t.clone() for t in tabs], [t.clone() for t in dense], ex.last_bytes)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_row_sparse_table_exchange_equals_dense_allreduce(world):
    """Touched-row exchange of entity-table gradients == dense all-reduce (sum), on every rank, with far fewer
    bytes on the wire."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sparse_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        got, dense, nbytes = out[r]
        for a, b in zip(got, dense):
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-6, atol=1e-6)    # (four summands: the two sums' orders differ)
        assert 0 < nbytes < 57 * 8 * 4 * world
    for t in range(2):      # replicas agree bit for bit
        for r in range(1, world):
            np.testing.assert_array_equal(out[0][0][t].numpy(), out[r][0][t].numpy())


def test_peer_exchange_kernels_phase_by_phase():
    """csrc/p2p.hip on the CPU emulator, all ranks in ONE process: the buffers are plain host arrays, the phases of the
    exchange run rank after rank (push of every rank, then reduce of every rank, then wait) -- what the kernels index and
    sum, ragged sizes included; the cross-process part (IPC mapping, concurrent kernels) is tests/test_parallel_gpu.py's."""
    import ctypes
    from tests.kernel_backend import EmuBackend
    be = EmuBackend()
    rng = np.random.RandomState(5)
    for world, cap, n in ((2, 1000, 1000), (3, 4099, 4097), (8, 5000, 37), (4, 64, 64)):
        so, fo = ctypes.c_int64(), ctypes.c_int64()
        nbytes = be.lib.mpqe_p2p_buffer_bytes(cap, world, ctypes.byref(so), ctypes.byref(fo))
        assert nbytes > 0 and so.value % 256 == 0 and fo.value % 256 == 0
        raw = [np.zeros(nbytes // 4 + 64, np.float32) for _ in range(world)]
        ptrs = [(r.ctypes.data + 255) // 256 * 256 for r in raw]
        views = [np.frombuffer((ctypes.c_char * nbytes).from_address(p), dtype=np.float32) for p in ptrs]
        data = [rng.randn(n).astype(np.float32) for _ in range(world)]
        for epoch in (1, 2):
            for r in range(world):
                views[r][:n] = data[r] * epoch
                views[r][n:cap] = 7.0                      # beyond n: not part of the exchange
            bufs = (ctypes.c_void_p * world)(*ptrs)
            err = be.zeros((1,), np.int32)
            for phase in (1, 2, 4):
                for r in range(world):
                    be.check(be.lib.mpqe_p2p_allreduce(bufs, r, world, cap, n, epoch, phase, be.ptr(err), be.stream), 'p2p')
            assert int(be.get(err)[0]) == 0
            ref = data[0] * epoch
            for r in range(1, world):
                ref = ref + data[r] * epoch                # rank order: the kernel's order
            for r in range(world):
                np.testing.assert_array_equal(views[r][:n], ref)
                assert (views[r][n:cap] == 7.0).all()


def test_peer_exchange_late_rank_is_flagged_not_waited_for_forever():
    """World 8, one rank never arrives (its push is missing): every wait on it runs out of its bound, ORs
    MPQE_FLAG_INTERNAL | 0x4000 into the error word and returns -- never a hang --; the waiting rank does NOT publish a sum
    over a missing slot (its shard stays as it was everywhere, its 'shard landed' flag is never raised), and the ranks that
    wait for that shard are flagged too. What the host does with the word -- StepExchange.check(): agree between the ranks,
    fall back to RCCL for good, raise -- is in mpqe_amd/parallel.py; this pins the device half on the CPU emulator."""
    import ctypes
    from mpqe_amd import _capi
    from tests.kernel_backend import EmuBackend
    be = EmuBackend()
    world, cap, n, late = 8, 4096, 4001, 5
    so, fo = ctypes.c_int64(), ctypes.c_int64()
    nbytes = be.lib.mpqe_p2p_buffer_bytes(cap, world, ctypes.byref(so), ctypes.byref(fo))
    raw = [np.zeros(nbytes // 4 + 64, np.float32) for _ in range(world)]
    ptrs = [(r.ctypes.data + 255) // 256 * 256 for r in raw]
    views = [np.frombuffer((ctypes.c_char * nbytes).from_address(p), dtype=np.float32) for p in ptrs]
    rng = np.random.RandomState(9)
    data = [rng.randn(n).astype(np.float32) for _ in range(world)]
    for r in range(world):
        views[r][:n] = data[r]
    bufs = (ctypes.c_void_p * world)(*ptrs)
    errs = [be.zeros((1,), np.int32) for _ in range(world)]
    for r in range(world):
        if r != late:
            be.check(be.lib.mpqe_p2p_allreduce(bufs, r, world, cap, n, 1, 1, be.ptr(errs[r]), be.stream), 'push')
    for r in range(world):
        if r != late:
            be.check(be.lib.mpqe_p2p_allreduce(bufs, r, world, cap, n, 1, 2, be.ptr(errs[r]), be.stream), 'reduce')
    want = _capi.FLAG_INTERNAL | 0x4000
    for r in range(world):
        if r != late:       # every rank waited for the late rank's slot of its own shard: flagged, nothing published
            assert int(be.get(errs[r])[0]) & want == want, (r, int(be.get(errs[r])[0]))
            np.testing.assert_array_equal(views[r][:n], data[r])
    be.check(be.lib.mpqe_p2p_allreduce(bufs, 0, world, cap, n, 1, 4, be.ptr(errs[0]), be.stream), 'wait')      # returns
    assert int(be.get(errs[0])[0]) & want == want

"""TEST INFRASTRUCTURE ONLY -- stand-ins for the third-party modules the
reference imports but which are neither under /root/reference nor installed
here (no network): PyTorch Geometric (unpinned; API evidence dates it to
~1.3-1.4) and torch_scatter (unpinned, ~1.4-2.0).

Call sites in the reference that these satisfy:
  mpqe/model.py:199      from torch_scatter import scatter_add, scatter_max, scatter_mean, scatter_min
  mpqe/model.py:201      from torch_geometric.nn.conv import MessagePassing
  mpqe/model.py:203      from torch_geometric.nn import inits
  mpqe/data_utils.py:9   from torch_geometric.data import Data, Batch

Semantics restated from the libraries' published documentation:
  * MessagePassing(aggr).propagate(edge_index, **kw): flow source_to_target;
    a `message` argument named foo_j receives kw['foo'].index_select(0, edge_index[0]),
    foo_i receives index_select(0, edge_index[1]), other names pass through;
    messages are scattered over edge_index[1] with dim_size = x.size(0);
    `update(aggr_out, ...)` receives remaining args by name.
  * inits.uniform(size, tensor): tensor ~ U(-1/sqrt(size), 1/sqrt(size)); no-op for None.
  * Data: attribute bag with .to(device); Batch.from_data_list: concatenates
    edge_index along dim -1 with a cumulative num_nodes offset, other
    attributes along dim 0, and adds `batch` = graph id repeated num_nodes times.
  * scatter_add(src, index, dim, dim_size): zeros + index_add;
    scatter_mean: add / clamp(count, 1); scatter_max/min: (values, argindex).

Nothing in the product path (mpqe_amd/) may import this file.
"""
import inspect
import math
import sys
import types

import numpy as np
import torch


# --------------------------------------------------------------------------- torch_scatter
def _dim_size(index, dim_size):
    if dim_size is not None:
        return int(dim_size)
    return int(index.max().item()) + 1 if index.numel() > 0 else 0


def scatter_add(src, index, dim=0, out=None, dim_size=None, fill_value=0):
    assert dim == 0
    n = _dim_size(index, dim_size)
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return res.index_add(0, index, src)


def scatter_mean(src, index, dim=0, out=None, dim_size=None, fill_value=0):
    assert dim == 0
    n = _dim_size(index, dim_size)
    total = scatter_add(src, index, dim, dim_size=n)
    count = torch.zeros(n, dtype=src.dtype, device=src.device).index_add(
        0, index, torch.ones_like(index, dtype=src.dtype))
    count = count.clamp(min=1)
    return total / count.view((-1,) + (1,) * (src.dim() - 1))


class _ScatterArg(torch.autograd.Function):
    """torch_scatter's scatter_max / scatter_min: values + arg index, and the BACKWARD torch_scatter has -- each output
    element's gradient goes to its arg row only (grad_src = grad_out gathered where arg == row). torch's own
    scatter_reduce('amax') backward, which this stand-in relied on until round 5, splits the gradient evenly between rows that
    tie exactly; torch_scatter does not. Identical wherever no two rows of a segment hold the same value."""

    @staticmethod
    def forward(ctx, src, index, n, reduce):
        res, arg = _scatter_arg_values(src.detach(), index, n, reduce)
        ctx.save_for_backward(arg)
        ctx.rows = src.shape[0]
        ctx.mark_non_differentiable(arg)
        return res, arg

    @staticmethod
    def backward(ctx, gres, _garg):
        arg, = ctx.saved_tensors
        g = torch.zeros((ctx.rows + 1,) + tuple(gres.shape[1:]), dtype=gres.dtype, device=gres.device)
        g.scatter_add_(0, torch.where(arg >= 0, arg, torch.full_like(arg, ctx.rows)), gres)
        return g[:ctx.rows], None, None, None


def _scatter_arg(src, index, dim, dim_size, reduce):
    assert dim == 0
    return _ScatterArg.apply(src, index, _dim_size(index, dim_size), reduce)


def _scatter_arg_values(src, index, n, reduce):
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    init = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    res = init.scatter_reduce(0, idx, src, reduce=reduce, include_self=False)
    # arg index: lowest source row attaining the extremum (torch_scatter's CPU
    # loop updates on strict improvement only). Rows with no source get -1 in
    # the argindex here; torch_scatter versions differ on that sentinel and
    # the reference discards the argindex (model.py:384-385, 512-513).
    arg = torch.full(res.shape, -1, dtype=torch.long, device=src.device)
    hit = src == res.index_select(0, index)
    rows = torch.arange(src.shape[0], device=src.device).view(
        (-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    big = torch.full_like(rows, src.shape[0])
    cand = torch.where(hit, rows, big)
    first = torch.full(res.shape, src.shape[0], dtype=torch.long, device=src.device)
    first = first.scatter_reduce(0, idx, cand, reduce='amin', include_self=True)
    arg = torch.where(first < src.shape[0], first, arg)
    return res, arg


def scatter_max(src, index, dim=0, out=None, dim_size=None, fill_value=None):
    return _scatter_arg(src, index, dim, dim_size, 'amax')


def scatter_min(src, index, dim=0, out=None, dim_size=None, fill_value=None):
    return _scatter_arg(src, index, dim, dim_size, 'amin')


# --------------------------------------------------------------------------- torch_geometric.nn
class MessagePassing(torch.nn.Module):
    def __init__(self, aggr='add', flow='source_to_target'):
        super().__init__()
        assert aggr in ('add', 'mean', 'max')
        assert flow == 'source_to_target'
        self.aggr = aggr
        self._msg_args = inspect.getfullargspec(self.message)[0][1:]
        self._upd_args = inspect.getfullargspec(self.update)[0][2:]

    def propagate(self, edge_index, size=None, **kwargs):
        dim_size = None
        msg_in = []
        for name in self._msg_args:
            if name.endswith('_j') or name.endswith('_i'):
                t = kwargs[name[:-2]]
                sel = 0 if name.endswith('_j') else 1
                if t is not None:
                    dim_size = t.size(0)
                    t = t.index_select(0, edge_index[sel])
                msg_in.append(t)
            else:
                msg_in.append(kwargs[name])
        if dim_size is None:
            dim_size = int(edge_index.max().item()) + 1
        out = self.message(*msg_in)
        if self.aggr == 'add':
            out = scatter_add(out, edge_index[1], 0, dim_size=dim_size)
        elif self.aggr == 'mean':
            out = scatter_mean(out, edge_index[1], 0, dim_size=dim_size)
        else:
            out = scatter_max(out, edge_index[1], 0, dim_size=dim_size)[0]
        return self.update(out, *[kwargs[n] for n in self._upd_args])

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


def _uniform(size, tensor):
    bound = 1.0 / math.sqrt(size)
    if tensor is not None:
        tensor.data.uniform_(-bound, bound)


# --------------------------------------------------------------------------- torch_geometric.data
class Data(object):
    def __init__(self, x=None, edge_index=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


class Batch(Data):
    @staticmethod
    def from_data_list(data_list):
        batch = Batch()
        keys = [k for k, v in data_list[0].__dict__.items() if torch.is_tensor(v)]
        cols = {k: [] for k in keys}
        ids = []
        offset = 0
        for g, d in enumerate(data_list):
            n = int(d.num_nodes)
            for k in keys:
                v = getattr(d, k)
                cols[k].append(v + offset if k == 'edge_index' else v)
            ids.append(torch.full((n,), g, dtype=torch.long))
            offset += n
        for k in keys:
            setattr(batch, k, torch.cat(cols[k], dim=-1 if k == 'edge_index' else 0))
        batch.batch = torch.cat(ids, dim=0)
        batch.num_nodes = offset
        return batch


def install():
    """Put the stand-ins into sys.modules so `import mpqe.model` resolves, and
    restore the `np.int` alias data_utils.py:382,392 relies on (removed in
    numpy >= 1.24)."""
    if not hasattr(np, 'int'):
        np.int = int
    ts = types.ModuleType('torch_scatter')
    ts.scatter_add, ts.scatter_max = scatter_add, scatter_max
    ts.scatter_mean, ts.scatter_min = scatter_mean, scatter_min
    tg = types.ModuleType('torch_geometric')
    tg_nn = types.ModuleType('torch_geometric.nn')
    tg_conv = types.ModuleType('torch_geometric.nn.conv')
    tg_inits = types.ModuleType('torch_geometric.nn.inits')
    tg_data = types.ModuleType('torch_geometric.data')
    tg_conv.MessagePassing = MessagePassing
    tg_inits.uniform = _uniform
    tg_nn.conv, tg_nn.inits = tg_conv, tg_inits
    tg_data.Data, tg_data.Batch = Data, Batch
    tg.nn, tg.data = tg_nn, tg_data
    for name, mod in [('torch_scatter', ts), ('torch_geometric', tg),
                      ('torch_geometric.nn', tg_nn),
                      ('torch_geometric.nn.conv', tg_conv),
                      ('torch_geometric.nn.inits', tg_inits),
                      ('torch_geometric.data', tg_data)]:
        sys.modules.setdefault(name, mod)
    # sacred is imported by mpqe/train_helpers.py only; not needed for model/data_utils.

"""Optimiser step of the training loop (reference train.py:83-88) on flat buffers.

The reference builds `optim.Adam(filter(requires_grad, params), lr=lr)` (or `optim.SGD(..., momentum=0)`) over
EVERY parameter, the dense entity tables included. `FusedTrainStep` already keeps all gradients in one flat
fp32 buffer; `FlatOptimizer` re-homes the parameters into one flat buffer too (every `p.data` becomes a
view of it: names, shapes and `state_dict()` are unchanged) and updates everything with one launch of
`mpqe_adam_step` / `mpqe_sgd_step` -- the update rule of torch.optim.Adam / SGD at the reference's settings.

    step = FusedTrainStep(model)
    opt = FlatOptimizer(step, lr=0.01, opt='adam')
    loss = step.run(packed); opt.step()

Row-sparse tables. Dense Adam over the entity tables streams 16 bytes per table element per step
whether or not a row was touched (AM-sized tables: 191 MB, 1M entities at D = 256: 1 GB, against a few thousand touched
rows), and needs the whole table gradient zero-filled first. `FlatOptimizer(step, sparse_tables=True)` with
`FusedTrainStep(model, sparse_tables=True)` updates ONLY the rows the step's ids touched (`mpqe_adam_rows_step`, the
rule of torch.optim.SparseAdam on the per-row gradient sums) and everything else densely as before:

    step = FusedTrainStep(model, sparse_tables=True)
    opt = FlatOptimizer(step, lr=0.01, sparse_tables=True)
    loss = step.run(packed); opt.step(packed)          # the packed step's touch plan names the rows

Deviation from the reference's dense Adam, stated: a row no step touches keeps its moments undecayed and does not move
(dense Adam decays every row's m, v every step and keeps moving rows whose m is non-zero).
"""
import ctypes

import torch

from . import _capi, ops


class FlatOptimizer(object):
    def __init__(self, fused_step, lr=0.01, opt='adam', betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 sparse_tables=False):
        if opt not in ('adam', 'sgd'):
            raise ValueError('opt must be adam or sgd')           # reference train.py:83-88
        if sparse_tables and (opt != 'adam' or weight_decay != 0.0):
            raise ValueError('sparse_tables: Adam without weight decay (torch.optim.SparseAdam has none either)')
        if sparse_tables and not fused_step.sparse_tables:
            raise ValueError('sparse_tables needs FusedTrainStep(model, sparse_tables=True)')
        self.sparse_tables = bool(sparse_tables)
        self.fused = fused_step
        self.lr, self.opt, self.betas, self.eps, self.weight_decay = float(lr), opt, betas, float(eps), float(weight_decay)
        self.t = 0
        params = fused_step.params
        dev = fused_step.device
        total = fused_step.flat_grad.numel()
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        offsets = {}
        with torch.no_grad():
            for p in params:                                      # same order as the gradient views
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + n].view(p.shape)
                offsets[id(p)] = (off, n)
                off += n
        fused_step._refresh_pointers()                             # the parameters moved
        self.exp_avg = torch.zeros_like(self.flat_param) if opt == 'adam' else None
        self.exp_avg_sq = torch.zeros_like(self.flat_param) if opt == 'adam' else None
        # sparse tables: the entity tables' slices of the flat buffers (row-sparse update) and the runs in between
        # (dense update, one launch per run: the tables are the first parameters, so normally one run)
        self.table_slices, self.dense_runs = [], [(0, total)]
        if self.sparse_tables:
            m = fused_step.model
            self.table_slices = [offsets[id(m.enc.table(mode))] for mode in fused_step.modes]
            taken = sorted(self.table_slices)
            self.dense_runs, cur = [], 0
            for o, n in taken:
                if o > cur:
                    self.dense_runs.append((cur, o - cur))
                cur = o + n
            if cur < total:
                self.dense_runs.append((cur, total - cur))
            arr = ctypes.c_void_p * len(self.table_slices)
            esz = 4
            self._tab_p = arr(*[self.flat_param.data_ptr() + esz * o for o, _ in self.table_slices])
            self._tab_g = arr(*[fused_step.flat_grad.data_ptr() + esz * o for o, _ in self.table_slices])
            self._tab_m = arr(*[self.exp_avg.data_ptr() + esz * o for o, _ in self.table_slices])
            self._tab_v = arr(*[self.exp_avg_sq.data_ptr() + esz * o for o, _ in self.table_slices])

    def step(self, packed=None, rows_plan=None):
        """packed: with sparse_tables, the packed step whose gradients are being applied (its touch plan lists the
        table rows to update). rows_plan: under data parallelism the (plan pointer, entries) of the row exchange
        (mpqe_amd.parallel.StepExchange): the rows ANY rank touched -- every replica updates the same rows."""
        self.t += 1
        self.fused.param_epoch += 1            # (the parameters are written behind autograd's version counters: dropin.py's lanes)
        g = self.fused.flat_grad
        with torch.cuda.device(self.fused.device):
            stream = torch.cuda.current_stream().cuda_stream
            if self.opt == 'adam' and self.sparse_tables:
                if packed is None or packed.touch_ptr is None:
                    raise ValueError('sparse_tables: step(packed) needs the packed step (with its touch plan)')
                L = ops.lib()
                st = 0
                for o, n in self.dense_runs:
                    st = st or L.mpqe_adam_step(self.flat_param.data_ptr() + 4 * o, g.data_ptr() + 4 * o,
                                                self.exp_avg.data_ptr() + 4 * o, self.exp_avg_sq.data_ptr() + 4 * o, n,
                                                self.lr, self.betas[0], self.betas[1], self.eps, 0.0, self.t, stream)
                plan_ptr, entries = rows_plan if rows_plan is not None else (packed.touch_ptr, packed.touch_entries)
                st = st or L.mpqe_adam_rows_step(plan_ptr, entries, self._tab_p, self._tab_g,
                                                 self._tab_m, self._tab_v, len(self.table_slices),
                                                 self.fused.model.emb_dim, self.lr, self.betas[0], self.betas[1],
                                                 self.eps, self.t, stream)
            elif self.opt == 'adam':
                st = ops.lib().mpqe_adam_step(self.flat_param.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), g.numel(), self.lr, self.betas[0],
                                              self.betas[1], self.eps, self.weight_decay, self.t, stream)
            else:
                st = ops.lib().mpqe_sgd_step(self.flat_param.data_ptr(), g.data_ptr(), g.numel(), self.lr,
                                             self.weight_decay, stream)
        _capi.check(ops.lib(), st, 'optimiser step')

    def zero_grad(self, set_to_none=True):
        """torch.optim's call (reference train_helpers.py:78) for the drop-in entry points (mpqe_amd/dropin.py): the next
        backward pass's fused step zero-fills the flat gradient buffer itself -- nothing is written here, and no p.grad is
        touched (re-binding ~30 parameters' .grad costs more interpreter time than the step's call)."""
        self.fused.zero_next = True

    def state_dict(self):
        return {'t': self.t, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq}

    def load_state_dict(self, sd):
        self.t = int(sd['t'])
        if self.opt == 'adam':
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])


# ------------------------------------------------------------------------------------------------------------------------
# torch.optim's constructors, as the reference's training script calls them (train.py:83-88):
#     optimizer = optim.SGD([p for p in enc_dec.parameters() if p.requires_grad], lr=args.lr, momentum=0)
#     optimizer = optim.Adam([p for p in enc_dec.parameters() if p.requires_grad], lr=args.lr)
# `from mpqe_amd import optim` in place of `from torch import optim` keeps those lines as they are: when the parameters are
# exactly those of ONE model on the fused step (mpqe_amd/dropin.py) the update is FlatOptimizer's one launch over the flat
# buffers (and zero_grad() a flag), otherwise the torch optimiser itself.
import weakref

_OWNERS = weakref.WeakValueDictionary()           # id(parameter) -> its model (mpqe_amd/model.py registers them)


def register_model(model):
    for p in model.parameters():
        _OWNERS[id(p)] = model


class _Switch(object):
    def __init__(self, kind, params, torch_cls, lr, kwargs):
        params = list(params)
        if params and isinstance(params[0], dict):
            raise ValueError('mpqe_amd.optim.%s takes a flat list of parameters (the reference builds no parameter groups); '
                             'use torch.optim.%s for groups' % (torch_cls.__name__, torch_cls.__name__))
        self._impl, self._flat = None, False
        owners = {id(_OWNERS.get(id(p))) for p in params}
        model = _OWNERS.get(id(params[0])) if params and len(owners) == 1 else None
        d = model.dropin() if model is not None and hasattr(model, 'dropin') else None
        plain = ((kind == 'adam' and not kwargs.get('amsgrad') and not kwargs.get('maximize'))
                 or (kind == 'sgd' and not kwargs.get('momentum') and not kwargs.get('nesterov') and not kwargs.get('dampening')
                     and not kwargs.get('maximize')))
        known = {'betas', 'eps', 'weight_decay', 'momentum', 'nesterov', 'dampening', 'amsgrad', 'maximize'}
        if (d is not None and plain and set(kwargs) <= known and len(params) == len(d.step.params)
                and {id(p) for p in params} == {id(p) for p in d.step.params}):
            self._impl = FlatOptimizer(d.step, lr=lr, opt=kind, betas=kwargs.get('betas', (0.9, 0.999)),
                                       eps=kwargs.get('eps', 1e-8), weight_decay=kwargs.get('weight_decay', 0.0))
            self._flat = True
            self.param_groups = [dict(params=params, lr=lr)]
        else:
            self._impl = torch_cls(params, lr=lr, **kwargs)
            self.param_groups = self._impl.param_groups

    @property
    def flat(self):
        """True: one launch over the model's flat buffers; False: the torch optimiser."""
        return self._flat

    def zero_grad(self, set_to_none=True):
        self._impl.zero_grad(set_to_none=set_to_none) if not self._flat else self._impl.zero_grad()

    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('closures are not supported')
        if self._flat:
            self._impl.lr = float(self.param_groups[0]['lr'])        # (a scheduler writes param_groups[0]['lr'])
        self._impl.step()

    def state_dict(self):
        return self._impl.state_dict()

    def load_state_dict(self, sd):
        self._impl.load_state_dict(sd)


class Adam(_Switch):
    """torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) at the reference's call."""

    def __init__(self, params, lr=1e-3, **kwargs):
        super(Adam, self).__init__('adam', params, torch.optim.Adam, lr, kwargs)


class SGD(_Switch):
    """torch.optim.SGD(params, lr, momentum=0, weight_decay=0) at the reference's call."""

    def __init__(self, params, lr=1e-3, **kwargs):
        super(SGD, self).__init__('sgd', params, torch.optim.SGD, lr, kwargs)

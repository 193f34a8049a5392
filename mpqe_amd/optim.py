"""Optimiser step of the training loop (reference train.py:83-88) on flat buffers.

The reference builds `optim.Adam(filter(requires_grad, params), lr=lr)` (or `optim.SGD(..., momentum=0)`) over
EVERY parameter, the dense entity tables included. `FusedTrainStep` already keeps all gradients in one flat
fp32 buffer; `FlatOptimizer` re-homes the parameters into one flat buffer too (every `p.data` becomes a
view of it: names, shapes and `state_dict()` are unchanged) and updates everything with one launch of
`mpqe_adam_step` / `mpqe_sgd_step` -- the update rule of torch.optim.Adam / SGD at the reference's settings.

    step = FusedTrainStep(model)
    opt = FlatOptimizer(step, lr=0.01, opt='adam')
    loss = step.run(packed); opt.step()
"""
import torch

from . import _capi, ops


class FlatOptimizer(object):
    def __init__(self, fused_step, lr=0.01, opt='adam', betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if opt not in ('adam', 'sgd'):
            raise ValueError('opt must be adam or sgd')           # reference train.py:83-88
        self.fused = fused_step
        self.lr, self.opt, self.betas, self.eps, self.weight_decay = float(lr), opt, betas, float(eps), float(weight_decay)
        self.t = 0
        params = fused_step.params
        dev = fused_step.device
        total = fused_step.flat_grad.numel()
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:                                      # same order as the gradient views
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + n].view(p.shape)
                off += n
        fused_step._refresh_pointers()                             # the parameters moved
        self.exp_avg = torch.zeros_like(self.flat_param) if opt == 'adam' else None
        self.exp_avg_sq = torch.zeros_like(self.flat_param) if opt == 'adam' else None

    def step(self):
        self.t += 1
        g = self.fused.flat_grad
        with torch.cuda.device(self.fused.device):
            stream = torch.cuda.current_stream().cuda_stream
            if self.opt == 'adam':
                st = ops.lib().mpqe_adam_step(self.flat_param.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), g.numel(), self.lr, self.betas[0],
                                              self.betas[1], self.eps, self.weight_decay, self.t, stream)
            else:
                st = ops.lib().mpqe_sgd_step(self.flat_param.data_ptr(), g.data_ptr(), g.numel(), self.lr,
                                             self.weight_decay, stream)
        _capi.check(ops.lib(), st, 'optimiser step')

    def state_dict(self):
        return {'t': self.t, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq}

    def load_state_dict(self, sd):
        self.t = int(sd['t'])
        if self.opt == 'adam':
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])

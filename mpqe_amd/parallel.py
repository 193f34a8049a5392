"""Data parallelism for the training step: one process per GPU, query graphs sharded by
graph, replicas of every parameter, and ONE all-reduce (sum) of the flattened gradients per
step over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests). The reference has no
distributed code at all; the encoder's forward/backward needs no communication because query
graphs never interact -- the only coupling is the mean in the hinge loss, which
the 1/world scale restores.

Ranks may touch different parameters in a step (different formulas use different relation
matrices and entity tables), so every parameter takes part in the bucket with an implicit zero
gradient; the bucket layout is therefore identical on all ranks by construction.
"""
import torch
import torch.distributed as dist


class GradReducer(object):
    def __init__(self, model, group=None, average=True):
        self.group = group
        self.params = [p for p in model.parameters() if p.requires_grad]
        # shared layers appear once in .parameters(); keep that de-duplication
        self.numel = [p.numel() for p in self.params]
        total = sum(self.numel)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p, n in zip(self.params, self.numel):
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.scale = 1.0 / self.world if average else 1.0

    def bucket_bytes(self):
        return self.flat.numel() * 4

    def all_reduce(self):
        """grad <- (1/world) * sum over ranks of grad, for every parameter."""
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        self.flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.scale != 1.0:
            self.flat.mul_(self.scale)
        for v, p in zip(self.views, self.params):
            p.grad = v
        return self.flat


def shard_slice(n_items, rank, world):
    """Contiguous [lo, hi) slice of a formula batch for `rank` (graph sharding)."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)

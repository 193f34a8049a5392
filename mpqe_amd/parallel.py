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
        """grad <- (1/world) * sum over ranks of grad, for every parameter. After the first call every p.grad IS
        its view of the bucket, and autograd (zero_grad(set_to_none=False), gradient accumulation) keeps
        accumulating into it in place: such gradients are already where the all-reduce reads them and must
        not be cleared; only parameters without a gradient contribute zeros, and gradients autograd allocated
        elsewhere are copied in."""
        copy_to, copy_from = [], []
        for v, p in zip(self.views, self.params):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr() or g.shape != v.shape:
                copy_to.append(v)
                copy_from.append(g)
        if copy_to:
            torch._foreach_copy_(copy_to, copy_from)
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


class RowSparseExchange(object):
    """Entity-table gradients under data parallelism WITHOUT a dense all-reduce.

    The reference's tables are dense nn.Embedding parameters (data_utils.py:31), so a literal port would
    all-reduce the whole table every step: 191 MB at AM size, 1 GB for the 1M-entity KG, against at most
    B * (A + 2) touched rows per batch. Here every rank sends only the rows it touched: all-gather of
    (row id, gradient row) pairs, then every rank sums the gathered rows in rank order -- the result equals
    the dense all-reduce (sum), bit-identical on every rank; rows nobody touched stay zero.

        ex = RowSparseExchange([table_a.grad, table_b.grad])          # dense per-rank gradient buffers
        ex.exchange([rows_touched_in_a, rows_touched_in_b])           # int64 row ids (duplicates allowed)
    """

    def __init__(self, table_grads, group=None, scale=1.0):
        self.grads = list(table_grads)
        self.group = group
        self.scale = float(scale)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dim = self.grads[0].shape[1]
        for g in self.grads:
            if g.dim() != 2 or g.shape[1] != self.dim or g.dtype != torch.float32:
                raise ValueError('table gradients must be fp32 [rows, dim] with one common dim')
        self.last_bytes = 0

    def exchange(self, touched):
        dev = self.grads[0].device
        ids, vals = [], []
        for t, (g, rows) in enumerate(zip(self.grads, touched)):
            rows = torch.unique(torch.as_tensor(rows, dtype=torch.long, device=dev))
            if rows.numel() and (int(rows.min()) < 0 or int(rows.max()) >= g.shape[0]):
                raise IndexError('touched row outside the table')
            ids.append(rows + (t << 40))                      # table number in the high bits
            vals.append(g.index_select(0, rows))
        ids = torch.cat(ids) if ids else torch.zeros(0, dtype=torch.long, device=dev)
        vals = torch.cat(vals) if vals else torch.zeros(0, self.dim, device=dev)
        if self.scale != 1.0:
            for g in self.grads:
                g.mul_(self.scale)
            vals = vals * self.scale
        if self.world == 1:
            return
        n = torch.tensor([ids.numel()], dtype=torch.long, device=dev)
        counts = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(counts, n, group=self.group)
        counts = [int(c.item()) for c in counts]
        cap = max(max(counts), 1)
        pad_ids = torch.zeros(cap, dtype=torch.long, device=dev)
        pad_vals = torch.zeros(cap, self.dim, dtype=torch.float32, device=dev)
        pad_ids[:ids.numel()] = ids
        pad_vals[:vals.shape[0]] = vals
        all_ids = [torch.empty_like(pad_ids) for _ in range(self.world)]
        all_vals = [torch.empty_like(pad_vals) for _ in range(self.world)]
        dist.all_gather(all_ids, pad_ids, group=self.group)
        dist.all_gather(all_vals, pad_vals, group=self.group)
        self.last_bytes = self.world * cap * (8 + 4 * self.dim)
        # every rank rebuilds the touched rows from zero in rank order (its own rows included, from the gathered
        # copy): the same additions in the same order everywhere, so the replicas stay bit-identical
        own_tab = ids >> 40
        for t, g in enumerate(self.grads):
            sel = own_tab == t
            if bool(sel.any()):
                g.index_fill_(0, ids[sel] & ((1 << 40) - 1), 0.0)
        for r in range(self.world):
            if counts[r] == 0:
                continue
            rid, rv = all_ids[r][:counts[r]], all_vals[r][:counts[r]]
            tab = rid >> 40
            for t, g in enumerate(self.grads):
                sel = tab == t
                if bool(sel.any()):
                    g.index_add_(0, rid[sel] & ((1 << 40) - 1), rv[sel])

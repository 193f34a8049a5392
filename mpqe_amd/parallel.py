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


def _touch_bits(v):
    b = 1
    while (1 << b) <= v and b < 40:
        b += 1
    return b


class ExchangePlan(object):
    __slots__ = ('form', 'grad_views', 'bucket', 'bucket_views', 'bucket_bytes', 'rows', 'n_own', 'cap', 'gidx', 'send',
                 'recv', 'plan', 'plan_ptr', 'entries', 'wire_bytes', 'union', 'desc_hash', 'send_keys', 'all_keys', 'plan_ws',
                 'plan_sizes', 'in_step', 'spans_in', 'spans_out', 'span_blocks', 'nspans')


class StepExchange(object):
    """The per-step gradient exchange of FusedTrainStep under data parallelism (one process per GPU, query graphs
    sharded by rank, replicas of every parameter). The reference has no distributed code; a literal
    data-parallel port would all-reduce EVERY parameter's dense gradient: 19 MB per step for the AIFB model, of which
    a step touches ~35 of 270 relation matrices, and 191 MB - 1 GB of entity tables at the AM / 1M-entity sizes. Here:

      * relation matrices: which (layer, relation) matrices a rank's step touches is known from its formulas; the
        ranks exchange those lists ONCE per formula set (plan(packed, key=...): cached under the caller's key, no
        collective and no host work when the set recurs), and only the union -- with the root matrices, biases and mode
        rows -- goes through ONE all-reduce (sum; the 1 / world of the mean is already in the batch weights): copied
        into a contiguous bucket and back ('bucket' form), or, when the union is most of the gradient anyway (>= 60 %:
        many ranks, few relations), all-reduced IN PLACE in the flat gradient buffer with no copy at all ('dense'
        form). Matrices no rank touched are zero on every rank and stay home.
      * entity tables: small tables (<= 32 MB in all: AIFB 1.3 MB, MUTAG 23 MB) ride in the same all-reduce, dense.
        Large ones (AM 191 MB, 1M entities 1 GB) exchange only rows: every rank all-gathers the (table, row) keys of
        its touch plan (a plan built at pack time: FusedTrainStep(touch='pack')) and all build the same plan over all
        of them (mpqe_rows_plan_build); per step a rank all-gathers its gradient rows and mpqe_table_rows_sum adds the
        gathered rows per key in plan order -- equal to the dense all-reduce, the same additions in the same order on
        every rank.
      No host synchronisation and no object collectives in reduce(): one all-reduce (+ one all-gather and one kernel
      with the row exchange).

        ex = StepExchange(fused_step)
        plan = ex.plan(packed, key=formula_set_id)     # collective on a key's first use
        loss = fused_step.run(packed); ex.reduce(plan); optimizer.step(packed, rows_plan=ex.rows_plan(plan))
    """
    DENSE_TABLE_BYTES = 32 << 20
    DENSE_FRACTION = 0.6

    def __init__(self, fused_step, group=None, tables='auto', transport='rccl'):
        """transport: how the bucket is summed -- 'rccl': torch.distributed's all-reduce (RCCL over xGMI; gloo in the
        one-GPU tests); 'p2p': the library's one-hop reduce-scatter + all-gather over peer-mapped buffers
        (mpqe_amd/p2p.py, csrc/p2p.hip: every link carries one shard each way instead of a ring's 2 (w - 1) serial hops),
        set up and self-tested against the all-reduce here (collective), falling back to 'rccl' -- `transport_note` says
        why -- if the buffers cannot be mapped or the self-test fails."""
        self.fused = fused_step
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        m = fused_step.model
        self.D = m.emb_dim
        self.dev = fused_step.device
        self.off = {}
        o = 0
        for p in fused_step.params:
            self.off[id(p)] = o
            o += p.numel()
        self.tables = [m.enc.table(mode) for mode in fused_step.modes]
        self.table_ids = set(id(t) for t in self.tables)
        # the tables as ONE [rows, D] view of the flat gradient buffer (they are the first parameters)
        lo = min(self.off[id(t)] for t in self.tables)
        hi = max(self.off[id(t)] + t.numel() for t in self.tables)
        if hi - lo != sum(t.numel() for t in self.tables):
            raise RuntimeError('entity tables are not contiguous in the flat gradient buffer')
        self.tab2d = fused_step.flat_grad[lo:hi].view(-1, self.D)
        self.row_base = [(self.off[id(t)] - lo) // self.D for t in self.tables]
        self.row_bits = _touch_bits(max(t.shape[0] for t in self.tables))
        if tables not in ('auto', 'dense', 'rows'):
            raise ValueError("tables: 'auto', 'dense' or 'rows'")
        self.table_mode = tables if tables != 'auto' else ('dense' if 4 * (hi - lo) <= self.DENSE_TABLE_BYTES else 'rows')
        # the row exchange plans with the keys of the step's touch plan: built by pack() (touch='pack': plan() reads them
        # before the step runs) or by the step itself (touch='step', the default: the keys are an OUTPUT of run() and
        # reduce() plans from them, stream-ordered -- no object collective, no host read, nothing id-dependent in pack())
        if self.table_mode == 'rows' and fused_step.touch_mode not in ('pack', 'step'):
            raise ValueError("the row exchange plans with the touch plan's keys: FusedTrainStep(touch='step' | 'pack')")
        self._plans = {}
        if transport not in ('rccl', 'p2p'):
            raise ValueError("transport: 'rccl' or 'p2p'")
        self.transport, self.transport_note, self.peer = 'rccl', None, None
        if transport == 'p2p' and self.world > 1:
            from .p2p import PeerExchange
            cap = sum(p.numel() for p in fused_step.params
                      if self.table_mode == 'dense' or id(p) not in self.table_ids)
            # (its bounded waits report into the STEP's error word: check() below and FusedTrainStep.check() see them)
            self.peer = PeerExchange(cap, group=group, device=self.dev, err=fused_step.err)
            if self.peer.ok:
                self.transport = 'p2p'
            else:
                self.transport_note = 'p2p exchange unavailable (%s): RCCL all-reduce used' % (self.peer.reason or 'a peer failed')
                self.peer = None
        import ctypes
        self._tab_g = (ctypes.c_void_p * len(self.tables))(
            *[fused_step.flat_grad.data_ptr() + 4 * self.off[id(t)] for t in self.tables])

    # ---- collectives that also work with the gloo backend on CUDA tensors (tests): through the host there
    def _all_reduce(self, t):
        if self.world == 1:
            return
        if self.backend == 'gloo' and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)

    def _all_gather(self, out, t):
        if self.world == 1:
            out.copy_(t.reshape(out.shape))
            return
        if self.backend == 'gloo' and t.is_cuda:
            parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world)]
            dist.all_gather(parts, t.cpu(), group=self.group)
            out.copy_(torch.cat(parts).reshape(out.shape))
        else:
            dist.all_gather_into_tensor(out, t, group=self.group)

    def _span_tables(self, ep, merged):
        """Device tables for mpqe_spans_copy: the merged (offset, floats) spans of the flat gradient buffer <-> the bucket."""
        recs_in, recs_out, o, blk = [], [], 0, 0
        for off, n in merged:
            recs_in += [o, off, n, blk]            # bucket[o ..] <- flat[off ..]
            recs_out += [off, o, n, blk]           # flat[off ..] <- bucket[o ..]
            o += n
            blk += (n + 4095) // 4096
        ep.spans_in = torch.tensor(recs_in, dtype=torch.int64, device=self.dev)
        ep.spans_out = torch.tensor(recs_out, dtype=torch.int64, device=self.dev)
        ep.span_blocks, ep.nspans = blk, len(merged)

    def _copy_spans(self, dst, src, table, ep):
        from . import _capi, ops
        L = ops.lib()
        with torch.cuda.device(self.dev):
            st = L.mpqe_spans_copy(dst.data_ptr(), src.data_ptr(), table.data_ptr(), ep.nspans, ep.span_blocks,
                                   torch.cuda.current_stream().cuda_stream)
        _capi.check(L, st, 'mpqe_spans_copy')

    @staticmethod
    def descriptor_hash(packed):
        """64-bit hash of a packed step's descriptor set (formulas, relation ids, passes, sizes, weights): what a plan
        key must stand for."""
        import hashlib
        return int.from_bytes(hashlib.blake2b(bytes(packed.batches), digest_size=8).digest(), 'little') & ((1 << 63) - 1)

    def plan(self, packed, key=None, verify=False):
        """Collective on a key's first use: every rank calls it with ITS packed step (same number of calls in the same
        order). key: the caller's name for what recurs across steps -- the formula sets of ALL ranks at this step (e.g.
        the index into a common schedule of formula sets). A key seen before returns its cached plan with no collective.
        What the key is trusted for, and what is checked:
          * first use: the ranks exchange (key, descriptor hash) next to their matrix lists; ranks that arrive with
            DIFFERENT keys raise (a schedule that has drifted apart);
          * every later use: this rank's descriptor set must hash to what the key was planned for, else ValueError
            (a key re-used for another formula set would reduce the wrong matrices -- silently diverging replicas);
          * verify=True (a collective: every rank must pass it in the same call): additionally all-reduces (min, max)
            of the combined hash, so a rank whose OWN set still matches learns that another rank's does not.
        With the row exchange and a pack-time touch plan (touch='pack') the ids take part: no caching."""
        import ctypes
        from . import _capi, ops
        rows = self.table_mode == 'rows'
        # (the STEP OBJECT's configuration decides -- the same on every rank. A rank whose descriptor set fell back to
        # pack-time plans after a recovered in-step sort, FusedTrainStep.run(checked=True), still holds sorted keys in the same
        # plan buffer: it keeps to the cached in-step exchange plan like its peers, or the ranks' collectives would differ)
        in_step = rows and self.fused.touch_mode == 'step' and getattr(packed, 'touch_mode', None) in ('step', 'pack')
        dh = self.descriptor_hash(packed)
        if key is not None and (not rows or in_step) and key in self._plans:
            ep = self._plans[key]
            ok = ep.desc_hash[self.rank] == dh
            if verify and self.world > 1:
                import hashlib
                mine = int.from_bytes(hashlib.blake2b(repr((ok, ep.desc_hash)).encode(), digest_size=7).digest(), 'little')
                t = torch.tensor([mine, -mine], dtype=torch.int64)
                if self.backend != 'gloo':
                    t = t.to(self.dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                lo_hi = t.tolist()
                if lo_hi[0] != -lo_hi[1]:
                    raise ValueError('StepExchange.plan(key=%r): the ranks disagree about what this key stands for' % (key,))
            if not ok:
                raise ValueError('StepExchange.plan(key=%r): this packed step is not the descriptor set the key was planned '
                                 'for (a plan key must name ONE formula set per rank)' % (key,))
            return ep
        f = self.fused
        m = f.model
        layers = list(m.layers)
        # (basis parameter, relation) pairs this rank's step can touch: every template edge at every executed pass
        mine = set()
        for i in range(packed.nb):
            b = packed.batches[i]
            info = ops.template_info(list(_capi.QUERY_TYPE_IDS.keys())[b.query_type])
            L = int(b.num_passes)
            for p in range(L):
                li = p if p < L - 1 else len(layers) - 1
                kk = [k for k, l in enumerate(layers) if l.basis is layers[li].basis][0]     # shared layers: one buffer
                for e in range(info.num_edges):
                    mine.add((kk, int(b.edge_type[e])))
        gathered = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(gathered, (repr(key), dh, sorted(mine)), group=self.group)
        else:
            gathered = [(repr(key), dh, sorted(mine))]
        if len(set(g[0] for g in gathered)) != 1:
            raise ValueError('StepExchange.plan: the ranks arrived with different keys %r' % ([g[0] for g in gathered],))
        desc_hash = tuple(g[1] for g in gathered)
        union = sorted(set().union(*[set(map(tuple, g[2])) for g in gathered]))
        segs, full = [], []
        DD = self.D * self.D
        for p in f.params:
            o = self.off[id(p)]
            if id(p) in self.table_ids:
                if not rows:
                    segs.append((o, p.numel()))
                    full.append((o, p.numel()))
                continue
            full.append((o, p.numel()))
            owner = [k for k, l in enumerate(layers) if l.basis is p]
            if owner:
                segs.extend((o + rel * DD, DD) for kk, rel in union if kk == owner[0])
            else:
                segs.append((o, p.numel()))

        def merge(ss):
            out = []
            for o, n in sorted(ss):
                if out and out[-1][0] + out[-1][1] == o:
                    out[-1][1] += n
                else:
                    out.append([o, n])
            return out
        merged, spans = merge(segs), merge(full)
        ep = ExchangePlan()
        ep.union = union
        ep.rows = rows
        ep.desc_hash = desc_hash
        ep.in_step = in_step
        ep.send_keys = ep.all_keys = ep.plan_ws = ep.plan_sizes = None
        ep.spans_in = ep.spans_out = None
        ep.span_blocks = ep.nspans = 0
        nsel, nfull = sum(n for _, n in merged), sum(n for _, n in spans)
        if self.transport == 'p2p':
            # the bucket IS the head of the peer-mapped communication buffer: copied in, summed in place by the one-hop
            # exchange, copied back
            ep.form = 'p2p'
            ep.grad_views = [f.flat_grad[o:o + n] for o, n in merged]
            ep.bucket = self.peer.bucket[:nsel]
            ep.bucket_views, o = [], 0
            for _, n in merged:
                ep.bucket_views.append(ep.bucket[o:o + n])
                o += n
            ep.bucket_bytes = 4 * nsel
        elif nsel >= self.DENSE_FRACTION * nfull:
            # most of the gradient is touched by some rank: all-reduce the parameter spans where they lie, no copies
            ep.form = 'dense'
            ep.grad_views = [f.flat_grad[o:o + n] for o, n in spans]
            ep.bucket, ep.bucket_views, ep.bucket_bytes = None, None, 4 * nfull
        else:
            ep.form = 'bucket'
            ep.grad_views = [f.flat_grad[o:o + n] for o, n in merged]
            ep.bucket = torch.zeros(nsel, dtype=torch.float32, device=self.dev)
            ep.bucket_views, o = [], 0
            for _, n in merged:
                ep.bucket_views.append(ep.bucket[o:o + n])
                o += n
            ep.bucket_bytes = 4 * nsel
        if ep.form != 'dense' and merged:
            self._span_tables(ep, merged)
        w = self.world
        ep.wire_bytes = int(2 * (w - 1) / max(w, 1) * ep.bucket_bytes)      # (ring and one-hop move the same bytes; the hops differ)
        ep.n_own = ep.cap = ep.entries = 0
        ep.gidx = ep.send = ep.recv = ep.plan = ep.plan_ptr = None
        if not rows:
            if key is not None:
                if len(self._plans) > 4096:
                    self._plans.clear()
                self._plans[key] = ep
            return ep
        if in_step:
            # ---- rows, the step builds its own touch plan: its sorted keys exist only after run(). Everything id-dependent
            # happens in reduce(packed=...), on the device, at FIXED sizes: every rank sends cap = (its looked-up ids) key
            # slots -- the first key of every run of equal keys, the other slots invalid --, so no count has to be exchanged
            # or read back. Here: the buffers (a function of the descriptor set alone -> cached under the key).
            L = ops.lib()
            M = int(packed.touch_entries)
            caps = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(caps, M, group=self.group)      # (once per key)
            else:
                caps = [M]
            ep.cap = max(max(caps), 1)
            ep.n_own = M
            ep.entries = self.world * ep.cap
            nbytes = L.mpqe_rows_plan_bytes(ep.entries)
            wbytes = L.mpqe_rows_plan_workspace_bytes(ep.entries, self.row_bits + 5)
            ep.plan_sizes = (nbytes, wbytes)
            ep.plan = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.dev)
            ep.plan_ptr = (ep.plan.data_ptr() + 255) // 256 * 256
            ep.plan_ws = torch.empty(wbytes + 256, dtype=torch.uint8, device=self.dev)
            ep.send_keys = torch.full((ep.cap,), -1, dtype=torch.int64, device=self.dev)
            ep.all_keys = torch.empty(self.world * ep.cap, dtype=torch.int64, device=self.dev)
            ep.gidx = torch.zeros(ep.cap, dtype=torch.int64, device=self.dev)
            ep.send = torch.zeros(ep.cap, self.D, dtype=torch.float32, device=self.dev)
            ep.recv = torch.empty(self.world * ep.cap, self.D, dtype=torch.float32, device=self.dev)
            ep.wire_bytes += int((w - 1) * ep.cap * (self.D * 4 + 8))
            if key is not None:
                if len(self._plans) > 4096:
                    self._plans.clear()
                self._plans[key] = ep
            return ep
        # ---- rows: this rank's distinct keys from its touch plan, all ranks' keys, one plan over all of them
        if packed.touch is None or packed.touch_mode != 'pack':
            raise ValueError("the row exchange needs a packed step whose touch plan was built by pack() "
                             "(FusedTrainStep(touch='pack')) or by the step itself (touch='step')")
        f.check_touch(packed)
        base = packed.touch_ptr - packed.touch.data_ptr()
        M = packed.touch_entries
        keys = packed.touch[base + 256: base + 256 + 8 * M].view(torch.int64)
        uk = torch.unique_consecutive(keys[keys != -1])
        ep.n_own = int(uk.numel())                        # (pack time: a host read is fine here)
        counts = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(counts, ep.n_own, group=self.group)
        else:
            counts = [ep.n_own]
        ep.cap = max(max(counts), 1)
        send_keys = torch.full((ep.cap,), -1, dtype=torch.int64, device=self.dev)
        send_keys[:ep.n_own] = uk
        allk = torch.empty(self.world * ep.cap, dtype=torch.int64, device=self.dev)
        self._all_gather(allk, send_keys)
        L = ops.lib()
        ep.entries = self.world * ep.cap
        nbytes = L.mpqe_rows_plan_bytes(ep.entries)
        wbytes = L.mpqe_rows_plan_workspace_bytes(ep.entries, self.row_bits + 5)
        ep.plan = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.dev)
        ep.plan_ptr = (ep.plan.data_ptr() + 255) // 256 * 256
        ws = torch.empty(wbytes + 256, dtype=torch.uint8, device=self.dev)
        with torch.cuda.device(self.dev):
            st = L.mpqe_rows_plan_build(allk.data_ptr(), ep.entries, self.row_bits, self.row_bits + 5, ep.plan_ptr, nbytes,
                                        (ws.data_ptr() + 255) // 256 * 256, wbytes, torch.cuda.current_stream().cuda_stream)
        _capi.check(L, st, 'mpqe_rows_plan_build')
        ws.record_stream(torch.cuda.current_stream())
        tab = uk >> self.row_bits
        row = uk & ((1 << self.row_bits) - 1)
        base_t = torch.tensor(self.row_base, dtype=torch.int64, device=self.dev)
        ep.gidx = base_t[tab] + row
        ep.send = torch.zeros(ep.cap, self.D, dtype=torch.float32, device=self.dev)
        ep.recv = torch.empty(self.world * ep.cap, self.D, dtype=torch.float32, device=self.dev)
        ep.wire_bytes += int((w - 1) * ep.cap * self.D * 4)
        return ep

    def check(self):
        """COLLECTIVE (every rank, same point of every step -- after reduce(), before the optimiser consumes the gradients).
        Reads this rank's error word (one 4-byte device-to-host read: waits for the step and its exchange) and all-reduces
        what it says, so that EVERY rank learns what ANY rank met and all act alike:
          * a peer of the p2p exchange did not arrive within its bound (MPQE_FLAG_INTERNAL | 0x4000) on any rank: that
            exchange is incomplete everywhere -- every rank switches to the RCCL all-reduce for good (the peer buffers are
            released, cached plans dropped: the next plan() of a key is a first use again) and raises RuntimeError; the
            step's gradients must not be used (run the step again);
          * anything else (a bad entity id, a failed in-launch hand-off, an unrecovered touch plan) on any rank: every rank
            raises -- the rank that met it what FusedTrainStep.check() raises, the others RuntimeError naming the rank.
        Returns None when every rank is clean. The reference is single-process: none of this has a counterpart there."""
        from . import _capi, ops
        flags = int(self.fused.err.item())
        p2p_bad = 1 if (flags & _capi.FLAG_INTERNAL and flags & 0x4000) else 0
        other = flags & ~0x4000 if p2p_bad else flags
        if p2p_bad and not (other & 0xff00):
            other &= ~_capi.FLAG_INTERNAL
        t = torch.tensor([p2p_bad, 1 if other else 0, self.rank if (p2p_bad or other) else -1], dtype=torch.int64)
        if self.world > 1:
            if self.backend != 'gloo':
                t = t.to(self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        any_p2p, any_other, who = [int(v) for v in t.tolist()]
        if any_p2p:
            self.fused.err.fill_(other)
            self._fall_back('a peer did not arrive within the bound of the p2p exchange (seen by rank %d)' % who)
            if not other:
                raise RuntimeError('mpqe_amd: the p2p gradient exchange of this step is incomplete (%s); RCCL is used from now '
                                   'on -- run the step again' % self.transport_note)
        if other:
            ops.raise_on_flags(self.fused.err)
        if any_other:
            raise RuntimeError('mpqe_amd: rank %d reported a fault in this step (see its exception); the gradients of this '
                               'step must not be used' % who)

    def _fall_back(self, why):
        """p2p -> RCCL, for good, on every rank (called collectively)."""
        if self.transport != 'p2p':
            return
        self.transport = 'rccl'
        self.transport_note = 'p2p exchange abandoned (%s): RCCL all-reduce used' % why
        self._plans.clear()
        peer, self.peer = self.peer, None
        if peer is not None:
            peer.close()

    def rows_plan(self, ep):
        """(plan pointer, entries) for FlatOptimizer.step(packed, rows_plan=...): the rows ANY rank touched (row exchange
        only; None with dense tables)."""
        return (ep.plan_ptr, ep.entries) if ep.rows else None

    def _plan_rows_in_step(self, ep, packed):
        """Row exchange of a step that built its own touch plan (touch='step'): called AFTER run(packed), stream-ordered,
        no host read, no object collective. This rank's sorted keys are in the plan buffer the step has just written;
        the first key of every run goes out (the other slots invalid: fixed size), the keys of all ranks are gathered and
        every rank sorts them into the same row plan (stable: equal keys stay in rank order)."""
        from . import _capi, ops
        if packed is None or packed.touch is None or getattr(packed, 'touch_mode', None) not in ('step', 'pack'):
            raise ValueError("reduce(plan, packed=...): the packed step that has just run (its touch plan holds the keys)")
        M = int(packed.touch_entries)
        if M != ep.n_own:
            raise ValueError('reduce: this packed step has %d looked-up ids, the plan was made for %d' % (M, ep.n_own))
        import ctypes
        L = ops.lib()
        # the first key of every run of the plan's sorted keys + the flat table row it names, one launch (a failed in-step
        # sort -- flagged in the plan's header, MPQE_FLAG_TOUCH_RETRY in the error word -- leaves whatever keys the buffer
        # held: keys outside the tables are masked there)
        i64 = ctypes.c_int64 * len(self.tables)
        with torch.cuda.device(self.dev):
            st = L.mpqe_rows_prepare(packed.touch_ptr + 256, M, ep.cap, self.row_bits, i64(*[t.shape[0] for t in self.tables]),
                                     i64(*self.row_base), len(self.tables), ep.send_keys.data_ptr(), ep.gidx.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
        _capi.check(L, st, 'mpqe_rows_prepare')
        self._all_gather(ep.all_keys, ep.send_keys)
        nbytes, wbytes = ep.plan_sizes
        with torch.cuda.device(self.dev):
            st = L.mpqe_rows_plan_build(ep.all_keys.data_ptr(), ep.entries, self.row_bits, self.row_bits + 5, ep.plan_ptr,
                                        nbytes, (ep.plan_ws.data_ptr() + 255) // 256 * 256, wbytes,
                                        torch.cuda.current_stream().cuda_stream)
        _capi.check(L, st, 'mpqe_rows_plan_build')

    def reduce(self, ep, packed=None):
        """After fused_step.run(packed): every p.grad <- sum over ranks (stream-ordered, no host read). packed: the step
        that has just run -- needed (only) by the row exchange of a step that builds its own touch plan."""
        from . import _capi, ops
        if ep.form == 'dense':
            for v in ep.grad_views:
                self._all_reduce(v)
        elif ep.form == 'p2p':
            if ep.bucket.numel():
                self._copy_spans(ep.bucket, self.fused.flat_grad, ep.spans_in, ep)       # (one launch each way: the library's own)
                self.peer.all_reduce(ep.bucket.numel())
                self._copy_spans(self.fused.flat_grad, ep.bucket, ep.spans_out, ep)
        elif ep.bucket.numel():
            self._copy_spans(ep.bucket, self.fused.flat_grad, ep.spans_in, ep)
            self._all_reduce(ep.bucket)
            self._copy_spans(self.fused.flat_grad, ep.bucket, ep.spans_out, ep)
        if not ep.rows:
            return
        L = ops.lib()
        if ep.in_step:
            self._plan_rows_in_step(ep, packed)
            n_send = ep.cap                                             # (invalid slots: row 0, never summed)
        else:
            n_send = ep.n_own
        if n_send:
            with torch.cuda.device(self.dev):
                st = L.mpqe_rows_gather(self.tab2d.data_ptr(), ep.gidx.data_ptr(), n_send, self.D, ep.send.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
            _capi.check(L, st, 'mpqe_rows_gather')
        self._all_gather(ep.recv, ep.send)
        with torch.cuda.device(self.dev):
            st = L.mpqe_table_rows_sum(ep.plan_ptr, ep.entries, ep.recv.data_ptr(), self.D, self._tab_g, len(self.tables), 1,
                                       torch.cuda.current_stream().cuda_stream)
        _capi.check(L, st, 'mpqe_table_rows_sum')

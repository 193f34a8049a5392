"""The reference's own entry points on the fused step (outer boundary).

`RGCNEncoderDecoder.margin_loss(formula, queries, anchor_ids, var_ids, q_graphs, hard_negatives, margin)`
(reference model.py:464-494) and `.forward(...)` under no_grad (model.py:400-462) keep their signatures, their python
`random` stream and their return values, and run on `mpqe_step_forward_backward` instead of one launch per op:

  margin_loss   the ids of the call (anchors, targets, the negatives drawn with python's own Mersenne-Twister stream) are
                appended to a pinned arena the kernels read in place; ONE forward-only library call (chain launch + loss)
                gives the loss value, returned through a torch.autograd.Function. Nothing else happens until backward.
  backward      `loss = l_0 + w_1 * l_1 + ...; loss.backward()` (reference train_helpers.py:81-119) reaches every
                margin_loss node with its upstream gradient, a 0-dim device tensor. The nodes only queue themselves; a
                callback the autograd engine runs at the end of the pass executes ALL of them as ONE fused step
                (forward + backward of every batch: three launches) whose per-batch weights are those device scalars
                (include/mpqe_amd.h: mpqe_step_extra_t.batch_weight) -- no device-to-host read, and p.grad of every
                parameter then holds the gradient (added to what it held; p.grad is a view of one flat buffer).
  forward       (torch.no_grad: the evaluation loops, reference utils.py:34-95) one forward-only call; one negative per
                query (eval_auc_queries) comes back from the same call, ragged negative lists (eval_perc_queries) score
                against the query embeddings the call also writes (mpqe_step_extra_t.query_out) with mpqe_cosine_fwd.

Gradients reach the parameters through `p.grad` only (torch.autograd.grad over a drop-in loss sees no inputs);
`model.fused = False` restores the per-op module path, which is also what foreign encoders and forward() with autograd
take.
"""
import random

import numpy as np
import torch

from . import _capi, ops
from .data_utils import RGCNQueryDataset
from .fused import FusedTrainStep, _TEMPLATES

MAX_CALLS = _capi.STEP_MAX_BATCHES
MAX_IDS = _capi.TSORT_MAX_ENTRIES          # looked-up ids of one fused step whose touch plan the step builds itself


class _Arena(object):
    """Pinned host memory for the ids of the margin_loss calls between two backward passes, in the layout the step reads:
    anchors [per call: A x B, slot-major], targets [per call: B], negatives [per call: B]. The kernels read it in place
    (pinned memory is mapped into the device's address space). Re-used once no autograd node refers to it and the last
    launch that read it has run."""

    def __init__(self, cap_g):
        self.cap_g, self.cap_a = cap_g, 3 * cap_g
        self.a = torch.empty(self.cap_a, dtype=torch.long, pin_memory=True)
        self.t = torch.empty(self.cap_g, dtype=torch.long, pin_memory=True)
        self.n = torch.empty(self.cap_g, dtype=torch.long, pin_memory=True)
        self.a_np, self.t_np, self.n_np = self.a.numpy(), self.t.numpy(), self.n.numpy()
        self.a_ptr, self.t_ptr, self.n_ptr = self.a.data_ptr(), self.t.data_ptr(), self.n.data_ptr()
        self.event = torch.cuda.Event()
        self.reset()

    def reset(self):
        self.na = self.ng = self.calls = 0
        self.live = 0              # autograd nodes that may still run their backward over these ids
        self.dirty = False         # a launch has read it since the event was recorded

    def fits(self, a, g):
        return self.calls < MAX_CALLS and self.na + a <= self.cap_a and self.ng + g <= self.cap_g


class _Call(object):
    """One margin_loss call as its autograd node remembers it."""
    __slots__ = ('arena', 'oa', 'og', 'idx', 'key', 'B', 'A', 'margin', 'seq', 'g', '__weakref__')

    def __del__(self):
        arena = getattr(self, 'arena', None)
        if arena is not None:
            arena.live -= 1


class _MarginLossNode(torch.autograd.Function):
    """The loss value came from the forward-only library call; backward queues the call for the pass' one fused step."""

    @staticmethod
    def forward(ctx, hook, owner, call, buf):
        ctx.owner, ctx.call = owner, call
        # a tensor of its own over the call's loss word: the reference's `loss += w * margin_loss(...)` writes its
        # left-hand side in place, which autograd refuses on a view made inside a Function
        return buf.new_empty(()).set_(buf.untyped_storage(), buf.storage_offset(), ())

    @staticmethod
    def backward(ctx, g):
        ctx.owner._on_backward(ctx.call, g)
        return None, None, None, None


class DropIn(object):
    """Per model: the fused step (FusedTrainStep: flat gradient buffer, descriptor / plan caches) and the bookkeeping of the
    calls in flight. Raises ValueError / NotImplementedError when the model's configuration has no fused form (the model
    then keeps the module path)."""

    def __init__(self, model):
        self.model = model
        held = [(p, p.grad) for p in model.parameters() if p.requires_grad and p.grad is not None]
        self.step = FusedTrainStep(model, margin=1.0)          # (binds every p.grad to its view of one flat buffer)
        for p, g in held:                                       # gradients the module path left there stay
            p.grad.copy_(g)
        self.device = self.step.device
        self._sig = self._signature()
        self.lib = ops.lib()
        self._hook = torch.zeros((), dtype=torch.float32, device=self.device, requires_grad=True)
        self._one = {}             # (formula, passes, B) -> packed one-batch step (forward-only calls)
        self._multi = {}           # tuple of those keys -> packed step of a whole backward pass
        self._arena = None
        self._free = []
        self._pending = []         # calls whose nodes ran in the current backward pass
        self._seq = 0
        self._full_lists = {}
        self._cursor = np.zeros(2, dtype=np.int64)
        self._err_host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        self._err_np = self._err_host.numpy()
        self.checked = False       # True: every backward pass reads the error word before it returns (one sync) and recovers
        self.steps = 0             # fused backward steps run
        self.fast_sampled = 0      # calls whose negatives were drawn by the library replay of python's stream

    def _signature(self):
        ps = self.step.params
        return (len(ps), ps[0].data_ptr(), ps[-1].data_ptr()) if ps else (0,)

    def stale(self):
        """The parameters moved (model.cuda() after the first call, FlatOptimizer re-homing them into one flat buffer)."""
        return self._sig != self._signature()

    def refresh(self):
        """Take the parameters' addresses again. False: they left this step's device (the model builds a new DropIn)."""
        if any(p.device != self.device for p in self.step.params):
            return False
        self.step._refresh_pointers()
        self._sig = self._signature()
        return True

    # ------------------------------------------------------------------------------------------- packed steps
    def _passes(self, formula):
        m = self.model
        if m.adaptive:
            passes = RGCNQueryDataset.query_diameters[formula.query_type]
            if passes > len(m.layers):
                raise ValueError(f'RGCN is adaptive with {len(m.layers)}'
                                 f' layers, but query requires {passes}.')
            return passes
        return m.num_layers

    def _packed(self, keys):
        """The packed step (descriptors, descriptor table, plan buffer; ids external) of the batches `keys`."""
        cache = self._one if len(keys) == 1 else self._multi
        ps = cache.get(keys)
        if ps is None:
            if len(cache) > 4096:
                cache.clear()
            ps = cache[keys] = self.step.pack([dict(formula=f, batch_size=B, weight=1.0) for (f, _p, B) in keys],
                                              ids='external')
        return ps

    # ------------------------------------------------------------------------------------------- arenas
    def _arena_for(self, a, g):
        ar = self._arena
        if ar is not None and ar.fits(a, g):
            return ar
        if ar is not None:
            self._retire(ar)
        need = max(g * 4, 1 << 14)
        for i, cand in enumerate(self._free):
            if cand.live == 0 and cand.cap_g >= need and cand.event.query():
                ar = self._free.pop(i)
                ar.reset()
                break
        else:
            ar = _Arena(max(need, 1 << 14))
            if len(self._free) > 16:               # (arenas whose nodes never ran: let them go with their graphs)
                self._free = [c for c in self._free if c.live > 0][-16:]
        self._arena = ar
        return ar

    def _retire(self, ar):
        if ar.dirty:
            ar.event.record(torch.cuda.current_stream(self.device))
            ar.dirty = False
        self._free.append(ar)
        if self._arena is ar:
            self._arena = None

    # ------------------------------------------------------------------------------------------- negatives
    def _choice(self, lens, len_all, base, cand, B, out_ptr):
        """random.choice per query over the candidate lists, python's own stream (include/mpqe_amd.h:
        mpqe_host_random_choice): raw Mersenne-Twister outputs are taken from the interpreter's generator in rounds of
        exactly as many as the queries still open need at least, so the generator ends where the reference's loop
        leaves it."""
        cur = self._cursor
        cur[0] = 0
        fn = self.lib.mpqe_host_random_choice
        lens_p = None if lens is None else lens.ctypes.data
        base_p = None if base is None else base.ctypes.data
        cand_p = cand.ctypes.data
        cur_p = cur.ctypes.data
        done = 0
        while done < B:
            n = B - done
            words = random.getrandbits(32 * n).to_bytes(4 * n, 'little')
            st = fn(words, n, lens_p, len_all, base_p, cand_p, B, cur_p, out_ptr)
            if st != 0:
                raise IndexError('Cannot choose from an empty sequence')
            done = int(cur[0])

    def _full_list(self, mode):
        lst = self.model.graph.full_lists[mode]
        c = self._full_lists.get(mode)
        if c is None or c[0] is not lst or c[1].shape[0] != len(lst):
            c = self._full_lists[mode] = (lst, np.asarray(lst, dtype=np.int64))
        return c[1]

    def _fill_ids(self, ar, formula, queries, anchor_ids, ids, hard_negatives, B, A):
        """Anchors, targets and freshly drawn negatives of one call into the arena (reference model.py:466-477,
        data_utils.py:382-383)."""
        oa, og = ar.na, ar.ng
        m = self.model
        fast = ids is not None and ids.end - ids.start == B and ids.fi.A == A
        # anchors, slot-major
        if fast and (anchor_ids is None or anchor_ids is ids.anchor_ref):
            sm = ids.fi.anchors_sm
            for i in range(A):
                ar.a_np[oa + i * B: oa + (i + 1) * B] = sm[i, ids.start:ids.end]
        else:
            if anchor_ids is None:
                for i in range(A):
                    ar.a_np[oa + i * B: oa + (i + 1) * B] = [q.anchor_nodes[i] for q in queries]
            else:
                a = anchor_ids.detach().cpu().numpy() if torch.is_tensor(anchor_ids) else np.asarray(anchor_ids)
                if a.shape != (B, A):
                    raise ValueError('anchor_ids must be [%d, %d] for %s' % (B, A, formula.query_type))
                np.copyto(ar.a_np[oa: oa + A * B].reshape(A, B), a.T, casting='same_kind')
        # targets
        if fast:
            ar.t_np[og: og + B] = ids.fi.targets[ids.start:ids.end]
        else:
            ar.t_np[og: og + B] = [q.target_node for q in queries]
        # negatives: same draws as the reference's list comprehensions
        if "inter" not in formula.query_type and hard_negatives:
            raise Exception("Hard negative examples can only be used with "
                            "intersection queries")
        out_ptr = ar.n_ptr + 8 * og
        if hard_negatives:
            csr = ids.fi.hard if fast else None
        elif formula.query_type == "1-chain":
            full = self._full_list(formula.target_mode)
            self._choice(None, full.shape[0], None, full, B, out_ptr)
            self.fast_sampled += 1
            return oa, og
        else:
            csr = ids.fi.neg if fast else None
        if csr is not None:
            flat, off, lens = csr
            self._choice(lens[ids.start:ids.end], 0, off[ids.start:ids.end], flat, B, out_ptr)
            self.fast_sampled += 1
        else:
            ar.n_np[og: og + B] = m.sample_negatives(formula, queries, hard_negatives)
        return oa, og

    # ------------------------------------------------------------------------------------------- margin_loss
    def _check_mirror(self):
        if self._err_np[0]:
            torch.cuda.current_stream(self.device).synchronize()
            self._err_np[0] = 0
            ops.raise_on_flags(self.step.err)

    def margin_loss(self, formula, queries, anchor_ids=None, var_ids=None, q_graphs=None, hard_negatives=False, margin=1):
        m = self.model
        if m.validate:
            self._check_mirror()
        B = len(queries)
        A = _TEMPLATES[formula.query_type][0]
        if A != len(formula.anchor_modes):
            raise ValueError('formula %s has %d anchor modes, template expects %d' % (formula, len(formula.anchor_modes), A))
        key = (formula, self._passes(formula), B)
        ps = self._packed((key,))
        ar = self._arena_for(A * B, B)
        oa, og = self._fill_ids(ar, formula, queries, anchor_ids, getattr(q_graphs, 'ids', None), hard_negatives, B, A)
        step = self.step
        step.margin = float(margin)
        loss = torch.empty(2, dtype=torch.float32, device=self.device)
        step.run(ps, backward=False, id_ptrs=(ar.a_ptr + 8 * oa, ar.t_ptr + 8 * og, ar.n_ptr + 8 * og),
                 out=(loss, None, None))
        ar.dirty = True
        idx = ar.calls
        ar.na, ar.ng, ar.calls = oa + A * B, og + B, idx + 1
        if not torch.is_grad_enabled() or not step.params:
            return loss[0]
        call = _Call()
        call.arena, call.oa, call.og, call.idx, call.key, call.B, call.A = ar, oa, og, idx, key, B, A
        call.margin, call.g = float(margin), None
        self._seq += 1
        call.seq = self._seq
        ar.live += 1
        return _MarginLossNode.apply(self._hook, self, call, loss)

    # ------------------------------------------------------------------------------------------- backward
    def _on_backward(self, call, g):
        if not self._pending:
            torch.autograd.Variable._execution_engine.queue_callback(self._flush)
        if g.dtype != torch.float32 or g.device != self.device:
            g = g.to(device=self.device, dtype=torch.float32)
        if call.g is not None:                 # (the node ran twice in one pass: retain_graph inside a pass cannot happen, but be exact)
            g = call.g + g
        call.g = g.detach()
        if call not in self._pending:
            self._pending.append(call)

    def _flush(self):
        calls, self._pending = self._pending, []
        if not calls:
            return
        calls.sort(key=lambda c: c.seq)
        step = self.step
        # p.grad: None everywhere (optimizer.zero_grad()) -> the step's own zero fill; otherwise added to what is there
        params, views = step.params, step._views
        zero = all(p.grad is None for p in params)
        if not zero:
            for p, v in zip(params, views):
                if p.grad is None:
                    v.zero_()
                elif p.grad is not v:
                    v.copy_(p.grad)
        # groups the library can run as one step: consecutive calls of one arena with one margin, at most MAX_CALLS
        groups, cur, nids = [], [], 0
        for c in calls:
            if cur and (c.arena is not cur[-1].arena or c.idx != cur[-1].idx + 1 or c.margin != cur[-1].margin
                        or len(cur) == MAX_CALLS or nids + (c.A + 2) * c.B > MAX_IDS):
                groups.append(cur)
                cur, nids = [], 0
            cur.append(c)
            nids += (c.A + 2) * c.B
        groups.append(cur)
        stream = torch.cuda.current_stream(self.device)
        for grp in groups:
            ps = self._packed(tuple(c.key for c in grp))
            extra = _capi.StepExtra()
            for i, c in enumerate(grp):
                extra.batch_weight[i] = c.g.data_ptr()
            ar, c0 = grp[0].arena, grp[0]
            step.margin = c0.margin
            loss = torch.empty(1 + len(grp), dtype=torch.float32, device=self.device)
            step.run(ps, backward=True, zero_grad=zero, checked=self.checked,
                     id_ptrs=(ar.a_ptr + 8 * c0.oa, ar.t_ptr + 8 * c0.og, ar.n_ptr + 8 * c0.og), extra=extra,
                     out=(loss, None, None))
            ar.dirty = True
            zero = False
            self.steps += 1
        for c in calls:
            c.g = None
        # the pass is over: its arena is closed (a node kept alive by retain_graph still finds its ids there)
        if self._arena is not None and self._arena.calls:
            self._retire(self._arena)
        for ar in set(c.arena for c in calls):
            if ar is not self._arena and ar.dirty:
                ar.event.record(stream)
                ar.dirty = False
        if self.model.validate and not self.checked:
            self._err_host.copy_(step.err, non_blocking=True)      # read at the next call: a bad id raises one step late

    # ------------------------------------------------------------------------------------------- forward (evaluation)
    def forward(self, formula, queries, target_nodes, anchor_ids=None, var_ids=None, q_graphs=None, neg_nodes=None,
                neg_lengths=None):
        """reference model.py:400-462 without autograd: scores [B] or [B + sum(neg_lengths)]."""
        m = self.model
        B = len(queries)
        A = _TEMPLATES[formula.query_type][0]
        if A != len(formula.anchor_modes):
            raise ValueError('formula %s has %d anchor modes, template expects %d' % (formula, len(formula.anchor_modes), A))
        key = (formula, self._passes(formula), B)
        ps = self._packed((key,))
        ar = self._arena_for(A * B, B)
        oa, og = ar.na, ar.ng
        if anchor_ids is None:
            for i in range(A):
                ar.a_np[oa + i * B: oa + (i + 1) * B] = [q.anchor_nodes[i] for q in queries]
        else:
            a = anchor_ids.detach().cpu().numpy() if torch.is_tensor(anchor_ids) else np.asarray(anchor_ids)
            if a.shape != (B, A):
                raise ValueError('anchor_ids must be [%d, %d] for %s' % (B, A, formula.query_type))
            np.copyto(ar.a_np[oa: oa + A * B].reshape(A, B), a.T, casting='same_kind')
        t = target_nodes.detach().cpu().numpy() if torch.is_tensor(target_nodes) else target_nodes
        ar.t_np[og: og + B] = t
        ragged = False
        if neg_nodes is None:
            ar.n_np[og: og + B] = ar.t_np[og: og + B]
        else:
            lengths = neg_lengths.tolist() if hasattr(neg_lengths, 'tolist') else list(neg_lengths)
            if len(lengths) != B:
                raise ValueError('neg_lengths must have one entry per query')
            if all(l == 1 for l in lengths):
                n = neg_nodes.detach().cpu().numpy() if torch.is_tensor(neg_nodes) else neg_nodes
                ar.n_np[og: og + B] = n
            else:
                ragged = True
                ar.n_np[og: og + B] = ar.t_np[og: og + B]
        scores = torch.empty(2 * B, dtype=torch.float32, device=self.device)
        loss = torch.empty(2, dtype=torch.float32, device=self.device)
        extra = q = None
        if ragged:
            if not self.step.uses_chain(ps):
                raise NotImplementedError('ragged negatives on the fused forward need the chain form')
            q = torch.empty(B, m.emb_dim, dtype=torch.float32, device=self.device)
            extra = _capi.StepExtra()
            extra.query_out = q.data_ptr()
        self.step.margin = 1.0
        self.step.run(ps, backward=False, id_ptrs=(ar.a_ptr + 8 * oa, ar.t_ptr + 8 * og, ar.n_ptr + 8 * og),
                      extra=extra, out=(loss, scores[:B], scores[B:]))
        ar.dirty = True
        ar.na, ar.ng, ar.calls = oa + A * B, og + B, ar.calls + 1
        if neg_nodes is None:
            out = scores[:B]
        elif not ragged:
            out = scores
        else:
            # targets and negatives through ONE cosine launch over the query embeddings the step wrote: a negative that IS
            # the target scores exactly like it, as in the reference, where both come from the same cosine_similarity op
            # (the percentile rank counts such ties, utils.py:25-32)
            enc = m.enc
            ids = np.empty(B + sum(lengths), dtype=np.int64)
            ids[:B] = ar.t_np[og: og + B]
            ids[B:] = neg_nodes.detach().cpu().numpy() if torch.is_tensor(neg_nodes) else neg_nodes
            rows = np.empty(ids.shape[0], dtype=np.int64)
            rows[:B] = np.arange(B)
            rows[B:] = np.repeat(np.arange(B), lengths)
            both = torch.from_numpy(np.stack((ids, rows))).to(self.device)
            embeds = ops.embed_l2norm(enc.table(formula.target_mode), enc.node_maps, both[0], self.step.err)
            out = ops.cosine(q, embeds, q_row=both[1])
        if m.validate:
            ops.raise_on_flags(self.step.err)
        return out

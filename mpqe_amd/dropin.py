"""The reference's own entry points on the fused step (outer boundary).

`RGCNEncoderDecoder.margin_loss(formula, queries, anchor_ids, var_ids, q_graphs, hard_negatives, margin)`
(reference model.py:464-494) and `.forward(...)` under no_grad (model.py:400-462) keep their signatures, their python
`random` stream and their return values, and run on `mpqe_step_forward_backward_ex` instead of one launch per op:

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

Host side: per margin_loss call one numpy copy per id array, one call into the CPython extension for the negatives
(csrc/host/pyhost.c: python's generator, the library's replay of random.choice) and one for the library call (its 24
arguments live in a block that is rewritten in the few fields that change), then torch's Function bookkeeping.

Between two calls the drop-in notices that parameters were written by their autograd version counters (torch's optimisers,
load_state_dict, any in-place op on the Parameter) and by FlatOptimizer's own epoch; that is when the side streams wait for
the caller's stream again and a learned readout's regulariser norms are formed again. Writes THROUGH `p.data` bypass the
counters: call `model.dropin().params_changed()` after them.

Gradients reach the parameters through `p.grad` only (torch.autograd.grad over a drop-in loss sees no inputs);
`model.fused = False` restores the per-op module path, which is also what foreign encoders and forward() with autograd
take.
"""
import ctypes
import random

import numpy as np
import torch

from . import _capi, _lib, ops
from .data_utils import RGCNQueryDataset
from .fused import FusedTrainStep, _TEMPLATES

MAX_CALLS = _capi.STEP_MAX_BATCHES
MAX_LANES = 7                              # side streams of the forward-only calls (channel 0 is the caller's stream)
# A chain launch's workgroups wait (bounded) for prologue workgroups of the SAME launch, which are dispatched ahead of them --
# per XCD. That is safe while every launch in flight finds its workgroups resident at once: launches of one stream follow
# each other, but launches on the lanes run side by side, and together they must stay well inside the chip's 256 - 512
# workgroup slots or they could hold each other's prologue out (seen between two PROCESSES on one GPU).
# 1 024 graphs = 64 graph blocks per call: at most 7 x ~100 workgroups in flight.
LANE_MAX_GRAPHS = 1024
MAX_IDS = _capi.TSORT_MAX_ENTRIES          # looked-up ids of one fused step whose touch plan the step builds itself
_P, _L, _U = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64


class StepCall(ctypes.Structure):
    """csrc/host/pyhost.c: StepCall -- the arguments of mpqe_step_forward_backward_ex, in order, as one block."""
    _fields_ = [('params', _P), ('batches', _P), ('num_batches', _L), ('anchor_ids', _P), ('targets', _P), ('negs', _P),
                ('margin', ctypes.c_double), ('grads', _P), ('backward', _L), ('loss', _P), ('scores_pos', _P),
                ('scores_neg', _P), ('desc', _P), ('desc_bytes', _U), ('upload_desc', _L), ('workspace', _P),
                ('workspace_bytes', _U), ('err', _P), ('lanes', _P), ('events', _P), ('num_events', _L), ('touch', _P),
                ('stream', _P), ('extra', _P)]


class _Arena(object):
    """Pinned host memory for the ids of the margin_loss calls between two backward passes, in the layout the step reads:
    anchors [per call: A x B, slot-major], targets [per call: B], negatives [per call: B]. The kernels read it in place
    (pinned memory is mapped into the device's address space). Re-used once no autograd node refers to it and the last
    library call that read it has run -- which the call itself reports in a pinned word (mpqe_step_extra_t.notify): no
    events, no synchronisation."""

    def __init__(self, cap_g):
        self.cap_g, self.cap_a = cap_g, 3 * cap_g
        self.a = torch.empty(self.cap_a, dtype=torch.long, pin_memory=True)
        self.t = torch.empty(self.cap_g, dtype=torch.long, pin_memory=True)
        self.n = torch.empty(self.cap_g, dtype=torch.long, pin_memory=True)
        self.a_np, self.t_np, self.n_np = self.a.numpy(), self.t.numpy(), self.n.numpy()
        self.a_ptr, self.t_ptr, self.n_ptr = self.a.data_ptr(), self.t.data_ptr(), self.n.data_ptr()
        # per channel (0: the caller's stream, 1 ..: the lanes) the number of the last library call that reads it (None: none)
        self.last = [None] * (1 + MAX_LANES)
        self.reset()

    def reset(self):
        self.na = self.ng = self.calls = 0
        for ch in range(len(self.last)):       # (taken again only once every channel's last reader has reported)
            self.last[ch] = None
        self.live = 0              # autograd nodes that may still run their backward over these ids

    def fits(self, a, g):
        return self.calls < MAX_CALLS and self.na + a <= self.cap_a and self.ng + g <= self.cap_g


class _Call(object):
    """One margin_loss call as its autograd node remembers it."""
    __slots__ = ('arena', 'oa', 'og', 'idx', 'key', 'rid', 'B', 'A', 'margin', 'seq', 'g')

    def __del__(self):
        arena = getattr(self, 'arena', None)
        if arena is not None:
            arena.live -= 1


class _Rec(object):
    """A packed step (descriptors, descriptor table, plan buffer; ids named per run) and the argument block of its library
    call. One per (formula, batch size) for the forward-only calls, one per sequence of those for a backward pass."""
    __slots__ = ('ps', 'call', 'addr', 'extra', 'key', 'rid', 'A', 'B', 'nb', 'P_seen', 'ws_seen', 'fb')


class _Lane(object):
    """A side stream for forward-only margin_loss calls. A call is 32 workgroups per 512 graphs and mostly latency: with a
    learned readout it takes 45 - 60 us on the device, three times the host's pace. Consecutive calls of a pass are
    independent, so call k runs on lane k mod K -- its own stream, workspace, packed steps (descriptor tables: their hand-off
    epochs must not be shared by two launches in flight), notification words and XCD (mpqe_step_extra_t.xcd_shift) -- and the
    caller's stream waits for it before the value is used (mpqe_step_extra_t.join_event). The lane waits for the caller's
    stream once per pass and whenever a parameter's version has changed (the optimiser's writes)."""
    __slots__ = ('ch', 'stream', 'raw', 'event', 'event_raw', 'fork', 'ws', 'note_ptr', 'shift', 'pass_id', 'ver', 'epoch',
                 'slots', 'next')

    _streams = {}          # (device index, channel) -> stream: one set per process, whatever the number of models

    def __init__(self, ch, device, note_ptr):
        self.ch = ch
        key = (device.index, ch)
        if key not in _Lane._streams:
            _Lane._streams[key] = torch.cuda.Stream(device=device)
        self.stream = _Lane._streams[key]
        self.raw = self.stream.cuda_stream
        self.event = torch.cuda.Event()
        self.event.record(self.stream)              # (creates the handle)
        self.event_raw = self.event.cuda_event
        self.fork = torch.cuda.Event()
        self.ws = None
        self.note_ptr = note_ptr
        self.shift = (2 * ch - 1) % 8               # lanes 1, 2, 3, 4 -> XCDs 1, 3, 5, 7 on from the plan's
        self.pass_id = self.ver = -1
        # the calls' loss words: a pool of the lane's own. torch's allocator orders a block's re-use within ONE stream; these
        # are written on the lane's stream and read on the caller's, so a word is taken again only when (a) nobody else holds
        # its storage any more and (b) the lane has waited for the caller's stream since it was last handed out (`epoch`
        # counts those waits) -- its readers are then in front of the lane's next kernel
        self.epoch = 0
        self.slots = []            # [tensor, epoch when handed out]
        self.next = 0


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
        self.dev_index = self.device.index
        self._sig = self._signature()
        self.lib = ops.lib()
        self.host = _lib.load_pyhost()
        self._step_fn = self.host.step_fn
        self._hook = torch.zeros((), dtype=torch.float32, device=self.device, requires_grad=True)
        # the calls' autograd nodes: C++ (csrc/host/autograd_node.cpp: the engine stays out of the interpreter until the
        # pass' one callback) when built, else the torch.autograd.Function above
        self._node_ext = _lib.load_autograd_node()
        self._pass = self._node_ext.Pass(self._flush_ext) if self._node_ext is not None else None
        self._calls_by_id = {}
        self.node_impl = 'c++' if self._pass is not None else 'python'
        self._one = {}             # (formula, B) -> _Rec of the one-batch step (forward-only calls)
        self._multi = {}           # tuple of one-batch keys -> _Rec of a whole backward pass
        self._arena = None
        self._free = []
        self._pending = []         # calls whose nodes ran in the current backward pass
        self._seq = 0
        self._full_lists = {}
        self._grb = self._rng = None
        # [0] the number of the last library call whose id reads are over, [1] the error word as that call left it: written
        # by the device (mpqe_step_extra_t.notify), read here without a call
        self._note = torch.zeros(2 * (1 + MAX_LANES), dtype=torch.int32, pin_memory=True)     # (a pair per channel)
        self._note_np = self._note.numpy().view(np.uint32)
        self._note_ptr = self._note.data_ptr()
        self._calls = 0            # library calls issued (their notify values, modulo 2^32)
        self.checked = False       # True: every backward pass reads the error word before it returns (one sync) and recovers
        self.steps = 0             # fused backward steps run
        self.fast_sampled = 0      # calls whose negatives were drawn by the library replay of python's stream
        # forward-only calls on side streams (_Lane): where a call's device time exceeds the host's pace -- the learned
        # readouts (every node of every graph through two more layers, no liveness pruning); set_lanes(n) overrides
        # a learned readout's regulariser (model.py:486-490): the four norms once per parameter version, not once per call
        # (mpqe_step_extra_t.readout_norms / mpqe_step_readout_norms; a forward-only call then has no launch for it)
        self._reg = (torch.zeros(1, dtype=torch.float32, device=self.device)
                     if self.step.learned and self.step.P.readout_weight_decay > 0 else None)
        self._reg_ver = None
        self.lanes = []
        self._want_lanes = 3 if self.step.learned else 0       # (the runtime gives a process four hardware queues: the null stream's and three more)
        self._pass_id = 0

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

    def _record(self, cache, key, batches):
        if len(cache) > 4096:
            cache.clear()
        step = self.step
        r = _Rec()
        r.ps = ps = step.pack([dict(formula=f, batch_size=B, weight=1.0) for (f, B) in batches], ids='external')
        r.key, r.nb = key, len(batches)
        r.fb = batches[0]                       # (a one-batch record's (formula, batch size))
        self._nrec = r.rid = getattr(self, '_nrec', 0) + 1
        r.extra = _capi.StepExtra()
        if self._reg is not None:
            r.extra.readout_norms = self._reg.data_ptr()
        c = r.call = StepCall()
        c.params, c.grads = ctypes.addressof(step.P), ctypes.addressof(step.G)
        c.batches, c.num_batches = ctypes.addressof(ps.batches), ps.nb
        c.desc, c.desc_bytes = ps.desc_ptr, ps.desc_bytes
        c.workspace_bytes = ps.ws_bytes
        c.err = step.err.data_ptr()
        c.touch = ps.touch_ptr
        r.addr = ctypes.addressof(c)
        # (margin_loss' one-call host path, csrc/host/pyhost.c: margin_call, takes the block as _launch last left it: valid
        # while the step's parameter struct and workspace are the ones seen then)
        r.P_seen = r.ws_seen = None
        cache[key] = r
        return r

    def set_lanes(self, n):
        """Forward-only margin_loss calls on n side streams (0: on the caller's stream), from the next call on."""
        self._want_lanes = max(0, min(int(n), MAX_LANES))

    def _make_lanes(self):
        n = self._want_lanes if self._pass is not None else 0        # (the loss words' pool asks the C++ extension who holds them)
        self._want_lanes = n
        while len(self.lanes) < n:
            ch = len(self.lanes) + 1
            self.lanes.append(_Lane(ch, self.device, self._note_ptr + 8 * ch))
        return self.lanes[:n]

    def _lane_ws(self, lane, nbytes):
        if lane.ws is None or lane.ws.numel() < nbytes + 256:
            if lane.ws is not None:
                lane.stream.synchronize()          # (nothing in flight in the buffer that goes)
            lane.ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        return (lane.ws.data_ptr() + 255) // 256 * 256

    def _fork(self, lane, cur):
        """The lane's next kernel runs behind everything the caller's stream holds now."""
        lane.fork.record(cur)
        lane.stream.wait_event(lane.fork)
        lane.epoch += 1

    def _lane_loss(self, lane, cur):
        """A loss word of the lane's pool that is free (see _Lane), else a new one."""
        slots, n, count = lane.slots, len(lane.slots), self._node_ext.storage_use_count
        for attempt in (0, 1):
            i = lane.next
            for _ in range(n):
                s = slots[i]
                i = i + 1 if i + 1 < n else 0
                if s[1] < lane.epoch and count(s[0]) == 1:
                    s[1] = lane.epoch
                    lane.next = i
                    return s[0]
            if attempt or n < 32:
                break
            self._fork(lane, cur)             # (a long run of calls without a backward pass: one wait frees the pool)
        t = torch.empty(2, dtype=torch.float32, device=self.device)
        self._fork(lane, cur)                 # (the block's earlier life on the caller's stream is in front of the lane now)
        slots.append([t, lane.epoch])
        return t

    def params_changed(self):
        """Tell the drop-in that parameters were written behind the version counters (through `p.data`)."""
        self.step.param_epoch += 1

    def _param_version(self):
        """Changes whenever a parameter is written in place (torch's optimisers, load_state_dict) or by FlatOptimizer."""
        v = self.step.param_epoch
        for p in self.step.params:
            v += p._version
        return v

    def _one_rec(self, formula, B, ch=0):
        r = self._one.get((ch, formula, B))
        if r is None:
            A = _TEMPLATES[formula.query_type][0]
            if A != len(formula.anchor_modes):
                raise ValueError('formula %s has %d anchor modes, template expects %d'
                                 % (formula, len(formula.anchor_modes), A))
            self._passes(formula)                  # (the reference's ValueError for a diameter beyond the layers)
            r = self._record(self._one, (ch, formula, B), [(formula, B)])
            r.A, r.B = A, B
        return r

    def _launch(self, r, backward, zero_grad, a_ptr, t_ptr, n_ptr, loss, sp=None, sn=None, extra=None, lane=None):
        """One library call on the current stream (lane: on the lane's, the current stream waiting for it): the argument
        block's changing fields, then csrc/host/pyhost.c: step_call."""
        step, ps, c = self.step, r.ps, r.call
        step.P.flags = step.flags | ps.step_flags | (_capi.STEP_ZERO_GRADS if (backward and zero_grad) else 0)
        c.params, c.grads = ctypes.addressof(step.P), ctypes.addressof(step.G)      # (refresh() makes new structs)
        c.anchor_ids, c.targets, c.negs = a_ptr, t_ptr, n_ptr
        c.margin = step.margin
        c.backward = 1 if backward else 0
        c.loss = loss.data_ptr()
        c.scores_pos = None if sp is None else sp.data_ptr()
        c.scores_neg = None if sn is None else sn.data_ptr()
        bufs = ps.bufs
        c.upload_desc = 0 if bufs.desc_resident else 1
        if extra is None:
            extra = r.extra
        self._calls = seq = (self._calls + 1) & 0xffffffff
        if lane is None:
            c.workspace = step._workspace(ps.ws_bytes)
            c.stream = torch._C._cuda_getCurrentRawStream(self.dev_index)
            extra.notify, extra.notify_value = self._note_ptr, seq
            extra.join_event = extra.join_stream = None
        else:
            c.workspace = self._lane_ws(lane, ps.ws_bytes)
            c.stream = lane.raw
            extra.notify, extra.notify_value = lane.note_ptr, seq
            extra.xcd_shift = lane.shift
            extra.join_event, extra.join_stream = lane.event_raw, torch._C._cuda_getCurrentRawStream(self.dev_index)
        c.extra = ctypes.addressof(extra)
        if torch._C._cuda_getDevice() != self.dev_index:
            with torch.cuda.device(self.device):
                st = self.host.step_call(self._step_fn, r.addr)
        else:
            st = self.host.step_call(self._step_fn, r.addr)
        if st != 0:
            _capi.check(self.lib, st, 'mpqe_step_forward_backward_ex')
        bufs.desc_resident = True
        if not backward and sp is None and extra is r.extra:
            c.upload_desc = 0
            r.P_seen, r.ws_seen = step.P, (step._ws if lane is None else lane.ws)
        return seq

    def _done(self, ar):
        """True once the last library call of every channel that reads arena `ar` (or a later call of that channel) has
        reported that its id reads are over."""
        note = self._note_np
        for ch, seq in enumerate(ar.last):
            if seq is not None and ((int(note[2 * ch]) - seq) & 0xffffffff) >= 0x80000000:
                return False
        return True

    # ------------------------------------------------------------------------------------------- arenas
    def _arena_for(self, a, g):
        ar = self._arena
        if ar is not None and ar.fits(a, g):
            return ar
        if ar is not None:
            self._retire(ar)
        self._drain_dead()
        need = max(g * 4, 1 << 14)
        for i, cand in enumerate(self._free):
            if cand.live == 0 and cand.cap_g >= need and self._done(cand):
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
        self._free.append(ar)
        if self._arena is ar:
            self._arena = None

    # ------------------------------------------------------------------------------------------- negatives
    def _full_list(self, mode):
        lst = self.model.graph.full_lists[mode]
        c = self._full_lists.get(mode)
        if c is None or c[0] is not lst or c[1].shape[0] != len(lst):
            arr = np.asarray(lst, dtype=np.int64)
            c = self._full_lists[mode] = (lst, arr, arr.ctypes.data)
        return c

    def _choice(self, lens, len_all, base, cand, B, out_ptr):
        """random.choice per query over python's own generator (csrc/host/pyhost.c): its state read in place when the
        interpreter passed the self test (mpqe_amd/_lib.py: _mt_selftest), else raw outputs through getrandbits."""
        rng = self._mt()
        if rng is not None:
            self.host.choice_mt(rng, lens, len_all, base, cand, B, out_ptr)
        else:
            self.host.choice(random.getrandbits, lens, len_all, base, cand, B, out_ptr)

    def _mt(self):
        """python's generator when its state may be read in place (else None: raw outputs through getrandbits)."""
        grb = random.getrandbits
        if grb is not self._grb:               # (first call, or somebody re-bound random.getrandbits)
            rng = getattr(grb, '__self__', None)
            ok = (self.host.mt_ok and isinstance(rng, random.Random)
                  and type(rng).getrandbits is random.Random.getrandbits)         # (not SystemRandom or another override)
            self._grb, self._rng = grb, rng if ok else None
        return self._rng

    def _fill_ids(self, ar, formula, queries, anchor_ids, ids, hard_negatives, B, A):
        """Anchors, targets and freshly drawn negatives of one call into the arena (reference model.py:466-477,
        data_utils.py:382-383). The negatives are random.choice's own draws: csrc/host/pyhost.c: choice takes raw outputs
        from the interpreter's generator and the library replays CPython's rejection loop over them
        (mpqe_host_random_choice), so the generator ends where the reference's list comprehension leaves it."""
        oa, og = ar.na, ar.ng
        fast = ids is not None and ids.end - ids.start == B and ids.fi.A == A
        # anchors, slot-major
        if fast and (anchor_ids is None or anchor_ids is ids.anchor_ref):
            sm, lo, hi = ids.fi.anchors_sm, ids.start, ids.end
            for i in range(A):
                ar.a_np[oa + i * B: oa + (i + 1) * B] = sm[i, lo:hi]
        elif anchor_ids is None:
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
            _lst, full, full_ptr = self._full_list(formula.target_mode)
            self._choice(0, full.shape[0], 0, full_ptr, B, out_ptr)
            self.fast_sampled += 1
            return oa, og
        else:
            csr = ids.fi.neg if fast else None
        if csr is not None:
            lo = 8 * ids.start
            self._choice(csr[5] + lo, 0, csr[4] + lo, csr[3], B, out_ptr)
            self.fast_sampled += 1
        else:
            ar.n_np[og: og + B] = self.model.sample_negatives(formula, queries, hard_negatives)
        return oa, og

    # ------------------------------------------------------------------------------------------- margin_loss
    def _check_mirror(self):
        """Raise what a finished call's error word says (IndexError for a bad entity id, as the reference's lookup would)."""
        if self._note_np[1::2].any():
            torch.cuda.current_stream(self.device).synchronize()
            for lane in self.lanes:
                lane.stream.synchronize()
            self._note_np[1::2] = 0
            ops.raise_on_flags(self.step.err)

    def margin_loss(self, formula, queries, anchor_ids=None, var_ids=None, q_graphs=None, hard_negatives=False, margin=1):
        lanes = self.lanes
        if len(lanes) != self._want_lanes:
            lanes = self._make_lanes()
        if self.model.validate and (self._note_np[1::2].any() if lanes else self._note_np[1]):
            self._check_mirror()
        B = len(queries)
        ar = self._arena
        if lanes and B > LANE_MAX_GRAPHS:
            lanes = ()              # (see LANE_MAX_GRAPHS: this call runs on the caller's stream)
        if ar is not None and lanes:
            lane = lanes[ar.calls % len(lanes)]
            ch = lane.ch
        else:
            lane, ch = (lanes[0], 1) if lanes else (None, 0)
        r = self._one.get((ch, formula, B)) or self._one_rec(formula, B, ch)
        A = r.A
        if ar is None or not ar.fits(A * B, B):
            ar = self._arena_for(A * B, B)
        step = self.step
        step.margin = margin = float(margin)
        idx = ar.calls
        ver = self._param_version() if (lane is not None or self._reg is not None) else 0
        if self._reg is not None and ver != self._reg_ver:
            # (on the caller's stream, in front of the lanes' wait for it below)
            with torch.cuda.device(self.device):
                st = self.lib.mpqe_step_readout_norms(ctypes.byref(step.P), self._reg.data_ptr(),
                                                      torch._C._cuda_getCurrentRawStream(self.dev_index))
            if st != 0:
                _capi.check(self.lib, st, 'mpqe_step_readout_norms')
            self._reg_ver = ver
        # (a buffer of its own per call: a caller may keep the value -- `loss.detach()` for a log -- beyond its graph)
        if lane is None:
            cur_raw = torch._C._cuda_getCurrentRawStream(self.dev_index)
            loss = torch.empty(2, dtype=torch.float32, device=self.device)
        else:
            cur = torch.cuda.current_stream(self.device)
            cur_raw = cur.cuda_stream
            # the lane waits for the caller's stream where the parameters may have been written since it last did: once per
            # backward pass, and whenever a parameter's version has moved
            if lane.pass_id != self._pass_id or lane.ver != ver:
                # (EVERY lane at once: a lane that waited only at its own first call of the pass would find the caller's
                # stream already waiting for the calls before it, and the pass' first calls would run one after the other)
                lane.fork.record(cur)
                for ln in lanes:
                    ln.stream.wait_event(lane.fork)
                    ln.epoch += 1
                    ln.pass_id, ln.ver = self._pass_id, ver
            loss = self._lane_loss(lane, cur)
        ids = None if q_graphs is None else q_graphs.ids
        rng = self._mt()
        if (ids is not None and rng is not None and r.P_seen is step.P and r.ws_seen is (step._ws if lane is None else lane.ws)
                and ids.end - ids.start == B
                and ids.fi.A == A and (anchor_ids is None or anchor_ids is ids.anchor_ref)
                and torch._C._cuda_getDevice() == self.dev_index):
            # the batch is a window of its formula's id arrays and this record has run before: ONE host call copies the
            # window into the arena, draws the negatives and launches (csrc/host/pyhost.c: margin_call)
            fi, lo = ids.fi, 8 * ids.start
            if hard_negatives:
                if "inter" not in formula.query_type:
                    raise Exception("Hard negative examples can only be used with "
                                    "intersection queries")
                csr = fi.hard
            elif formula.query_type == "1-chain":
                _lst, full, full_ptr = self._full_list(formula.target_mode)
                csr = (None, None, None, full_ptr, 0, 0, full.shape[0])
            else:
                csr = fi.neg
        else:
            csr = None
        oa, og = ar.na, ar.ng
        if csr is not None:
            step.P.flags = step.flags | r.ps.step_flags
            self._calls = seq = (self._calls + 1) & 0xffffffff
            if len(csr) == 7:          # (one list for every query)
                lens_p, len_all, base_p = 0, csr[6], 0
            else:
                lens_p, len_all, base_p = csr[5] + lo, 0, csr[4] + lo
            st = self.host.margin_call(rng, self._step_fn, r.addr, A, B, fi.anchors_sm_ptr + lo, fi.anchors_sm_stride,
                                       fi.targets_ptr + lo, ar.a_ptr + 8 * oa, ar.t_ptr + 8 * og, ar.n_ptr + 8 * og,
                                       lens_p, len_all, base_p, csr[3], loss.data_ptr(),
                                       cur_raw if lane is None else lane.raw, margin, seq, 0 if lane is None else cur_raw)
            if st != 0:
                _capi.check(self.lib, st, 'mpqe_step_forward_backward_ex')
            ar.last[ch] = seq
            self.fast_sampled += 1
        else:
            self._fill_ids(ar, formula, queries, anchor_ids, ids, hard_negatives, B, A)
            ar.last[ch] = self._launch(r, False, False, ar.a_ptr + 8 * oa, ar.t_ptr + 8 * og, ar.n_ptr + 8 * og, loss, lane=lane)
        ar.na, ar.ng, ar.calls = oa + A * B, og + B, idx + 1
        if not torch.is_grad_enabled() or not step.params:
            return loss[0]
        call = _Call()
        call.arena, call.oa, call.og, call.idx, call.key, call.rid, call.B, call.A = ar, oa, og, idx, r.fb, r.rid, B, A
        call.margin, call.g = step.margin, None
        self._seq = seq = self._seq + 1
        call.seq = seq
        ar.live += 1
        if self._pass is not None:
            self._calls_by_id[seq] = call
            return self._node_ext.make_loss(self._pass, seq, loss)
        return _MarginLossNode.apply(self._hook, self, call, loss)

    # ------------------------------------------------------------------------------------------- backward
    def _on_backward(self, call, g):
        pending = self._pending
        if not pending:
            torch.autograd.Variable._execution_engine.queue_callback(self._flush)
        if g.dtype != torch.float32 or g.device != self.device:
            g = g.to(device=self.device, dtype=torch.float32)
        if call.g is None:
            pending.append(call)
            call.g = g
        else:                                  # (the node was reached twice in one pass)
            call.g = call.g + g

    def _drain_dead(self):
        """Calls whose C++ nodes are gone (their graphs were freed): drop the records -- their arenas' `live` counts fall."""
        if self._pass is not None:
            by_id = self._calls_by_id
            for i in self._pass.take_dead():
                by_id.pop(i, None)

    def _flush_ext(self, ids, grads):
        """The C++ nodes' one callback per backward pass (on the engine's thread, the GIL taken once)."""
        by_id, calls = self._calls_by_id, []
        for i, g in zip(ids, grads):
            c = by_id.get(i)
            if c is None:
                continue
            if g.dtype != torch.float32 or g.device != self.device:
                g = g.to(device=self.device, dtype=torch.float32)
            if c.g is None:
                calls.append(c)
                c.g = g
            else:
                c.g = c.g + g
        self._pending = calls
        self._flush()
        self._drain_dead()

    def _flush(self):
        calls, self._pending = self._pending, []
        if not calls:
            return
        calls.sort(key=lambda c: c.seq)
        step = self.step
        # p.grad: None everywhere (optimizer.zero_grad()) -> the step's own zero fill; otherwise added to what is there
        params, views = step.params, step._views
        if step.zero_next:                     # (FlatOptimizer.zero_grad(): every p.grad is still its view)
            zero, step.zero_next = True, False
        else:
            zero = True
            for p in params:
                if p.grad is not None:
                    zero = False
                    break
            if not zero:
                for p, v in zip(params, views):
                    if p.grad is None:
                        v.zero_()
                    elif p.grad is not v:
                        v.copy_(p.grad)
        step.bind_grads()
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
        for grp in groups:
            key = tuple(c.rid for c in grp)            # (ints: the one-batch records' numbers)
            r = self._multi.get(key) or self._record(self._multi, key, [c.key for c in grp])
            extra = r.extra
            for i, c in enumerate(grp):
                extra.batch_weight[i] = c.g.data_ptr()
            ar, c0 = grp[0].arena, grp[0]
            step.margin = c0.margin
            loss = torch.empty(1 + len(grp), dtype=torch.float32, device=self.device)
            ptrs = (ar.a_ptr + 8 * c0.oa, ar.t_ptr + 8 * c0.og, ar.n_ptr + 8 * c0.og)
            if self.checked:
                self._calls = seq = (self._calls + 1) & 0xffffffff
                extra.notify, extra.notify_value = self._note_ptr, seq
                step.run(r.ps, backward=True, zero_grad=zero, checked=True, id_ptrs=ptrs, extra=extra, out=(loss, None, None))
                ar.last[0] = seq
            else:
                ar.last[0] = self._launch(r, True, zero, ptrs[0], ptrs[1], ptrs[2], loss, extra=extra)
            zero = False
            self.steps += 1
        for c in calls:
            c.g = None
        self._pass_id += 1
        # the pass is over: its arena is closed (a node kept alive by retain_graph still finds its ids there)
        ar = self._arena
        if ar is not None and ar.calls:
            self._free.append(ar)
            self._arena = None

    # ------------------------------------------------------------------------------------------- forward (evaluation)
    def forward(self, formula, queries, target_nodes, anchor_ids=None, var_ids=None, q_graphs=None, neg_nodes=None,
                neg_lengths=None):
        """reference model.py:400-462 without autograd: scores [B] or [B + sum(neg_lengths)]."""
        m = self.model
        B = len(queries)
        r = self._one.get((0, formula, B)) or self._one_rec(formula, B)
        A = r.A
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
            if not self.step.uses_chain(r.ps):
                raise NotImplementedError('ragged negatives on the fused forward need the chain form')
            q = torch.empty(B, m.emb_dim, dtype=torch.float32, device=self.device)
            extra = _capi.StepExtra()
            extra.query_out = q.data_ptr()
        self.step.margin = 1.0
        ar.last[0] = self._launch(r, False, False, ar.a_ptr + 8 * oa, ar.t_ptr + 8 * og, ar.n_ptr + 8 * og, loss, scores[:B],
                                   scores[B:], extra=extra)
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

"""Whole-step execution of the encoder: every formula batch of one training step (the
reference's run_train inner loop, train_helpers.py:76-120: `loss = margin_loss(1-chain) +
path_weight * ... + inter_weight * ...; loss.backward()`) goes through ONE C-ABI call,
mpqe_step_forward_backward (~15 kernel launches), instead of one launch per op per batch.

    step = FusedTrainStep(model)                 # model: mpqe_amd.model.RGCNEncoderDecoder
    packed = step.pack(batches)                  # ids -> HBM, descriptors -> host structs
    loss = step.run(packed)                      # forward + backward; p.grad holds the gradients
    optimizer.step()

`p.grad` of every parameter is a view into one flat fp32 buffer, so the data-parallel gradient
exchange is a single all-reduce of that buffer with no packing copies (mpqe_amd/parallel.py).
"""
import ctypes
import sys
import time

import numpy as np
import torch

from . import _capi, ops
from .data_utils import RGCNQueryDataset


class StepBuffers(object):
    """Device buffers ONE packed step owns while it lives: the descriptor table (a function of the descriptor set alone:
    a recurring set finds it resident, hand-off epochs carried on), the id buffer (when the ids came from the host) and
    the touch plan. Handed back to the step object's pool by PackedStep.__del__ -- explicit ownership: a buffer set is
    either in the pool's free list or referenced by exactly one live PackedStep."""
    __slots__ = ('desc', 'desc_resident', 'ids', 'stage', 'last_use', 'touch', 'touch_ptr', 'skey')

    def __init__(self):
        self.desc = self.ids = self.stage = self.last_use = self.touch = self.touch_ptr = self.skey = None
        self.desc_resident = False


class PackedStep(object):
    __slots__ = ('batches', 'nb', 'anchor_ids', 'targets', 'negs', 'num_graphs', 'ws_bytes', 'sizes',
                 'desc', 'desc_bytes', 'desc_ptr', 'lanes', 'order', 'lane_begin', 'touch',
                 'touch_ptr', 'touch_entries', 'touch_sizes', 'bufs', 'owner', 'ids_ref', 'step_flags', 'touch_mode', 'captured')

    @property
    def desc_resident(self):
        return self.bufs.desc_resident

    def __del__(self):
        owner, bufs = getattr(self, 'owner', None), getattr(self, 'bufs', None)
        if owner is not None and bufs is not None:
            self.owner = self.bufs = None
            owner._release(bufs)


def _bsize(b):
    return int(b['batch_size']) if 'batch_size' in b else len(b['targets'])


_INFO = {}


def _template_info(query_type):
    if query_type not in _INFO:
        _INFO[query_type] = ops.template_info(query_type)
    return _INFO[query_type]


def _batch_work(query_type, passes):
    """Relative MFMA work of one batch per graph: passes * (edges + nodes) K-blocks."""
    e_n = {'1-chain': 3, '2-chain': 5, '3-chain': 7, '2-inter': 5, '3-inter': 7, '3-inter_chain': 7,
           '3-chain_inter': 7}[query_type]
    return passes * e_n


_TEMPLATES = {   # query type -> (num anchors, num nodes, [(src, dst)]) : reference data_utils.py:325-362
    '1-chain': (1, 2, [(0, 1)]), '2-chain': (1, 3, [(0, 2), (2, 1)]), '3-chain': (1, 4, [(0, 3), (3, 2), (2, 1)]),
    '2-inter': (2, 3, [(0, 2), (1, 2)]), '3-inter': (3, 4, [(0, 3), (1, 3), (2, 3)]),
    '3-inter_chain': (2, 4, [(0, 2), (1, 3), (3, 2)]), '3-chain_inter': (2, 4, [(0, 3), (1, 3), (3, 2)])}
CHAIN_MAX_GRAPHS = 1 << 20          # csrc/step.hip


def live_units(query_type, passes, readout, prune=True, uniform=False):
    """GEMM units ([B, D] x [D, D] products) the fused step executes per pass for one formula batch:
    units[p] = (source slot, destination slot) pairs of pass p -- edges and self terms -- whose destination
    state is live after pass p and whose source state is a per-graph one. A state is live when it can reach the
    readout (host mirror of the liveness masks in csrc/step.hip); without pruning every pass has E + N units
    (SURVEY.md 8d). uniform: states no anchor has reached yet are one vector per batch (csrc/step.hip: UOp) and
    their products are matrix-vector work done once per batch, not counted here. Forward, backward-x and the
    weight gradient each execute exactly these units."""
    A, N, edges = _TEMPLATES[query_type]
    full = (1 << N) - 1
    live = [0] * (passes + 1)
    live[passes] = (1 << A) if (prune and readout == 'mp') else full
    for p in range(passes - 1, -1, -1):
        m = live[p + 1]
        for s, d in edges:
            if (live[p + 1] >> d) & 1:
                m |= 1 << s
        live[p] = m if prune else full
    uni = [0] * (passes + 1)
    uni[0] = (full & ~((1 << A) - 1)) if uniform else 0
    for p in range(passes):
        m = uni[p]
        for s, d in edges:
            if not (uni[p] >> s) & 1:
                m &= ~(1 << d)
        uni[p + 1] = m
    units = [0] * passes
    for p in range(passes):
        pairs = [(s, d) for s, d in edges] + [(n, n) for n in range(N)]
        units[p] = sum(1 for s, d in pairs if (live[p + 1] >> d) & 1 and not (uni[p] >> s) & 1)
    return units


class CapturedStep(object):
    """A fused step recorded into a hipGraph together with everything its kernel nodes point at."""
    __slots__ = ('graph', 'loss', 'workspace', 'packed')

    def __init__(self, graph, loss, workspace, packed):
        self.graph, self.loss, self.workspace, self.packed = graph, loss, workspace, packed

    def replay(self):
        self.graph.replay()
        return self.loss


class FusedTrainStep(object):
    """lanes: number of HIP streams a step is spread over (1 = everything on the current stream).
    A step is a chain of ~13 dependent, very short launches; with lanes > 1 the batches are split
    into groups whose chains run concurrently on their own streams and meet before the weight
    gradients (include/mpqe_amd.h, mpqe_step_lanes_t). The split balances MFMA work and keeps
    batches of equal depth together (longest chains first)."""

    def __init__(self, model, margin=1.0, lanes=1, prune=True, chain=True, ksplit=True, eight_waves=False,
                 uniform=True, touch=True, sparse_tables=False, merge_tail=None, host_ids='direct'):
        """touch: how the entity-table gradients are accumulated (chain form; include/mpqe_amd.h) --
        True / 'step': per-entry gradient rows summed per table row in a fixed order, the plan (which looked-up ids share
        a row) built INSIDE the step by workgroups of its first launch (MPQE_STEP_BUILD_TOUCH): nothing id-dependent
        happens outside run(); steps too large for it fall back to 'pack'. 'pack': the same sums, the plan built by pack()
        (mpqe_step_touch_build) -- for callers that need the plan's keys before the step runs (StepExchange's row
        exchange). False: fp32 atomics.
        host_ids: how ids that arrive in host memory reach the kernels -- 'direct': the step reads them from the packed
        step's PINNED host buffer in place (176 KB per AIFB step, each id read twice: no copy to launch or wait for);
        'copy': one host-to-device copy per pack, in stream order (~20 us of the stream's time per step)."""
        enc = model.enc
        if not hasattr(enc, 'table') or getattr(enc, 'node_maps', None) is None:
            raise ValueError('FusedTrainStep needs a DirectEncoder built with node_maps')
        # learned readouts (reference model.py:441-446, 497-553): inside the same library call. `mlp` on the chain form (its
        # two Linear layers are two more levels of every graph block's programme: csrc/step_chain.h) when the dimension is
        # one of the chain kernel's and two layer slots are free; `targetmlp`, `concat` and the rest on the level form --
        # gather, the two Linear layers on the dense-layer kernels, the reduction over each graph's rows, and back
        # (csrc/step_readout.h; the node states must be in HBM)
        self.learned = model.readout_str in _capi.LEARNED_READOUT_IDS
        if not self.learned and model.readout_str not in _capi.READOUT_IDS:
            raise NotImplementedError('fused step: unknown readout %r' % model.readout_str)
        if self.learned:
            if model.emb_dim % 4:
                raise NotImplementedError('fused step with a learned readout: embedding dimension must be a multiple of 4')
            on_chain = (chain and model.emb_dim in (64, 128, 256) and model.num_layers <= 3 and not eight_waves and
                        (model.readout_str != 'concat' or not model.adaptive))
            if on_chain:
                lanes = 1
            else:
                if sparse_tables:
                    raise ValueError('sparse_tables needs the chain form (readouts sum / max / mp, mlp)')
                chain, touch, lanes = False, False, 1
        self.model = model
        self.margin = float(margin)
        # speed switches of the library call (include/mpqe_amd.h): identical loss / scores / gradients
        self.flags = ((0 if prune else _capi.STEP_NO_PRUNE) | (0 if chain else _capi.STEP_NO_CHAIN) |
                      (0 if ksplit else _capi.STEP_NO_KSPLIT) | (_capi.STEP_EIGHT_WAVES if eight_waves else 0) |
                      (0 if uniform else _capi.STEP_NO_UNIFORM) | (0 if merge_tail is None else _capi.STEP_MERGE_TAIL if merge_tail else _capi.STEP_SPLIT_TAIL))
        # chain form: weight-gradient tiles + backward post-pass as workgroups of the chain launch (True), as a launch of
        # their own (False), or the library's choice by step size (None; include/mpqe_amd.h MPQE_STEP_MERGE_TAIL)
        self.merge_tail = merge_tail
        # (concat on the chain form reads every node's state after every layer: the library leaves none to the pre-pass)
        self.uniform = bool(uniform and chain) and not (self.learned and model.readout_str == 'concat')
        if touch not in (True, False, 'step', 'pack'):
            raise ValueError("touch: True / 'step', 'pack' or False")
        if host_ids not in ('direct', 'copy'):
            raise ValueError("host_ids: 'direct' or 'copy'")
        self.host_ids = host_ids
        self.touch = bool(touch and chain)
        self.touch_mode = ('pack' if touch == 'pack' or eight_waves or lanes > 1 else 'step') if self.touch else None
        # row-sparse entity-table gradients (include/mpqe_amd.h: MPQE_STEP_SPARSE_TABLES): only the rows a step's ids touch
        # are written -- for FlatOptimizer(sparse_tables=True) / the row exchange; p.grad of a table is then NOT a
        # dense gradient (untouched rows hold whatever they held)
        self.sparse_tables = bool(sparse_tables)
        if self.sparse_tables and not self.touch:
            raise ValueError('sparse_tables needs the chain form with the touch plan')
        if self.sparse_tables:
            self.flags |= _capi.STEP_SPARSE_TABLES
        self.device = next(model.parameters()).device
        if self.device.type != 'cuda':
            raise RuntimeError('mpqe_amd: the model must be on the GPU -- there is no CPU path')
        self.modes = list(model.mode_ids.keys())          # table index = mode id order
        if any(not p.requires_grad for p in model.parameters()):
            # (the step writes every gradient buffer; a frozen table or layer has none: the module path handles that model)
            raise ValueError('FusedTrainStep: every parameter of the model must require grad')
        self.params = [p for p in model.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._views = []
        off = 0
        for p in self.params:
            self._views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.bind_grads()
        self.err = ops.new_error_word(self.device)
        self._ws = None
        self._desc_cache, self._size_cache, self._pool = {}, {}, {}
        # descriptor sets whose in-step touch plan failed once (run(checked=True)): their plans are built by pack() from then on
        self._pack_touch_sets = set()
        self.param_epoch = 0               # bumped by writers of the parameters that bypass autograd's version counters (FlatOptimizer.step)
        self.zero_next = False             # FlatOptimizer.zero_grad(): the drop-in's next backward pass zero-fills (mpqe_amd/dropin.py)
        self.touch_retries = 0
        self.handoff_retries = 0           # steps run(checked=True) ran again in the level form (an in-launch hand-off timed out)
        self.num_lanes = max(1, min(int(lanes), _capi.STEP_MAX_LANES))
        self._streams = [None] + [torch.cuda.Stream(device=self.device) for _ in range(self.num_lanes - 1)]
        self._fork = torch.cuda.Event()
        self._joins = [None] + [torch.cuda.Event() for _ in range(self.num_lanes - 1)]
        with torch.cuda.device(self.device):
            self._fork.record()            # creates the underlying hipEvent_t handles
            for e in self._joins[1:]:
                e.record()
        self._refresh_pointers()

    POOL_PER_SET = 8

    def _release(self, bufs):
        """A packed step is gone: its buffers serve the next step of the same descriptor set (PackedStep.__del__)."""
        free = self._pool.get(bufs.skey)
        if free is not None and len(free) < self.POOL_PER_SET:
            free.append(bufs)

    def bind_grads(self):
        """Make every p.grad the parameter's view of the flat gradient buffer (again). Anything that
        replaced p.grad in between -- autograd after zero_grad(set_to_none=True), an optimizer that
        sets grads to None -- is undone here; run() calls it, so the kernels always write where
        p.grad reads."""
        for p, v in zip(self.params, self._views):
            if p.grad is not v:
                p.grad = v

    def _refresh_pointers(self):
        m = self.model
        tabs = [m.enc.table(mode) for mode in self.modes]
        for t in tabs + [m.mode_embeddings.weight]:
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise RuntimeError('parameters must be contiguous fp32 CUDA tensors')
        layers = list(m.layers)
        gv = {id(p): v for p, v in zip(self.params, self._views)}       # (the gradient views, whatever p.grad is bound to now)

        def gptr(t):
            return gv[id(t)].data_ptr()
        self.P = _capi.make_step_params(
            m.emb_dim, layers[0].num_relations,
            _capi.LEARNED_READOUT_IDS[m.readout_str] if self.learned else m.readout_str, [t.data_ptr() for t in tabs],
            [t.shape[0] for t in tabs], m.enc.node_maps.data_ptr(), m.enc.node_maps.shape[0],
            m.mode_embeddings.weight.data_ptr(), [l.basis.data_ptr() for l in layers],
            [l.root.data_ptr() for l in layers], [l.bias.data_ptr() for l in layers], flags=self.flags)
        self.G = _capi.make_step_grads(
            [gptr(t) for t in tabs], gptr(m.mode_embeddings.weight),
            [gptr(l.basis) for l in layers], [gptr(l.root) for l in layers],
            [gptr(l.bias) for l in layers])
        if self.learned:
            lay = m.readout.layers            # nn.Sequential(Linear, ReLU, Linear): state_dict keys layers.0 / layers.2
            ro = (lay[0].weight, lay[0].bias, lay[2].weight, lay[2].bias)
            for t in ro:
                if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.requires_grad):
                    raise RuntimeError('readout parameters must be trainable contiguous fp32 CUDA tensors')
            for field, t in zip(('readout_w0', 'readout_b0', 'readout_w2', 'readout_b2'), ro):
                setattr(self.P, field, t.data_ptr())
                setattr(self.G, field, gptr(t))
            self.P.readout_scatter = _capi.SCATTER_IDS[{ops.scatter_add: 'add', ops.scatter_max: 'max',
                                                        ops.scatter_mean: 'mean'}[m.readout.scatter_fn]]
            self.P.readout_weight_decay = float(m.weight_decay)
        self._keep = (tabs, layers)

    def flatten_ids(self, batches, out=None):
        """The ids of a step in the layout the library reads: [anchors of batch 0 (slot-major: [A, B]) | ... | targets of
        all batches | negatives of all batches], as one int64 numpy array (`out`: write there). batches: dicts with
        formula, anchor_ids [B, A], targets [B], negs [B]."""
        acols = [_TEMPLATES[b['formula'].query_type][0] for b in batches]
        sizes = [len(b['targets']) for b in batches]
        na, ngr = sum(B * A for B, A in zip(sizes, acols)), sum(sizes)
        snp = np.empty(na + 2 * ngr, dtype=np.int64) if out is None else out
        if snp.shape != (na + 2 * ngr,):
            raise ValueError('flatten_ids: out must hold %d ids' % (na + 2 * ngr))
        oa = 0
        tl, nl_ = [], []
        for b, B, A in zip(batches, sizes, acols):
            a, t, n = b['anchor_ids'], b['targets'], b['negs']
            if not isinstance(a, np.ndarray):
                a = a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
            if not isinstance(t, np.ndarray):
                t = t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
            if not isinstance(n, np.ndarray):
                n = n.cpu().numpy() if torch.is_tensor(n) else np.asarray(n)
            if a.shape != (B, A):
                raise ValueError('anchor_ids must be [B, %d] for %s' % (A, b['formula'].query_type))
            if t.shape != (B,) or n.shape != (B,):
                raise ValueError('targets / negs must have one id per query')
            if A == 1:
                snp[oa:oa + B] = a[:, 0]
            else:
                np.copyto(snp[oa:oa + B * A].reshape(A, B), a.T)
            tl.append(t)
            nl_.append(n)
            oa += B * A
        # (targets and negatives of all batches: one concatenation each, straight into the buffer)
        np.concatenate(tl, out=snp[na:na + ngr], casting='unsafe')
        np.concatenate(nl_, out=snp[na + ngr:na + 2 * ngr], casting='unsafe')
        return snp

    def pack(self, batches, ids=None):
        """batches: list of dicts with keys formula, anchor_ids ([B, A] int64 tensor/array),
        targets, negs ([B] ids), weight. Returns the HBM-resident packed step.
        ids: the step's ids as ONE int64 array in flatten_ids' layout -- the batches then need only formula, weight and
        their size (`batch_size`, or `targets` for its length). A CUDA tensor (a loader that stages ids itself, ids drawn
        on the device): pack() does no device work at all; the step reads the tensor when it runs, keep its contents
        until then. A numpy array / CPU tensor (a collate function that writes this layout directly): one copy into
        pinned staging + one host-to-device copy, in stream order."""
        m = self.model
        nb = len(batches)
        if nb == 0 or nb > _capi.STEP_MAX_BATCHES:
            raise ValueError('a step holds 1..%d batches' % _capi.STEP_MAX_BATCHES)
        # lane assignment: deepest chains first, each batch to the lane with the least work so far;
        # the library wants every lane to own a contiguous range of batches, so the batches are
        # re-ordered (packed.order[i] = caller's index of library batch i)
        passes_of = []
        for b in batches:
            qt = b['formula'].query_type
            passes_of.append(RGCNQueryDataset.query_diameters[qt] if m.adaptive else m.num_layers)
        if self.learned and m.readout_str == 'concat' and any(p != m.num_layers for p in passes_of):
            raise ValueError('concat readout: every batch must run num_layers passes (adaptive=False)')
        nl = min(self.num_lanes, nb)
        graphs = sum(_bsize(b) for b in batches)
        chain = (not (self.flags & _capi.STEP_NO_CHAIN) and m.emb_dim in (64, 128, 256)
                 and graphs <= CHAIN_MAX_GRAPHS and max(passes_of) <= 5)
        if chain:
            # chain form: one launch sequence per step on the caller's stream whatever the split (the library ignores it;
            # batch order is kept)
            members = [list(range(nb))]
        else:
            load = [0.0] * nl
            members = [[] for _ in range(nl)]
            for i in sorted(range(nb), key=lambda i: (-passes_of[i], -_batch_work(batches[i]['formula'].query_type,
                                                                                    passes_of[i]))):
                l = min(range(nl), key=lambda l: load[l])
                members[l].append(i)
                load[l] += _batch_work(batches[i]['formula'].query_type, passes_of[i]) * _bsize(batches[i])
        members = [sorted(mm) for mm in members if mm]
        order = [i for mm in members for i in mm]
        lane_begin = [0]
        for mm in members:
            lane_begin.append(lane_begin[-1] + len(mm))
        batches = [batches[i] for i in order]
        # descriptors: per (formula, passes) everything that does not depend on the ids is cached (a training run
        # draws from a finite set of formulas); ids go straight into ONE pinned staging buffer (numpy views, no
        # per-batch tensors) and to the device in ONE copy: [anchors (slot-major per batch) | targets | negatives]
        prof = getattr(self, '_prof', None)       # (tools/pack_profile.py: seconds per section, or None)
        t0 = time.perf_counter() if prof is not None else 0.0
        SB = (_capi.StepBatch * nb)()
        sizes, acols = [], []
        for i, b in enumerate(batches):
            f = b['formula']
            passes = passes_of[order[i]]
            key = (f, passes)
            proto = self._desc_cache.get(key)
            if proto is None:
                info = _template_info(f.query_type)
                if m.adaptive and passes > len(m.layers):
                    raise ValueError(f'RGCN is adaptive with {len(m.layers)}'
                                     f' layers, but query requires {passes}.')
                nodes, rels = f.get_nodes(), f.get_rels()
                edge_type = [m.rel_ids[(rels[info.rel_label[e]][2], rels[info.rel_label[e]][1],
                                        rels[info.rel_label[e]][0])] for e in range(info.num_edges)]
                var_ids = [m.mode_ids[nodes[info.var_node[k]]] for k in range(info.num_vars)]
                proto = (_capi.make_step_batch(f.query_type, passes, 1, edge_type, var_ids,
                                               [m.mode_ids[x] for x in f.anchor_modes], m.mode_ids[f.target_mode], 1.0),
                         info.num_anchors)
                if len(self._desc_cache) > 65536:
                    self._desc_cache.clear()
                self._desc_cache[key] = proto
            B = _bsize(b)
            ctypes.memmove(ctypes.addressof(SB[i]), ctypes.addressof(proto[0]), ctypes.sizeof(_capi.StepBatch))
            SB[i].batch_size = B
            SB[i].weight = float(b.get('weight', 1.0))
            sizes.append(B)
            acols.append(proto[1])
        if prof is not None:
            t1 = time.perf_counter(); prof['descriptors'] = prof.get('descriptors', 0.0) + t1 - t0; t0 = t1
        na, ngr = sum(B * A for B, A in zip(sizes, acols)), sum(sizes)
        ps = PackedStep()
        ps.lanes = None
        if len(members) > 1:
            L = _capi.StepLanes()
            L.num_lanes = len(members)
            for i, v in enumerate(lane_begin):
                L.batch_begin[i] = v
            L.fork_event = self._fork.cuda_event
            for l in range(1, len(members)):
                L.aux_stream[l] = self._streams[l].cuda_stream
                L.join_event[l] = self._joins[l].cuda_event
            ps.lanes = ctypes.pointer(L)
        # sizes: functions of the descriptors alone -- cached per descriptor set (one planning pass on a miss; the
        # step's first run takes that plan over)
        # where the touch plan is built: inside the step (nothing id-dependent left for pack), or here
        mode = self.touch_mode
        external = isinstance(ids, str)
        if mode == 'step' and (not chain or na + 2 * ngr > _capi.TSORT_MAX_ENTRIES):
            # (the level form has no use for a plan: a step whose ids are named per run carries none)
            mode = None if (external and not chain) else 'pack'
        if mode == 'step' and self._pack_touch_sets and (bytes(SB), tuple(lane_begin)) in self._pack_touch_sets:
            mode = 'pack'
        ps.touch_mode = mode
        ps.step_flags = _capi.STEP_BUILD_TOUCH if mode == 'step' else 0
        skey = (bytes(SB), tuple(lane_begin), mode)
        sz = self._size_cache.get(skey)
        if sz is None:
            lib = ops.lib()
            self.P.flags = self.flags | ps.step_flags         # (the launch plan, hence the sizes, depend on it)
            sz = (lib.mpqe_step_workspace_bytes(ctypes.byref(self.P), SB, nb, ps.lanes),
                  lib.mpqe_step_desc_bytes(ctypes.byref(self.P), SB, nb, ps.lanes),
                  lib.mpqe_step_touch_bytes(ctypes.byref(self.P), SB, nb),
                  lib.mpqe_step_touch_workspace_bytes(ctypes.byref(self.P), SB, nb),
                  int(lib.mpqe_step_touch_entries(SB, nb)))
            if sz[0] == 0:
                raise _capi.MpqeError('mpqe_step_workspace_bytes rejected the step descriptors')
            if len(self._size_cache) > 4096:
                self._size_cache.clear()
            self._size_cache[skey] = sz
        ps.ws_bytes, ps.desc_bytes = sz[0], sz[1]
        # Buffers of this packed step (StepBuffers): from the free list of its descriptor set, or new. The descriptor table
        # is a function of the set alone (not of the ids), so a buffer set whose packed step is gone serves the next step
        # as it is -- table resident, hand-off epochs carried on (they only ever grow): a training loop that draws fresh
        # ids for a recurring set of formulas uploads nothing. (One stream per FusedTrainStep, as for the workspace: a
        # re-used buffer's previous step is ordered before the next one.)
        free = self._pool.get(skey)
        if free is None:
            if len(self._pool) > 1024:
                self._pool.clear()
            free = self._pool[skey] = []
        # (FIFO, and at least three sets take turns before one is used again: plan buffers a caller may still want to read
        # -- the optimiser's row plan of the step before -- are not overwritten by the very next pack)
        if len(free) >= 3:
            bufs = free.pop(0)
        else:
            bufs = StepBuffers()
            bufs.skey = skey
            bufs.desc = torch.empty(ps.desc_bytes + 256, dtype=torch.uint8, device=self.device)
        ps.bufs, ps.owner = bufs, self
        ps.desc = bufs.desc
        ps.desc_ptr = (ps.desc.data_ptr() + 255) // 256 * 256
        ps.batches, ps.nb, ps.sizes = SB, nb, sizes
        ps.order, ps.lane_begin = order, lane_begin
        ps.num_graphs = int(ngr)
        ps.touch_entries = sz[4]
        ps.touch_sizes = (sz[2], sz[3])
        ps.touch, ps.touch_ptr = None, None
        ps.captured = False
        n_ids = na + 2 * ngr
        if isinstance(ids, str):
            if ids != 'external':
                raise ValueError("ids: an id array, or 'external' (the ids' addresses are handed to run(id_ptrs=...))")
            # descriptors and buffers only: every run names the three id arrays itself (mpqe_amd/dropin.py: the calls of one
            # training step append their ids to pinned arenas the kernels read in place)
            ps.ids_ref = dev = None
        elif ids is not None and torch.is_tensor(ids) and ids.is_cuda:
            if not (ids.dtype == torch.long and ids.is_contiguous() and ids.numel() == n_ids and ids.device == self.device):
                raise ValueError('ids: a contiguous int64 tensor of %d ids (flatten_ids layout)' % n_ids)
            ps.ids_ref = dev = ids
        else:
            # ids from the host. 'direct': into the PINNED buffer this packed step owns (pooled with its other buffers); the
            # kernels read it in place -- pinned host memory is mapped into the device's address space --, so there is no
            # copy to launch, to order or to wait for. The buffer is refilled only after the step that read it last has
            # finished (an event per buffer set; three sets take turns, so the wait is over before it is asked for).
            # 'copy': pinned staging ring -> ONE host-to-device copy in stream order (behind the step that read this id
            # buffer last, in front of the one that will). (A separate copy stream with events both ways was measured
            # too: the loop is host-bound either way and the copy sometimes queued behind the running step -- 0.078 or
            # 0.185 ms per step by run.)
            direct = self.host_ids == 'direct'
            if direct:
                if bufs.stage is None or bufs.stage.numel() < n_ids:
                    bufs.stage = torch.empty(max(n_ids, 1 << 12), dtype=torch.long, pin_memory=True)
                    bufs.last_use = None
                if bufs.last_use is not None:
                    bufs.last_use.synchronize()
                stage = bufs.stage[:n_ids]
            else:
                stage = self._staging(n_ids)
            if ids is None:
                self.flatten_ids(batches, out=stage.numpy())
            else:
                h = ids.numpy() if torch.is_tensor(ids) else ids
                if not (isinstance(h, np.ndarray) and h.dtype == np.int64 and h.shape == (n_ids,)):
                    raise ValueError('ids: an int64 array of %d ids (flatten_ids layout)' % n_ids)
                np.copyto(stage.numpy(), h)
            if prof is not None:
                t1 = time.perf_counter(); prof['ids to staging'] = prof.get('ids to staging', 0.0) + t1 - t0; t0 = t1
            if direct:
                dev = stage
                ps.ids_ref = bufs.stage
            else:
                if bufs.ids is None or bufs.ids.numel() < n_ids:
                    bufs.ids = torch.empty(n_ids, dtype=torch.long, device=self.device)
                dev = bufs.ids[:n_ids]
                cs = torch.cuda.current_stream(self.device)
                # (the library's own hipMemcpyAsync wrapper: copy_ through torch is ~3x the host time)
                st = ops.lib().mpqe_copy_to_device(dev.data_ptr(), stage.data_ptr(), 8 * n_ids, cs.cuda_stream)
                _capi.check(ops.lib(), st, 'mpqe_copy_to_device')
                self._stage_events[self._stage_next].record(cs)      # the staging buffer is free again once this copy has run
                ps.ids_ref = bufs.ids
                if prof is not None:
                    t1 = time.perf_counter(); prof['copy to device'] = prof.get('copy to device', 0.0) + t1 - t0; t0 = t1
        if dev is None:
            ps.anchor_ids = ps.targets = ps.negs = None
        else:
            ps.anchor_ids, ps.targets, ps.negs = dev[:na], dev[na:na + ngr], dev[na + ngr:]
        if self.touch and mode is not None:
            if bufs.touch is None:
                # (zero-filled once: the step reads plan entries before it knows whether its own build finished, include/mpqe_amd.h)
                bufs.touch = torch.zeros(sz[2] + 256, dtype=torch.uint8, device=self.device)
                bufs.touch_ptr = (bufs.touch.data_ptr() + 255) // 256 * 256
            ps.touch, ps.touch_ptr = bufs.touch, bufs.touch_ptr
            if mode == 'pack':
                if dev is None:
                    raise ValueError("ids='external' needs the in-step touch plan (touch='step')")
                self.build_touch(ps)
            if prof is not None:
                t1 = time.perf_counter(); prof['touch plan'] = prof.get('touch plan', 0.0) + t1 - t0
        return ps

    def build_touch(self, ps, library_sort=False, id_ptrs=None):
        """The touch plan of the packed step's ids (include/mpqe_amd.h: mpqe_step_touch_build): which looked-up
        entities share a table row, sorted once here, so that the step adds their gradient rows in a fixed order
        instead of with float atomics. Stream-ordered on the current stream, no synchronisation. Call it again
        after refilling ps.anchor_ids / targets / negs in place with new ids. library_sort: the multi-launch library
        sort, which does not depend on how many workgroups the device holds at once (the recovery path)."""
        L = ops.lib()
        self.P.flags = self.flags | (_capi.STEP_TOUCH_LIBRARY_SORT if library_sort else 0)
        nbytes, wbytes = ps.touch_sizes
        if nbytes == 0:
            raise _capi.MpqeError('mpqe_step_touch_bytes rejected the step descriptors')
        if ps.touch is None:
            ps.touch = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
            ps.touch_ptr = (ps.touch.data_ptr() + 255) // 256 * 256
        # (build workspace: ONE buffer per step object, grown on demand -- every build runs on the current stream, so the
        # next build's kernels are ordered behind this one's; pack time is host time, and an allocation + record_stream
        # per pack was a quarter of the touch plan's)
        ws = getattr(self, '_touch_ws', None)
        if ws is None or ws.numel() < wbytes + 256:
            ws = self._touch_ws = torch.empty(max(wbytes + 256, 1 << 20), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        if getattr(self, '_touch_ws_stream', None) not in (None, stream.cuda_stream):
            stream.wait_stream(self._touch_ws_stream_obj)      # (a caller that switched streams: order the re-use)
        self._touch_ws_stream, self._touch_ws_stream_obj = stream.cuda_stream, stream
        if id_ptrs is None:
            id_ptrs = (ps.anchor_ids.data_ptr(), ps.targets.data_ptr(), ps.negs.data_ptr())
        with torch.cuda.device(self.device):
            st = L.mpqe_step_touch_build(ctypes.byref(self.P), ps.batches, ps.nb, id_ptrs[0], id_ptrs[1], id_ptrs[2],
                                         ps.touch_ptr, nbytes, (ws.data_ptr() + 255) // 256 * 256, wbytes, stream.cuda_stream)
        _capi.check(L, st, 'mpqe_step_touch_build')

    def _staging(self, n):
        """A pinned host buffer of >= n int64 from a small ring (allocating pinned memory costs more than the rest of
        pack()); a buffer is handed out again only after the copy that read it last has completed."""
        if not hasattr(self, '_stage_ring'):
            self._stage_ring, self._stage_events, self._stage_next = [None] * 4, [None] * 4, -1
        k = self._stage_next = (self._stage_next + 1) % 4
        if self._stage_events[k] is not None:
            self._stage_events[k].synchronize()
        else:
            self._stage_events[k] = torch.cuda.Event()
        if self._stage_ring[k] is None or self._stage_ring[k].numel() < n:
            self._stage_ring[k] = torch.empty(max(n, 1 << 15), dtype=torch.long, pin_memory=True)
        return self._stage_ring[k][:n]

    def uses_chain_dims(self):
        """True when steps of this model run the chain kernels (whatever the batches: at most 5 passes assumed)."""
        return not (self.flags & _capi.STEP_NO_CHAIN) and self.model.emb_dim in (64, 128, 256)

    def uses_chain(self, packed):
        """True when the library runs the graph-block chain kernels for this step (csrc/step.hip)."""
        passes = max(int(packed.batches[i].num_passes) for i in range(packed.nb))
        return (not (self.flags & _capi.STEP_NO_CHAIN) and self.model.emb_dim in (64, 128, 256)
                and packed.num_graphs <= CHAIN_MAX_GRAPHS and passes <= (3 if self.learned else 5))

    def merged(self, packed):
        """True when the weight-gradient tiles and the backward post-pass ride in the chain launch (two launches per
        step; include/mpqe_amd.h MPQE_STEP_MERGE_TAIL -- the library's rule, mirrored for the bench's accounting)."""
        if not self.uses_chain(packed) or len(packed.lane_begin) > 2 or self.merge_tail is False:
            return False
        blocks = sum((int(b) + 15) // 16 for b in packed.sizes)
        return bool(self.merge_tail) or blocks <= 256 + 256 // 8

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes + 256:
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        return (self._ws.data_ptr() + 255) // 256 * 256

    def run(self, packed, backward=True, zero_grad=True, scores=False, events=None, workspace=None, checked=False,
            id_ptrs=None, extra=None, out=None):
        """Returns loss [1 + nb] on the device: loss[0] = weighted step loss, loss[1 + i] = mean hinge
        of library batch i = the caller's batch packed.order[i] (identity with one lane); scores come
        back in the same library order. With backward=True every p.grad then holds d loss[0] / d p
        (accumulated on top of the previous content unless zero_grad). workspace: a uint8 tensor of at
        least packed.ws_bytes + 256 bytes to run in instead of the step object's shared arena (a
        captured graph owns its arena: see capture()).
        checked: read the error word before returning (ONE 4-byte device-to-host read: the call then waits for the
        step) -- the gradients are then known to be whole before an optimiser can consume them. A step that could not
        build its own touch plan (MPQE_FLAG_TOUCH_RETRY: the sort's workgroups were not all resident at once, e.g. on
        a GPU shared with another process; the reference's embedding backward, encoders.py:40-43, has no such failure
        mode) is RECOVERED here: the plan is rebuilt with the library sort, the entity-table rows are summed again
        from the per-entry rows the step left in its workspace (mpqe_step_table_rows), and this descriptor set takes
        pack-time plans from then on; a bad id raises IndexError, a timed-out hand-off RuntimeError, as check() does
        -- but now, not a step later.
        id_ptrs: (anchor ids, targets, negatives) as three addresses the DEVICE can read (a step packed with ids='external').
        extra: a _capi.StepExtra (include/mpqe_amd.h: mpqe_step_extra_t -- per-batch loss weights as device scalars, the query
        embeddings out). out: (loss [1 + nb], scores_pos, scores_neg) tensors to write instead of new ones (scores: None = not
        wanted)."""
        if backward:
            self.bind_grads()
        # the library zero-fills the gradient buffers itself (one launch with its other prologue work)
        self.P.flags = self.flags | packed.step_flags | (_capi.STEP_ZERO_GRADS if (backward and zero_grad) else 0)
        bufs = packed.bufs
        stream = torch.cuda.current_stream(self.device)
        sp = sn = None
        if out is not None:
            loss, sp, sn = out
            scores = sp is not None
        else:
            loss = torch.empty(1 + packed.nb, dtype=torch.float32, device=self.device)
            if scores:
                sp = torch.empty(packed.num_graphs, dtype=torch.float32, device=self.device)
                sn = torch.empty_like(sp)
        if id_ptrs is None:
            if packed.anchor_ids is None:
                raise ValueError("a step packed with ids='external' runs with id_ptrs=(anchors, targets, negatives)")
            id_ptrs = (packed.anchor_ids.data_ptr(), packed.targets.data_ptr(), packed.negs.data_ptr())
        elif packed.touch_mode == 'pack':
            raise ValueError('id_ptrs: steps with the in-step touch plan only')
        if workspace is None:
            wptr = self._workspace(packed.ws_bytes)
        else:
            if workspace.numel() < packed.ws_bytes + 256 or workspace.device != self.device:
                raise ValueError('workspace too small for this packed step')
            wptr = (workspace.data_ptr() + 255) // 256 * 256
        L = ops.lib()
        args = (ctypes.byref(self.P), packed.batches, packed.nb, id_ptrs[0], id_ptrs[1], id_ptrs[2], self.margin,
                ctypes.byref(self.G),
                1 if backward else 0, loss.data_ptr(), None if sp is None else sp.data_ptr(),
                None if sn is None else sn.data_ptr(), packed.desc_ptr, packed.desc_bytes,
                0 if bufs.desc_resident else 1, wptr, packed.ws_bytes, self.err.data_ptr(), packed.lanes,
                events, 0 if events is None else len(events), packed.touch_ptr,
                stream.cuda_stream)
        fn = L.mpqe_step_forward_backward
        if extra is not None:
            fn, args = L.mpqe_step_forward_backward_ex, args + (ctypes.byref(extra),)
        if torch.cuda.current_device() != self.device.index:        # (the context manager costs ~10 us of host time)
            with torch.cuda.device(self.device):
                st = fn(*args)
        else:
            st = fn(*args)
        _capi.check(L, st, 'mpqe_step_forward_backward')
        bufs.desc_resident = True
        if checked:
            flags = int(self.err.item())                  # (synchronises)
            if flags & _capi.FLAG_TOUCH_RETRY and backward and packed.touch_ptr is not None:
                self.err.fill_(flags & ~_capi.FLAG_TOUCH_RETRY)
                self.touch_retries += 1
                self._pack_touch_sets.add((bytes(packed.batches), tuple(packed.lane_begin)))
                self.build_touch(packed, library_sort=True, id_ptrs=id_ptrs)
                self.P.flags = self.flags | packed.step_flags | (_capi.STEP_ZERO_GRADS if zero_grad else 0)
                with torch.cuda.device(self.device):
                    st = L.mpqe_step_table_rows(ctypes.byref(self.P), packed.batches, packed.nb, ctypes.byref(self.G),
                                                packed.desc_ptr, wptr, packed.ws_bytes, packed.touch_ptr, stream.cuda_stream)
                _capi.check(L, st, 'mpqe_step_table_rows')
                flags &= ~_capi.FLAG_TOUCH_RETRY
            if flags & _capi.FLAG_INTERNAL and not (flags & 0xf) and backward and zero_grad and self.uses_chain(packed):
                # An in-launch hand-off between workgroups timed out (a producer lost its CU to another process: one process
                # per GPU is what the chain form is built for). What the launches left in the gradient buffers is not to be
                # trusted; the LEVEL form (one launch per message-passing level, include/mpqe_amd.h MPQE_STEP_NO_CHAIN) hands
                # nothing from workgroup to workgroup inside a launch: the step runs again there, into the same buffers --
                # slower, never wrong. (Without the call's own zero fill the garbage cannot be taken back: that raises.)
                self.err.fill_(flags & ~(_capi.FLAG_INTERNAL | 0xff00))
                self.handoff_retries += 1
                self._rerun_level_form(packed, loss, sp, sn, stream, id_ptrs, extra)
                flags = int(self.err.item())
            if flags:
                ops.raise_on_flags(self.err)
        if packed.ids_ref is not None and packed.ids_ref is bufs.stage and bufs.stage is not None and not packed.captured:
            if bufs.last_use is None:
                bufs.last_use = torch.cuda.Event()
            bufs.last_use.record(stream)                # (the pinned id buffer may be refilled once this step has run)
        if scores:
            return loss, sp, sn
        return loss

    def _rerun_level_form(self, packed, loss, sp, sn, stream, id_ptrs, extra=None):
        """The packed step once more through the level form (no in-launch hand-offs), forward + backward with the call's own
        zero fill, into the same loss / score / gradient buffers. Its plan, descriptor table and workspace are its own."""
        L = ops.lib()
        self.P.flags = (self.flags | _capi.STEP_NO_CHAIN | _capi.STEP_ZERO_GRADS) & ~(_capi.STEP_SPARSE_TABLES | _capi.STEP_MERGE_TAIL |
                                                                                    _capi.STEP_SPLIT_TAIL)
        if self.sparse_tables:
            raise _capi.MpqeError('an in-launch hand-off timed out and row-sparse table gradients have no level form to fall back to')
        wsb = L.mpqe_step_workspace_bytes(ctypes.byref(self.P), packed.batches, packed.nb, None)
        dsb = L.mpqe_step_desc_bytes(ctypes.byref(self.P), packed.batches, packed.nb, None)
        if wsb == 0 or dsb == 0:
            raise _capi.MpqeError('the level form rejected the step descriptors')
        ws = torch.empty(wsb + 256, dtype=torch.uint8, device=self.device)
        desc = torch.empty(dsb + 256, dtype=torch.uint8, device=self.device)
        args = (ctypes.byref(self.P), packed.batches, packed.nb, id_ptrs[0], id_ptrs[1], id_ptrs[2], self.margin,
                ctypes.byref(self.G), 1, loss.data_ptr(),
                None if sp is None else sp.data_ptr(), None if sn is None else sn.data_ptr(),
                (desc.data_ptr() + 255) // 256 * 256, dsb, 1, (ws.data_ptr() + 255) // 256 * 256, wsb, self.err.data_ptr(), None,
                None, 0, None, stream.cuda_stream, None if extra is None else ctypes.byref(extra))
        with torch.cuda.device(self.device):
            st = L.mpqe_step_forward_backward_ex(*args)
        _capi.check(L, st, 'mpqe_step_forward_backward (level form)')
        torch.cuda.synchronize(self.device)          # (ws / desc are this call's own: they must outlive its launches)

    def capture(self, packed, backward=True, zero_grad=True):
        """Record one step on `packed` into a hipGraph. Returns a CapturedStep: .replay() re-runs the step on
        the buffers of `packed` (refill them in place for a new set of queries of the same formulas) and
        leaves the losses in .loss. The library call is capturable because it neither allocates nor
        synchronises once the descriptor table is resident, which the warm run ensures.

        Every capture owns its workspace: the graph's kernel nodes hold the arena's ADDRESS, so it must stay
        mapped for as long as the graph may be replayed. (Round 1 captured into the step object's shared arena,
        which the next, larger packed step re-allocated -- torch.cuda.graph's empty_cache() then returned the
        old block to the driver and an earlier graph replayed into unmapped memory: the 'memory access fault'
        the chain form showed. The level form had the same exposure.)"""
        ws = torch.empty(packed.ws_bytes + 256, dtype=torch.uint8, device=self.device)
        # (the graph replays on this packed step's buffers for as long as it lives: they never go back to the pool, and
        # no event of the eager path is recorded inside the capture)
        packed.captured = True
        packed.bufs.skey = None
        self.run(packed, backward, zero_grad, workspace=ws)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = self.run(packed, backward, zero_grad, workspace=ws)
        return CapturedStep(graph, loss, ws, packed)

    def check_touch(self, packed):
        """Raise if the touch plan of `packed` could not be built (its sort's workgroups were not all resident at once:
        csrc/step_touch.h). One D2H read of the plan's header: for callers that read the plan's keys themselves."""
        if packed.touch is None:
            return
        base = packed.touch_ptr - packed.touch.data_ptr()
        if int(packed.touch[base + 16: base + 20].view(torch.int32).item()) != 0:
            raise _capi.MpqeError('touch plan: the one-launch sort could not finish (MPQE_FLAG_INTERNAL)')

    def check(self):
        """Raise IndexError if any kernel of a previous run saw an invalid entity id (one D2H read)."""
        ops.raise_on_flags(self.err)

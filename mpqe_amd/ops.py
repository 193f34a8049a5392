"""torch.autograd bindings of the C ABI (include/mpqe_amd.h). PyTorch here is plumbing:
device memory (caching allocator), the current HIP stream and autograd bookkeeping. All
arithmetic runs in the gfx950 kernels; tensors must be CUDA fp32/int64 and contiguous,
anything else raises -- there is no CPU or eager-PyTorch path.
"""
import numpy as np
import torch

from . import _capi, _lib
from ._capi import QUERY_TYPE_IDS, READOUT_IDS, SCATTER_IDS


def lib():
    return _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _need(t, dtype, what):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise RuntimeError('mpqe_amd: %s must be a CUDA (ROCm) tensor -- there is no CPU path' % what)
    if t.dtype != dtype:
        raise TypeError('mpqe_amd: %s must be %s, got %s' % (what, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _f(t, what):
    return _need(t, torch.float32, what)


def _i(t, what):
    return _need(t, torch.int64, what)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _ck(st, what):
    _capi.check(lib(), st, what)


def new_error_word(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def raise_on_flags(err):
    """One 4-byte D2H read (synchronises). Mirrors the IndexError the reference's
    index_select / nn.Embedding raise on a bad id."""
    flags = int(err.item())
    if flags:
        err.zero_()
        names = [n for bit, n in ((1, 'entity id outside node_map / not of this mode'),
                                  (2, 'edge endpoint outside [0, num_nodes)'),
                                  (4, 'edge type outside [0, num_relations)'),
                                  (8, 'scatter index outside [0, dim_size)')) if flags & bit]
        if flags & 16:
            sites = [n for bit, n in ((0x100, 'a pre-pass vector'), (0x200, 'the transposed weight copies'),
                                      (0x400, 'a completion counter'), (0x800, 'a vector op\'s inputs'),
                                      (0x1000, 'the fused tail\'s arrivals'), (0x2000, 'the touch plan\'s sort'),
                                      (0x4000, 'a peer of the p2p gradient exchange')) if flags & bit]
            raise RuntimeError('mpqe_amd: an in-launch hand-off between workgroups timed out (library fault): '
                               + (', '.join(sites) or 'unknown site'))
        if flags & 32:
            raise RuntimeError('mpqe_amd: a step could not build its own touch plan (the sort\'s workgroups were not all '
                               'resident at once -- a shared GPU?) and its entity-table gradients were NOT accumulated; '
                               'FusedTrainStep.run(..., checked=True) rebuilds the plan and recovers them')
        raise IndexError('mpqe_amd: ' + '; '.join(names))


# --------------------------------------------------------------------------------------------- templates
class Template(object):
    """Host-side description of a batch of B replicas of one query template."""
    __slots__ = ('query_type', 'qid', 'B', 'N', 'E', 'A', 'V', 'diameter', 'edge_type', '_et')

    def __init__(self, query_type, batch_size, edge_type):
        if query_type not in QUERY_TYPE_IDS:
            raise ValueError('unknown query type %r' % (query_type,))
        info = _capi.TemplateInfo()
        _ck(lib().mpqe_template_info(QUERY_TYPE_IDS[query_type], info), 'mpqe_template_info')
        self.query_type = query_type
        self.qid = QUERY_TYPE_IDS[query_type]
        self.B = int(batch_size)
        self.N, self.E = info.num_nodes, info.num_edges
        self.A, self.V = info.num_anchors, info.num_vars
        self.diameter = info.diameter
        self.edge_type = tuple(int(e) for e in edge_type)
        if len(self.edge_type) != self.E:
            raise ValueError('%s has %d edges, got %d edge types' % (query_type, self.E, len(self.edge_type)))
        self._et = np.ascontiguousarray(np.array(self.edge_type, dtype=np.int64))

    @property
    def et_ptr(self):
        return self._et.ctypes.data


def template_info(query_type):
    info = _capi.TemplateInfo()
    _ck(lib().mpqe_template_info(QUERY_TYPE_IDS[query_type], info), 'mpqe_template_info')
    return info


def collate_template(tmpl, device):
    """(a1) device-side expansion of the template: edge_index [2,B*E], edge_type [B*E], batch [B*N]."""
    ei = torch.empty((2, tmpl.B * tmpl.E), dtype=torch.int64, device=device)
    et = torch.empty((tmpl.B * tmpl.E,), dtype=torch.int64, device=device)
    bt = torch.empty((tmpl.B * tmpl.N,), dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        _ck(lib().mpqe_collate_template(tmpl.qid, tmpl.B, tmpl.et_ptr, _p(ei), _p(et), _p(bt), _stream()),
            'mpqe_collate_template')
    return ei, et, bt


# --------------------------------------------------------------------------------------------- general-graph plan
class GraphPlan(object):
    """Sorted views of an arbitrary edge list (by relation / destination / source), built once
    per graph on the device. Raises IndexError for out-of-range endpoints or edge types."""

    def __init__(self, edge_index, edge_type, num_nodes, num_relations):
        edge_index = _i(edge_index, 'edge_index')
        edge_type = _i(edge_type, 'edge_type')
        if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_type.shape[0] != edge_index.shape[1]:
            raise ValueError('edge_index must be [2, E] and edge_type [E]')
        self.Nn, self.E, self.R = int(num_nodes), int(edge_index.shape[1]), int(num_relations)
        L = lib()
        dev = edge_index.device
        with torch.cuda.device(dev):
            pb = L.mpqe_rgcn_plan_bytes(self.Nn, self.E, self.R)
            pw = L.mpqe_rgcn_plan_workspace_bytes(self.Nn, self.E, self.R)
            self.buf = _ws(pb, dev)
            ws = _ws(pw, dev)
            err = new_error_word(dev)
            _ck(L.mpqe_rgcn_plan_build(_p(edge_index), _p(edge_type), self.Nn, self.E, self.R, _p(self.buf), pb,
                                       _p(ws), pw, _p(err), _stream()), 'mpqe_rgcn_plan_build')
            raise_on_flags(err)


# --------------------------------------------------------------------------------------------- R-GCN layer
class _RGCNLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, basis, root, bias, graph, relu):
        x, basis, root = _f(x, 'x'), _f(basis, 'basis'), _f(root, 'root')
        bias = None if bias is None else _f(bias, 'bias')
        R, Din, Dout = basis.shape
        if x.dim() != 2 or x.shape[1] != Din or tuple(root.shape) != (Din, Dout):
            raise ValueError('shape mismatch: x %s basis %s root %s' % (tuple(x.shape), tuple(basis.shape),
                                                                        tuple(root.shape)))
        L = lib()
        out = torch.empty((x.shape[0], Dout), dtype=torch.float32, device=x.device)
        mask_bits = None
        with torch.cuda.device(x.device):
            if isinstance(graph, Template):
                if x.shape[0] != graph.B * graph.N:
                    raise ValueError('x has %d rows, template batch needs %d' % (x.shape[0], graph.B * graph.N))
                _ck(L.mpqe_rgcn_template_fwd(graph.qid, graph.B, graph.et_ptr, _p(x), _p(basis), R, _p(root),
                                             _p(bias), Din, Dout, int(relu), _p(out), _stream()),
                    'mpqe_rgcn_template_fwd')
            else:
                if graph.Nn != x.shape[0] or graph.R != R:
                    raise ValueError('plan was built for %d nodes / %d relations' % (graph.Nn, graph.R))
                wb = L.mpqe_rgcn_general_workspace_bytes(graph.Nn, graph.E, R, Din, Dout, 0)
                ws = _ws(wb, x.device)
                # (the ReLU mask as bit words, 1 / 32 of `out`: the backward masks with them instead of gathering `out` rows)
                mb = L.mpqe_rgcn_general_mask_bytes(graph.Nn, Dout) if relu else 0
                if mb:
                    mask_bits = torch.empty(mb // 8, dtype=torch.int64, device=x.device)
                _ck(L.mpqe_rgcn_general_fwd(_p(graph.buf), graph.Nn, graph.E, R, _p(x), _p(basis), _p(root),
                                            _p(bias), Din, Dout, int(relu), _p(out), _p(mask_bits), _p(ws), wb, _stream()),
                    'mpqe_rgcn_general_fwd')
        ctx.graph, ctx.relu, ctx.has_bias = graph, bool(relu), bias is not None
        ctx.save_for_backward(x, basis, root, out if relu else None, mask_bits)
        return out

    @staticmethod
    def backward(ctx, g):
        x, basis, root, out, mask_bits = ctx.saved_tensors
        g = _f(g, 'grad_out')
        graph, relu = ctx.graph, ctx.relu
        R, Din, Dout = basis.shape
        L = lib()
        need_x, need_b, need_r, need_bias = ctx.needs_input_grad[:4]
        gx = torch.empty_like(x) if need_x else None
        general = not isinstance(graph, Template)
        # (general graphs: the call writes every gradient buffer whole -- no zero fill, no read-modify-write)
        gb = (torch.empty_like(basis) if general else torch.zeros_like(basis)) if need_b else None
        gr = (torch.empty_like(root) if general else torch.zeros_like(root)) if need_r else None
        gbias = ((torch.empty if general else torch.zeros)(Dout, dtype=torch.float32, device=x.device)
                 if (need_bias and ctx.has_bias) else None)
        with torch.cuda.device(x.device):
            if isinstance(graph, Template):
                wb = L.mpqe_rgcn_template_bwd_workspace_bytes(graph.qid, graph.B, Din, Dout)
                ws = _ws(wb, x.device)
                _ck(L.mpqe_rgcn_template_bwd(graph.qid, graph.B, graph.et_ptr, _p(x), _p(out), _p(g), _p(basis), R,
                                             _p(root), Din, Dout, int(relu), _p(gx), _p(gb), _p(gr), _p(gbias),
                                             _p(ws), wb, _stream()), 'mpqe_rgcn_template_bwd')
            else:
                wb = L.mpqe_rgcn_general_workspace_bytes(graph.Nn, graph.E, R, Din, Dout, 1)
                ws = _ws(wb, x.device)
                _ck(L.mpqe_rgcn_general_bwd(_p(graph.buf), graph.Nn, graph.E, R, _p(x), _p(out), _p(mask_bits), _p(g), _p(basis),
                                            _p(root), Din, Dout, int(relu), 1, _p(gx), _p(gb), _p(gr), _p(gbias),
                                            _p(ws), wb, _stream()), 'mpqe_rgcn_general_bwd')
        return gx, gb, gr, gbias, None, None


def rgcn_layer(x, basis, root, bias, graph, relu=False):
    """out = [relu](sum_e x[src_e].basis[type_e] + x.root + bias); graph is a Template or a GraphPlan."""
    return _RGCNLayer.apply(x, basis, root, bias, graph, relu)


# --------------------------------------------------------------------------------------------- dense layer
class _Linear(torch.autograd.Function):
    """y = [relu](sum_k x_k . W[:, off_k : off_k + w_k]^T + bias): nn.Linear's arithmetic on the library's own MFMA tiles
    (mpqe_linear_fwd / mpqe_linear_bwd). One block = a plain dense layer; several = one wide matrix applied block by
    block to inputs that are never concatenated (Encoder.forward's compress product, reference encoders.py:120-124);
    the column blocks of W -- and of its gradient -- are read and written in place through the row stride."""

    @staticmethod
    def forward(ctx, W, bias, relu, blocks, *xs):
        W = _f(W, 'weight')
        xs = [_f(x, 'x') for x in xs]
        bias = None if bias is None else _f(bias, 'bias')
        if W.dim() != 2 or len(xs) != len(blocks) or not xs:
            raise ValueError('one column block of the weight per input')
        dout, total = W.shape
        rows = xs[0].shape[0]
        for x, (off, width) in zip(xs, blocks):
            if x.dim() != 2 or x.shape[0] != rows or x.shape[1] != width or off < 0 or off + width > total:
                raise ValueError('shape mismatch: x %s, weight block [%d, %d:%d]' % (tuple(x.shape), dout, off, off + width))
        y = torch.empty((rows, dout), dtype=torch.float32, device=W.device)
        L = lib()
        with torch.cuda.device(y.device):
            for k, (x, (off, width)) in enumerate(zip(xs, blocks)):
                last = k == len(xs) - 1
                _ck(L.mpqe_linear_fwd(_p(x), rows, W.data_ptr() + 4 * off, total, _p(bias) if last else None, width, dout,
                                      int(bool(relu) and last), int(k > 0), _p(y), _stream()), 'mpqe_linear_fwd')
        ctx.relu, ctx.blocks, ctx.has_bias = bool(relu), list(blocks), bias is not None
        ctx.save_for_backward(W, y if relu else None, *xs)
        return y

    @staticmethod
    def backward(ctx, g):
        W, y = ctx.saved_tensors[:2]
        xs = ctx.saved_tensors[2:]
        g = _f(g, 'grad_out')
        rows, dout = g.shape
        total = W.shape[1]
        L = lib()
        need_w, need_bias = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and ctx.has_bias
        pos, covered = 0, True
        for o, w in sorted(ctx.blocks):
            covered, pos = covered and o == pos, o + w
        covered = covered and pos == total
        gW = ((torch.empty_like(W) if covered else torch.zeros_like(W)) if need_w else None)    # (blocks that tile W: written whole)
        gbias = torch.empty(dout, dtype=torch.float32, device=g.device) if need_bias else None
        gxs = []
        with torch.cuda.device(g.device):
            for k, (x, (off, width)) in enumerate(zip(xs, ctx.blocks)):
                gx = torch.empty_like(x) if ctx.needs_input_grad[4 + k] else None
                wb = L.mpqe_linear_bwd_workspace_bytes(rows, width, dout)
                ws = _ws(wb, g.device)
                _ck(L.mpqe_linear_bwd(_p(x), rows, W.data_ptr() + 4 * off, total, _p(y), _p(g), width, dout, int(ctx.relu), 1,
                                      _p(gx), None if gW is None else gW.data_ptr() + 4 * off, total,
                                      _p(gbias) if k == 0 else None, _p(ws), wb, _stream()), 'mpqe_linear_bwd')
                gxs.append(gx)
        return (gW, gbias, None, None) + tuple(gxs)


def linear(x, weight, bias=None, relu=False):
    """[relu](x . weight^T + bias), weight [out, in] as nn.Linear stores it."""
    return _Linear.apply(weight, bias, relu, [(0, weight.shape[1])], x)


def blocks_linear(xs, weight, blocks, bias=None, relu=False):
    """[relu](sum_k xs[k] . weight[:, off_k : off_k + w_k]^T + bias), blocks = [(off_k, w_k)]: one wide matrix applied block
    by block, the inputs never concatenated."""
    return _Linear.apply(weight, bias, relu, list(blocks), *xs)


# --------------------------------------------------------------------------------------------- embeddings
class _EmbedL2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, node_map, ids, err):
        table, ids = _f(table, 'embedding table'), _i(ids, 'ids')
        node_map = None if node_map is None else _i(node_map, 'node_map')
        n, D = ids.shape[0], table.shape[1]
        out = torch.empty((n, D), dtype=torch.float32, device=table.device)
        with torch.cuda.device(table.device):
            _ck(lib().mpqe_embed_l2norm_fwd(_p(table), table.shape[0], D, _p(node_map),
                                            0 if node_map is None else node_map.shape[0], _p(ids), n, _p(out), D,
                                            None, _p(err), _stream()), 'mpqe_embed_l2norm_fwd')
        ctx.save_for_backward(table, ids)
        ctx.node_map, ctx.err = node_map, err
        return out

    @staticmethod
    def backward(ctx, g):
        table, ids = ctx.saved_tensors
        g = _f(g, 'grad')
        node_map = ctx.node_map
        gt = torch.zeros_like(table)
        with torch.cuda.device(table.device):
            _ck(lib().mpqe_embed_l2norm_bwd(_p(g), g.shape[1], _p(table), table.shape[0], table.shape[1],
                                            _p(node_map), 0 if node_map is None else node_map.shape[0], _p(ids),
                                            ids.shape[0], _p(gt), _p(ctx.err), _stream()), 'mpqe_embed_l2norm_bwd')
        return gt, None, None, None


def embed_l2norm(table, node_map, ids, err=None):
    """(a2) rows = node_map[ids]; table[rows] / ||.||_2   -> [n, D]"""
    return _EmbedL2Norm.apply(table, node_map, ids, err)


class _AssembleX(torch.autograd.Function):
    """(a3) x[b, i] = normalise(table_i[node_map[anchor_ids[i, b]]]); x[b, A+k] = mode_emb[var_ids[k]].
    anchor_ids_t is [A, B] (one contiguous id row per anchor slot)."""

    @staticmethod
    def forward(ctx, mode_emb, node_map, anchor_ids_t, var_ids, slot_table, err, *tables):
        mode_emb = _f(mode_emb, 'mode_embeddings.weight')
        anchor_ids_t, var_ids = _i(anchor_ids_t, 'anchor_ids'), _i(var_ids, 'var_ids')
        node_map = None if node_map is None else _i(node_map, 'node_map')
        tables = [_f(t, 'embedding table') for t in tables]
        A, B = anchor_ids_t.shape
        V = var_ids.shape[0]
        N, D = A + V, mode_emb.shape[1]
        L = lib()
        x = torch.empty((B * N, D), dtype=torch.float32, device=mode_emb.device)
        with torch.cuda.device(x.device):
            for i in range(A):
                t = tables[slot_table[i]]
                if t.shape[1] != D:
                    raise ValueError('embedding dim mismatch')
                _ck(L.mpqe_embed_l2norm_fwd(_p(t), t.shape[0], D, _p(node_map),
                                            0 if node_map is None else node_map.shape[0],
                                            anchor_ids_t.data_ptr() + 8 * i * B, B, x.data_ptr() + 4 * i * D,
                                            N * D, None, _p(err), _stream()), 'mpqe_embed_l2norm_fwd')
            _ck(L.mpqe_var_rows_fwd(_p(mode_emb), mode_emb.shape[0], D, _p(var_ids), V, B, N, A, _p(x), _p(err),
                                    _stream()), 'mpqe_var_rows_fwd')
        ctx.save_for_backward(mode_emb, anchor_ids_t, var_ids, *tables)
        ctx.node_map, ctx.slot_table, ctx.err = node_map, slot_table, err
        return x

    @staticmethod
    def backward(ctx, g):
        mode_emb, anchor_ids_t, var_ids = ctx.saved_tensors[:3]
        tables = ctx.saved_tensors[3:]
        g = _f(g, 'grad_x')
        node_map, err = ctx.node_map, ctx.err
        A, B = anchor_ids_t.shape
        V = var_ids.shape[0]
        N, D = A + V, mode_emb.shape[1]
        L = lib()
        gmode = torch.zeros_like(mode_emb) if ctx.needs_input_grad[0] else None
        gtabs = [torch.zeros_like(t) if ctx.needs_input_grad[6 + j] else None for j, t in enumerate(tables)]
        with torch.cuda.device(g.device):
            for i in range(A):
                j = ctx.slot_table[i]
                if gtabs[j] is None:
                    continue
                t = tables[j]
                _ck(L.mpqe_embed_l2norm_bwd(g.data_ptr() + 4 * i * D, N * D, _p(t), t.shape[0], D, _p(node_map),
                                            0 if node_map is None else node_map.shape[0],
                                            anchor_ids_t.data_ptr() + 8 * i * B, B, _p(gtabs[j]), _p(err),
                                            _stream()), 'mpqe_embed_l2norm_bwd')
            if gmode is not None:
                _ck(L.mpqe_var_rows_bwd(_p(g), mode_emb.shape[0], D, _p(var_ids), V, B, N, A, _p(gmode), _p(err),
                                        _stream()), 'mpqe_var_rows_bwd')
        return (gmode, None, None, None, None, None) + tuple(gtabs)


def assemble_x(mode_emb, node_map, anchor_ids_t, var_ids, slot_table, tables, err=None):
    return _AssembleX.apply(mode_emb, node_map, anchor_ids_t, var_ids, tuple(slot_table), err, *tables)


# --------------------------------------------------------------------------------------------- readouts
class _Readout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, kind, B, N, A):
        h = _f(h, 'embs')
        D = h.shape[1]
        if h.shape[0] != B * N:
            raise ValueError('embs has %d rows, expected batch_size*num_nodes = %d' % (h.shape[0], B * N))
        out = torch.empty((B, D), dtype=torch.float32, device=h.device)
        arg = torch.empty((B, D), dtype=torch.int32, device=h.device) if kind == 'max' else None
        with torch.cuda.device(h.device):
            _ck(lib().mpqe_readout_fwd(READOUT_IDS[kind], _p(h), B, N, A, D, _p(out), _p(arg), _stream()),
                'mpqe_readout_fwd')
        ctx.meta = (kind, B, N, A, D)
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        kind, B, N, A, D = ctx.meta
        (arg,) = ctx.saved_tensors
        g = _f(g, 'grad')
        gh = torch.empty((B * N, D), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _ck(lib().mpqe_readout_bwd(READOUT_IDS[kind], _p(g), _p(arg), B, N, A, D, _p(gh), _stream()),
                'mpqe_readout_bwd')
        return gh, None, None, None, None


def readout(kind, h, batch_size, num_nodes, num_anchors):
    """(a5) sum / max / mp(TM) over regular batches (row = b*N + n)."""
    return _Readout.apply(h, kind, int(batch_size), int(num_nodes), int(num_anchors))


class _Scatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, op, dim_size, err):
        src, index = _f(src, 'src'), _i(index, 'index')
        shape = src.shape
        src2 = src.reshape(shape[0], -1)
        n, D = src2.shape
        if index.dim() != 1 or index.shape[0] != n:
            raise ValueError('index must be 1-D with src.size(0) entries')
        L = lib()
        out = torch.empty((dim_size, D), dtype=torch.float32, device=src.device)
        arg = torch.empty((dim_size, D), dtype=torch.int64, device=src.device) if op == 'max' else None
        with torch.cuda.device(src.device):
            wb = L.mpqe_scatter_workspace_bytes(n, dim_size)
            ws = _ws(wb, src.device)
            _ck(L.mpqe_scatter_fwd(SCATTER_IDS[op], _p(src2), _p(index), n, D, dim_size, _p(out), _p(arg), _p(ws),
                                   wb, _p(err), _stream()), 'mpqe_scatter_fwd')
        ctx.meta = (op, n, D, dim_size, shape)
        ctx.save_for_backward(index, arg)
        out = out.reshape((dim_size,) + tuple(shape[1:]))
        if op == 'max':
            arg = arg.reshape(out.shape)
            ctx.mark_non_differentiable(arg)
            return out, arg
        return out

    @staticmethod
    def backward(ctx, g, *unused):
        op, n, D, dim_size, shape = ctx.meta
        index, arg = ctx.saved_tensors
        g = _f(g, 'grad').reshape(dim_size, D)
        L = lib()
        gs = torch.empty((n, D), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            wb = L.mpqe_scatter_workspace_bytes(n, dim_size)
            ws = _ws(wb, g.device)
            _ck(L.mpqe_scatter_bwd(SCATTER_IDS[op], _p(g), _p(index), _p(arg), n, D, dim_size, _p(gs), _p(ws), wb,
                                   _stream()), 'mpqe_scatter_bwd')
        return gs.reshape(shape), None, None, None, None


def _scatter(op, src, index, dim, dim_size):
    if dim != 0:
        raise NotImplementedError('mpqe_amd scatter ops reduce along dim 0 (all reference call sites do)')
    if dim_size is None:
        # the reference lets torch_scatter size the output from index.max() (one sync there too)
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    err = new_error_word(src.device)
    res = _Scatter.apply(src, index, op, int(dim_size), err)
    raise_on_flags(err)
    return res


def scatter_add(src, index, dim=0, out=None, dim_size=None, fill_value=0):
    """torch_scatter.scatter_add stand-in (reference call sites model.py:351, 381)."""
    return _scatter('add', src, index, dim, dim_size)


def scatter_mean(src, index, dim=0, out=None, dim_size=None, fill_value=0):
    return _scatter('mean', src, index, dim, dim_size)


def scatter_max(src, index, dim=0, out=None, dim_size=None, fill_value=None):
    """Returns (values, argmax) like torch_scatter (reference model.py:384)."""
    return _scatter('max', src, index, dim, dim_size)


# --------------------------------------------------------------------------------------------- (a9) LayerNorm + ReLU
class _LayerNormReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu):
        x, gamma, beta = _f(x, 'x'), _f(gamma, 'gamma'), _f(beta, 'beta')
        rows, D = x.shape
        if gamma.shape != (D,) or beta.shape != (D,):
            raise ValueError('layer norm: gamma / beta must be [%d]' % D)
        y = torch.empty_like(x)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _ck(lib().mpqe_layernorm_relu_fwd(_p(x), rows, D, _p(gamma), _p(beta), eps, int(relu), _p(y), _p(stats), _stream()),
                'mpqe_layernorm_relu_fwd')
        ctx.eps, ctx.relu = eps, bool(relu)
        ctx.save_for_backward(x, y, gamma, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, stats = ctx.saved_tensors
        gy = _f(gy, 'grad_out')
        rows, D = x.shape
        gx = torch.empty_like(x)
        gg = torch.zeros(D, dtype=torch.float32, device=x.device)
        gb = torch.zeros(D, dtype=torch.float32, device=x.device)
        L = lib()
        with torch.cuda.device(x.device):
            wb = L.mpqe_layernorm_relu_bwd_workspace_bytes(rows, D)
            ws = _ws(wb, x.device)
            _ck(L.mpqe_layernorm_relu_bwd(_p(gy), _p(x), _p(y), rows, D, _p(gamma), _p(stats), ctx.eps, int(ctx.relu), _p(gx),
                                          _p(gg), _p(gb), _p(ws), wb, _stream()), 'mpqe_layernorm_relu_bwd')
        return gx, gg, gb, None, None


def layernorm_relu(x, gamma, beta, eps=1e-6, relu=True):
    """(a9) act(gamma * (x - mean) / (std + eps) + beta) per row, std UNBIASED (reference encoders.py:143-146), one kernel
    with the Encoder's ReLU (encoders.py:127-128)."""
    return _LayerNormReLU.apply(x, gamma, beta, float(eps), bool(relu))


# --------------------------------------------------------------------------------------------- score / loss
class _Cosine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, t, q_row, eps):
        q, t = _f(q, 'query embeddings'), _f(t, 'target embeddings')
        q_row = None if q_row is None else _i(q_row, 'q_row')
        n, D = t.shape
        if q.shape[1] != D or (q_row is None and q.shape[0] != n):
            raise ValueError('cosine: shapes %s vs %s' % (tuple(q.shape), tuple(t.shape)))
        s = torch.empty((n,), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            _ck(lib().mpqe_cosine_fwd(_p(q), _p(q_row), _p(t), n, D, eps, _p(s), _stream()), 'mpqe_cosine_fwd')
        ctx.eps = eps
        ctx.save_for_backward(q, t, q_row)
        return s

    @staticmethod
    def backward(ctx, gs):
        q, t, q_row = ctx.saved_tensors
        gs = _f(gs, 'grad_scores')
        n, D = t.shape
        gq = (torch.zeros_like(q) if q_row is not None else torch.empty_like(q)) if ctx.needs_input_grad[0] else None
        gt = torch.empty_like(t) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(q.device):
            _ck(lib().mpqe_cosine_bwd(_p(gs), _p(q), _p(q_row), _p(t), n, D, ctx.eps, _p(gq), _p(gt), _stream()),
                'mpqe_cosine_bwd')
        return gq, gt, None, None


def cosine(q, t, q_row=None, eps=1e-8):
    """(a6) F.cosine_similarity(q[q_row], t, dim=1)"""
    return _Cosine.apply(q, t, q_row, float(eps))


class _Hinge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, neg, margin):
        pos, neg = _f(pos, 'pos scores'), _f(neg, 'neg scores')
        if pos.shape != neg.shape or pos.dim() != 1 or pos.shape[0] == 0:
            raise ValueError('hinge: pos/neg must be equal non-empty vectors')
        # (0-dim from the start: a reshape here would make the result a VIEW created inside a Function, and the reference's
        # `loss += w * margin_loss(...)`, train_helpers.py:103-110, writes the first loss in place -- autograd refuses that)
        loss = torch.empty((), dtype=torch.float32, device=pos.device)
        with torch.cuda.device(pos.device):
            _ck(lib().mpqe_hinge_fwd(_p(pos), _p(neg), pos.shape[0], margin, _p(loss), _stream()), 'mpqe_hinge_fwd')
        ctx.margin = margin
        ctx.save_for_backward(pos, neg)
        return loss

    @staticmethod
    def backward(ctx, gl):
        pos, neg = ctx.saved_tensors
        gl = _f(gl.reshape(1), 'grad_loss')
        gp, gn = torch.empty_like(pos), torch.empty_like(neg)
        with torch.cuda.device(pos.device):
            _ck(lib().mpqe_hinge_bwd(_p(pos), _p(neg), pos.shape[0], ctx.margin, _p(gl), _p(gp), _p(gn), _stream()),
                'mpqe_hinge_bwd')
        return gp, gn, None


class _L2Norms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *params):
        import ctypes
        ps = [_f(p, 'parameter') for p in params]
        out = torch.zeros((), dtype=torch.float32, device=ps[0].device)
        with torch.cuda.device(ps[0].device):
            for i in range(0, len(ps), 4):
                chunk = ps[i:i + 4]
                arr = (ctypes.c_void_p * len(chunk))(*[_p(p) for p in chunk])
                n = (ctypes.c_int64 * len(chunk))(*[p.numel() for p in chunk])
                _ck(lib().mpqe_l2_norms(arr, n, len(chunk), None, _p(out), None, _stream()), 'mpqe_l2_norms')
        ctx.save_for_backward(*ps)
        return out

    @staticmethod
    def backward(ctx, gl):
        import ctypes
        ps = ctx.saved_tensors
        gl = _f(gl.reshape(1), 'grad_out')
        grads = [torch.zeros_like(p) for p in ps]
        with torch.cuda.device(ps[0].device):
            for i in range(0, len(ps), 4):
                chunk, gch = ps[i:i + 4], grads[i:i + 4]
                arr = (ctypes.c_void_p * len(chunk))(*[_p(p) for p in chunk])
                garr = (ctypes.c_void_p * len(chunk))(*[_p(g) for g in gch])
                n = (ctypes.c_int64 * len(chunk))(*[p.numel() for p in chunk])
                _ck(lib().mpqe_l2_norms(arr, n, len(chunk), _p(gl), None, garr, _stream()), 'mpqe_l2_norms')
        return tuple(grads)


def l2_norms(params):
    """(a7) sum_i ||p_i||_2 over parameter tensors (unsquared: reference model.py:486-490), fixed order of every sum."""
    return _L2Norms.apply(*[p for p in params])


def hinge(pos, neg, margin=1.0):
    """(a7) mean(clamp(margin - (pos - neg), min=0))"""
    return _Hinge.apply(pos, neg, float(margin))

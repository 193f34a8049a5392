"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32 + numpy int64) of the
MPQE R-GCN query-graph encoder hot path, op for op in the reference's order.
It is the checker for the HIP path and the timed `cpu_baseline` ("port") in
bench.py. Nothing under mpqe_amd/ imports it.

PINNING: the reference has no numerical tests for this path,
so this file is pinned against the reference ITSELF: oracle/gen_golden.py
imports /root/reference/mpqe (with oracle/standins.py for the absent
torch_geometric / torch_scatter), runs it, and commits the input/output vectors
under tests/golden/; tests/test_oracle_golden.py checks every function here
against them (ints exact, floats rtol 1e-5 / atol 1e-6, grads rtol 1e-4).

Functions work on plain tensors and a `params` dict keyed like the reference's
state_dict ('enc.feat-<mode>.weight', 'mode_embeddings.weight',
'layers.<i>.basis|root|bias', 'readout.layers.<0|2>.weight|bias').
"""
import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- templates
# reference: data_utils.py:325-362. Per query type: edges (src row, dst row)
# inside one graph, which flattened relation labels each edge (before reversal),
# which entries of Formula.get_nodes() are variable nodes, and the diameter.
TEMPLATES = {
    '1-chain':       dict(src=[0], dst=[1], rel=[0], var=[0], diam=1),
    '2-chain':       dict(src=[0, 2], dst=[2, 1], rel=[1, 0], var=[0, 2], diam=2),
    '3-chain':       dict(src=[0, 3, 2], dst=[3, 2, 1], rel=[2, 1, 0], var=[0, 2, 4], diam=3),
    '2-inter':       dict(src=[0, 1], dst=[2, 2], rel=[0, 1], var=[0], diam=1),
    '3-inter':       dict(src=[0, 1, 2], dst=[3, 3, 3], rel=[0, 1, 2], var=[0], diam=1),
    '3-inter_chain': dict(src=[0, 1, 3], dst=[2, 3, 2], rel=[0, 2, 1], var=[0, 3], diam=2),
    '3-chain_inter': dict(src=[0, 1, 3], dst=[3, 3, 2], rel=[1, 2, 0], var=[0, 2], diam=2),
}


def build_ids(relations, mode_weights):
    """mode -> id in mode_weights order; typed relation -> id in nested
    iteration order of `relations` (reference: model.py:326-338)."""
    mode_ids = {m: i for i, m in enumerate(mode_weights)}
    rel_ids = {}
    for m in relations:
        for (to, name) in relations[m]:
            rel_ids[(m, name, to)] = len(rel_ids)
    return mode_ids, rel_ids


def collate(formula, queries, rel_ids, mode_ids):
    """reference: RGCNQueryDataset.get_query_graph (data_utils.py:377-409) plus
    PyG Batch.from_data_list. All outputs int64 numpy, bit-exact contract."""
    t = TEMPLATES[formula.query_type]
    B = len(queries)
    A = len(formula.anchor_modes)
    V = len(t['var'])
    N, E = A + V, len(t['src'])
    anchor_ids = np.empty((B, A), dtype=np.int64)
    for b, q in enumerate(queries):
        for i in range(A):
            anchor_ids[b, i] = q.anchor_nodes[i]
    nodes = formula.get_nodes()
    var_ids = np.array([mode_ids[nodes[i]] for i in t['var']], dtype=np.int64)
    rels = formula.get_rels()
    etype = np.array([rel_ids[(rels[i][2], rels[i][1], rels[i][0])] for i in t['rel']],
                     dtype=np.int64)
    offs = (np.arange(B, dtype=np.int64) * N)[:, None]
    src = (np.array(t['src'], dtype=np.int64)[None, :] + offs).reshape(-1)
    dst = (np.array(t['dst'], dtype=np.int64)[None, :] + offs).reshape(-1)
    edge_index = np.stack([src, dst], axis=0)
    edge_type = np.tile(etype, B)
    batch = np.repeat(np.arange(B, dtype=np.int64), N)
    return dict(anchor_ids=anchor_ids, var_ids=var_ids, edge_index=edge_index,
                edge_type=edge_type, batch=batch, B=B, A=A, V=V, N=N, E=E)


# --------------------------------------------------------------------------- embedding
def direct_encode(table, node_map, ids):
    """reference: DirectEncoder.forward (encoders.py:40-43) with the `features`
    closure of data_utils.py:35. Returns [B, D] (the reference returns the
    [D, B] transpose and transposes back at every call site)."""
    ids = torch.as_tensor(ids, dtype=torch.long)
    rows = node_map[ids]
    emb = table[rows].t()                       # [D, B]
    norm = emb.norm(p=2, dim=0, keepdim=True)   # no eps
    return emb.div(norm.expand_as(emb)).t()


# --------------------------------------------------------------------------- R-GCN layer
def rgcn_layer_refseq(x, edge_index, edge_type, basis, root, bias):
    """reference sequence: model.py:277-305 (per-edge weight copy + bmm +
    scatter-add + root + bias). edge_norm is always None (model.py:436, 441)."""
    x_j = x.index_select(0, edge_index[0])
    w = basis.index_select(0, edge_type)                      # [E, D, D]
    msg = torch.bmm(x_j.unsqueeze(1), w).squeeze(-2)
    agg = torch.zeros(x.shape[0], basis.shape[2], dtype=x.dtype).index_add(0, edge_index[1], msg)
    out = agg + torch.matmul(x, root)
    if bias is not None:
        out = out + bias
    return out


def rgcn_layer_grouped(x, edge_index, edge_type, basis, root, bias):
    """Same maths without the [E, D, D] copy: one GEMM per used relation."""
    out = torch.matmul(x, root)
    if bias is not None:
        out = out + bias
    agg = torch.zeros_like(out)
    for r in torch.unique(edge_type).tolist():
        sel = (edge_type == r).nonzero(as_tuple=True)[0]
        msg = x.index_select(0, edge_index[0][sel]) @ basis[r]
        agg = agg.index_add(0, edge_index[1][sel], msg)
    return out + agg


# --------------------------------------------------------------------------- scatter fns / readouts
def scatter_add(src, index, dim_size):
    return torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype).index_add(0, index, src)


def scatter_mean(src, index, dim_size):
    cnt = torch.zeros(dim_size, dtype=src.dtype).index_add(
        0, index, torch.ones(index.shape[0], dtype=src.dtype)).clamp(min=1)
    return scatter_add(src, index, dim_size) / cnt[:, None]


def _scatter_max_arg(src, index, dim_size):
    idx = index[:, None].expand_as(src)
    val = torch.zeros((dim_size, src.shape[1]), dtype=src.dtype).scatter_reduce(
        0, idx, src, reduce='amax', include_self=False)
    rows = torch.arange(src.shape[0])[:, None].expand_as(src)
    cand = torch.where(src == val.index_select(0, index), rows,
                       torch.full_like(rows, src.shape[0]))
    first = torch.full((dim_size, src.shape[1]), src.shape[0], dtype=torch.long).scatter_reduce(
        0, idx, cand, reduce='amin', include_self=True)
    arg = torch.where(first < src.shape[0], first, torch.full_like(first, -1))
    return val, arg


class _ScatterMax(torch.autograd.Function):
    """torch_scatter's scatter_max as the reference uses it (model.py:384, 509, 547): the backward routes each output
    element's gradient to ITS argmax row only (torch_scatter: grad_src = grad_out gathered where arg == row). torch's own
    scatter_reduce('amax') backward would split the gradient evenly between tied rows instead -- different at exact ties
    (two nodes with identical states), identical everywhere else."""

    @staticmethod
    def forward(ctx, src, index, dim_size):
        val, arg = _scatter_max_arg(src.detach(), index, dim_size)
        ctx.save_for_backward(arg)
        ctx.n = src.shape[0]
        ctx.mark_non_differentiable(arg)
        return val, arg

    @staticmethod
    def backward(ctx, gval, _garg):
        arg, = ctx.saved_tensors
        g = torch.zeros((ctx.n + 1, gval.shape[1]), dtype=gval.dtype)
        g.scatter_add_(0, torch.where(arg >= 0, arg, torch.full_like(arg, ctx.n)), gval)
        return g[:ctx.n], None, None


def scatter_max(src, index, dim_size):
    """Returns (values, argmax); argmax = lowest source row attaining the max;
    rows with no source: value 0 / arg -1. The tie-break is this build's
    statement -- the reference discards argmax (model.py:384-385)."""
    return _ScatterMax.apply(src, index, dim_size)


_SCATTER = {'add': scatter_add, 'mean': scatter_mean,
            'max': lambda s, i, n: scatter_max(s, i, n)[0]}


def _mlp(x, params, prefix):
    h = F.linear(x, params[prefix + 'layers.0.weight'], params[prefix + 'layers.0.bias'])
    h = F.relu(h)
    return F.linear(h, params[prefix + 'layers.2.weight'], params[prefix + 'layers.2.bias'])


def readout(kind, scatter_op, params, embs, batch_idx, B, N, A):
    """reference: model.py:380-398 (sum / max / mp), 497-515 (mlp, concat),
    518-553 (targetmlp)."""
    if kind == 'sum':
        return scatter_add(embs, batch_idx, B)
    if kind == 'max':
        return scatter_max(embs, batch_idx, B)[0]
    if kind == 'mp':
        return embs.reshape(B, N, -1)[:, A, :]
    if kind in ('mlp', 'concat'):
        return _SCATTER[scatter_op](_mlp(embs, params, 'readout.'), batch_idx, B)
    if kind == 'targetmlp':
        keep = [n for n in range(N) if n != A]
        e3 = embs.reshape(B, N, -1)
        non_t = e3[:, keep, :]
        tgt = e3[:, A:A + 1, :].expand_as(non_t)
        xin = torch.cat((tgt, non_t), dim=-1).reshape(B * (N - 1), -1)
        bidx = batch_idx.reshape(B, N)[:, keep].reshape(-1)
        return _SCATTER[scatter_op](_mlp(xin, params, 'readout.'), bidx, B)
    raise ValueError('Unknown readout function %s' % kind)


# --------------------------------------------------------------------------- encoder
def num_passes(cfg, query_type):
    """reference: model.py:425-431."""
    if cfg['adaptive']:
        n = TEMPLATES[query_type]['diam']
        if n > cfg['num_layers']:
            raise ValueError('RGCN is adaptive with %d layers, but query requires %d.'
                             % (cfg['num_layers'], n))
        return n
    return cfg['num_layers']


def encode_queries(params, cfg, node_map, formula, col, layer_fn=rgcn_layer_refseq,
                   keep=None):
    """reference: RGCNEncoderDecoder.forward, model.py:414-449. `col` is the
    dict from collate(). Returns the query embeddings [B, D]; fills `keep`
    (a dict) with x0, every layer output and the readout when given."""
    B, A, N = col['B'], col['A'], col['N']
    D = params['mode_embeddings.weight'].shape[1]
    anchor_ids = torch.as_tensor(col['anchor_ids'])
    x = torch.empty(B, N, D)
    cols = []
    for i, mode in enumerate(formula.anchor_modes):
        cols.append(direct_encode(params['enc.feat-%s.weight' % mode], node_map,
                                  anchor_ids[:, i]))
    var = params['mode_embeddings.weight'][torch.as_tensor(col['var_ids'])]
    x = torch.cat([c[:, None, :] for c in cols] + [var[None].expand(B, -1, -1)], dim=1)
    x = x.reshape(B * N, D)
    ei = torch.as_tensor(col['edge_index'])
    et = torch.as_tensor(col['edge_type'])
    if keep is not None:
        keep['x0'] = x
    L = num_passes(cfg, formula.query_type)
    last = cfg['num_layers'] - 1
    h = x
    hs = []
    for i in range(L - 1):
        h = layer_fn(h, ei, et, params['layers.%d.basis' % i], params['layers.%d.root' % i],
                     params['layers.%d.bias' % i])
        h = F.relu(h)
        hs.append(h)
    h = layer_fn(h, ei, et, params['layers.%d.basis' % last], params['layers.%d.root' % last],
                 params['layers.%d.bias' % last])
    hs.append(h)
    if keep is not None:
        keep['layers'] = list(hs)
    if cfg['readout'] == 'concat':
        h = torch.cat(hs, dim=1)
    out = readout(cfg['readout'], cfg['scatter_op'], params, h,
                  torch.as_tensor(col['batch']), B, N, A)
    if keep is not None:
        keep['readout'] = out
    return out


def score(params, node_map, formula, q_emb, target_nodes, neg_nodes=None, neg_lengths=None):
    """reference: model.py:451-462."""
    table = params['enc.feat-%s.weight' % formula.target_mode]
    t = direct_encode(table, node_map, target_nodes)
    scores = F.cosine_similarity(q_emb, t, dim=1)
    if neg_nodes is not None:
        n = direct_encode(table, node_map, neg_nodes)
        rep = q_emb.repeat_interleave(torch.as_tensor(neg_lengths), dim=0)
        scores = torch.cat((scores, F.cosine_similarity(rep, n)), dim=0)
    return scores


def forward(params, cfg, node_map, formula, col, target_nodes, neg_nodes=None,
            neg_lengths=None, layer_fn=rgcn_layer_refseq, keep=None):
    q = encode_queries(params, cfg, node_map, formula, col, layer_fn, keep)
    return score(params, node_map, formula, q, target_nodes, neg_nodes, neg_lengths)


def margin_loss(params, cfg, node_map, formula, col, target_nodes, neg_nodes, margin=1.0,
                layer_fn=rgcn_layer_refseq, encode_twice=True):
    """reference: model.py:478-494 with the negatives given explicitly (the
    reference draws them with python `random.choice`, model.py:470-476).
    encode_twice=True keeps the reference's two full encoder passes."""
    if encode_twice:
        pos = forward(params, cfg, node_map, formula, col, target_nodes, layer_fn=layer_fn)
        neg = forward(params, cfg, node_map, formula, col, neg_nodes, layer_fn=layer_fn)
    else:
        q = encode_queries(params, cfg, node_map, formula, col, layer_fn)
        pos = score(params, node_map, formula, q, target_nodes)
        neg = score(params, node_map, formula, q, neg_nodes)
    loss = torch.clamp(margin - (pos - neg), min=0).mean()
    if cfg['readout'] in ('mlp', 'concat', 'targetmlp') and cfg.get('weight_decay', 0) > 0:
        reg = 0
        for k in ('readout.layers.0.weight', 'readout.layers.0.bias',
                  'readout.layers.2.weight', 'readout.layers.2.bias'):
            reg = reg + torch.norm(params[k])
        loss = loss + cfg['weight_decay'] * reg
    return loss


# --------------------------------------------------------------------------- negative sampling
_M64 = (1 << 64) - 1


def _mix64(x):
    """splitmix64 finaliser, python ints (mpqe_amd/csrc/optim.hip: mpqe_mix64)."""
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def sample_negatives(cand, offsets, qidx, nq, seed):
    """One uniform draw per batch position from its candidate list (reference model.py:466-476 draws with
    python's random.choice; the device path uses this counter-based stream instead -- same distribution).
    cand: int64 array; offsets: CSR over cand or None (every query draws from all of cand); qidx: list index
    per position or None. Returns (int64 [nq], bad) with -1 / bad=True for empty lists."""
    out = np.empty(nq, dtype=np.int64)
    bad = False
    for i in range(nq):
        lo, hi = 0, len(cand)
        if offsets is not None:
            q = int(qidx[i]) if qidx is not None else i
            if q < 0 or q >= len(offsets) - 1:
                out[i], bad = -1, True
                continue
            lo, hi = int(offsets[q]), int(offsets[q + 1])
        if hi <= lo or lo < 0 or hi > len(cand):
            out[i], bad = -1, True
            continue
        r = _mix64((seed & _M64) ^ _mix64(i))
        out[i] = cand[lo + r % (hi - lo)]
    return out, bad

"""Query-graph collation with the reference's interface (mpqe/data_utils.py:268-426).

get_query_graph returns what the reference returns -- (anchor_ids [B,A] int64, var_ids [V]
int64, graph) -- but the graph's B-fold replicated tensors (edge_index, edge_type, batch) are
expanded ON THE DEVICE from the <= 3-edge template by one kernel instead of B-way torch.cat
on the host (reference: PyG Batch.from_data_list, data_utils.py:402-405).
"""
import os
import pickle
from collections import OrderedDict, defaultdict

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import ops
from .graph import Graph, Query, reverse_relation


# ------------------------------------------------------------------------------------- dataset files
def load_graph(data_dir, embed_dim):
    """reference: data_utils.py:18-37. `graph_data.pkl` = (relations {mode: [(to_mode, name)]}, adjacency
    {(mode, name, to_mode): {entity: neighbours}}, node_maps {mode: [entity ids]}). Returns (graph,
    feature_modules, node_maps) like the reference: per mode an nn.Embedding(count + 1, D) initialised N(0, 1/D) in
    the relations' mode order (the same torch draws as the reference under the same seed), the int64 LUT global
    entity id -> row of its mode's table (-1: of no mode), and the Graph container whose `features` closure is the
    LUT lookup. Hand feature_modules + node_maps to DirectEncoder(graph.features, feature_modules, node_maps)."""
    with open(os.path.join(data_dir, 'graph_data.pkl'), 'rb') as f:
        rels, adj_lists, node_ids = pickle.load(f)
    num_nodes = sum(len(ids) for ids in node_ids.values())
    node_maps = torch.full((num_nodes + 1,), -1, dtype=torch.long)
    for mode, ids in node_ids.items():
        ids = torch.as_tensor(list(ids), dtype=torch.long)
        # (the reference asserts per element that the id is not mapped yet, line 27: an id listed under two modes, or twice
        # under one, fails there)
        if ids.numel() and (bool((node_maps[ids] != -1).any()) or ids.unique().numel() != ids.numel()):
            raise AssertionError('entity id listed twice (under two modes, or repeated inside one)')
        node_maps[ids] = torch.arange(ids.shape[0])
    feature_dims = {m: embed_dim for m in rels}
    feature_modules = {m: torch.nn.Embedding(len(node_ids[m]) + 1, embed_dim) for m in rels}
    for mode in rels:
        feature_modules[mode].weight.data.normal_(0, 1. / embed_dim)

    def features(nodes, mode):
        w = feature_modules[mode].weight
        return feature_modules[mode](node_maps.to(w.device)[torch.as_tensor(nodes, dtype=torch.long, device=w.device)])
    graph = Graph(features, feature_dims, rels, adj_lists)
    return graph, feature_modules, node_maps


def _raw(data_file):
    with open(data_file, 'rb') as f:
        return pickle.load(f)


def load_queries(data_file, keep_graph=False):
    """reference: data_utils.py:150-152."""
    return [Query.deserialize(info, keep_graph=keep_graph) for info in _raw(data_file)]


def load_queries_by_formula(data_file):
    """reference: data_utils.py:155-161: {query type: {formula: [Query]}} in file order."""
    queries = defaultdict(lambda: defaultdict(list))
    for raw_query in _raw(data_file):
        query = Query.deserialize(raw_query)
        queries[query.formula.query_type][query.formula].append(query)
    return queries


def load_queries_by_type(data_file, keep_graph=True):
    """reference: data_utils.py:164-170."""
    queries = defaultdict(list)
    for raw_query in _raw(data_file):
        query = Query.deserialize(raw_query, keep_graph=keep_graph)
        queries[query.formula.query_type].append(query)
    return queries


def load_test_queries_by_formula(data_file):
    """reference: data_utils.py:173-182: split by whether a query carries one negative or a full list."""
    queries = {'full_neg': defaultdict(lambda: defaultdict(list)), 'one_neg': defaultdict(lambda: defaultdict(list))}
    for raw_query in _raw(data_file):
        neg_type = 'full_neg' if len(raw_query[1]) > 1 else 'one_neg'
        query = Query.deserialize(raw_query)
        queries[neg_type][query.formula.query_type][query.formula].append(query)
    return queries


class QueryGraphBatch(object):
    """Stand-in for the PyG `Batch` the reference passes around: attributes edge_index
    [2, B*E], edge_type [B*E], batch [B*N], x, num_nodes and .to(device). Holds the template
    so the encoder can take the scatter-free fused path."""

    def __init__(self, template):
        self.template = template
        self.num_nodes = template.B * template.N
        self.ids = None            # BatchIds of a collated batch (RGCNQueryDataset.collate_fn)
        self.x = None
        self.device = None
        self._tensors = None

    def to(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('mpqe_amd: query graphs live on the GPU; .to(%s) is not supported' % device)
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if self._tensors is None or self.device != device:
            ei, et, bt = ops.collate_template(self.template, device)
            ei._mpqe_graph = self.template          # lets RGCNConv.forward recognise the batch
            self._tensors = (ei, et, bt)
            self.device = device
        return self

    def _get(self, i):
        if self._tensors is None:
            raise RuntimeError('mpqe_amd: call .to("cuda") first; graph tensors are built on the device')
        return self._tensors[i]

    edge_index = property(lambda self: self._get(0))
    edge_type = property(lambda self: self._get(1))
    batch = property(lambda self: self._get(2))


class FormulaIds(object):
    """Every id the queries of ONE formula carry, as arrays built once (the reference re-reads the Query objects with
    python list comprehensions per batch: data_utils.py:382-383, model.py:470-477): targets [Nq], anchors [Nq, A], and the
    candidate lists of the negative draws as CSR (flat ids + offsets). A batch is a window [start, end) of them."""

    def __init__(self, queries):
        n = len(queries)
        A = len(queries[0].anchor_nodes) if n else 0
        self.n, self.A = n, A
        self.targets = np.fromiter((q.target_node for q in queries), dtype=np.int64, count=n)
        self.anchors = np.empty((n, A), dtype=np.int64)
        for i in range(A):
            self.anchors[:, i] = [q.anchor_nodes[i] for q in queries]
        self.anchors_sm = np.ascontiguousarray(self.anchors.T)          # slot-major [A, Nq]: the fused step's id layout
        # (+ addresses: a window of these arrays is plain pointer arithmetic for the host routines, csrc/host/pyhost.c)
        self.targets_ptr = self.targets.ctypes.data
        self.anchors_sm_ptr, self.anchors_sm_stride = self.anchors_sm.ctypes.data, 8 * n
        self.neg = self._csr([q.neg_samples for q in queries])
        self.hard = self._csr([getattr(q, 'hard_neg_samples', None) for q in queries])

    @staticmethod
    def _csr(lists):
        """(flat ids, offsets [n + 1], lengths [n], their three addresses); None when any query has no list (the python path then raises what the
        reference raises)."""
        if any(l is None for l in lists):
            return None
        lens = np.fromiter((len(l) for l in lists), dtype=np.int64, count=len(lists))
        off = np.zeros(len(lists) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        flat = np.fromiter((v for l in lists for v in l), dtype=np.int64, count=int(off[-1]))
        # (+ the arrays' addresses: a window's slices are then plain pointer arithmetic for the host routines)
        return flat, off, lens, flat.ctypes.data, off.ctypes.data, lens.ctypes.data


class BatchIds(object):
    """The window of a FormulaIds one collated batch covers; rides on the batch's query-graph object (`q_graphs.ids`)."""
    __slots__ = ('fi', 'start', 'end', 'anchor_ref')

    def __init__(self, fi, start, end, anchor_ref=None):
        self.fi, self.start, self.end, self.anchor_ref = fi, start, end, anchor_ref


class QueryDataset(Dataset):
    """reference: data_utils.py:268-311. One formula per batch, drawn with probability
    proportional to its number of queries; the index window wraps around the formula's list."""

    def __init__(self, queries, *args, **kwargs):
        self.queries = queries
        self.num_formula_queries = OrderedDict((f, len(qs)) for f, qs in queries.items())
        self.num_queries = sum(self.num_formula_queries.values())
        self.max_num_queries = max(self.num_formula_queries.values())

    def __len__(self):
        return self.max_num_queries

    def __getitem__(self, index):
        return index

    def _window(self, idx_list):
        w = self.__dict__.get('_pick_tables')
        if w is None:       # (the same probability vector the reference builds per batch, built once)
            counts = np.array(list(self.num_formula_queries.values()))
            formulas = list(self.num_formula_queries.keys())
            w = self.__dict__['_pick_tables'] = (counts / float(self.num_queries), formulas, [None] * len(formulas))
        pick = np.argmax(np.random.multinomial(1, w[0]))
        formula = w[1][pick]
        n = self.num_formula_queries[formula]
        lo, hi = idx_list[0], idx_list[-1]
        start = lo % n
        end = min((hi + 1) % n, n)
        if end <= start:
            end = n
        return formula, start, end

    def collate_fn(self, idx_list):
        formula, start, end = self._window(idx_list)
        return formula, self.queries[formula][start:end]


class RGCNQueryDataset(QueryDataset):
    """reference: data_utils.py:314-409."""
    # kept as class attributes because the reference exposes them (model.py:426)
    query_diameters = {'1-chain': 1, '2-chain': 2, '3-chain': 3, '2-inter': 1, '3-inter': 1,
                       '3-inter_chain': 2, '3-chain_inter': 2}

    def __init__(self, queries, enc_dec):
        super(RGCNQueryDataset, self).__init__(queries)
        self.mode_ids = enc_dec.mode_ids
        self.rel_ids = enc_dec.rel_ids

        self._formula_ids = {}

    def formula_ids(self, formula):
        fi = self._formula_ids.get(formula)
        if fi is None:
            fi = self._formula_ids[formula] = FormulaIds(self.queries[formula])
        return fi

    def collate_fn(self, idx_list):
        """reference: data_utils.py:369-375. Same draws (one np.random.multinomial per batch), same window, same return
        values; the ids of the window come from the formula's arrays (FormulaIds, built on the formula's first batch)
        instead of per-query python loops, and ride along on the graph object so that margin_loss need not walk the
        Query objects again (mpqe_amd/dropin.py). Everything that depends on the formula alone sits in one record per
        formula, found by the multinomial draw's index (no hashing of Formula objects on the way)."""
        tabs = self.__dict__.get('_pick_tables')
        if tabs is None:
            tabs = self._tables()
        pick = int(np.random.multinomial(1, tabs[0]).argmax())
        rec = tabs[2][pick]
        if rec is None:
            rec = tabs[2][pick] = self._formula_record(tabs[1][pick])
        formula, n, all_queries, fi, var_ids, edge_type, templates = rec
        lo, hi = idx_list[0], idx_list[-1]
        start = lo % n
        end = min((hi + 1) % n, n)
        if end <= start:
            end = n
        tmpl = templates.get(end - start)
        if tmpl is None:        # (immutable: one per formula and batch size)
            tmpl = templates[end - start] = ops.Template(formula.query_type, end - start, edge_type)
        graph = QueryGraphBatch(tmpl)
        anchor_ids = torch.from_numpy(fi.anchors[start:end])
        graph.ids = BatchIds(fi, start, end, anchor_ids)
        return formula, all_queries[start:end], anchor_ids, torch.from_numpy(var_ids.copy()), graph

    def _tables(self):
        counts = np.array(list(self.num_formula_queries.values()))
        formulas = list(self.num_formula_queries.keys())
        tabs = self.__dict__['_pick_tables'] = (counts / float(self.num_queries), formulas, [None] * len(formulas))
        return tabs

    def _formula_record(self, formula):
        fi = self.formula_ids(formula)
        info = ops.template_info(formula.query_type)
        if fi.A != info.num_anchors or info.num_anchors != len(formula.anchor_modes):
            raise ValueError('formula %s has %d anchor modes, template expects %d'
                             % (formula, len(formula.anchor_modes), info.num_anchors))
        var_ids, edge_type = self._formula_consts(formula, info)
        return (formula, self.num_formula_queries[formula], self.queries[formula], fi, var_ids, edge_type, {})

    def _formula_consts(self, formula, info):
        c = getattr(self, '_consts', None)
        if c is None:
            c = self._consts = {}
        v = c.get(formula)
        if v is None:
            nodes, rels = formula.get_nodes(), formula.get_rels()
            var_ids = np.array([self.mode_ids[nodes[info.var_node[k]]] for k in range(info.num_vars)], dtype=np.int64)
            edge_type = [self.rel_ids[reverse_relation(rels[info.rel_label[e]])] for e in range(info.num_edges)]
            v = c[formula] = (var_ids, edge_type)
        return v

    @staticmethod
    def get_query_graph(formula, queries, rel_ids, mode_ids):
        info = ops.template_info(formula.query_type)
        B, A, V, E = len(queries), info.num_anchors, info.num_vars, info.num_edges
        if A != len(formula.anchor_modes):
            raise ValueError('formula %s has %d anchor modes, template expects %d'
                             % (formula, len(formula.anchor_modes), A))
        anchor_ids = np.empty((B, A), dtype=np.int64)
        for i in range(A):
            anchor_ids[:, i] = [q.anchor_nodes[i] for q in queries]
        nodes = formula.get_nodes()
        var_ids = np.array([mode_ids[nodes[info.var_node[k]]] for k in range(V)], dtype=np.int64)
        rels = formula.get_rels()
        edge_type = [rel_ids[reverse_relation(rels[info.rel_label[e]])] for e in range(E)]
        graph = QueryGraphBatch(ops.Template(formula.query_type, B, edge_type))
        return torch.from_numpy(anchor_ids), torch.from_numpy(var_ids), graph


def make_data_iterator(data_loader):
    while True:
        for item in data_loader:
            yield item


def _sequential_batches(dataset, batch_size):
    """What `make_data_iterator(DataLoader(dataset, batch_size, shuffle=False, collate_fn=dataset.collate_fn))` yields
    (reference data_utils.py:412-426), without the loader: a sequential sampler's index windows [lo, lo + batch_size) over
    range(len(dataset)), the last one short, epoch after epoch; collate_fn reads only the window's first and last index
    (data_utils.py:293-311), so it gets a range. The loader fetched `dataset[i]` for every index of a window first (512
    interpreter calls that return i) and wrapped every batch in profiler scopes: ~100 us per batch of the ~120.
    The ONE thing the loader does besides: each epoch's iterator draws a base seed from torch's global generator
    (`torch.empty((), dtype=torch.int64).random_()`, torch/utils/data/dataloader.py) -- repeated here so that torch's random
    stream stays where the reference's run would leave it."""
    n = len(dataset)
    if batch_size is None or batch_size <= 0:
        raise ValueError('batch_size should be a positive integer value, but got batch_size=%r' % (batch_size,))
    while True:
        torch.empty((), dtype=torch.int64).random_()
        for lo in range(0, n, batch_size):
            yield dataset.collate_fn(range(lo, min(lo + batch_size, n)))


def get_queries_iterator(queries, batch_size, enc_dec=None):
    """reference: data_utils.py:422-426."""
    dataset = RGCNQueryDataset(queries, enc_dec)
    return _sequential_batches(dataset, batch_size)


def make_feature_modules(node_ids_by_mode, embed_dim, num_entities=None):
    """The embedding-table layout of load_graph (reference data_utils.py:20-35) from
    {mode: list of global entity ids}: per mode an nn.Embedding(count + 1, D) initialised
    N(0, 1/D), and the int64 LUT global id -> row (-1 = not of any mode)."""
    if num_entities is None:
        num_entities = sum(len(v) for v in node_ids_by_mode.values())
    node_maps = torch.full((num_entities + 1,), -1, dtype=torch.long)
    feature_modules = {}
    for mode, ids in node_ids_by_mode.items():
        ids = torch.as_tensor(np.asarray(ids), dtype=torch.long)
        if (node_maps[ids] != -1).any():
            raise ValueError('entity id listed under two modes')
        node_maps[ids] = torch.arange(ids.shape[0])
        feature_modules[mode] = torch.nn.Embedding(ids.shape[0] + 1, embed_dim)
        feature_modules[mode].weight.data.normal_(0, 1. / embed_dim)
    return feature_modules, node_maps

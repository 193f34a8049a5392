"""Seeded synthetic knowledge graphs and query batches of the shapes
BASELINE.json names (the datasets themselves are not in the reference tree and
cannot be downloaded: reference README.md:26-40).

A schema is plain data (mode names, typed relations closed under inverse,
global entity ids per mode) so that the same numbers can be fed to the
reference's own Graph/Query classes by oracle/gen_golden.py and to this
package's harness types.
"""
from collections import OrderedDict

import numpy as np

from .graph import Formula, Query, reverse_relation

# name -> (entities, modes, relation names); SURVEY.md section 8(d)
KG_SHAPES = {
    'aifb': (2601, 6, 49),
    'mutag': (22372, 4, 8),
    'am': (372584, 5, 19),
    'stress': (1000000, 8, 64),
    'tiny': (60, 3, 4),
}


class Schema(object):
    """modes: list of names; relations: OrderedDict mode -> [(to_mode, name)];
    ids[mode]: int64 array of the global entity ids of that mode."""

    def __init__(self, modes, relations, ids, num_entities):
        self.modes = modes
        self.relations = relations
        self.ids = ids
        self.num_entities = num_entities

    def typed_relations(self):
        out = []
        for m in self.relations:
            for (to, name) in self.relations[m]:
                out.append((m, name, to))
        return out


def make_schema(num_entities, num_modes, num_rel_names, seed=0):
    rng = np.random.RandomState(seed)
    modes = ['m%d' % i for i in range(num_modes)]
    relations = OrderedDict((m, []) for m in modes)
    seen = set()

    def add(m1, name, m2):
        for a, b in ((m1, m2), (m2, m1)):
            key = (a, name, b)
            if key not in seen:
                seen.add(key)
                relations[a].append((b, name))

    # a ring first so every mode has an outgoing relation, then random pairs
    for k in range(num_rel_names):
        if k < num_modes:
            i, j = k, (k + 1) % num_modes
        else:
            i, j = int(rng.randint(num_modes)), int(rng.randint(num_modes))
        add(modes[i], 'r%d' % k, modes[j])
    perm = rng.permutation(num_entities).astype(np.int64)
    ids = OrderedDict()
    bounds = np.linspace(0, num_entities, num_modes + 1).astype(np.int64)
    for i, m in enumerate(modes):
        ids[m] = np.sort(perm[bounds[i]:bounds[i + 1]])
    return Schema(modes, relations, ids, num_entities)


def make_adjacency(schema, degree=2, seed=0):
    """adj_lists[(m1,name,m2)][node] -> set(neighbours), consistent with the
    inverse relation. Only for small schemas (Python dict of sets)."""
    rng = np.random.RandomState(seed + 17)
    adj = OrderedDict()
    for rel in schema.typed_relations():
        adj[rel] = {int(n): set() for n in schema.ids[rel[0]]}
    done = set()
    for rel in schema.typed_relations():
        if rel in done:
            continue
        inv = reverse_relation(rel)
        done.add(rel)
        done.add(inv)
        src, dst = schema.ids[rel[0]], schema.ids[rel[2]]
        for n in src:
            for d in rng.choice(dst, size=min(degree, len(dst)), replace=False):
                adj[rel][int(n)].add(int(d))
                adj[inv][int(d)].add(int(n))
    return adj


class SchemaGraph(object):
    """Adjacency-free stand-in for graph.Graph for large synthetic KGs: carries
    exactly the attributes the encoder reads (see graph.Graph docstring)."""

    def __init__(self, schema, embed_dim):
        self.schema = schema
        self.relations = schema.relations
        self.feature_dims = {m: embed_dim for m in schema.modes}
        self.rel_edges = OrderedDict((r, 1.0) for r in schema.typed_relations())
        self.mode_weights = OrderedDict()
        for r in self.rel_edges:
            self.mode_weights.setdefault(r[0], 0.0)
            self.mode_weights[r[0]] += 1.0 / len(self.rel_edges)
        self.full_lists = {m: schema.ids[m] for m in schema.modes}
        self.features = None
        self.adj_lists = None


# ----------------------------------------------------------------------------- formulas / queries
def sample_formula(schema, query_type, rng):
    rel = schema.relations

    def step(m):
        to, name = rel[m][int(rng.randint(len(rel[m])))]
        return (m, name, to)

    t = schema.modes[int(rng.randint(len(schema.modes)))]
    if query_type.endswith('-chain'):
        k = int(query_type[0])
        rels, m = [], t
        for _ in range(k):
            r = step(m)
            rels.append(r)
            m = r[2]
        return Formula(query_type, tuple(rels))
    if query_type.endswith('-inter'):
        k = int(query_type[0])
        return Formula(query_type, tuple(step(t) for _ in range(k)))
    r0 = step(t)
    if query_type == '3-inter_chain':
        r1 = step(t)
        r2 = step(r1[2])
    elif query_type == '3-chain_inter':
        r1 = step(r0[2])
        r2 = step(r0[2])
    else:
        raise ValueError('unknown query type %r' % (query_type,))
    return Formula(query_type, (r0, (r1, r2)))


def _pick(schema, mode, rng, size=None):
    ids = schema.ids[mode]
    return ids[rng.randint(len(ids), size=size)]


def query_graph_tuple(formula, target, anchors, variables):
    """Grounded ('type', edge, ...) tuple; `variables` are the entity ids bound
    to the non-target variable nodes (never read by the encoder)."""
    qt, rels = formula.query_type, formula.rels
    if qt.endswith('-chain'):
        nodes = [target] + list(variables) + [anchors[0]]
        return (qt,) + tuple((nodes[i], rels[i], nodes[i + 1]) for i in range(len(rels)))
    if qt.endswith('-inter'):
        return (qt,) + tuple((target, rels[i], anchors[i]) for i in range(len(rels)))
    if qt == '3-inter_chain':
        v = variables[0]
        return (qt, (target, rels[0], anchors[0]),
                ((target, rels[1][0], v), (v, rels[1][1], anchors[1])))
    v = variables[0]
    return (qt, (target, rels[0], v),
            ((v, rels[1][0], anchors[0]), (v, rels[1][1], anchors[1])))


def num_variables(query_type):
    return {'1-chain': 0, '2-chain': 1, '3-chain': 2, '2-inter': 0, '3-inter': 0,
            '3-inter_chain': 1, '3-chain_inter': 1}[query_type]


def sample_queries(schema, formula, batch_size, rng, n_neg=1, n_hard=1):
    """`batch_size` Query objects of one formula with uniformly drawn
    entities of the right modes."""
    out = []
    var_modes = []
    if formula.query_type.endswith('-chain'):
        var_modes = [r[2] for r in formula.get_rels()[:-1]]
    elif formula.query_type == '3-inter_chain':
        var_modes = [formula.rels[1][0][2]]
    elif formula.query_type == '3-chain_inter':
        var_modes = [formula.rels[0][2]]
    for _ in range(batch_size):
        tgt = int(_pick(schema, formula.target_mode, rng))
        anchors = [int(_pick(schema, m, rng)) for m in formula.anchor_modes]
        vs = [int(_pick(schema, m, rng)) for m in var_modes]
        neg = [int(x) for x in _pick(schema, formula.target_mode, rng, size=n_neg)]
        hard = None
        if 'inter' in formula.query_type:
            hard = [int(x) for x in _pick(schema, formula.target_mode, rng, size=n_hard)]
        out.append(Query(query_graph_tuple(formula, tgt, anchors, vs), neg, hard,
                         keep_graph=True))
    return out


# The post-burn-in training step of the reference draws these batches
# (train_helpers.py:81, 97-112): one 1-chain, one each of the other chains, and
# for every intersection type a normal and a hard-negative batch.
FULL_MIX = [('1-chain', False), ('2-chain', False), ('3-chain', False),
            ('2-inter', False), ('2-inter', True), ('3-inter', False), ('3-inter', True),
            ('3-inter_chain', False), ('3-inter_chain', True),
            ('3-chain_inter', False), ('3-chain_inter', True)]


# ----------------------------------------------------------------------------- grounded queries on a real adjacency
# For end-task runs (tools/train_synthetic.py): queries whose target really answers them on `adj`, negatives that do
# not. A minimal stand-in for the reference's KG sampler (mpqe/graph.py:227-473, out of scope): one random
# walk per query, the answer set by following the inverse relations back from the anchors.
def _answers(adj, formula, anchors):
    """All entities that answer the query (formula, anchors) on adj."""
    def back(rel, nodes):          # {x : some y in nodes with y in adj[rel][x]} = union of the inverse lists
        inv = adj[reverse_relation(rel)]
        out = set()
        for y in nodes:
            out |= inv.get(y, set())
        return out
    qt, rels = formula.query_type, formula.rels
    if qt.endswith('-chain'):
        s = {anchors[0]}
        for r in reversed(rels):
            s = back(r, s)
        return s, [s]
    if qt.endswith('-inter'):
        br = [back(rels[i], {anchors[i]}) for i in range(len(rels))]
        return set.intersection(*br), br
    if qt == '3-inter_chain':
        br = [back(rels[0], {anchors[0]}), back(rels[1][0], back(rels[1][1], {anchors[1]}))]
        return br[0] & br[1], br
    v = back(rels[1][0], {anchors[0]}) & back(rels[1][1], {anchors[1]})         # 3-chain_inter
    s = back(rels[0], v)
    return s, [s]


def sample_grounded_queries(schema, adj, formula, count, rng, n_neg=8, n_hard=4, max_tries=50):
    """`count` Query objects of `formula` grounded on adj: the target reaches every anchor along the formula's relations;
    neg_samples = entities of the target mode that are NOT answers; hard_neg_samples (intersection types) = entities
    that satisfy some branch but not the query (falling back to plain negatives where there are none)."""
    def nb(rel, x):
        s = adj[rel].get(int(x))
        return None if not s else sorted(s)[int(rng.randint(len(s)))]
    qt, rels = formula.query_type, formula.rels
    pool = schema.ids[formula.target_mode]
    out = []
    for _ in range(count):
        for _try in range(max_tries):
            t = int(_pick(schema, formula.target_mode, rng))
            ok = True
            if qt.endswith('-chain'):
                x, vs = t, []
                for r in rels:
                    x = nb(r, x)
                    if x is None:
                        ok = False
                        break
                    vs.append(x)
                anchors, variables = (vs[-1:] if ok else []), vs[:-1]
            elif qt.endswith('-inter'):
                anchors = [nb(r, t) for r in rels]
                variables = []
                ok = all(a is not None for a in anchors)
            elif qt == '3-inter_chain':
                a0, v = nb(rels[0], t), nb(rels[1][0], t)
                a1 = nb(rels[1][1], v) if v is not None else None
                anchors, variables = [a0, a1], [v]
                ok = a0 is not None and a1 is not None
            else:
                v = nb(rels[0], t)
                a0 = nb(rels[1][0], v) if v is not None else None
                a1 = nb(rels[1][1], v) if v is not None else None
                anchors, variables = [a0, a1], [v]
                ok = a0 is not None and a1 is not None
            if not ok:
                continue
            ans, branches = _answers(adj, formula, anchors)
            assert t in ans
            cand = [int(x) for x in pool if int(x) not in ans]
            if not cand:
                continue
            neg = [cand[int(i)] for i in rng.randint(len(cand), size=n_neg)]
            hard = None
            if 'inter' in qt:
                some = set().union(*branches) - ans
                hc = sorted(some) if some else cand
                hard = [hc[int(i)] for i in rng.randint(len(hc), size=n_hard)]
            out.append(Query(query_graph_tuple(formula, t, anchors, variables), neg, hard, keep_graph=True))
            break
        else:
            raise RuntimeError('no grounded query of %s found' % (formula,))
    return out

"""Harness types the query-graph encoder reads: formulas, queries and the KG
schema container. Only the attributes the hot path touches are mirrored
(reference: mpqe/graph.py:11-58 Formula, 60-123 Query, 127-170 Graph); the
reference's samplers, negative mining and self-checks are out of scope.

The encoder is duck-typed: objects of the reference's own classes work too.
"""
import random
from collections import OrderedDict

CHAIN_TYPES = ('1-chain', '2-chain', '3-chain')
INTER_TYPES = ('2-inter', '3-inter')
QUERY_TYPES = CHAIN_TYPES + INTER_TYPES + ('3-inter_chain', '3-chain_inter')


def reverse_relation(rel):
    """(m1, name, m2) -> (m2, name, m1)   (reference: graph.py:4-5)."""
    return (rel[2], rel[1], rel[0])


def _flat_triples(rels):
    """Depth-first list of the (from_mode, name, to_mode) triples nested in
    `rels` (reference: Formula.flatten + get_rels, graph.py:26-39)."""
    out = []

    def walk(item):
        if len(item) == 3 and not isinstance(item[0], (tuple, list)):
            out.append(tuple(item))
        else:
            for sub in item:
                walk(sub)
    walk(rels)
    return out


class Formula(object):
    """A query shape: `query_type` plus the typed relations on its edges.

    rels layout per type (target mode first in every triple):
      k-chain        (r_0, ..., r_{k-1})      target -r_0-> v ... -r_{k-1}-> anchor
      k-inter        (r_0, ..., r_{k-1})      target -r_i-> anchor_i
      3-inter_chain  (r_0, (r_1, r_2))        target -r_0-> a_0 ; target -r_1-> v -r_2-> a_1
      3-chain_inter  (r_0, (r_1, r_2))        target -r_0-> v ; v -r_1-> a_0 ; v -r_2-> a_1
    """

    def __init__(self, query_type, rels):
        if query_type not in QUERY_TYPES:
            raise ValueError('unknown query type %r' % (query_type,))
        self.query_type = query_type
        self.rels = rels
        self.target_mode = rels[0][0]
        if query_type in CHAIN_TYPES:
            self.anchor_modes = (rels[-1][-1],)
        elif query_type in INTER_TYPES:
            self.anchor_modes = tuple(r[-1] for r in rels)
        elif query_type == '3-inter_chain':
            self.anchor_modes = (rels[0][-1], rels[1][-1][-1])
        else:  # 3-chain_inter
            self.anchor_modes = (rels[1][0][-1], rels[1][1][-1])

    def get_rels(self):
        return _flat_triples(self.rels)

    def get_nodes(self):
        nodes = []
        for t in _flat_triples(self.rels):
            nodes.append(t[0])
            nodes.append(t[2])
        return nodes

    def _key(self):
        return (self.query_type, self.rels)

    def __hash__(self):
        h = self.__dict__.get('_hash')          # (formulas are dictionary keys on the packing path: hash once)
        if h is None:
            h = self.__dict__['_hash'] = hash(self._key())
        return h

    def __eq__(self, other):
        return self._key() == (other.query_type, other.rels)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __str__(self):
        return '%s: %s' % (self.query_type, self.rels)

    __repr__ = __str__


class Query(object):
    """One grounded query: ('type', edge, ...) with edge = (node, rel, node).
    Reads out the anchors/target the same way the reference does
    (graph.py:62-77). Negative lists are kept as given (no sub-sampling)."""

    def __init__(self, query_graph, neg_samples=None, hard_neg_samples=None,
                 keep_graph=False, neg_sample_max=None):
        qt = query_graph[0]
        edges = query_graph[1:]
        if qt in CHAIN_TYPES or qt in INTER_TYPES:
            rels = tuple(e[1] for e in edges)
            if qt in CHAIN_TYPES:
                self.anchor_nodes = (edges[-1][-1],)
            else:
                self.anchor_nodes = tuple(e[-1] for e in edges)
        elif qt in ('3-inter_chain', '3-chain_inter'):
            rels = (edges[0][1], (edges[1][0][1], edges[1][1][1]))
            if qt == '3-inter_chain':
                self.anchor_nodes = (edges[0][-1], edges[1][-1][-1])
            else:
                self.anchor_nodes = (edges[1][0][-1], edges[1][1][-1])
        else:
            raise ValueError('unknown query type %r' % (qt,))
        self.formula = Formula(qt, rels)
        self.target_node = edges[0][0]
        self.query_graph = query_graph if keep_graph else None
        # neg_sample_max (reference graph.py:81-88): lists at / above the cap are drawn WITHOUT replacement with
        # python's `random` -- also when the cap equals the list's length, which is what deserialize() passes: the
        # loaded negatives are a seeded permutation of the stored ones. None keeps the lists as given.
        if neg_samples is None:
            self.neg_samples = None
        elif neg_sample_max is None or len(neg_samples) < neg_sample_max:
            self.neg_samples = list(neg_samples)
        else:
            self.neg_samples = random.sample(list(neg_samples), neg_sample_max)
        if hard_neg_samples is None:
            self.hard_neg_samples = None
        elif neg_sample_max is None or len(hard_neg_samples) <= neg_sample_max:
            self.hard_neg_samples = list(hard_neg_samples)
        else:
            self.hard_neg_samples = random.sample(list(hard_neg_samples), neg_sample_max)

    def serialize(self):
        """(query_graph, neg_samples, hard_neg_samples): the record the reference's *_queries_*.pkl files hold
        (graph.py:116-120)."""
        if self.query_graph is None:
            raise Exception('Cannot serialize query loaded with query graph!')
        return (self.query_graph, self.neg_samples, self.hard_neg_samples)

    @staticmethod
    def deserialize(serial_info, keep_graph=False):
        """reference graph.py:121-123: the cap on the negative lists is the stored list's own length."""
        return Query(serial_info[0], serial_info[1], serial_info[2], keep_graph=keep_graph,
                     neg_sample_max=None if serial_info[1] is None else len(serial_info[1]))

    def __hash__(self):
        return hash((self.formula, self.target_node, self.anchor_nodes))

    def __eq__(self, other):
        return ((self.formula, self.target_node, self.anchor_nodes) ==
                (other.formula, other.target_node, other.anchor_nodes))


class Graph(object):
    """KG schema + adjacency container. What the encoder needs from it
    (reference: graph.py:131-170; model.py:320-338, 345, 474):
      feature_dims  {mode: D}
      relations     {mode: [(to_mode, rel_name), ...]}   iteration order = rel id order
      rel_edges     {(mode, rel_name, to_mode): edge count}   len() = num_relations
      mode_weights  ordered {mode: weight}                iteration order = mode id order
      full_lists    {mode: [entity ids]}                  1-chain negative pool
    """

    def __init__(self, features, feature_dims, relations, adj_lists):
        self.features = features
        self.feature_dims = feature_dims
        self.relations = relations
        self.adj_lists = adj_lists
        self.full_sets = {}
        for rel, adjs in adj_lists.items():
            self.full_sets.setdefault(rel[0], set()).update(adjs.keys())
        self.full_lists = {m: list(s) for m, s in self.full_sets.items()}
        self._count_edges()

    def _count_edges(self):
        self.rel_edges = OrderedDict()
        n_sources = 0.0
        for m_from in self.relations:
            for (m_to, name) in self.relations[m_from]:
                rel = (m_from, name, m_to)
                lists = self.adj_lists[rel].values()
                self.rel_edges[rel] = float(sum(len(l) for l in lists))
                n_sources += float(len(lists))
        self.edges = n_sources
        per_mode = OrderedDict()
        self.rel_weights = OrderedDict()
        for rel, cnt in self.rel_edges.items():
            self.rel_weights[rel] = cnt / self.edges
            per_mode[rel[0]] = per_mode.get(rel[0], 0.0) + cnt
        self.mode_edges = per_mode
        self.mode_weights = OrderedDict(
            (m, cnt / self.edges) for m, cnt in per_mode.items())

"""Dataset loaders (reference data_utils.py:18-37, 150-182; graph.py:116-123) on files the reference's own classes
wrote (oracle/gen_golden.py: run_loader_case): what this package loads from tests/golden/dataset/*.pkl must equal
what the reference's loaders returned there, under the same python / torch seeds -- formulas, grouping and order,
anchors, targets, the seeded permutation of the negative lists, the LUT, the embedding tables' draws, the relation /
mode orderings the encoder derives its ids from."""
import json
import os
import random

import numpy as np
import torch

from mpqe_amd import data_utils
from mpqe_amd.graph import Query

HERE = os.path.dirname(os.path.abspath(__file__))
DS = os.path.join(HERE, 'golden', 'dataset')
EXPECT = json.load(open(os.path.join(DS, 'expect.json')))


def _tup(o):
    return tuple(_tup(x) for x in o) if isinstance(o, list) else o


def _check_query(q, rec):
    assert q.formula.query_type == rec['type']
    assert q.formula.rels == _tup(rec['rels'])
    assert q.formula.target_mode == rec['target_mode']
    assert list(q.formula.anchor_modes) == rec['anchor_modes']
    assert list(q.anchor_nodes) == rec['anchors']
    assert q.target_node == rec['target']
    assert q.neg_samples == rec['neg']
    assert q.hard_neg_samples == rec['hard']


def _check_grouped(got, expect):
    assert list(got.keys()) == [qt for qt, _ in expect]
    for qt, formulas in expect:
        assert [f.rels for f in got[qt].keys()] == [_tup(r) for r, _ in formulas]
        for (f, qs), (_, recs) in zip(got[qt].items(), formulas):
            assert len(qs) == len(recs)
            for q, rec in zip(qs, recs):
                assert q.formula == f
                _check_query(q, rec)


def test_load_queries_by_formula_equals_reference():
    random.seed(11)
    got = data_utils.load_queries_by_formula(os.path.join(DS, 'train_queries.pkl'))
    _check_grouped(got, EXPECT['train'])


def test_load_test_queries_by_formula_equals_reference():
    random.seed(12)
    got = data_utils.load_test_queries_by_formula(os.path.join(DS, 'test_queries.pkl'))
    assert set(got.keys()) == {'full_neg', 'one_neg'}
    for neg in ('full_neg', 'one_neg'):
        _check_grouped(got[neg], EXPECT['test'][neg])
    # the split itself: one stored negative vs a list
    assert all(len(q.neg_samples) == 1 for fs in got['one_neg'].values() for qs in fs.values() for q in qs)
    assert all(len(q.neg_samples) > 1 for fs in got['full_neg'].values() for qs in fs.values() for q in qs)


def test_load_queries_and_by_type_and_round_trip():
    random.seed(3)
    flat = data_utils.load_queries(os.path.join(DS, 'train_queries.pkl'), keep_graph=True)
    random.seed(3)
    by_type = data_utils.load_queries_by_type(os.path.join(DS, 'train_queries.pkl'))
    assert sum(len(v) for v in by_type.values()) == len(flat) == 42
    k = {}
    for q in flat:
        i = k.get(q.formula.query_type, 0)
        assert by_type[q.formula.query_type][i] == q and by_type[q.formula.query_type][i].neg_samples == q.neg_samples
        k[q.formula.query_type] = i + 1
    # serialize -> deserialize keeps the query; the negatives come back as a permutation (reference graph.py:81-83)
    for q in flat[:10]:
        q2 = Query.deserialize(q.serialize(), keep_graph=True)
        assert q2 == q and sorted(q2.neg_samples) == sorted(q.neg_samples) and q2.hard_neg_samples == q.hard_neg_samples


def test_load_graph_equals_reference():
    g = EXPECT['graph']
    torch.manual_seed(13)
    graph, feature_modules, node_maps = data_utils.load_graph(DS, 8)
    assert node_maps.dtype == torch.long and node_maps.tolist() == g['node_map']
    assert list(feature_modules.keys()) == g['modes']
    for m in g['modes']:
        w = feature_modules[m].weight
        assert w.shape == (g['feature_rows'][m], 8)
        np.testing.assert_allclose(w[0].tolist(), g['feature_first_row'][m], rtol=0, atol=0)      # same torch draws
        np.testing.assert_allclose(float(w.double().sum()), g['feature_sum'][m], rtol=1e-12)
    assert [[list(k), v] for k, v in graph.rel_edges.items()] == g['rel_edges']
    assert [k for k, _ in graph.mode_weights.items()] == [k for k, _ in g['mode_weights']]
    np.testing.assert_allclose([v for _, v in graph.mode_weights.items()], [v for _, v in g['mode_weights']], rtol=1e-12)
    assert {m: sorted(v) for m, v in graph.full_lists.items()} == g['full_lists']
    first_mode = g['modes'][0]
    some = [i for i, r in enumerate(g['node_map']) if r == 1]
    ent = [e for e in some if e in g['full_lists'].get(first_mode, []) or True][0]
    # the features closure: LUT lookup into the mode's table
    row = graph.features(torch.tensor([ent]), first_mode) if node_maps[ent] >= 0 else None
    assert row is not None and row.shape == (1, 8)


def test_loaded_graph_feeds_the_encoder_ids():
    """mode / relation id orderings derived from the loaded graph == the reference's (model.py:326-338 iterate
    graph.mode_weights and graph.relations)."""
    from oracle import ref_cpu
    graph, feature_modules, node_maps = data_utils.load_graph(DS, 8)
    mode_ids, rel_ids = ref_cpu.build_ids(graph.relations, graph.mode_weights)
    assert list(mode_ids.keys()) == [k for k, _ in EXPECT['graph']['mode_weights']]
    assert len(rel_ids) == len(EXPECT['graph']['rel_edges'])
    assert [list(k) for k in rel_ids.keys()] == [k for k, _ in EXPECT['graph']['rel_edges']]


def test_load_graph_rejects_ids_listed_twice(tmp_path):
    """reference data_utils.py:27 asserts, id by id, that an entity is not mapped yet: an id under two modes fails there,
    and so does an id repeated inside ONE mode's list."""
    import pickle
    import pytest
    rels = {'a': [('b', 'r')], 'b': [('a', 'r')]}
    for node_ids in ({'a': [0, 1, 2], 'b': [2, 3]}, {'a': [0, 1, 1], 'b': [2, 3]}):
        d = tmp_path / ('g%d' % len(node_ids['a']) + str(node_ids['b'][0] == 2 and node_ids['a'][2]))
        d.mkdir()
        with open(d / 'graph_data.pkl', 'wb') as f:
            pickle.dump((rels, {}, node_ids), f)
        with pytest.raises(AssertionError):
            data_utils.load_graph(str(d), 8)

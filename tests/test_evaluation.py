"""The evaluation loop (mpqe_amd/evaluation.py; reference utils.py:25-95): the numpy metrics against sklearn /
scipy, and the loop's batch slicing / negative sampling / ragged score layout against a scripted model."""
import random

import numpy as np
import pytest
import torch

from mpqe_amd import evaluation


def test_metrics_match_sklearn_and_scipy():
    sklearn_metrics = pytest.importorskip('sklearn.metrics')
    stats = pytest.importorskip('scipy.stats')
    rng = np.random.RandomState(0)
    for n in (2, 7, 200):
        labels = rng.randint(0, 2, size=n)
        labels[0], labels[1] = 0, 1
        scores = np.round(rng.randn(n), 1)                    # plenty of ties
        assert abs(evaluation.roc_auc(labels, scores) - sklearn_metrics.roc_auc_score(labels, scores)) < 1e-12
        for s in (scores[0], 0.05, -9.0, 9.0):
            assert abs(evaluation.percentile_of_score(scores, s) - stats.percentileofscore(scores, s)) < 1e-12
    with pytest.raises(ValueError):
        evaluation.roc_auc([1, 1], [0.1, 0.2])


class _Q(object):
    def __init__(self, target, negs, hard):
        self.target_node, self.neg_samples, self.hard_neg_samples = target, negs, hard


class _Scripted(object):
    """forward() returns score(node) = node / 100: targets first, then the flat negatives, and checks the
    ragged layout it is given."""
    def __init__(self):
        self.calls = []

    def forward(self, formula, queries, targets, neg_nodes=None, neg_lengths=None):
        assert len(targets) == len(queries) == len(neg_lengths) and sum(neg_lengths) == len(neg_nodes)
        self.calls.append((formula, len(queries), list(neg_lengths)))
        return torch.tensor([t / 100.0 for t in targets] + [n / 100.0 for n in neg_nodes])


def test_eval_loops_slice_and_score_like_the_reference():
    rng = np.random.RandomState(1)
    queries = {'f1': [_Q(int(rng.randint(40, 90)), [int(v) for v in rng.randint(0, 100, size=rng.randint(1, 6))],
                         [int(v) for v in rng.randint(0, 100, size=2)]) for _ in range(300)],
               'f2': [_Q(50, [10, 60, 50], [70]) for _ in range(5)]}
    m = _Scripted()
    auc, per = evaluation.eval_auc_queries(queries, m, batch_size=128, seed=3)
    assert [c[1] for c in m.calls] == [128, 128, 44, 5] and all(set(c[2]) == {1} for c in m.calls)
    # replay the reference's draw order: random.seed(seed), one choice per query in order
    random.seed(3)
    labels, preds = [], []
    for f in ('f1', 'f2'):
        qs = queries[f]
        for lo in range(0, len(qs), 128):
            batch = qs[lo:lo + 128]
            negs = [random.choice(q.neg_samples) for q in batch]
            labels += [1] * len(batch) + [0] * len(batch)
            preds += [q.target_node / 100.0 for q in batch] + [n / 100.0 for n in negs]
    assert abs(auc - evaluation.roc_auc(labels, preds)) < 1e-12 and set(per) == {'f1', 'f2'}
    m = _Scripted()
    perc = evaluation.eval_perc_queries(queries, m, batch_size=128)
    want = [evaluation.percentile_of_score([n / 100.0 for n in q.neg_samples], q.target_node / 100.0)
            for f in ('f1', 'f2') for q in queries[f]]
    assert abs(perc - np.mean(want)) < 1e-12
    assert m.calls[-1][2] == [3] * 5
    hard = evaluation.eval_perc_queries(queries, _Scripted(), batch_size=64, hard_negatives=True)
    assert 0.0 <= hard <= 100.0

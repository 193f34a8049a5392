"""Evaluation loop of the reference (utils.py:25-95): ROC-AUC with one sampled negative per
query and percentile rank of the target among all of a query's negatives, both through
`enc_dec.forward(formula, queries, targets, neg_nodes=flat, neg_lengths=lengths)` -- the ragged scoring form
whose `repeat_interleave` row map is mpqe_cosine_fwd's `q_row` (include/mpqe_amd.h).

Same call signatures, same python `random` stream for the sampled negatives (seeded per call, as the
reference does), same batch slicing. The two metrics are restated in numpy so nothing here needs sklearn or
scipy; tests/test_evaluation.py checks them against both.
"""
import random

import numpy as np


def roc_auc(labels, scores):
    """Area under the ROC curve = P(score of a positive > score of a negative) + 0.5 P(tie): the
    Mann-Whitney statistic on average ranks (what sklearn.metrics.roc_auc_score returns)."""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    n_pos, n_neg = int(labels.sum()), int((~labels).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError('Only one class present in y_true. ROC AUC score is not defined in that case.')
    order = np.argsort(scores, kind='mergesort')
    s = scores[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):                       # average rank over each run of equal scores
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return (ranks[labels].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def percentile_of_score(a, score):
    """scipy.stats.percentileofscore(a, score) with its default kind='rank'."""
    a = np.asarray(a, dtype=np.float64)
    n = len(a)
    if n == 0:
        return np.nan
    left, right = int((a < score).sum()), int((a <= score).sum())
    return (left + right + (1 if right > left else 0)) * 50.0 / n


def _get_perc_scores(scores, lengths):
    out, start = [], 0
    neg_scores = scores[len(lengths):]
    for i, length in enumerate(lengths):
        out.append(percentile_of_score(neg_scores[start:start + length], scores[i]))
        start += length
    return out


def _batches(formula_queries, batch_size):
    for offset in range(0, len(formula_queries), batch_size):
        yield offset, min(offset + batch_size, len(formula_queries))


def eval_auc_queries(test_queries, enc_dec, batch_size=128, hard_negatives=False, seed=0):
    """-> (overall AUC, {formula: AUC}). One negative per query, drawn with random.choice after
    random.seed(seed) (reference utils.py:34-69)."""
    predictions, labels, formula_aucs = [], [], {}
    random.seed(seed)
    for formula in test_queries:
        formula_labels, formula_predictions = [], []
        formula_queries = test_queries[formula]
        for lo, hi in _batches(formula_queries, batch_size):
            batch_queries = formula_queries[lo:hi]
            pool = 'hard_neg_samples' if hard_negatives else 'neg_samples'
            negatives = [random.choice(getattr(formula_queries[j], pool)) for j in range(lo, hi)]
            lengths = [1] * (hi - lo)
            formula_labels.extend([1] * len(lengths) + [0] * len(negatives))
            targets = [q.target_node for q in batch_queries]
            scores = enc_dec.forward(formula, batch_queries, targets, neg_nodes=negatives, neg_lengths=lengths)
            formula_predictions.extend(scores.detach().cpu().tolist())
        formula_aucs[formula] = roc_auc(formula_labels, np.nan_to_num(formula_predictions))
        labels.extend(formula_labels)
        predictions.extend(formula_predictions)
    return roc_auc(labels, np.nan_to_num(predictions)), formula_aucs


def eval_perc_queries(test_queries, enc_dec, batch_size=128, hard_negatives=False):
    """-> mean percentile rank of the target's score among ALL negatives of its query (reference utils.py:72-95)."""
    perc_scores = []
    for formula in test_queries:
        formula_queries = test_queries[formula]
        for lo, hi in _batches(formula_queries, batch_size):
            batch_queries = formula_queries[lo:hi]
            pool = 'hard_neg_samples' if hard_negatives else 'neg_samples'
            lengths = [len(getattr(formula_queries[j], pool)) for j in range(lo, hi)]
            negatives = [n for j in range(lo, hi) for n in getattr(formula_queries[j], pool)]
            targets = [q.target_node for q in batch_queries]
            scores = enc_dec.forward(formula, batch_queries, targets, neg_nodes=negatives, neg_lengths=lengths)
            perc_scores.extend(_get_perc_scores(scores.detach().cpu().tolist(), lengths))
    return np.mean(perc_scores)

"""Wall time of the reference's evaluation loops (utils.py:34-95: eval_auc_queries with one sampled negative per query,
eval_perc_queries with every negative of every query) through `enc_dec.forward(..., neg_nodes, neg_lengths)` under
torch.no_grad -- on the fused forward (mpqe_amd/dropin.py: one library call per batch of 128 queries, ragged negatives scored
against the query embeddings the call writes) and on the per-op module path the same calls took before round 5.

    python tools/eval_bench.py [--queries 2048] [--negs 100]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--queries', type=int, default=2048)
    ap.add_argument('--negs', type=int, default=100)
    a = ap.parse_args()
    import dropin_loop_bench as dl
    from mpqe_amd import synthetic
    from mpqe_amd.evaluation import eval_auc_queries, eval_perc_queries
    torch.cuda.set_device(0)
    model, _ = dl.build('mp', per_formula=8, n_formulas=1)
    schema = model.graph.schema
    rng = np.random.RandomState(3)
    test_queries = {}
    for qt in ('1-chain', '2-chain', '3-chain', '2-inter', '3-inter', '3-inter_chain', '3-chain_inter'):
        f = synthetic.sample_formula(schema, qt, rng)
        test_queries[f] = synthetic.sample_queries(schema, f, a.queries, rng, n_neg=a.negs, n_hard=8)
    model = model.to('cuda:0').eval()
    out = {'workload': '7 formulas x %d queries, %d negatives each, batches of 128 (reference utils.py:34-95), AIFB-shaped KG, D=128, TM'
                       % (a.queries, a.negs)}
    with torch.no_grad():
        for fused in (True, False):
            model.fused = fused
            eval_auc_queries(test_queries, model)           # warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            auc, _ = eval_auc_queries(test_queries, model)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            perc = eval_perc_queries(test_queries, model)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            out['fused' if fused else 'module_path'] = {'eval_auc_queries_s': t1 - t0, 'eval_perc_queries_s': t2 - t1,
                                                        'auc': float(auc), 'percentile': float(perc)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()

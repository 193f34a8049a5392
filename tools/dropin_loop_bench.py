"""The reference's training-loop body (train_helpers.py:76-120, post-burn-in phase) timed over pre-collated batches:

    optimizer.zero_grad()
    loss = margin_loss(1-chain batch)
    for the other six query types: loss += path_weight * margin_loss(...)   (chains)
                                   loss += inter_weight * margin_loss(...); loss += inter_weight * margin_loss(..., hard_negatives=True)
    loss.item(); loss.backward(); optimizer.step()

exactly as a maintainer who only changed the imports would run it -- model.margin_loss with the
reference's signature, python's `random` stream for the negatives, torch arithmetic on the loss, `.item()` every iteration.
Collation (get_queries_iterator -> collate_fn) is excluded, as SURVEY.md 8d excludes it: the batches are drawn from the
iterators beforehand. `fused=True` is the drop-in on the fused step (mpqe_amd/dropin.py), `fused=False` the per-op module
path the same calls took before round 5.

    python tools/dropin_loop_bench.py [--readout mp] [--iters 200] [--optimizer flat|torch|none]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(readout='mp', D=128, B=512, per_formula=2048, n_formulas=3, seed=0, kg='aifb'):
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.model import RGCNEncoderDecoder
    torch.manual_seed(seed)
    schema = synthetic.make_schema(*synthetic.KG_SHAPES[kg], seed=seed)
    graph = synthetic.SchemaGraph(schema, D)
    graph.full_lists = {m: [int(v) for v in ids] for m, ids in graph.full_lists.items()}
    fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
    adaptive = readout == 'mp'
    model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=readout, num_layers=3,
                               shared_layers=False, adaptive=adaptive, weight_decay=1e-3 if readout not in ('mp', 'sum', 'max') else 0)
    rng = np.random.RandomState(seed + 1)
    train_queries = {}
    for qt in ('1-chain', '2-chain', '3-chain', '2-inter', '3-inter', '3-inter_chain', '3-chain_inter'):
        by = {}
        while len(by) < n_formulas:
            f = synthetic.sample_formula(schema, qt, rng)
            if f not in by:
                by[f] = synthetic.sample_queries(schema, f, per_formula, rng, n_neg=32, n_hard=8)
        train_queries[qt] = by
    return model, train_queries


def precollate(model, train_queries, B, steps):
    """`steps` draws of the 11 batches of one iteration, in the loop's order: [(batch, hard_negatives, weight)]."""
    from mpqe_amd.data_utils import get_queries_iterator
    its = {qt: get_queries_iterator(train_queries[qt], B, model) for qt in train_queries}
    out = []
    for _ in range(steps):
        step = [(next(its['1-chain']), False, None)]
        for qt in train_queries:
            if qt == '1-chain':
                continue
            if 'inter' in qt:
                step.append((next(its[qt]), False, 0.005))
                step.append((next(its[qt]), True, 0.005))
            else:
                step.append((next(its[qt]), False, 0.01))
        out.append(step)
    return out


class _Live(object):
    """The 11 draws of an iteration taken from the live iterators inside the timed loop (run_batch_v2 as it is,
    train_helpers.py:157-162: collation included)."""

    def __init__(self, model, train_queries, B):
        from mpqe_amd.data_utils import get_queries_iterator
        self.its = {qt: get_queries_iterator(train_queries[qt], B, model) for qt in train_queries}
        self.order = [('1-chain', False, None)]
        for qt in train_queries:
            if qt == '1-chain':
                continue
            if 'inter' in qt:
                self.order += [(qt, False, 0.005), (qt, True, 0.005)]
            else:
                self.order.append((qt, False, 0.01))

    def __len__(self):
        return 1

    def __getitem__(self, i):
        its = self.its
        return ((next(its[qt]), hard, w) for qt, hard, w in self.order)


def loop(model, optimizer, steps, iters, warmup):
    """-> seconds per iteration (wall, host + device: the loop synchronises at loss.item() as the reference's does)."""
    def body(step):
        if optimizer is not None:
            optimizer.zero_grad()
        else:
            for p in model.parameters():
                p.grad = None
        loss = None
        for batch, hard, w in step:
            l = model.margin_loss(*batch, hard_negatives=hard)
            if loss is None:
                loss = l
            else:
                loss += w * l
        value = loss.item()
        loss.backward()
        if optimizer is not None:
            optimizer.step()
        return value
    n = len(steps)
    for i in range(warmup):
        body(steps[i % n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        body(steps[(warmup + i) % n])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


class _FlatAdapter(object):
    """mpqe_amd.optim.FlatOptimizer behind torch.optim's two calls (one launch per step over the flat buffers)."""

    def __init__(self, model, lr):
        from mpqe_amd.optim import FlatOptimizer
        self.model = model
        self.opt = FlatOptimizer(model.dropin().step, lr=lr, opt='adam')

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self):
        self.opt.step()


def run(readout='mp', D=128, B=512, iters=200, warmup=20, optimizer='flat', module_iters=10, lr=0.001, seed=0):
    dev = torch.device('cuda', torch.cuda.current_device())
    out = {'workload': 'reference train_helpers.py:76-120 loop body over pre-collated batches: 11 margin_loss calls (B=%d, D=%d, '
                       'readout %s), loss.item(), backward, optimizer step' % (B, D, readout), 'optimizer': optimizer}
    graphs = 11 * B
    for fused in ((True, False) if module_iters > 0 else (True,)):          # (--module-iters 0: the drop-in alone)
        model, train_queries = build(readout, D, B, seed=seed)
        model = model.to(dev)
        model.fused = fused
        model.validate = fused          # (module path: its per-call D2H flag read off, as the round-1..4 bench ran it)
        np.random.seed(seed)
        random.seed(seed)
        n = (iters if fused else module_iters) + warmup
        steps = precollate(model, train_queries, B, min(n, 64))
        if optimizer == 'torch':
            opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr)
        elif optimizer == 'flat' and fused:
            opt = _FlatAdapter(model, lr)
        elif optimizer == 'flat':
            opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr)
        else:
            opt = None
        t = loop(model, opt, steps, iters if fused else module_iters, warmup if fused else 3)
        key = 'fused' if fused else 'module_path'
        out[key] = {'ms_per_step': t * 1e3, 'query_graphs_per_s': graphs / t}
        if fused:
            # the same iteration with its collation: every batch drawn from the live iterators inside the timed loop
            # (get_queries_iterator -> collate_fn: windows of per-formula id arrays, no DataLoader machinery)
            tl = loop(model, opt, _Live(model, train_queries, B), iters, warmup)
            out[key]['ms_per_step_with_collation'] = tl * 1e3
        if fused:
            d = model.dropin()
            out[key]['fused_backward_steps'] = d.steps
            out[key]['negatives_by_library_replay'] = d.fast_sampled
    if 'module_path' in out:
        out['speedup'] = out['module_path']['ms_per_step'] / out['fused']['ms_per_step']
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--readout', default='mp')
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--module-iters', type=int, default=10)
    ap.add_argument('--optimizer', default='flat', choices=['flat', 'torch', 'none'])
    ap.add_argument('--profile', action='store_true', help='cProfile of the fused loop (host side)')
    a = ap.parse_args()
    torch.cuda.set_device(0)
    if a.profile:
        import cProfile
        import pstats
        model, tq = build(a.readout)
        model = model.to('cuda:0')
        steps = precollate(model, tq, 512, 64)
        opt = _FlatAdapter(model, 0.001) if a.optimizer == 'flat' else None
        loop(model, opt, steps, 20, 10)
        pr = cProfile.Profile()
        pr.enable()
        t = loop(model, opt, steps, a.iters, 0)
        pr.disable()
        print('ms per step under the profiler: %.3f' % (t * 1e3))
        pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
    else:
        print(json.dumps(run(a.readout, iters=a.iters, optimizer=a.optimizer, module_iters=a.module_iters)))

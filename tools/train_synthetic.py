"""End-task run on a synthetic KG with a REAL adjacency: the reference's training loop (train_helpers.py:76-120: per
step one 1-chain batch + path_weight x {2,3}-chain + inter_weight x (normal + hard-negative batch of each intersection
type), loss.backward(), Adam; evaluation by ROC-AUC with one sampled negative per query, utils.py:34-69) through

  * the product path: FusedTrainStep (one C-ABI call per step) + FlatOptimizer(adam) + NegativeSampler (device draws) +
    mpqe_amd.evaluation.eval_auc_queries on the drop-in modules, and
  * (--oracle) the CPU oracle in the reference's op sequence + torch.optim.Adam on the SAME schedule: same formulas, same
    query indices, same negatives (the sampler's counter-based stream is reproduced on the CPU bit for bit).

    python tools/train_synthetic.py --kg aifb --embed-dim 128 --batch-size 512 --steps 300            # GPU only
    python tools/train_synthetic.py --kg small --embed-dim 64 --batch-size 64 --steps 300 --oracle    # both, compared

Prints one JSON line: loss curves, AUC before / after training on held-out queries of the same KG (both sides).
Only tests/ and this tool's --oracle leg use oracle/ (the checker, never the thing trained or shipped).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PATH_WEIGHT, INTER_WEIGHT = 0.01, 0.005          # reference train_helpers.py:60-61 (defaults of run_train)
KG_EXTRA = {'small': (480, 4, 8)}


def build(args, device):
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.model import RGCNEncoderDecoder
    shape = KG_EXTRA.get(args.kg) or synthetic.KG_SHAPES[args.kg]
    schema = synthetic.make_schema(*shape, seed=args.seed)
    adj = synthetic.make_adjacency(schema, degree=args.degree, seed=args.seed)
    torch.manual_seed(args.seed)
    graph = synthetic.SchemaGraph(schema, args.embed_dim)
    fm, node_maps = make_feature_modules(schema.ids, args.embed_dim, schema.num_entities)
    model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=args.readout, num_layers=3,
                               shared_layers=False, adaptive=args.readout == 'mp', weight_decay=0)
    rng = np.random.RandomState(args.seed + 1)
    train, test = {}, {}
    for qt in sorted(set(q for q, _ in synthetic.FULL_MIX)):
        train[qt], test[qt] = [], []
        for _ in range(args.formulas):
            f = synthetic.sample_formula(schema, qt, rng)
            train[qt].append((f, synthetic.sample_grounded_queries(schema, adj, f, args.train_queries, rng)))
            test[qt].append((f, synthetic.sample_grounded_queries(schema, adj, f, args.test_queries, rng)))
    return schema, graph, node_maps, model, train, test


def schedule(args, train):
    """Per step the 11 batches of the post-burn-in mix: (query type, hard, formula index, query positions, seed, weight)."""
    from mpqe_amd import synthetic
    rng = np.random.RandomState(args.seed + 2)
    steps = []
    for it in range(args.steps):
        row = []
        for k, (qt, hard) in enumerate(synthetic.FULL_MIX):
            fi = int(rng.randint(len(train[qt])))
            idx = rng.randint(len(train[qt][fi][1]), size=args.batch_size).astype(np.int64)
            w = 1.0 if qt == '1-chain' else (INTER_WEIGHT if 'inter' in qt else PATH_WEIGHT)
            row.append((qt, hard, fi, idx, 100000 * (it + 1) + k, w * args.weight_scale if qt != '1-chain' else w))
        steps.append(row)
    return steps


def test_dict(test):
    out = {}
    for qt in test:
        for f, qs in test[qt]:
            out.setdefault(f, []).extend(qs)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--kg', default='small')
    ap.add_argument('--embed-dim', type=int, default=64)
    ap.add_argument('--batch-size', type=int, default=64)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--lr', type=float, default=0.01)
    ap.add_argument('--readout', default='mp')
    ap.add_argument('--degree', type=int, default=2)
    ap.add_argument('--formulas', type=int, default=2, help='formulas per query type')
    ap.add_argument('--train-queries', type=int, default=256, help='per formula')
    ap.add_argument('--test-queries', type=int, default=96, help='per formula')
    ap.add_argument('--weight-scale', type=float, default=1.0,
                    help='multiplies the reference path / inter weights (0.01 / 0.005): 100 makes every batch type count')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--oracle', action='store_true', help='also train through the CPU oracle + torch.optim.Adam and compare')
    ap.add_argument('--eval-every', type=int, default=0)
    ap.add_argument('--dropin', action='store_true',
                    help="train through the reference's own loop body and entry points (model.margin_loss, loss.backward(): "
                         'mpqe_amd/dropin.py) instead of FusedTrainStep.pack / run')
    args = ap.parse_args(argv)
    out = run_dropin(args) if args.dropin else run(args)
    print(json.dumps(out))
    return out


def run(args):
    from mpqe_amd import evaluation
    from mpqe_amd.fused import FusedTrainStep
    from mpqe_amd.optim import FlatOptimizer
    from mpqe_amd.sampling import NegativeSampler
    device = torch.device('cuda:0')
    schema, graph, node_maps, model, train, test = build(args, device)
    cpu_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    steps = schedule(args, train)
    tq = test_dict(test)
    samplers = {}
    for qt in train:
        for fi, (f, qs) in enumerate(train[qt]):
            samplers[(qt, fi)] = NegativeSampler(qs, device, full_list=graph.full_lists[f.target_mode] if qt == '1-chain' else None)
    anchors = {(qt, fi): np.array([q.anchor_nodes for q in qs], dtype=np.int64) for qt in train for fi, (f, qs) in enumerate(train[qt])}
    targets = {(qt, fi): np.array([q.target_node for q in qs], dtype=np.int64) for qt in train for fi, (f, qs) in enumerate(train[qt])}

    fstep = FusedTrainStep(model)
    opt = FlatOptimizer(fstep, lr=args.lr, opt='adam')
    with torch.no_grad():
        auc0, _ = evaluation.eval_auc_queries(tq, model, batch_size=128, seed=0)
    losses, negs_used, aucs = [], [], []
    t0 = time.perf_counter()
    for it, row in enumerate(steps):
        batches, negs_row = [], []
        for qt, hard, fi, idx, seed, w in row:
            f = train[qt][fi][0]
            neg = samplers[(qt, fi)].sample(idx, seed, hard_negatives=hard).cpu().numpy()
            negs_row.append(neg)
            batches.append(dict(formula=f, anchor_ids=anchors[(qt, fi)][idx], targets=targets[(qt, fi)][idx], negs=neg, weight=w))
        packed = fstep.pack(batches)
        assert fstep.uses_chain(packed) or model.emb_dim not in (64, 128, 256)
        loss = fstep.run(packed)
        opt.step()
        losses.append(loss)
        negs_used.append(negs_row)
        if args.eval_every and (it + 1) % args.eval_every == 0:
            with torch.no_grad():
                aucs.append((it + 1, evaluation.eval_auc_queries(tq, model, batch_size=128, seed=0)[0]))
    fstep.check()
    for s in samplers.values():
        s.check()
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    losses = torch.stack(losses).cpu().numpy()            # [steps, 1 + 11]
    with torch.no_grad():
        auc1, per1 = evaluation.eval_auc_queries(tq, model, batch_size=128, seed=0)
    out = dict(kg=args.kg, embed_dim=args.embed_dim, batch_size=args.batch_size, steps=args.steps, readout=args.readout,
               chain_form=bool(fstep.uses_chain(packed)), train_seconds=train_s,
               loss_first=float(losses[0, 0]), loss_last20=float(losses[-20:, 0].mean()), auc_before=float(auc0),
               auc_after=float(auc1), auc_curve=aucs, loss_curve=[float(v) for v in losses[:, 0]])
    if args.oracle:
        out.update(run_oracle(args, graph, node_maps, model, cpu_state, train, tq, steps, negs_used, anchors, targets))
    return out


def run_oracle(args, graph, node_maps, model, cpu_state, train, tq, steps, negs_used, anchors, targets):
    """The same schedule through the CPU oracle (reference op sequence, two encoder passes) + torch.optim.Adam."""
    from mpqe_amd import evaluation
    from oracle import ref_cpu            # the checker
    params = {k: v.clone().requires_grad_(True) for k, v in cpu_state.items()}
    cfg = dict(readout=args.readout, scatter_op='add', num_layers=3, adaptive=args.readout == 'mp', weight_decay=0)
    node_map = node_maps.cpu() if torch.is_tensor(node_maps) else node_maps
    opt = torch.optim.Adam(list(params.values()), lr=args.lr)
    # the reference's era (torch 1.4) zeroes gradients in place: a parameter that has had a gradient once keeps being
    # updated (zero gradient: its moments decay, its momentum moves it). The flat optimiser is dense Adam over every
    # parameter from step 1 -- the same thing when every parameter TENSOR is touched in the first step, which the
    # 11-batch mix does; the zero gradients are materialised here so that both sides are that rule exactly.
    for p in params.values():
        p.grad = torch.zeros_like(p)

    class Q(object):
        def __init__(self, a):
            self.anchor_nodes = tuple(int(v) for v in a)

    class Oracle(object):
        def forward(self, formula, queries, tg, neg_nodes=None, neg_lengths=None):
            col = ref_cpu.collate(formula, queries, model.rel_ids, model.mode_ids)
            return ref_cpu.forward(params, cfg, node_map, formula, col, tg, neg_nodes, neg_lengths)
    with torch.no_grad():
        auc0, _ = evaluation.eval_auc_queries(tq, Oracle(), batch_size=128, seed=0)
    losses = []
    t0 = time.perf_counter()
    for it, row in enumerate(steps):
        opt.zero_grad(set_to_none=False)
        total = 0
        for k, (qt, hard, fi, idx, seed, w) in enumerate(row):
            f, qs = train[qt][fi]
            # (the device sampler's draws, reproduced by the oracle's CPU stream: tests/test_fused_gpu.py checks the equality)
            col = ref_cpu.collate(f, [Q(a) for a in anchors[(qt, fi)][idx]], model.rel_ids, model.mode_ids)
            l = ref_cpu.margin_loss(params, cfg, node_map, f, col, targets[(qt, fi)][idx], negs_used[it][k])
            total = total + w * l
        total.backward()
        opt.step()
        losses.append(float(total.item()))
    with torch.no_grad():
        auc1, _ = evaluation.eval_auc_queries(tq, Oracle(), batch_size=128, seed=0)
    return dict(oracle_seconds=time.perf_counter() - t0, oracle_loss_first=losses[0],
                oracle_loss_last20=float(np.mean(losses[-20:])), oracle_auc_before=float(auc0), oracle_auc_after=float(auc1),
                oracle_loss_curve=losses)


def _reference_loop(model_like, iterators, query_types, optimizer, steps, margin_loss):
    """The body of the reference's run_train (train_helpers.py:76-120, post-burn-in phase), verbatim in structure:
    margin_loss(batch) per query type, `loss += w * ...`, loss.item(), backward, optimizer step. -> the losses."""
    losses = []
    for _ in range(steps):
        optimizer.zero_grad()
        loss = margin_loss(next(iterators['1-chain']), False)
        for qt in query_types:
            if qt == '1-chain':
                continue
            if 'inter' in qt:
                loss += INTER_WEIGHT * margin_loss(next(iterators[qt]), False)
                loss += INTER_WEIGHT * margin_loss(next(iterators[qt]), True)
            else:
                loss += PATH_WEIGHT * margin_loss(next(iterators[qt]), False)
        losses.append(loss.item())
        loss.backward()
        optimizer.step()
    return losses


def run_dropin(args):
    """End-task run through the reference's OWN entry points (mpqe_amd/dropin.py): the run_train loop body over
    get_queries_iterator batches, model.margin_loss with python's random negatives, loss.backward(), FlatOptimizer --
    beside the same loop through the CPU oracle + torch.optim.Adam. Both sides seed numpy (formula draws) and python's
    `random` (negative draws) alike, and the drop-in replays python's stream exactly, so the two runs see the SAME batches
    and the SAME negatives without any of them being recorded."""
    import random
    from mpqe_amd import evaluation
    from mpqe_amd.data_utils import get_queries_iterator
    from mpqe_amd.optim import FlatOptimizer
    from oracle import ref_cpu            # the checker
    device = torch.device('cuda:0')
    schema, graph, node_maps, model, train, test = build(args, device)
    graph.full_lists = {m: [int(v) for v in ids] for m, ids in graph.full_lists.items()}
    cpu_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    tq = test_dict(test)
    by_type = {qt: {f: qs for f, qs in train[qt]} for qt in train}
    order = [qt for qt, _ in __import__('mpqe_amd.synthetic', fromlist=['FULL_MIX']).FULL_MIX]
    query_types = list(dict.fromkeys(order))

    def iterators(m):
        np.random.seed(args.seed + 11)
        return {qt: get_queries_iterator(by_type[qt], args.batch_size, m) for qt in by_type}
    # ---- the product path
    with torch.no_grad():
        auc0, _ = evaluation.eval_auc_queries(tq, model, batch_size=128, seed=0)
    assert model.dropin() is not None, 'the model did not take the fused step'
    opt = FlatOptimizer(model.dropin().step, lr=args.lr, opt='adam')
    random.seed(args.seed + 12)
    t0 = time.perf_counter()
    losses = _reference_loop(model, iterators(model), query_types, opt, args.steps,
                             lambda batch, hard: model.margin_loss(*batch, hard_negatives=hard))
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    model.dropin()._check_mirror()
    with torch.no_grad():
        auc1, _ = evaluation.eval_auc_queries(tq, model, batch_size=128, seed=0)
    out = dict(kg=args.kg, embed_dim=args.embed_dim, batch_size=args.batch_size, steps=args.steps, readout=args.readout,
               train_seconds=train_s, fused_backward_steps=model.dropin().steps, node_impl=model.dropin().node_impl,
               loss_first=losses[0], loss_last20=float(np.mean(losses[-20:])), auc_before=float(auc0), auc_after=float(auc1),
               loss_curve=losses)
    if not args.oracle:
        return out
    # ---- the same loop through the oracle
    params = {k: v.clone().requires_grad_(True) for k, v in cpu_state.items()}
    cfg = dict(readout=args.readout, scatter_op='add', num_layers=3, adaptive=args.readout == 'mp', weight_decay=0)
    node_map = node_maps.cpu() if torch.is_tensor(node_maps) else node_maps
    oopt = torch.optim.Adam(list(params.values()), lr=args.lr)
    for p in params.values():               # (dense Adam over every parameter from step 1: see run_oracle)
        p.grad = torch.zeros_like(p)

    class ZeroInPlace(object):
        def zero_grad(self):
            oopt.zero_grad(set_to_none=False)

        def step(self):
            oopt.step()

    def oracle_margin_loss(batch, hard):
        formula, queries = batch[0], batch[1]
        if hard:                                                        # reference model.py:470-476, literally
            negs = [random.choice(q.hard_neg_samples) for q in queries]
        elif formula.query_type == '1-chain':
            negs = [random.choice(graph.full_lists[formula.target_mode]) for _ in queries]
        else:
            negs = [random.choice(q.neg_samples) for q in queries]
        col = ref_cpu.collate(formula, queries, model.rel_ids, model.mode_ids)
        return ref_cpu.margin_loss(params, cfg, node_map, formula, col, np.array([q.target_node for q in queries]), np.array(negs))

    class Oracle(object):
        def forward(self, formula, queries, tg, neg_nodes=None, neg_lengths=None):
            col = ref_cpu.collate(formula, queries, model.rel_ids, model.mode_ids)
            return ref_cpu.forward(params, cfg, node_map, formula, col, tg, neg_nodes, neg_lengths)
    with torch.no_grad():
        oauc0, _ = evaluation.eval_auc_queries(tq, Oracle(), batch_size=128, seed=0)
    random.seed(args.seed + 12)
    t0 = time.perf_counter()
    olosses = _reference_loop(None, iterators(model), query_types, ZeroInPlace(), args.steps, oracle_margin_loss)
    with torch.no_grad():
        oauc1, _ = evaluation.eval_auc_queries(tq, Oracle(), batch_size=128, seed=0)
    out.update(oracle_seconds=time.perf_counter() - t0, oracle_loss_first=olosses[0], oracle_loss_last20=float(np.mean(olosses[-20:])),
               oracle_auc_before=float(oauc0), oracle_auc_after=float(oauc1), oracle_loss_curve=olosses)
    return out


if __name__ == '__main__':
    main()

"""The reference's OWN entry points on the fused step (mpqe_amd/dropin.py; SURVEY.md 8b outer boundary): everything here is
driven only through `model.margin_loss(...)` / `loss.backward()` and `model.forward(...)`, the calls of
reference train_helpers.py:76-120, 157-162 and utils.py:34-95 -- against the reference-generated goldens, and at
BASELINE configs[1] (the 11-batch post-burn-in step, B = 512, D = 128, TM) against the CPU oracle in the reference's op
sequence, with the negatives python's own `random` stream draws."""
import random

import numpy as np
import pytest
import torch

from tests.conftest import build_model

pytestmark = pytest.mark.gpu
FWD = dict(rtol=1e-5, atol=1e-6)
BWD = dict(rtol=1e-4, atol=2e-6)


def _np(t):
    return t.detach().cpu().numpy()


def test_goldens_through_margin_loss_use_the_fused_step(enc_case):
    """loss + every gradient of the reference's margin_loss (tests/golden/enc_*: 42 cases) through model.margin_loss /
    backward alone, the call routed through mpqe_step_forward_backward (one forward-only call, one fused backward step)."""
    c = enc_case
    model = build_model(c, torch.device('cuda:0'))
    d = model.dropin()
    if c.cfg['readout'] == 'concat' and c.cfg['adaptive'] and d is not None and \
            d._passes(c.formula) != c.cfg['num_layers']:
        pytest.skip('concat with fewer passes than layers fails in the reference itself (module path keeps its error)')
    assert d is not None
    random.seed(4242 + c.meta['seed'])
    loss = model.margin_loss(c.formula, c.queries, hard_negatives=c.hard_negatives)
    assert loss.requires_grad and loss.dim() == 0
    np.testing.assert_allclose(loss.item(), float(c.arrays['loss']), rtol=1e-5, atol=1e-6)
    assert d.steps == 0
    loss.backward()
    assert d.steps == 1
    got = dict(model.named_parameters())
    for k, g in c.grads().items():
        np.testing.assert_allclose(_np(got[k].grad), g, err_msg=k, **BWD)
    # a second pass without zeroing adds (what autograd's AccumulateGrad does), scaled by the upstream gradient
    random.seed(4242 + c.meta['seed'])
    (0.5 * model.margin_loss(c.formula, c.queries, hard_negatives=c.hard_negatives)).backward()
    for k, g in c.grads().items():
        np.testing.assert_allclose(_np(got[k].grad), 1.5 * g, err_msg=k, **BWD)
    model.dropin()._check_mirror()


def test_goldens_through_forward_without_autograd(enc_case):
    """scores of reference model.py:451-462 under torch.no_grad (utils.py:34-95 evaluate this way): positives alone, one
    negative per query (eval_auc_queries) and ragged negative lists (eval_perc_queries)."""
    c = enc_case
    model = build_model(c, torch.device('cuda:0'))
    d = model.dropin()
    if c.cfg['readout'] == 'concat' and c.cfg['adaptive'] and d._passes(c.formula) != c.cfg['num_layers']:
        pytest.skip('concat with fewer passes than layers fails in the reference itself')
    targets = c.arrays['targets'].tolist()
    with torch.no_grad():
        s_pos = model.forward(c.formula, c.queries, targets)
        s_one = model.forward(c.formula, c.queries, targets, neg_nodes=c.arrays['neg_nodes'].tolist(),
                              neg_lengths=[1] * len(targets))
        s_eval = model.forward(c.formula, c.queries, targets, neg_nodes=c.arrays['eval_negs'].tolist(),
                               neg_lengths=c.arrays['neg_lengths'].tolist())
    np.testing.assert_allclose(_np(s_pos), c.arrays['scores_pos'], **FWD)
    np.testing.assert_allclose(_np(s_one), np.concatenate([c.arrays['scores_pos'], c.arrays['scores_neg']]), **FWD)
    np.testing.assert_allclose(_np(s_eval), c.arrays['eval_scores'], **FWD)


def _aifb(readout='mp', adaptive=True, D=128, n_formulas=2, per_formula=700, seed=0, weight_decay=0.0, kg='aifb'):
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.model import RGCNEncoderDecoder
    torch.manual_seed(seed)
    schema = synthetic.make_schema(*synthetic.KG_SHAPES[kg], seed=seed)
    graph = synthetic.SchemaGraph(schema, D)
    graph.full_lists = {m: [int(v) for v in ids] for m, ids in graph.full_lists.items()}
    fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
    model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=readout, num_layers=3,
                               shared_layers=False, adaptive=adaptive, weight_decay=weight_decay)
    with torch.no_grad():
        for p in model.layers.parameters():
            p.mul_(4.0)
    rng = np.random.RandomState(seed + 5)
    train_queries = {}
    for qt in ('1-chain', '2-chain', '3-chain', '2-inter', '3-inter', '3-inter_chain', '3-chain_inter'):
        by_formula = {}
        while len(by_formula) < n_formulas:
            f = synthetic.sample_formula(schema, qt, rng)
            if f not in by_formula:
                by_formula[f] = synthetic.sample_queries(schema, f, per_formula, rng, n_neg=int(rng.randint(3, 40)),
                                                         n_hard=int(rng.randint(1, 9)))
        train_queries[qt] = by_formula
    return schema, node_maps, model, train_queries


def _reference_loop_body(model, iterators, train_queries, record, inter_weight=0.005, path_weight=0.01):
    """reference train_helpers.py:81-112 (edge_conv phase), run_batch_v2 inlined so the batches can be recorded."""
    def run_batch_v2(it, hard_negatives=False):
        model.train()
        batch = next(it)
        record.append((batch, hard_negatives))
        return model.margin_loss(*batch, hard_negatives=hard_negatives)
    loss = run_batch_v2(iterators['1-chain'])
    for query_type in train_queries:
        if query_type == '1-chain':
            continue
        if 'inter' in query_type:
            loss += inter_weight * run_batch_v2(iterators[query_type])
            loss += inter_weight * run_batch_v2(iterators[query_type], hard_negatives=True)
        else:
            loss += path_weight * run_batch_v2(iterators[query_type])
    return loss


def _oracle_loop_body(cpu_params, cfg, node_maps, model, record, inter_weight=0.005, path_weight=0.01):
    """The same step on the CPU oracle (reference op sequence), the negatives drawn by the reference's own expressions
    (model.py:470-476) from the re-seeded python stream."""
    from oracle import ref_cpu
    total = None
    for (formula, queries, anchor_ids, var_ids, q_graphs), hard in record:
        if hard:
            negs = [random.choice(q.hard_neg_samples) for q in queries]
        elif formula.query_type == '1-chain':
            negs = [random.choice(model.graph.full_lists[formula.target_mode]) for _ in queries]
        else:
            negs = [random.choice(q.neg_samples) for q in queries]
        targets = [q.target_node for q in queries]
        col = ref_cpu.collate(formula, queries, model.rel_ids, model.mode_ids)
        np.testing.assert_array_equal(col['anchor_ids'], anchor_ids.numpy())
        l = ref_cpu.margin_loss(cpu_params, cfg, node_maps, formula, col, np.array(targets), np.array(negs))
        qt = formula.query_type
        w = 1.0 if qt == '1-chain' else (inter_weight if 'inter' in qt else path_weight)
        total = l if total is None else total + w * l
    return total


@pytest.mark.parametrize('kg,D,readout,adaptive,wd', [('aifb', 128, 'mp', True, 0.0), ('aifb', 128, 'sum', False, 0.0),
                                                      ('aifb', 128, 'mlp', True, 1e-3),
                                                      ('mutag', 256, 'sum', False, 0.0)])      # configs[1] (+ variants), configs[2]
def test_reference_training_loop_body_against_oracle(kg, D, readout, adaptive, wd):
    """BASELINE configs[1] through the loop body of reference train_helpers.py:76-120: batches from
    get_queries_iterator (data_utils.py:422-426), 11 margin_loss calls, `loss += w * ...`, loss.backward() -- loss and
    every parameter gradient against the oracle; two iterations (fresh windows, p.grad = None in between as
    optimizer.zero_grad leaves it), and the python `random` stream ends where the reference's draws leave it."""
    from mpqe_amd.data_utils import get_queries_iterator
    schema, node_maps, model, train_queries = _aifb(readout, adaptive, D=D, weight_decay=wd, kg=kg)
    cpu_params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to('cuda:0')
    cfg = dict(readout=readout, scatter_op='add', num_layers=3, adaptive=adaptive, weight_decay=wd)
    np.random.seed(11)
    iterators = {qt: get_queries_iterator(train_queries[qt], 512, model) for qt in train_queries}
    torch.set_num_threads(min(16, max(1, len(__import__('os').sched_getaffinity(0)))))
    d = model.dropin()
    assert d is not None
    for it in range(2 if D <= 128 else 1):          # (the D = 256 oracle step takes ~20 s of CPU)
        for p in model.parameters():
            p.grad = None
        record = []
        random.seed(900 + it)
        loss = _reference_loop_body(model, iterators, train_queries, record)
        state_fast = random.getstate()
        value = loss.item()
        loss.backward()
        assert len(record) == 11 and d.steps == it + 1 and d.fast_sampled == 11 * (it + 1)
        random.seed(900 + it)
        for p in cpu_params.values():
            p.grad = None
        ref = _oracle_loop_body(cpu_params, cfg, node_maps, model, record)
        assert random.getstate() == state_fast          # the library replay consumed exactly the reference's draws
        np.testing.assert_allclose(value, ref.item(), rtol=1e-5, atol=1e-6)
        ref.backward()
        for k, p in model.named_parameters():
            g = cpu_params[k].grad
            g = torch.zeros_like(cpu_params[k]) if g is None else g
            np.testing.assert_allclose(_np(p.grad), g.numpy(), err_msg='%s (iteration %d)' % (k, it), **BWD)
    d._check_mirror()


def test_fused_and_module_paths_agree_and_options():
    """model.fused = False is the per-op module path: same draws, same loss, same gradients; margin and hard negatives
    behave as in reference model.py:464-494; torch.no_grad gives the plain value."""
    from mpqe_amd.data_utils import get_queries_iterator
    schema, node_maps, model, train_queries = _aifb('mp', True, D=64, per_formula=300)
    model = model.to('cuda:0')
    np.random.seed(3)
    it = get_queries_iterator(train_queries['3-inter_chain'], 256, model)
    batch = next(it)
    out = {}
    for fused in (True, False):
        model.fused = fused
        for p in model.parameters():
            p.grad = None
        random.seed(77)
        loss = model.margin_loss(*batch, hard_negatives=True, margin=0.7)
        loss *= 0.3             # (in place on the call's result, as the reference's `loss += ...` does on the first loss)
        loss.backward()
        out[fused] = (loss.item(), {k: (np.zeros(tuple(p.shape), np.float32) if p.grad is None else _np(p.grad).copy())
                                    for k, p in model.named_parameters()})
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-5, atol=1e-6)
    for k in out[True][1]:
        np.testing.assert_allclose(out[True][1][k], out[False][1][k], err_msg=k, **BWD)
    model.fused = True
    with pytest.raises(Exception, match='Hard negative examples'):
        it1 = get_queries_iterator(train_queries['2-chain'], 64, model)
        model.margin_loss(*next(it1), hard_negatives=True)
    with torch.no_grad():
        random.seed(77)
        v = model.margin_loss(*batch, hard_negatives=True, margin=0.7)
    assert not v.requires_grad
    np.testing.assert_allclose(0.3 * v.item(), out[True][0], rtol=1e-6)
    # queries given without the collated ids (the reference's run_batch form, train_helpers.py:143-154)
    formula, queries = batch[0], batch[1]
    random.seed(77)
    v2 = model.margin_loss(formula, queries, hard_negatives=True, margin=0.7)
    np.testing.assert_allclose(v2.item(), v.item(), rtol=1e-6)
    # a bad entity id raises IndexError (a step late on the training path, at once on the evaluation path)
    with torch.no_grad(), pytest.raises(IndexError):
        model.forward(formula, queries, [schema.num_entities + 7] * len(queries))


def test_evaluation_loops_on_the_fused_forward():
    """eval_auc_queries / eval_perc_queries (reference utils.py:34-95) give the same numbers on the fused forward as on the
    module path."""
    from mpqe_amd.evaluation import eval_auc_queries, eval_perc_queries
    schema, node_maps, model, train_queries = _aifb('mp', True, D=128, per_formula=300)
    model = model.to('cuda:0').eval()
    test_queries = {}
    for qt in ('2-chain', '3-inter', '3-chain_inter'):
        test_queries.update(train_queries[qt])
    res = {}
    with torch.no_grad():
        for fused in (True, False):
            model.fused = fused
            auc, per = eval_auc_queries(test_queries, model, hard_negatives=False)
            perc = eval_perc_queries(test_queries, model)
            res[fused] = (auc, perc)
    # (scores differ in the last bits between the paths; a near-tie between two DIFFERENT entities may fall the other way --
    # one such flip moves the mean percentile by ~3e-5 of its value; exact ties, a negative that is the target, do not flip)
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=1e-5)
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=2e-4)


def test_retained_graphs_copies_and_two_models_in_one_pass():
    """Less common uses of the same calls: backward twice over a retained graph (the ids of the calls are still in their
    arena: every gradient doubles), a deep copy of a model that has already stepped (its fused-step state is rebuilt, the
    two train independently), and the losses of two models summed into one backward pass (each model's calls become its own
    fused step)."""
    import copy
    from mpqe_amd.data_utils import get_queries_iterator
    schema, node_maps, model, train_queries = _aifb('mp', True, D=64, per_formula=200)
    model = model.to('cuda:0')
    np.random.seed(1)
    it = get_queries_iterator(train_queries['3-chain_inter'], 128, model)
    batch = next(it)
    random.seed(5)
    loss = 2.0 * model.margin_loss(*batch)
    loss.backward(retain_graph=True)
    g1 = {k: _np(p.grad).copy() for k, p in model.named_parameters()}
    loss.backward()
    for k, p in model.named_parameters():
        np.testing.assert_allclose(_np(p.grad), 2.0 * g1[k], err_msg=k, **BWD)
    assert model.dropin().steps == 2
    twin = copy.deepcopy(model)
    assert twin.dropin() is not model.dropin() and twin.dropin().steps == 0
    for m in (model, twin):
        for p in m.parameters():
            p.grad = None
    random.seed(5)
    la = model.margin_loss(*batch)
    random.seed(5)
    lb = twin.margin_loss(*batch)
    np.testing.assert_allclose(la.item(), lb.item(), rtol=1e-6)
    (la + 3.0 * lb).backward()                       # one pass, two models
    assert model.dropin().steps == 3 and twin.dropin().steps == 1
    for (k, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        np.testing.assert_allclose(_np(p.grad), 0.5 * g1[k], err_msg=k, **BWD)
        np.testing.assert_allclose(_np(q.grad), 1.5 * g1[k], err_msg=k, **BWD)


@pytest.mark.parametrize('readout,adaptive,wd', [('mp', True, 0.0), ('mlp', False, 1e-3)])
def test_forward_only_calls_on_side_streams_change_nothing(readout, adaptive, wd, monkeypatch):
    """DropIn.set_lanes(n): the forward-only margin_loss calls of a pass on n side streams (their own workspaces, packed
    steps, notification words and XCDs; the caller's stream waits for each before its value is used; the lanes wait for
    the caller's stream once per pass and when a parameter's version moves). Three iterations of the reference loop body
    with an optimiser step in between -- torch.optim.Adam (version counters) and FlatOptimizer (its own epoch) --: every
    loss value and every parameter after the last step equal the one-stream run's bit for bit; so does a no_grad loop that
    drops every value after adding it (the allocator must not hand a value's block to another lane too early)."""
    from mpqe_amd import dropin as dropin_mod
    from mpqe_amd.data_utils import get_queries_iterator
    from mpqe_amd.optim import FlatOptimizer
    out = {}
    for opt_kind in ('torch', 'flat'):
        # (second round: calls of more than LANE_MAX_GRAPHS graphs -- here every other batch size -- stay on the caller's stream)
        monkeypatch.setattr(dropin_mod, 'LANE_MAX_GRAPHS', 1024 if opt_kind == 'torch' else 200)
        for lanes in (0, 3):
            schema, node_maps, model, train_queries = _aifb(readout, adaptive, D=64, per_formula=600, weight_decay=wd)
            model = model.to('cuda:0')
            d = model.dropin()
            d.set_lanes(lanes)
            np.random.seed(5)
            iterators = {qt: get_queries_iterator(train_queries[qt], 256, model) for qt in train_queries}
            opt = (torch.optim.Adam(model.parameters(), lr=0.01) if opt_kind == 'torch'
                   else FlatOptimizer(d.step, lr=0.01, opt='adam'))
            random.seed(123)
            values = []
            for it in range(3):
                opt.zero_grad()
                loss = _reference_loop_body(model, iterators, train_queries, [])
                values.append(loss.item())
                loss.backward()
                opt.step()
            with torch.no_grad():
                tot = torch.zeros((), device='cuda:0')
                for k in range(40):
                    qt = ('3-inter', '2-chain', '3-chain_inter', '1-chain')[k % 4]
                    batch = next(iterators[qt])
                    if k % 2:           # (a shorter batch: the same queries without their collated ids)
                        batch = (batch[0], batch[1][:150])
                    tot = tot + model.margin_loss(*batch)
            torch.cuda.synchronize()
            d._check_mirror()
            assert len(d.lanes) >= lanes and (lanes == 0 or all(l.pass_id >= 0 for l in d.lanes[:lanes]))
            out[opt_kind, lanes] = (values, tot.item(), {k: _np(p).copy() for k, p in model.named_parameters()})
        a, b = out[opt_kind, 0], out[opt_kind, 3]
        assert a[0] == b[0] and a[1] == b[1], (a[0], b[0], a[1], b[1])
        for k in a[2]:
            np.testing.assert_array_equal(a[2][k], b[2][k], err_msg='%s (%s)' % (k, opt_kind))


@pytest.mark.parametrize('lanes', [0, 2])
def test_bad_entity_id_on_the_training_path_raises_at_the_next_call(lanes):
    """The reference raises IndexError inside forward (encoders.py:40-43: the embedding lookup). The drop-in's margin_loss does
    not wait for its launch: the call's last workgroup leaves the error word in pinned host memory
    (mpqe_step_extra_t.notify; a pair of words per side stream), and the NEXT margin_loss call -- or _check_mirror() -- raises.
    The model keeps working afterwards."""
    from mpqe_amd.data_utils import get_queries_iterator
    schema, node_maps, model, train_queries = _aifb('mp', True, D=64, per_formula=300)
    model = model.to('cuda:0')
    d = model.dropin()
    d.set_lanes(lanes)
    np.random.seed(1)
    it = get_queries_iterator(train_queries['2-inter'], 128, model)
    good = next(it)
    random.seed(5)
    v0 = model.margin_loss(*good).item()
    formula, queries = good[0], list(good[1])
    bad_anchors = good[2].clone()
    bad_anchors[7, 1] = schema.num_entities + 11                  # (beyond the id -> row table)
    with torch.no_grad():
        model.margin_loss(formula, queries, bad_anchors)        # queued, not waited for
        torch.cuda.synchronize()
        with pytest.raises(IndexError):
            model.margin_loss(*good)
        random.seed(5)
        v1 = model.margin_loss(*good).item()
    d._check_mirror()
    assert v1 == v0


def test_python_autograd_node_fallback_gives_the_same(monkeypatch):
    """Without csrc/host/autograd_node.cpp built (mpqe_amd/_lib.py: load_autograd_node() -> None) the calls carry a
    torch.autograd.Function instead of the C++ node: same loss, same gradients (and no side streams: their loss-word pool asks
    the extension who holds a word)."""
    from mpqe_amd import _lib
    from mpqe_amd.data_utils import get_queries_iterator
    out = {}
    for impl in ('c++', 'python'):
        if impl == 'python':
            monkeypatch.setattr(_lib, 'load_autograd_node', lambda: None)
        schema, node_maps, model, train_queries = _aifb('mlp', False, D=64, per_formula=400, weight_decay=1e-3)
        model = model.to('cuda:0')
        d = model.dropin()
        assert d.node_impl == impl
        np.random.seed(2)
        iterators = {qt: get_queries_iterator(train_queries[qt], 200, model) for qt in train_queries}
        random.seed(31)
        loss = _reference_loop_body(model, iterators, train_queries, [])
        loss.backward()
        assert (len(d.lanes) > 0) == (impl == 'c++')
        out[impl] = (loss.item(), {k: _np(p.grad).copy() for k, p in model.named_parameters()})
    assert out['c++'][0] == out['python'][0]
    for k in out['c++'][1]:
        np.testing.assert_array_equal(out['c++'][1][k], out['python'][1][k], err_msg=k)


def test_torch_optim_shaped_constructors_switch_to_the_flat_update():
    """`from mpqe_amd import optim` in place of `from torch import optim` (reference train.py:83-88): optim.Adam / optim.SGD
    over exactly one fused model's parameters are FlatOptimizer's one launch (zero_grad a flag); anything else is the torch
    optimiser. Three iterations of the reference loop give the parameters torch.optim.Adam gives on the same model."""
    from mpqe_amd import optim
    from mpqe_amd.data_utils import get_queries_iterator
    out = {}
    for kind in ('ours', 'torch'):
        schema, node_maps, model, train_queries = _aifb('mp', True, D=64, per_formula=400)
        model = model.to('cuda:0')
        params = [p for p in model.parameters() if p.requires_grad]
        opt = optim.Adam(params, lr=0.01) if kind == 'ours' else torch.optim.Adam(params, lr=0.01)
        if kind == 'ours':
            assert opt.flat and opt.param_groups[0]['lr'] == 0.01
        np.random.seed(4)
        iterators = {qt: get_queries_iterator(train_queries[qt], 128, model) for qt in train_queries}
        random.seed(8)
        for it in range(3):
            opt.zero_grad()
            loss = _reference_loop_body(model, iterators, train_queries, [])
            loss.backward()
            opt.step()
        out[kind] = {k: _np(p).copy() for k, p in model.named_parameters()}
    for k in out['ours']:
        np.testing.assert_allclose(out['ours'][k], out['torch'][k], rtol=2e-5, atol=2e-7, err_msg=k)
    # not one model's parameters / a model on the module path / options the flat update has not: the torch optimiser
    schema, node_maps, model, train_queries = _aifb('mp', True, D=64, per_formula=50)
    model = model.to('cuda:0')
    params = [p for p in model.parameters() if p.requires_grad]
    assert not optim.Adam(params[:-1], lr=0.01).flat
    assert not optim.Adam(params, lr=0.01, amsgrad=True).flat
    assert not optim.SGD(params, lr=0.1, momentum=0.9).flat
    assert optim.SGD(params, lr=0.1, momentum=0).flat
    model.fused = False
    assert not optim.Adam(params, lr=0.01).flat


@pytest.mark.parametrize('readout,adaptive,wd', [('mp', True, 0.0), ('mlp', False, 1e-3), ('max', False, 0.0)])
def test_ragged_batch_sizes_fused_against_module_path(readout, adaptive, wd):
    """Every query type at batch sizes around the 16-graph block (1, 15, 16, 17, 33, 129), with and without hard negatives,
    margins 1 and 0.3: the drop-in's loss and every parameter gradient against the per-op module path on the same draws."""
    schema, node_maps, model, train_queries = _aifb(readout, adaptive, D=64, n_formulas=1, per_formula=140, weight_decay=wd)
    model = model.to('cuda:0')
    cases = []
    for qt, by_formula in train_queries.items():
        formula, queries = next(iter(by_formula.items()))
        for B in (1, 15, 16, 17, 33, 129):
            cases.append((formula, queries[:B], 'inter' in qt and B % 2 == 1, 1 if B != 17 else 0.3))
    out = {}
    for fused in (True, False):
        model.fused = fused
        res = []
        random.seed(11)
        for formula, queries, hard, margin in cases:
            for p in model.parameters():
                p.grad = None
            loss = model.margin_loss(formula, queries, hard_negatives=hard, margin=margin)
            loss.backward()
            res.append((loss.item(), {k: (np.zeros(tuple(p.shape), np.float32) if p.grad is None else _np(p.grad).copy())
                                      for k, p in model.named_parameters()}))
        out[fused] = res
    if model.dropin() is not None:
        model.dropin()._check_mirror()
    for i, ((lf, gf), (lm, gm)) in enumerate(zip(out[True], out[False])):
        np.testing.assert_allclose(lf, lm, rtol=1e-5, atol=1e-6, err_msg='case %d' % i)
        for k in gf:
            np.testing.assert_allclose(gf[k], gm[k], err_msg='case %d (%s, B=%d) %s' % (i, cases[i][0].query_type, len(cases[i][1]), k),
                                       **BWD)


@pytest.mark.parametrize('readout,adaptive,wd', [('mp', True, 0.0), ('mlp', False, 1e-3)])
def test_training_inside_a_callers_stream(readout, adaptive, wd):
    """The caller's CURRENT stream is where the calls, the pass' fused step and the lanes' joins go: two iterations of the loop
    inside `with torch.cuda.stream(s)` give the default stream's losses and parameters bit for bit."""
    from mpqe_amd.data_utils import get_queries_iterator
    from mpqe_amd.optim import FlatOptimizer
    out = {}
    for where in ('default', 'side'):
        schema, node_maps, model, train_queries = _aifb(readout, adaptive, D=64, per_formula=300, weight_decay=wd)
        model = model.to('cuda:0')
        torch.cuda.synchronize()
        s = torch.cuda.Stream() if where == 'side' else torch.cuda.current_stream()
        with torch.cuda.stream(s):
            opt = FlatOptimizer(model.dropin().step, lr=0.01, opt='adam')
            np.random.seed(6)
            iterators = {qt: get_queries_iterator(train_queries[qt], 128, model) for qt in train_queries}
            random.seed(21)
            values = []
            for it in range(2):
                opt.zero_grad()
                loss = _reference_loop_body(model, iterators, train_queries, [])
                values.append(loss.item())
                loss.backward()
                opt.step()
            s.synchronize()
        torch.cuda.synchronize()
        model.dropin()._check_mirror()
        out[where] = (values, {k: _np(p).copy() for k, p in model.named_parameters()})
    assert out['default'][0] == out['side'][0]
    for k in out['default'][1]:
        np.testing.assert_array_equal(out['default'][1][k], out['side'][1][k], err_msg=k)

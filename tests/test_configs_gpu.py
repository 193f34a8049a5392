"""The other BASELINE.json configurations at their full shapes on one GPU. Where the CPU oracle would
take minutes, parity is established through size-independent properties of the path (plus the oracle
on a sub-batch at full embedding width and full KG size):
  * splitting: scores of a batch of B graphs == scores of its sub-batches, bit for bit (query graphs
    never interact; reference data_utils.py:405 builds a block-diagonal batch)
  * permutation equivariance over the graphs of a batch
  * gradient additivity: d(sum of sub-batch losses) == d(whole-batch loss) for every parameter
  * the fused step == the module path
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def build(kg, D, readout, adaptive, seed=0, num_layers=3):
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.model import RGCNEncoderDecoder
    torch.manual_seed(seed)
    schema = synthetic.make_schema(*synthetic.KG_SHAPES[kg], seed=seed)
    graph = synthetic.SchemaGraph(schema, D)
    fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
    model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=readout, num_layers=num_layers,
                               shared_layers=False, adaptive=adaptive, weight_decay=0)
    with torch.no_grad():
        for p in model.layers.parameters():
            p.mul_(4.0)
    return schema, node_maps, model


def draw(schema, qt, B, rng):
    from mpqe_amd import synthetic
    f = synthetic.sample_formula(schema, qt, rng)
    anchors = np.stack([synthetic._pick(schema, m, rng, size=B) for m in f.anchor_modes], axis=1)
    tg = synthetic._pick(schema, f.target_mode, rng, size=B)
    ng = synthetic._pick(schema, f.target_mode, rng, size=B)
    return dict(formula=f, anchor_ids=anchors, targets=tg, negs=ng, weight=1.0)


def sub(b, idx):
    return dict(formula=b['formula'], anchor_ids=b['anchor_ids'][idx], targets=b['targets'][idx],
                negs=b['negs'][idx], weight=b['weight'])


def close_enough(got, ref, readout, name):
    """rtol 1e-4 / atol 2e-6. With the max readout the two paths sum K in different orders, so a near-tie
    between two node states can pick the other node in one of them (a discrete, equally valid argmax): the
    gradient of those few (graph, column) entries is routed to another node slot, which moves every weight
    gradient a little. Then: norm-wise error below 5e-3 and no element off by more than 1e-3."""
    if readout != 'max':
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-6, err_msg=name)
        return
    denom = max(float(np.linalg.norm(ref)), 1e-12)
    assert float(np.linalg.norm(got - ref)) / denom < 5e-3, (name, float(np.linalg.norm(got - ref)) / denom)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-3, err_msg=name)


def grads(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize('kg,D,readout,adaptive', [('aifb', 128, 'mp', True),        # configs[1], the benchmarked one
                                                   ('mutag', 256, 'sum', False),      # configs[2]
                                                   ('am', 128, 'max', False)])        # configs[3]
def test_full_mix_fused_equals_modules_and_oracle(kg, D, readout, adaptive):
    from mpqe_amd import ops, synthetic
    from mpqe_amd.fused import FusedTrainStep
    from oracle import ref_cpu
    schema, node_maps, model = build(kg, D, readout, adaptive)
    cpu_params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to('cuda:0')
    rng = np.random.RandomState(3)
    batches = [draw(schema, qt, 512, rng) for qt, _ in synthetic.FULL_MIX]
    step = FusedTrainStep(model)
    loss, sp, sn = step.run(step.pack(batches), scores=True)
    step.check()
    g_fused = grads(model)
    # module path on the same batches
    for p in model.parameters():
        p.grad = None
    total, off = None, 0
    for b in batches:
        queries = [type('Q', (), {'anchor_nodes': tuple(int(v) for v in row)})() for row in b['anchor_ids']]
        out = model.encode(b['formula'], queries)
        pos = model.score(b['formula'], out, torch.from_numpy(b['targets']).cuda())
        neg = model.score(b['formula'], out, torch.from_numpy(b['negs']).cuda())
        np.testing.assert_allclose(sp[off:off + 512].cpu().numpy(), pos.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sn[off:off + 512].cpu().numpy(), neg.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
        off += 512
        l = ops.hinge(pos, neg, 1.0)
        total = l if total is None else total + l
    total.backward()
    np.testing.assert_allclose(loss[0].item(), total.item(), rtol=1e-5, atol=1e-6)
    for k, p in model.named_parameters():
        ref = torch.zeros_like(p) if p.grad is None else p.grad
        close_enough(g_fused[k].cpu().numpy(), ref.cpu().numpy(), readout, k)
    # the oracle (reference op sequence) on a sub-batch of one formula, full width, full KG
    b = sub(batches[5], np.arange(96))
    cfg = dict(readout=readout, scatter_op='add', num_layers=3, adaptive=adaptive, weight_decay=0)
    queries = [type('Q', (), {'anchor_nodes': tuple(int(v) for v in row)})() for row in b['anchor_ids']]
    col = ref_cpu.collate(b['formula'], queries, model.rel_ids, model.mode_ids)
    ref_loss = ref_cpu.margin_loss(cpu_params, cfg, node_maps, b['formula'], col, b['targets'], b['negs'])
    l2 = step.run(step.pack([b]))
    np.testing.assert_allclose(l2[0].item(), ref_loss.item(), rtol=1e-5, atol=1e-6)
    ref_loss.backward()
    for k, p in model.named_parameters():
        ref = cpu_params[k].grad
        ref = torch.zeros_like(cpu_params[k]) if ref is None else ref
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


def test_stress_shape_properties():
    """configs[4]: 1M entities / 8 modes / 64 relation names, D = 256, B = 8192, 3-chain + 3-inter."""
    from mpqe_amd.fused import FusedTrainStep
    schema, node_maps, model = build('stress', 256, 'sum', False)
    model = model.to('cuda:0')
    rng = np.random.RandomState(11)
    B = 8192
    big = [draw(schema, '3-chain', B, rng), draw(schema, '3-inter', B, rng)]
    step = FusedTrainStep(model)
    loss, sp, sn = step.run(step.pack(big), scores=True)
    step.check()
    g_big = grads(model)
    assert torch.isfinite(sp).all() and torch.isfinite(sn).all() and sp.abs().max() <= 1.0001
    # splitting: four quarter batches give the same scores bit for bit, and their summed gradient
    # (each loss is a mean over its own B/4 graphs -> weight 1/4) is the whole batch's gradient
    acc = None
    for q in range(4):
        idx = np.arange(q * B // 4, (q + 1) * B // 4)
        parts = [dict(sub(b, idx), weight=0.25) for b in big]
        l, p1, n1 = step.run(step.pack(parts), scores=True)
        for j in range(2):
            lo = j * B + q * B // 4
            assert torch.equal(p1[j * B // 4:(j + 1) * B // 4], sp[lo:lo + B // 4])
            assert torch.equal(n1[j * B // 4:(j + 1) * B // 4], sn[lo:lo + B // 4])
        g = grads(model)
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    for k in g_big:
        np.testing.assert_allclose(acc[k].cpu().numpy(), g_big[k].cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    # permutation equivariance
    perm = rng.permutation(B)
    shuffled = [sub(b, perm) for b in big]
    _, p2, n2 = step.run(step.pack(shuffled), scores=True)
    for j in range(2):
        assert torch.equal(p2[j * B:(j + 1) * B], sp[j * B:(j + 1) * B][torch.from_numpy(perm).cuda()])


BENCH_WEIGHTS = {'1-chain': 1.0, 'chain': 0.01, 'inter': 0.005}      # reference train_helpers.py:60-61, 97-112


def _queries(b):
    return [type('Q', (), {'anchor_nodes': tuple(int(v) for v in row)})() for row in b['anchor_ids']]


def drop_near_ties(batches, cpu_params, cfg, node_maps, model, tol=1e-6):
    """max readout: the reference takes, per graph and column, the largest of the N final node states (model.py:384);
    where the two largest are within `tol` of each other the fp32 summation order decides which node wins -- a discrete,
    equally valid choice that routes that column's gradient to another node. Those graphs (a handful per batch) are
    taken out of BOTH sides, so that everything left is compared at the plain tolerances."""
    from oracle import ref_cpu
    out, dropped = [], 0
    with torch.no_grad():
        for b in batches:
            col = ref_cpu.collate(b['formula'], _queries(b), model.rel_ids, model.mode_ids)
            keep = {}
            ref_cpu.encode_queries(cpu_params, cfg, node_maps, b['formula'], col, keep=keep)
            h = keep['layers'][-1].reshape(col['B'], col['N'], -1)
            top2 = torch.topk(h, 2, dim=1).values
            ok = ((top2[:, 0] - top2[:, 1]) > tol).all(dim=1).numpy()
            dropped += int((~ok).sum())
            out.append(sub(b, np.nonzero(ok)[0]))
    return out, dropped


@pytest.mark.parametrize('kg,D,readout,adaptive,flags', [
    ('aifb', 128, 'mp', True, 'default'), ('aifb', 128, 'mp', True, 'no_prune'), ('aifb', 128, 'mp', True, 'no_uniform'),
    ('aifb', 128, 'mp', True, 'no_chain'),            # configs[1], the benchmarked one, with the speed switches on and off
    ('aifb', 128, 'mp', True, 'device_ids'),          # ... and packed the way bench.py's timed loop packs: pack(descs, ids=<CUDA tensor>)
    ('mutag', 256, 'sum', False, 'default'),          # configs[2]
    ('am', 128, 'max', False, 'default'),             # configs[3]
    # the learned readouts at the benchmarked shape, the readout inside the call: on the chain form (their Linear layers as two
    # more levels of every block's programme: csrc/step_chain.h) and on the level form (csrc/step_readout.h)
    ('aifb', 128, 'mlp', True, 'default'), ('aifb', 128, 'targetmlp', False, 'default'), ('aifb', 128, 'concat', False, 'default'),
    ('aifb', 128, 'concat', False, 'no_chain')])
def test_benchmarked_workload_against_oracle(kg, D, readout, adaptive, flags):
    """BASELINE.json configs[1] exactly as bench.py times it -- AIFB-shaped KG, the 11-batch post-burn-in mix at
    B = 512, D = 128, readout mp (TM), adaptive, num_layers 3 unshared, the reference's loss weights -- through the
    fused step (the chain kernel with liveness pruning, XCD placement with holes, paired blocks; ids fresh, the touch
    plan built inside the step) against the CPU oracle in the reference's op sequence on ALL 11 batches: every score
    (rtol 1e-5), the loss, every parameter gradient (rtol 1e-4). And the same for configs[2] (MUTAG-shaped, D = 256,
    sum, 3 layers) and configs[3] (AM-shaped: 372 584 entities, D = 128, max) at their full shapes."""
    from mpqe_amd import synthetic
    from mpqe_amd.fused import FusedTrainStep
    from oracle import ref_cpu
    schema, node_maps, model = build(kg, D, readout, adaptive)
    cpu_params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to('cuda:0')
    rng = np.random.RandomState(1000)
    batches = []
    for qt, _ in synthetic.FULL_MIX:
        b = draw(schema, qt, 512, rng)
        b['weight'] = 1.0 if qt == '1-chain' else (BENCH_WEIGHTS['inter'] if 'inter' in qt else BENCH_WEIGHTS['chain'])
        batches.append(b)
    cfg = dict(readout=readout, scatter_op='add', num_layers=3, adaptive=adaptive, weight_decay=0)
    torch.set_num_threads(min(16, max(1, len(__import__('os').sched_getaffinity(0)))))
    if readout == 'max':
        batches, dropped = drop_near_ties(batches, cpu_params, cfg, node_maps, model)
        assert dropped < 0.05 * 11 * 512, dropped       # (a handful of graphs, not a loophole)
    kw = dict(default={}, no_prune=dict(prune=False), no_uniform=dict(uniform=False), no_chain=dict(chain=False),
              device_ids={})[flags]
    step = FusedTrainStep(model, **kw)
    if flags == 'device_ids':
        # the form the bench times: descriptors (formula, weight, size) + ONE device-resident id tensor in the library's
        # layout; pack() does no device work, the id -> row lookups and the touch plan happen inside the step
        ids = torch.from_numpy(step.flatten_ids(batches)).to('cuda:0')
        descs = [dict(formula=b['formula'], weight=b['weight'], batch_size=len(b['targets'])) for b in batches]
        packed = step.pack(descs, ids=ids)
        assert packed.touch_mode == 'step' and packed.ids_ref is ids
    else:
        packed = step.pack(batches)
    assert step.uses_chain(packed) == (flags != 'no_chain')       # (the learned readouts ride on the chain form too)
    loss, sp, sn = step.run(packed, scores=True)
    step.check()
    sp, sn, loss = sp.cpu().numpy(), sn.cpu().numpy(), loss.cpu().numpy()
    starts = np.concatenate([[0], np.cumsum([len(batches[i]['targets']) for i in packed.order])])
    total = 0
    for i, b in enumerate(batches):
        k = packed.order.index(i)                   # scores come back in library batch order
        off, n = int(starts[k]), len(b['targets'])
        col = ref_cpu.collate(b['formula'], _queries(b), model.rel_ids, model.mode_ids)
        q = ref_cpu.encode_queries(cpu_params, cfg, node_maps, b['formula'], col)
        pos = ref_cpu.score(cpu_params, node_maps, b['formula'], q, b['targets'])
        neg = ref_cpu.score(cpu_params, node_maps, b['formula'], q, b['negs'])
        np.testing.assert_allclose(sp[off:off + n], pos.detach().numpy(), rtol=1e-5, atol=1e-6, err_msg='batch %d' % i)
        np.testing.assert_allclose(sn[off:off + n], neg.detach().numpy(), rtol=1e-5, atol=1e-6, err_msg='batch %d' % i)
        l = torch.clamp(1.0 - (pos - neg), min=0).mean()
        np.testing.assert_allclose(loss[1 + k], l.item(), rtol=1e-5, atol=1e-6)
        total = total + b['weight'] * l
    np.testing.assert_allclose(loss[0], total.item(), rtol=1e-5, atol=1e-6)
    total.backward()
    for k, p in model.named_parameters():
        ref = cpu_params[k].grad
        ref = torch.zeros_like(cpu_params[k]) if ref is None else ref
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('qt', ['3-chain', '3-inter'])
def test_stress_shape_against_oracle(qt):
    """configs[4] (1M entities / 8 modes / 64 relation names, D = 256, sum, 3 layers): the oracle in the reference's op
    sequence on a 96-graph batch of each of its two query types, on the full-size KG -- scores, loss, every gradient
    (the entity tables' 1M x 256 included). The properties test above covers B = 8192."""
    from mpqe_amd.fused import FusedTrainStep
    from oracle import ref_cpu
    schema, node_maps, model = build('stress', 256, 'sum', False)
    cpu_params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to('cuda:0')
    rng = np.random.RandomState(5)
    b = draw(schema, qt, 96, rng)
    b['anchor_ids'][:8] = b['anchor_ids'][8:16]           # entities that occur in several graphs
    cfg = dict(readout='sum', scatter_op='add', num_layers=3, adaptive=False, weight_decay=0)
    step = FusedTrainStep(model)
    loss, sp, sn = step.run(step.pack([b]), scores=True)
    step.check()
    col = ref_cpu.collate(b['formula'], _queries(b), model.rel_ids, model.mode_ids)
    q = ref_cpu.encode_queries(cpu_params, cfg, node_maps, b['formula'], col)
    pos = ref_cpu.score(cpu_params, node_maps, b['formula'], q, b['targets'])
    neg = ref_cpu.score(cpu_params, node_maps, b['formula'], q, b['negs'])
    np.testing.assert_allclose(sp.cpu().numpy(), pos.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sn.cpu().numpy(), neg.detach().numpy(), rtol=1e-5, atol=1e-6)
    ref_loss = torch.clamp(1.0 - (pos - neg), min=0).mean()
    np.testing.assert_allclose(loss[0].item(), ref_loss.item(), rtol=1e-5, atol=1e-6)
    ref_loss.backward()
    for k, p in model.named_parameters():
        ref = cpu_params[k].grad
        if ref is None:
            assert float(p.grad.abs().max()) == 0.0, k
            continue
        g = p.grad
        if ref.shape[0] > 100000:                   # an entity table: compare the touched rows, the rest must be exactly zero
            rows = torch.nonzero(ref.abs().sum(1) > 0).flatten()
            np.testing.assert_allclose(g[rows.cuda()].cpu().numpy(), ref[rows].numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
            assert int((g.abs().sum(1) > 0).sum().item()) <= len(rows), k
        else:
            np.testing.assert_allclose(g.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


def test_config0_literal_against_oracle():
    """BASELINE.json configs[0] literally: 1-chain queries, embed_dim 128, readout sum, batch 512, the CLI's default
    num_layers = 2 (reference train.py:23-33), not adaptive -- fused step against the oracle in the reference's op
    sequence: scores, loss, every gradient."""
    from mpqe_amd.fused import FusedTrainStep
    from oracle import ref_cpu
    schema, node_maps, model = build('aifb', 128, 'sum', False, num_layers=2)
    assert len(model.layers) == 2
    cpu_params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to('cuda:0')
    b = draw(schema, '1-chain', 512, np.random.RandomState(21))
    cfg = dict(readout='sum', scatter_op='add', num_layers=2, adaptive=False, weight_decay=0)
    step = FusedTrainStep(model)
    packed = step.pack([b])
    assert step.uses_chain(packed) and int(packed.batches[0].num_passes) == 2
    loss, sp, sn = step.run(packed, scores=True)
    step.check()
    col = ref_cpu.collate(b['formula'], _queries(b), model.rel_ids, model.mode_ids)
    q = ref_cpu.encode_queries(cpu_params, cfg, node_maps, b['formula'], col)
    pos = ref_cpu.score(cpu_params, node_maps, b['formula'], q, b['targets'])
    neg = ref_cpu.score(cpu_params, node_maps, b['formula'], q, b['negs'])
    np.testing.assert_allclose(sp.cpu().numpy(), pos.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sn.cpu().numpy(), neg.detach().numpy(), rtol=1e-5, atol=1e-6)
    ref_loss = torch.clamp(1.0 - (pos - neg), min=0).mean()
    np.testing.assert_allclose(loss[0].item(), ref_loss.item(), rtol=1e-5, atol=1e-6)
    ref_loss.backward()
    for k, p in model.named_parameters():
        ref = cpu_params[k].grad
        ref = torch.zeros_like(cpu_params[k]) if ref is None else ref
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('flags', ['default', 'no_chain'])
def test_max_readout_exact_ties_at_full_shape(flags):
    """max readout with CONSTRUCTED exact ties at B = 512, D = 128: both anchors of every 2-inter / 3-inter graph are the
    SAME entity, so their node states are identical in every column at every level (an anchor's state depends on its own
    row, root and bias only) and the readout's max is attained twice wherever an anchor wins. torch_scatter routes the
    gradient to ONE of them (reference model.py:384: scatter_max's argmax); the build's statement is the lowest node row.
    The oracle's scatter_max has that rule in its backward; graphs with a NEAR tie between different values (the
    summation order decides there) are taken out of both sides as in the test above -- exact ties stay in."""
    from mpqe_amd import synthetic
    from mpqe_amd.fused import FusedTrainStep
    from oracle import ref_cpu
    schema, node_maps, model = build('aifb', 128, 'max', False)
    cpu_params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to('cuda:0')
    cfg = dict(readout='max', scatter_op='add', num_layers=3, adaptive=False, weight_decay=0)
    rng = np.random.RandomState(77)
    batches = []
    for qt in ('2-inter', '3-inter'):
        for _ in range(200):
            f = synthetic.sample_formula(schema, qt, rng)
            if f.anchor_modes[0] == f.anchor_modes[1]:
                break
        assert f.anchor_modes[0] == f.anchor_modes[1], 'no %s formula with two anchors of one mode' % qt
        anchors = np.stack([synthetic._pick(schema, m, rng, size=512) for m in f.anchor_modes], axis=1)
        anchors[:, 1] = anchors[:, 0]
        batches.append(dict(formula=f, anchor_ids=anchors, targets=synthetic._pick(schema, f.target_mode, rng, size=512),
                            negs=synthetic._pick(schema, f.target_mode, rng, size=512), weight=1.0))
    torch.set_num_threads(min(16, max(1, len(__import__('os').sched_getaffinity(0)))))
    kept, ties = [], 0
    with torch.no_grad():
        for b in batches:
            col = ref_cpu.collate(b['formula'], _queries(b), model.rel_ids, model.mode_ids)
            keep = {}
            ref_cpu.encode_queries(cpu_params, cfg, node_maps, b['formula'], col, keep=keep)
            h = keep['layers'][-1].reshape(col['B'], col['N'], -1)
            assert torch.equal(h[:, 0], h[:, 1])                       # the constructed ties are exact in the oracle
            top2 = torch.topk(h, 2, dim=1).values
            gap = top2[:, 0] - top2[:, 1]
            near = ((gap > 0) & (gap <= 1e-6)).any(dim=1).numpy()      # different values, too close: dropped
            ties += int(((gap == 0).any(dim=1).numpy() & ~near).sum())
            kept.append(sub(b, np.nonzero(~near)[0]))
    assert ties > 900, ties                                            # (nearly every graph has a tied column)
    step = FusedTrainStep(model, chain=flags != 'no_chain')
    packed = step.pack(kept)
    loss, sp, sn = step.run(packed, scores=True)
    step.check()
    sp, sn, loss = sp.cpu().numpy(), sn.cpu().numpy(), loss.cpu().numpy()
    total, off = 0, 0
    assert packed.order == list(range(len(kept)))
    for i, b in enumerate(kept):
        n = len(b['targets'])
        col = ref_cpu.collate(b['formula'], _queries(b), model.rel_ids, model.mode_ids)
        q = ref_cpu.encode_queries(cpu_params, cfg, node_maps, b['formula'], col)
        pos = ref_cpu.score(cpu_params, node_maps, b['formula'], q, b['targets'])
        neg = ref_cpu.score(cpu_params, node_maps, b['formula'], q, b['negs'])
        np.testing.assert_allclose(sp[off:off + n], pos.detach().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sn[off:off + n], neg.detach().numpy(), rtol=1e-5, atol=1e-6)
        total = total + torch.clamp(1.0 - (pos - neg), min=0).mean()
        off += n
    np.testing.assert_allclose(loss[0], total.item(), rtol=1e-5, atol=1e-6)
    total.backward()
    for k, p in model.named_parameters():
        ref = cpu_params[k].grad
        ref = torch.zeros_like(cpu_params[k]) if ref is None else ref
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)

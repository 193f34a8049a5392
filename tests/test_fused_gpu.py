"""FusedTrainStep (one C-ABI call per training step) against the drop-in module path on the GPU:
same loss, same scores, same gradient for every parameter."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(readout, adaptive, shared, D=64, B=96, seed=0, weight_decay=0, scatter_op='add'):
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.model import RGCNEncoderDecoder
    torch.manual_seed(seed)
    schema = synthetic.make_schema(*synthetic.KG_SHAPES['tiny'], seed=seed)
    graph = synthetic.SchemaGraph(schema, D)
    fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
    model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=readout, num_layers=3,
                               shared_layers=shared, adaptive=adaptive, weight_decay=weight_decay,
                               scatter_op=scatter_op).to('cuda:0')
    with torch.no_grad():
        for p in model.layers.parameters():
            p.mul_(5.0)
    rng = np.random.RandomState(seed + 1)
    batches = []
    for qt, hard in synthetic.FULL_MIX:
        f = synthetic.sample_formula(schema, qt, rng)
        qs = synthetic.sample_queries(schema, f, B, rng)
        batches.append(dict(formula=f, queries=qs, weight=float(rng.uniform(0.1, 1.0)),
                            anchor_ids=np.array([q.anchor_nodes for q in qs], dtype=np.int64),
                            targets=np.array([q.target_node for q in qs], dtype=np.int64),
                            negs=np.array([q.neg_samples[0] for q in qs], dtype=np.int64)))
    return model, batches


@pytest.mark.parametrize('lanes', [1, 2, 3])
@pytest.mark.parametrize('readout,adaptive,shared', [('mp', True, False), ('sum', False, False),
                                                     ('max', False, True)])
def test_fused_step_equals_module_path(readout, adaptive, shared, lanes):
    from mpqe_amd import ops
    from mpqe_amd.fused import FusedTrainStep
    model, batches = _setup(readout, adaptive, shared)
    # module path: one autograd graph over all batches
    model.zero_grad(set_to_none=True)
    total, sp_ref, sn_ref = None, [], []
    for b in batches:
        out = model.encode(b['formula'], b['queries'])
        pos = model.score(b['formula'], out, b['targets'].tolist())
        neg = model.score(b['formula'], out, b['negs'].tolist())
        l = ops.hinge(pos, neg, 1.0) * b['weight']
        total = l if total is None else total + l
        sp_ref.append(pos.detach())
        sn_ref.append(neg.detach())
    total.backward()
    ref = {k: (torch.zeros_like(p) if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}
    # fused path
    step = FusedTrainStep(model, lanes=lanes)
    packed = step.pack(batches)
    assert sorted(packed.order) == list(range(len(batches)))
    assert (packed.lanes is None) == (lanes == 1 or step.uses_chain(packed))      # (the chain form runs on one stream)
    loss, sp, sn = step.run(packed, scores=True)
    step.check()
    np.testing.assert_allclose(loss[0].item(), total.item(), rtol=1e-5, atol=1e-6)
    # scores come back in library batch order: packed.order maps to the caller's batches
    np.testing.assert_allclose(sp.cpu().numpy(), torch.cat([sp_ref[i] for i in packed.order]).cpu().numpy(),
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sn.cpu().numpy(), torch.cat([sn_ref[i] for i in packed.order]).cpu().numpy(),
                               rtol=1e-5, atol=1e-6)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref[k].cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
    # running it again gives bit-identical gradients for the layer weights (fixed reduction order)
    g1 = {k: p.grad.clone() for k, p in model.named_parameters() if k.startswith('layers')}
    step.run(packed)
    for k, p in model.named_parameters():
        if k.startswith('layers'):
            assert torch.equal(p.grad, g1[k]), k


@pytest.mark.parametrize('readout,adaptive,shared,scatter_op,D,host_ids', [
    ('mlp', True, False, 'add', 64, 'direct'), ('mlp', False, True, 'max', 128, 'copy'),
    ('targetmlp', True, True, 'add', 128, 'direct'), ('targetmlp', False, False, 'mean', 64, 'copy'),
    ('concat', False, False, 'add', 64, 'direct'), ('concat', False, True, 'max', 128, 'direct')])
def test_fused_step_with_learned_readout_equals_module_path(readout, adaptive, shared, scatter_op, D, host_ids):
    """MLPReadout / TargetMLPReadout / concat (reference model.py:441-446, 497-553) inside the fused step's one library
    call (csrc/step_readout.h). Against the module path's margin_loss arithmetic, regulariser included (model.py:486-490)."""
    from mpqe_amd import ops
    from mpqe_amd.fused import FusedTrainStep
    wd = 1e-3
    model, batches = _setup(readout, adaptive, shared, D=D, weight_decay=wd, scatter_op=scatter_op)
    model.zero_grad(set_to_none=True)
    total, sp_ref, sn_ref, per = None, [], [], []
    for b in batches:
        out = model.encode(b['formula'], b['queries'])
        pos = model.score(b['formula'], out, b['targets'].tolist())
        neg = model.score(b['formula'], out, b['negs'].tolist())
        h = ops.hinge(pos, neg, 1.0)
        per.append(h.item())
        l = (h + wd * sum(torch.norm(p) for p in model.readout.parameters())) * b['weight']
        total = l if total is None else total + l
        sp_ref.append(pos.detach())
        sn_ref.append(neg.detach())
    total.backward()
    ref = {k: (torch.zeros_like(p) if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}
    step = FusedTrainStep(model, host_ids=host_ids)
    assert step.learned
    packed = step.pack(batches)
    with torch.no_grad():
        step.flat_grad.fill_(3.0)                # zero_grad covers the readout's parameters too
    loss, sp, sn = step.run(packed, scores=True)
    step.check()
    np.testing.assert_allclose(loss[0].item(), total.item(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss[1:].cpu().numpy(), np.array(per, np.float32), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sp.cpu().numpy(), torch.cat(sp_ref).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sn.cpu().numpy(), torch.cat(sn_ref).cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref[k].cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
    # accumulate on top (zero_grad=False): twice the gradient; forward only: the same loss, gradients untouched
    step.run(packed, zero_grad=False)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), 2 * ref[k].cpu().numpy(), rtol=1e-4, atol=4e-6, err_msg=k)
    keep = step.flat_grad.clone()
    l2 = step.run(packed, backward=False)
    np.testing.assert_allclose(l2[0].item(), total.item(), rtol=1e-5, atol=1e-6)
    assert torch.equal(step.flat_grad, keep)
    # fresh ids through the same descriptor set (pooled buffers, cached index plan)
    rng = np.random.RandomState(5)
    for b in batches:
        rng.shuffle(b['negs'])
    p3 = step.pack(batches)
    l3 = step.run(p3)
    step.check()
    assert np.isfinite(l3.cpu().numpy()).all() and abs(l3[0].item() - total.item()) > 0
    # the whole step (both library calls and the readout's autograd between them) replayed from a hipGraph
    g3 = step.flat_grad.clone()
    cap = step.capture(p3)
    step.flat_grad.fill_(-1.0)
    l4 = cap.replay()
    torch.cuda.synchronize()
    np.testing.assert_allclose(l4.cpu().numpy(), l3.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(step.flat_grad.cpu().numpy(), g3.cpu().numpy(), rtol=1e-4, atol=2e-6)


def test_fused_step_bad_id_is_reported():
    from mpqe_amd.fused import FusedTrainStep
    model, batches = _setup('mp', True, False, B=16)
    batches[3]['negs'][5] = 10 ** 6
    step = FusedTrainStep(model)
    step.run(step.pack(batches))
    with pytest.raises(IndexError):
        step.check()


@pytest.mark.parametrize('readout', ['mp', 'mlp'])
def test_checked_run_survives_a_late_producer(readout):
    """The chain form hands vectors and transposed weights from workgroup to workgroup inside a launch; every wait is
    bounded. A producer that arrives ~1 s late (forced: HANDOFF_LATE -- what a co-tenant on the GPU can do) makes its
    consumers give up: run(checked=True) must then deliver the step all the same -- through the level form, which has no
    in-launch hand-off -- and the next, undisturbed step must run on the chain form as if nothing had happened."""
    from mpqe_amd import ops
    from mpqe_amd.fused import FusedTrainStep
    model, batches = _setup(readout, True, False, D=64, B=96, weight_decay=0 if readout == 'mp' else 1e-3)
    step = FusedTrainStep(model)
    packed = step.pack(batches)
    assert step.uses_chain(packed)          # (mlp: the readout's Linear layers ride in the chain launch)
    loss0 = step.run(packed, checked=True).clone()
    good = {k: p.grad.clone() for k, p in model.named_parameters()}
    lib = ops.lib()
    lib.mpqe_debug_option(b'HANDOFF_LATE', 1, 1)
    try:
        for p in model.parameters():
            p.grad.fill_(5.0)
        step.run(packed)
        with pytest.raises(RuntimeError, match='hand-off'):
            step.check()                                   # unchecked: named a step later, never silently consumed
        for p in model.parameters():
            p.grad.fill_(5.0)
        loss = step.run(packed, checked=True)
        assert step.handoff_retries == 1
    finally:
        lib.mpqe_debug_option(b'HANDOFF_LATE', 0, 0)
    np.testing.assert_allclose(loss.cpu().numpy(), loss0.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), good[k].cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
    for p in model.parameters():
        p.grad.fill_(5.0)
    again = step.run(packed, checked=True)                 # the late workgroup has counted itself in meanwhile
    assert step.handoff_retries == 1
    assert torch.equal(again, loss0)
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, good[k]), k


@pytest.mark.parametrize('sparse', [False, True])
def test_checked_run_recovers_a_failed_in_step_touch_plan(sparse):
    """FusedTrainStep.run(checked=True): a step whose own touch plan could not be built (forced: TSORT_FAIL) is recovered
    before the call returns -- plan rebuilt with the library sort, entity-table rows summed again -- and its descriptor set
    takes pack-time plans from then on; without `checked` the failure is named at check(), never silently consumed."""
    from mpqe_amd import ops
    from mpqe_amd.fused import FusedTrainStep
    model, batches = _setup('mp', True, False, D=64, B=96)
    step = FusedTrainStep(model, sparse_tables=sparse)
    packed = step.pack(batches)
    assert packed.touch_mode == 'step'
    for p in model.parameters():
        p.grad.fill_(5.0)                    # (row-sparse tables: rows no id touches keep what they hold)
    step.run(packed)
    step.check()
    good = {k: p.grad.clone() for k, p in model.named_parameters()}
    lib = ops.lib()
    lib.mpqe_debug_option(b'TSORT_FAIL', 1, 1)
    try:
        for p in model.parameters():
            p.grad.fill_(5.0)
        step.run(packed)
        with pytest.raises(RuntimeError, match='touch plan'):
            step.check()
        for p in model.parameters():
            p.grad.fill_(5.0)
        loss = step.run(packed, checked=True)
        assert step.touch_retries == 1
        step.check()
    finally:
        lib.mpqe_debug_option(b'TSORT_FAIL', 0, 0)
    for k, p in model.named_parameters():
        if sparse and k.startswith('enc.'):
            touched = good[k].ne(5.0).any(dim=1) | p.grad.ne(5.0).any(dim=1)
            assert torch.equal(p.grad[touched], good[k][touched]), k
            assert bool((p.grad[~touched] == 5.0).all()), k
        else:
            assert torch.equal(p.grad, good[k]), k
    assert step.pack(batches).touch_mode == 'pack'           # this descriptor set no longer relies on the in-step sort
    other = step.pack(batches[:3])
    assert other.touch_mode == 'step'


@pytest.mark.parametrize('opt,readout', [('adam', 'mp'), ('sgd', 'mp'), ('sgd', 'targetmlp'), ('adam', 'mlp')])
def test_training_loop_fused_step_plus_flat_optimizer(opt, readout):
    """Three training steps: FusedTrainStep + FlatOptimizer (one launch over the flat parameter buffer)
    against the module path + torch.optim (the reference's loop, train_helpers.py:76-120, train.py:83-88)."""
    import copy
    from mpqe_amd import ops
    from mpqe_amd.fused import FusedTrainStep
    from mpqe_amd.optim import FlatOptimizer
    model, batches = _setup(readout, True, False, weight_decay=0 if readout == 'mp' else 1e-3)
    wd = model.weight_decay
    ref_model = copy.deepcopy(model)
    ref_opt = (torch.optim.Adam(ref_model.parameters(), lr=0.01) if opt == 'adam'
               else torch.optim.SGD(ref_model.parameters(), lr=0.01, momentum=0))
    step = FusedTrainStep(model)
    fopt = FlatOptimizer(step, lr=0.01, opt=opt)
    keys = list(model.state_dict().keys())
    packed = step.pack(batches)
    for it in range(3):
        ref_opt.zero_grad()
        total = None
        for b in batches:
            out = ref_model.encode(b['formula'], b['queries'])
            l = ops.hinge(ref_model.score(b['formula'], out, b['targets'].tolist()),
                          ref_model.score(b['formula'], out, b['negs'].tolist()), 1.0)
            if readout != 'mp':      # margin_loss's regulariser on the readout's parameters (model.py:486-490)
                l = l + wd * sum(torch.norm(p) for p in ref_model.readout.parameters())
            l = l * b['weight']
            total = l if total is None else total + l
        total.backward()
        ref_opt.step()
        loss = step.run(packed)
        fopt.step()
        np.testing.assert_allclose(loss[0].item(), total.item(), rtol=2e-5, atol=1e-6)
    assert list(model.state_dict().keys()) == keys            # re-homing the parameters changes no name
    for (k, p), (_, q) in zip(model.named_parameters(), ref_model.named_parameters()):
        # Adam divides by sqrt(v): where a gradient is ~1e-8 the two paths' last bits decide the step's sign
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-3,
                                   atol=2e-3 if opt == 'adam' else 1e-6, err_msg=k)
    assert model.layers[0].basis.data_ptr() >= fopt.flat_param.data_ptr()


def test_negative_sampler_on_device():
    """NegativeSampler: draws come from each query's own list, match the CPU stream, and can be written
    straight into a packed step's id buffer."""
    from mpqe_amd import synthetic
    from mpqe_amd.sampling import NegativeSampler
    from oracle import ref_cpu
    schema = synthetic.make_schema(*synthetic.KG_SHAPES['tiny'], seed=0)
    rng = np.random.RandomState(2)
    f = synthetic.sample_formula(schema, '3-inter', rng)
    qs = synthetic.sample_queries(schema, f, 200, rng)
    for q in qs:
        q.neg_samples = [int(v) for v in rng.randint(1, 50, size=rng.randint(1, 6))]
        q.hard_neg_samples = [int(v) for v in rng.randint(50, 90, size=rng.randint(1, 4))]
    s = NegativeSampler(qs, 'cuda:0')
    idx = rng.permutation(200)[:128]
    got = s.sample(idx, seed=42).cpu().numpy()
    off = np.concatenate([[0], np.cumsum([len(q.neg_samples) for q in qs])])
    cand = np.array([v for q in qs for v in q.neg_samples], dtype=np.int64)
    np.testing.assert_array_equal(got, ref_cpu.sample_negatives(cand, off, idx, 128, 42)[0])
    assert all(got[i] in qs[idx[i]].neg_samples for i in range(128))
    hard = s.sample(idx, seed=43, hard_negatives=True).cpu().numpy()
    assert all(hard[i] in qs[idx[i]].hard_neg_samples for i in range(128))
    buf = torch.zeros(256, dtype=torch.long, device='cuda:0')
    s.sample(idx, seed=42, out=buf[64:192])
    assert torch.equal(buf[64:192].cpu(), torch.from_numpy(got)) and int(buf[:64].abs().sum()) == 0
    s.check()
    shared = NegativeSampler(qs, 'cuda:0', full_list=range(1000, 1010))
    d = shared.sample(np.arange(64), seed=1).cpu().numpy()
    assert d.min() >= 1000 and d.max() < 1010
    qs[3].neg_samples = []
    with pytest.raises(IndexError):
        bad = NegativeSampler(qs, 'cuda:0')
        bad.sample(np.array([3, 4]), seed=0)
        bad.check()


def test_evaluation_loop_on_gpu_matches_oracle():
    """eval_auc_queries / eval_perc_queries (reference utils.py:34-95) through the GPU modules against the same
    loops through the CPU oracle: ragged negatives per query (the `neg_lengths` scoring form)."""
    from mpqe_amd import evaluation, synthetic
    from oracle import ref_cpu
    model, _ = _setup('sum', False, False, D=32)
    schema = synthetic.make_schema(*synthetic.KG_SHAPES['tiny'], seed=0)
    rng = np.random.RandomState(9)
    test_queries = {}
    for qt in ('2-chain', '3-inter_chain'):
        f = synthetic.sample_formula(schema, qt, rng)
        qs = []
        for _ in range(150):
            q = synthetic.sample_queries(schema, f, 1, rng, n_neg=int(rng.randint(1, 12)), n_hard=3)[0]
            qs.append(q)
        test_queries[f] = qs
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = dict(readout='sum', scatter_op='add', num_layers=3, adaptive=False, weight_decay=0)
    node_map = model.enc.node_maps.cpu()

    class Oracle(object):
        def forward(self, formula, queries, targets, neg_nodes=None, neg_lengths=None):
            col = ref_cpu.collate(formula, queries, model.rel_ids, model.mode_ids)
            return ref_cpu.forward(params, cfg, node_map, formula, col, targets, neg_nodes, neg_lengths)
    with torch.no_grad():
        auc, per = evaluation.eval_auc_queries(test_queries, model, batch_size=64, seed=1)
        perc = evaluation.eval_perc_queries(test_queries, model, batch_size=64)
        auc_ref, per_ref = evaluation.eval_auc_queries(test_queries, Oracle(), batch_size=64, seed=1)
        perc_ref = evaluation.eval_perc_queries(test_queries, Oracle(), batch_size=64)
    assert abs(auc - auc_ref) < 1e-3 and abs(perc - perc_ref) < 0.2     # a near-tie may swap one pair
    for f in per:
        assert abs(per[f] - per_ref[f]) < 2e-3


@pytest.mark.parametrize('chain', [True, False])
def test_captured_steps_own_their_workspace(chain):
    """hipGraph replay of the fused step, chain form and level form. Two packed steps whose workspaces differ in
    size are captured one after the other (round 1 captured into ONE shared arena: capturing the second step
    re-allocated it and the first graph replayed into freed memory -- the 'memory access fault' of the chain form);
    then every graph is replayed, the first one last, and compared with a plain run of the same packed step."""
    from mpqe_amd.fused import FusedTrainStep
    model, batches = _setup('mp', True, False, D=64, B=96)
    step = FusedTrainStep(model, chain=chain)
    small = step.pack(batches[:4])
    large = step.pack([dict(b) for b in batches] )
    assert large.ws_bytes > small.ws_bytes
    assert step.uses_chain(small) == chain
    ref = {}
    for name, p in (('small', small), ('large', large)):
        loss = step.run(p)
        ref[name] = (loss.clone(), {k: q.grad.clone() for k, q in model.named_parameters()})
    cap_small = step.capture(small)
    cap_large = step.capture(large)          # a second, larger arena: the first graph's must stay alive
    torch.cuda.empty_cache()
    junk = torch.full((64 << 20,), float('nan'), device='cuda:0')      # whatever was freed gets overwritten
    del junk
    for name, cap in (('large', cap_large), ('small', cap_small), ('large', cap_large), ('small', cap_small)):
        for q in model.parameters():
            q.grad.fill_(7.0)                 # the step zero-fills the gradients itself
        loss = cap.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(loss.cpu().numpy(), ref[name][0].cpu().numpy())
        for k, q in model.named_parameters():
            if k.startswith('enc.') and not chain:
                # level form: entity-table gradients by fp32 atomics -- the order of the additions into a popular entity's
                # row differs from run to run (hundreds of terms: the usual gradient tolerance, not last-bit equality; a
                # tighter bound failed once in ~10 runs)
                np.testing.assert_allclose(q.grad.cpu().numpy(), ref[name][1][k].cpu().numpy(), rtol=1e-4, atol=1e-6)
            else:
                assert torch.equal(q.grad, ref[name][1][k]), (name, k)
    step.check()


def test_sparse_tables_training_loop():
    """Row-sparse entity tables: FusedTrainStep(sparse_tables=True) writes only the touched rows of the
    table gradients (no zero fill, no pass over the tables) and FlatOptimizer(sparse_tables=True) updates only those
    rows with torch.optim.SparseAdam's rule. Three steps against the module path + torch.optim.SparseAdam on the touched
    rows of its dense table gradients + torch.optim.Adam on everything else."""
    import copy
    from mpqe_amd import ops
    from mpqe_amd.fused import FusedTrainStep
    from mpqe_amd.optim import FlatOptimizer
    model, batches = _setup('mp', True, False)
    ref_model = copy.deepcopy(model)
    tab_names = [k for k, _ in ref_model.named_parameters() if k.startswith('enc.')]
    ref_tabs = [p for k, p in ref_model.named_parameters() if k.startswith('enc.')]
    ref_rest = [p for k, p in ref_model.named_parameters() if not k.startswith('enc.')]
    opt_tabs = torch.optim.SparseAdam(ref_tabs, lr=0.01)
    opt_rest = torch.optim.Adam(ref_rest, lr=0.01)
    step = FusedTrainStep(model, sparse_tables=True)
    opt = FlatOptimizer(step, lr=0.01, sparse_tables=True)
    packed = step.pack(batches)
    node_maps = model.enc.node_maps.cpu()
    # rows each table is touched in by this packed step: anchors of the batch's anchor modes, targets and negatives
    touched = {m: set() for m in step.modes}
    for b in batches:
        f = b['formula']
        for a, mode in enumerate(f.anchor_modes):
            touched[mode].update(node_maps[torch.from_numpy(b['anchor_ids'][:, a])].tolist())
        touched[f.target_mode].update(node_maps[torch.from_numpy(b['targets'])].tolist())
        touched[f.target_mode].update(node_maps[torch.from_numpy(b['negs'])].tolist())
    for it in range(3):
        for q in model.parameters():
            q.grad.fill_(7.0)                      # stale content: sparse mode must not rely on (or clear) it
        step.run(packed)
        step.check()
        if it == 0:
            for mode in step.modes:
                g = model.enc.table(mode).grad
                rows = sorted(touched[mode])
                rest = sorted(set(range(g.shape[0])) - touched[mode])
                assert bool((g[rest] == 7.0).all()), 'untouched rows of a sparse table gradient were written'
                assert not bool((g[rows] == 7.0).all(dim=1).any()), 'a touched row was not written'
        opt.step(packed)
        ref_model.zero_grad(set_to_none=True)
        total = None
        for b in batches:
            out = ref_model.encode(b['formula'], b['queries'])
            l = ops.hinge(ref_model.score(b['formula'], out, b['targets'].tolist()),
                          ref_model.score(b['formula'], out, b['negs'].tolist()), 1.0) * b['weight']
            total = l if total is None else total + l
        total.backward()
        for name, p, mode in zip(tab_names, ref_tabs, step.modes):
            assert name == 'enc.feat-%s.weight' % mode
            rows = torch.tensor(sorted(touched[mode]), dtype=torch.long, device=p.device)
            dense = torch.zeros_like(p) if p.grad is None else p.grad
            p.grad = torch.sparse_coo_tensor(rows[None], dense[rows], p.shape)
        for p in ref_rest:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        opt_tabs.step()
        opt_rest.step()
    ref = dict(ref_model.named_parameters())
    for k, p in model.named_parameters():
        # (Adam turns a last-bit difference of a near-zero gradient into lr-sized steps: the tolerance of the dense loop test)
        np.testing.assert_allclose(p.detach().cpu().numpy(), ref[k].detach().cpu().numpy(), rtol=1e-3, atol=2e-3, err_msg=k)


def test_fresh_ids_every_step_in_step_plan_equals_pack_time_plan():
    """The training-loop regime: fresh ids every step for a recurring descriptor set. The step that builds its touch
    plan itself (default) from (a) host arrays -- staged and copied in stream order -- and (b) a flat id tensor
    already on the device, against a step object that builds the plan in pack() (touch='pack'): losses, scores and
    every gradient bit for bit, 15 steps, two packed steps alive at a time (their buffers come from the step object's
    pool and go back to it when the packed step dies)."""
    from mpqe_amd.fused import FusedTrainStep
    model, batches = _setup('mp', True, False)
    at_pack = FusedTrainStep(model, touch='pack')
    step = FusedTrainStep(model)
    assert step.touch_mode == 'step' and at_pack.touch_mode == 'pack'
    rng = np.random.RandomState(3)
    prev, seen_bufs = None, set()
    for it in range(15):
        fresh = []
        for b in batches:                       # same formulas and sizes, ids permuted between the queries
            p = rng.permutation(len(b['targets']))
            fresh.append(dict(b, anchor_ids=b['anchor_ids'][p], targets=b['targets'][p], negs=b['negs'][rng.permutation(len(p))]))
        pk = step.pack(fresh)
        assert pk.step_flags != 0 and pk.bufs is not (prev.bufs if prev is not None else None)
        seen_bufs.add(id(pk.bufs))
        loss, sp, sn = step.run(pk, scores=True)
        got = {k: p.grad.clone() for k, p in model.named_parameters()}
        loss, sp, sn = loss.clone(), sp.clone(), sn.clone()
        # (b) the same ids as one device tensor, descriptors only
        flat = torch.from_numpy(step.flatten_ids(fresh)).cuda()
        desc = [dict(formula=b['formula'], weight=b.get('weight', 1.0), batch_size=len(b['targets'])) for b in fresh]
        dl, dsp, dsn = step.run(step.pack(desc, ids=flat), scores=True)
        assert torch.equal(loss, dl) and torch.equal(sp, dsp) and torch.equal(sn, dsn), it
        for k, p in model.named_parameters():
            assert torch.equal(got[k], p.grad), (it, k)
        rl, rsp, rsn = at_pack.run(at_pack.pack(fresh), scores=True)
        assert torch.equal(loss, rl) and torch.equal(sp, rsp) and torch.equal(sn, rsn), it
        for k, p in model.named_parameters():
            assert torch.equal(got[k], p.grad), (it, k)
        # the plan the step left behind == the plan pack() builds (keys and permutation)
        M = pk.touch_entries
        a0 = pk.touch_ptr - pk.touch.data_ptr()
        ref_pk = at_pack.pack(fresh)
        torch.cuda.synchronize()
        b0 = ref_pk.touch_ptr - ref_pk.touch.data_ptr()
        al = lambda n: (n + 255) // 256 * 256
        assert torch.equal(pk.touch[a0 + 256:a0 + 256 + 8 * M], ref_pk.touch[b0 + 256:b0 + 256 + 8 * M])
        o = 256 + al(8 * M)
        assert torch.equal(pk.touch[a0 + o:a0 + o + 4 * M], ref_pk.touch[b0 + o:b0 + o + 4 * M])
        prev = pk                               # (keeps the previous packed step alive: the next pack takes other buffers)
    assert len(seen_bufs) <= step.POOL_PER_SET  # the pool recycles: buffers of dead packed steps serve later ones
    step.check()
    at_pack.check()


def test_packed_steps_own_their_buffers():
    """Explicit ownership of the pooled buffers: a live packed step's descriptor table / ids / plan are never handed to
    another pack; they return to the pool exactly when the packed step dies."""
    from mpqe_amd.fused import FusedTrainStep
    model, batches = _setup('mp', True, False)
    step = FusedTrainStep(model)
    live = [step.pack(batches) for _ in range(5)]
    assert len(set(id(p.bufs) for p in live)) == 5
    skey = live[0].bufs.skey
    assert len(step._pool[skey]) == 0
    ref = step.run(live[0]).clone()
    g0 = step.flat_grad.clone()
    for p in live[1:]:
        assert torch.equal(step.run(p), ref) and torch.equal(step.flat_grad, g0)
    ids = [id(p.bufs) for p in live]
    del p
    live = live[:2]
    assert len(step._pool[skey]) == 3           # the three dead steps' buffers are back
    again = step.pack(batches)
    assert id(again.bufs) in ids[2:] and again.desc_resident      # ... table still resident: nothing to upload
    assert torch.equal(step.run(again), ref) and torch.equal(step.flat_grad, g0)
    step.check()


@pytest.mark.parametrize('readout,D', [('concat', 64), ('concat', 128), ('mlp', 64), ('targetmlp', 128)])
def test_learned_readout_on_the_chain_uses_this_steps_weights(readout, D):
    """A learned readout on the chain form multiplies by TRANSPOSED copies of its Linear weights that workgroups of the same
    launch make (csrc/step.hip: prep_transpose_block). Every op that reads a copy must wait for it: the concat readout's
    partial products sit in the FORWARD levels' programme (reference model.py:441-446: one input block per layer) and until
    round 5 ran ahead of that wait -- a step then used the copies the PREVIOUS step had left (weights one optimiser step
    old; harmless in a test that never changes the weights, which is why every test until then passed). Here the readout's
    weights change between two runs of one packed step, and each run is held against the module path with the weights of
    that run: loss and every gradient."""
    from mpqe_amd import ops
    from mpqe_amd.fused import FusedTrainStep
    model, batches = _setup(readout, False, False, D=D, B=48, weight_decay=1e-3)
    batches = batches[:7]
    step = FusedTrainStep(model)
    packed = step.pack(batches)
    assert step.uses_chain(packed) and step.learned
    gen = torch.Generator(device='cpu').manual_seed(5)
    for trial in range(4):
        if trial:           # new readout weights, in place: what an optimiser step does
            with torch.no_grad():
                for p in model.readout.parameters():
                    p.copy_((torch.rand(p.shape, generator=gen) * 2 - 1).to(p.device) * (0.5 / np.sqrt(p.shape[-1])))
        loss = step.run(packed)
        step.check()
        got = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        got_loss = loss[0].item()
        for p in model.parameters():
            p.grad = None
        total = None
        for b in batches:
            out = model.encode(b['formula'], b['queries'])
            pos = model.score(b['formula'], out, b['targets'].tolist())
            neg = model.score(b['formula'], out, b['negs'].tolist())
            l = (ops.hinge(pos, neg, 1.0) + model.weight_decay * ops.l2_norms(list(model.readout.parameters()))) * b['weight']
            total = l if total is None else total + l
        total.backward()
        np.testing.assert_allclose(got_loss, total.item(), rtol=1e-5, atol=1e-6, err_msg='trial %d' % trial)
        for k, p in model.named_parameters():
            ref = torch.zeros_like(p) if p.grad is None else p.grad
            np.testing.assert_allclose(got[k].cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg='%s (trial %d)' % (k, trial))
        step.bind_grads()

"""The fused training step (mpqe_step_forward_backward) against the CPU oracle: the weighted sum
of the reference's margin losses over a mix of formula batches, every score, and every parameter
gradient. Runs on the host emulator and (gpu) on the real library."""
import ctypes

import numpy as np
import pytest
import torch

from mpqe_amd import _capi, synthetic
from oracle import ref_cpu


@pytest.fixture(scope='module', params=['emu', pytest.param('hip', marks=pytest.mark.gpu)])
def be(request):
    from tests import kernel_backend
    return kernel_backend.EmuBackend() if request.param == 'emu' else kernel_backend.HipBackend()


def make_problem(seed, D, num_layers, shared, mix, readout, adaptive, scale=3.0):
    rng = np.random.RandomState(seed)
    torch.manual_seed(seed)
    schema = synthetic.make_schema(*synthetic.KG_SHAPES['tiny'], seed=seed)
    graph = synthetic.SchemaGraph(schema, D)
    mode_ids, rel_ids = ref_cpu.build_ids(schema.relations, graph.mode_weights)
    R = len(rel_ids)
    params = {}
    node_map = torch.full((schema.num_entities + 1,), -1, dtype=torch.long)
    for m in schema.modes:
        ids = torch.from_numpy(schema.ids[m])
        node_map[ids] = torch.arange(len(ids))
        params['enc.feat-%s.weight' % m] = torch.randn(len(ids) + 1, D) / D
    params['mode_embeddings.weight'] = torch.randn(len(schema.modes), D)
    bound = scale / np.sqrt(R * D)
    for l in range(num_layers):
        if shared and l > 0:
            for k in ('basis', 'root', 'bias'):
                params['layers.%d.%s' % (l, k)] = params['layers.0.%s' % k]
            continue
        params['layers.%d.basis' % l] = (torch.rand(R, D, D) * 2 - 1) * bound
        params['layers.%d.root' % l] = (torch.rand(D, D) * 2 - 1) * bound
        params['layers.%d.bias' % l] = (torch.rand(D) * 2 - 1) * bound
    if readout in ('mlp', 'targetmlp', 'concat'):        # reference model.py:497-553: Linear - ReLU - Linear
        din = 2 * D if readout == 'targetmlp' else (num_layers * D if readout == 'concat' else D)
        params['readout.layers.0.weight'] = (torch.rand(D, din) * 2 - 1) / np.sqrt(din)
        params['readout.layers.0.bias'] = (torch.rand(D) * 2 - 1) / np.sqrt(din)
        params['readout.layers.2.weight'] = (torch.rand(D, D) * 2 - 1) / np.sqrt(D)
        params['readout.layers.2.bias'] = (torch.rand(D) * 2 - 1) / np.sqrt(D)
    for v in params.values():
        v.requires_grad_(True)
    cfg = dict(readout=readout, scatter_op='add', num_layers=num_layers, adaptive=adaptive, weight_decay=0)
    batches = []
    for qt, B, w in mix:
        formula = synthetic.sample_formula(schema, qt, rng)
        queries = synthetic.sample_queries(schema, formula, B, rng)
        col = ref_cpu.collate(formula, queries, rel_ids, mode_ids)
        tg = np.array([q.target_node for q in queries], dtype=np.int64)
        ng = np.array([q.neg_samples[0] for q in queries], dtype=np.int64)
        batches.append(dict(formula=formula, col=col, targets=tg, negs=ng, weight=w, qt=qt, B=B))
    return schema, mode_ids, rel_ids, params, node_map, cfg, batches


def _gpu_only_when_heavy(be, heavy):
    """The fiber emulator runs a D = 128 chain launch of hundreds of node updates for tens of seconds: the heaviest
    parametrisations run on the GPU backend only (same test, same assertions); their lighter siblings cover the same code on
    the CPU (D = 64 of the same form, D = 128 with fewer ops)."""
    if heavy and be.name == 'emu':
        pytest.skip('tens of seconds on the CPU emulator: runs on the GPU backend (-m gpu)')


def oracle_step(params, cfg, node_map, batches, margin):
    total, per, sp, sn = 0, [], [], []
    for b in batches:
        q = ref_cpu.encode_queries(params, cfg, node_map, b['formula'], b['col'])
        pos = ref_cpu.score(params, node_map, b['formula'], q, b['targets'])
        neg = ref_cpu.score(params, node_map, b['formula'], q, b['negs'])
        # the reference's own sequence (two encoder passes) gives the same loss; checked below once
        l = torch.clamp(margin - (pos - neg), min=0).mean()
        total = total + b['weight'] * l
        per.append(l.item())
        sp.append(pos.detach().numpy())
        sn.append(neg.detach().numpy())
    total.backward()
    return total.item(), per, np.concatenate(sp), np.concatenate(sn)


def run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, backward=1, lanes=None, flags=0, touch='step', repeat=1,
             plan_out=None, between=None, recover=False, before_recovery=None, dev_weights=False, query_out=None,
             then_plain=False, xcd_shift=0, readout_norms=False):
    """between: the step in three calls around the CALLER's readout (MPQE_READOUT_CALLER, MPQE_STEP_PHASE_*): a generator
    function -- between(final_states) yields the query embeddings [graphs, D], is sent their gradients and yields d loss /
    d final states per batch; final_states[i]: batch i's [B N, D]. touch: 'step' = the step builds the touch plan of its ids itself (MPQE_STEP_BUILD_TOUCH, the product's default),
    'pack' = mpqe_step_touch_build in front of it, False = fp32 atomics. plan_out: a list that receives the plan's bytes."""
    D = params['mode_embeddings.weight'].shape[1]
    if touch == 'step' and ((flags & _capi.STEP_EIGHT_WAVES) or lanes is not None):
        touch = 'pack'
    if touch == 'step':
        flags |= _capi.STEP_BUILD_TOUCH
    L = cfg['num_layers']
    R = params['layers.0.basis'].shape[0]
    modes = list(schema.modes)
    dev = {}

    def put(key):
        if key not in dev:
            dev[key] = be.put(params[key].detach().numpy())
        return dev[key]
    tables = [put('enc.feat-%s.weight' % m) for m in modes]
    # shared layers must alias ONE buffer, exactly like the reference's ModuleList of one module
    uniq = {}
    lay = []
    for l in range(L):
        key = id(params['layers.%d.basis' % l])
        if key not in uniq:
            uniq[key] = tuple(put('layers.%d.%s' % (l, k)) for k in ('basis', 'root', 'bias'))
        lay.append(uniq[key])
    dnm = be.put(node_map.numpy())
    dmode = put('mode_embeddings.weight')
    learned = between is None and cfg['readout'] in _capi.LEARNED_READOUT_IDS
    if between is not None or (learned and not (D in (64, 128, 256) and (cfg['readout'] != 'concat' or not cfg['adaptive']))):
        touch = False           # (level form: it has no use for a touch plan)
    rid = _capi.READOUT_CALLER if between is not None else (_capi.LEARNED_READOUT_IDS[cfg['readout']] if learned else cfg['readout'])
    P = _capi.make_step_params(D, R, rid, [be.ptr(t) for t in tables],
                               [params['enc.feat-%s.weight' % m].shape[0] for m in modes], be.ptr(dnm),
                               node_map.shape[0], be.ptr(dmode), [be.ptr(x[0]) for x in lay],
                               [be.ptr(x[1]) for x in lay], [be.ptr(x[2]) for x in lay], flags=flags)
    gtabs = [be.zeros(tuple(params['enc.feat-%s.weight' % m].shape)) for m in modes]
    gmode = be.zeros(tuple(params['mode_embeddings.weight'].shape))
    glay_u = {k: (be.zeros((R, D, D)), be.zeros((D, D)), be.zeros((D,))) for k in uniq}
    glay = [glay_u[id(params['layers.%d.basis' % l])] for l in range(L)]
    if flags & _capi.STEP_ZERO_GRADS:     # the library must clear whatever is in the gradient buffers
        for t in gtabs + [gmode] + [x for u in glay_u.values() for x in u]:
            if be.name == 'emu':
                t.fill(7.5)
            else:
                t.fill_(7.5)
    G = _capi.make_step_grads([be.ptr(t) for t in gtabs], be.ptr(gmode), [be.ptr(x[0]) for x in glay],
                              [be.ptr(x[1]) for x in glay], [be.ptr(x[2]) for x in glay])
    gro = {}
    if learned:        # the readout's two Linear layers are the library's too (include/mpqe_amd.h: MPQE_READOUT_MLP ...)
        for field, key in (('readout_w0', 'readout.layers.0.weight'), ('readout_b0', 'readout.layers.0.bias'),
                           ('readout_w2', 'readout.layers.2.weight'), ('readout_b2', 'readout.layers.2.bias')):
            setattr(P, field, be.ptr(put(key)))
            gro[key] = be.zeros(tuple(params[key].shape))
            if flags & _capi.STEP_ZERO_GRADS:
                gro[key].fill(7.5) if be.name == 'emu' else gro[key].fill_(7.5)
            setattr(G, field, be.ptr(gro[key]))
        P.readout_scatter = _capi.SCATTER_IDS[cfg['scatter_op']]
        P.readout_weight_decay = float(cfg.get('weight_decay', 0))
    nb = len(batches)
    SB = (_capi.StepBatch * nb)()
    anchors = []
    for i, b in enumerate(batches):
        col, f = b['col'], b['formula']
        passes = ref_cpu.num_passes(cfg, b['qt'])
        E = col['E']
        SB[i] = _capi.make_step_batch(b['qt'], passes, b['B'], col['edge_type'][:E], col['var_ids'],
                                      [modes.index(m) for m in f.anchor_modes], modes.index(f.target_mode),
                                      2.0 if dev_weights else b['weight'])
        anchors.append(np.ascontiguousarray(col['anchor_ids'].T).reshape(-1))
    d_anchor = be.put(np.concatenate(anchors))
    d_tg = be.put(np.concatenate([b['targets'] for b in batches]))
    d_ng = be.put(np.concatenate([b['negs'] for b in batches]))
    Gtot = sum(b['B'] for b in batches)
    keep = []
    if lanes is not None:
        splits = lanes
        lanes = _capi.StepLanes()
        lanes.num_lanes = len(splits) - 1
        for i, v in enumerate(splits):
            lanes.batch_begin[i] = v
        if be.name == 'hip':
            ev = be.torch.cuda.Event()
            ev.record()
            keep.append(ev)
            lanes.fork_event = ev.cuda_event
            for l in range(1, lanes.num_lanes):
                st, je = be.torch.cuda.Stream(), be.torch.cuda.Event()
                je.record()
                keep.extend([st, je])
                lanes.aux_stream[l], lanes.join_event[l] = st.cuda_stream, je.cuda_event
        else:       # the emulator runs launches in program order; handles only need to be non-null
            lanes.fork_event = 1
            for l in range(1, lanes.num_lanes):
                lanes.aux_stream[l], lanes.join_event[l] = 1, 1
        lanes = ctypes.byref(lanes)
    wsb = be.lib.mpqe_step_workspace_bytes(ctypes.byref(P), SB, nb, lanes)
    if wsb == 0:
        raise _capi.MpqeError('mpqe_step_workspace_bytes rejected the step descriptors')
    ws = be.nbytes(wsb + 256)
    if be.name == 'emu':          # node states the step skips must never be read: poison the arena
        ws.fill(np.nan)
    else:
        ws.fill_(float('nan'))
    wptr = (be.ptr(ws) + 255) // 256 * 256
    dsb = be.lib.mpqe_step_desc_bytes(ctypes.byref(P), SB, nb, lanes)
    dbuf = be.nbytes(dsb + 256)
    dptr = (be.ptr(dbuf) + 255) // 256 * 256
    loss = be.empty((1 + nb,))
    sp, sn = be.empty((Gtot,)), be.empty((Gtot,))
    err = be.zeros((1,), np.int32)
    tptr = None
    if touch:
        tb = be.lib.mpqe_step_touch_bytes(ctypes.byref(P), SB, nb)
        twb = be.lib.mpqe_step_touch_workspace_bytes(ctypes.byref(P), SB, nb)
        assert tb > 0 and twb > 0
        tbuf, twbuf = be.nbytes(tb + 256), be.nbytes(twb + 256)
        tptr = (be.ptr(tbuf) + 255) // 256 * 256
        if touch == 'pack':
            be.check(be.lib.mpqe_step_touch_build(ctypes.byref(P), SB, nb, be.ptr(d_anchor), be.ptr(d_tg), be.ptr(d_ng), tptr, tb,
                                                  (be.ptr(twbuf) + 255) // 256 * 256, twb, be.stream), 'touch')
        keep.extend([tbuf, twbuf])
    def put_at(buf, start, arr):
        buf[start:start + arr.size] = arr.reshape(-1) if be.name == 'emu' else be.put(arr.reshape(-1))

    for rep in range(repeat if between is not None else 0):
        def call(phase, upload):
            be.check(be.lib.mpqe_step_forward_backward(ctypes.byref(P), SB, nb, be.ptr(d_anchor), be.ptr(d_tg), be.ptr(d_ng),
                                                       margin, ctypes.byref(G), phase, be.ptr(loss), be.ptr(sp), be.ptr(sn), dptr,
                                                       dsb, upload, wptr, wsb, be.ptr(err), lanes, None, 0, None, be.stream),
                     'step')
        i64 = ctypes.c_int64
        so, go, lstride, ro, qo, gqo = i64(), i64(), i64(), (i64 * (nb + 1))(), i64(), i64()
        be.check(be.lib.mpqe_step_states_layout(ctypes.byref(P), SB, nb, lanes, ctypes.byref(so), ctypes.byref(go),
                                                ctypes.byref(lstride), ro, ctypes.byref(qo), ctypes.byref(gqo)), 'layout')
        assert lstride.value == ro[nb] * D and all(v.value % 4 == 0 for v in (so, go, qo, gqo))
        call(_capi.STEP_PHASE_STATES, 1 if rep == 0 else 0)
        raw = np.asarray(be.get(ws))
        base = (wptr - be.ptr(ws)) // 4
        finals, levels = [], []
        for i, b in enumerate(batches):
            levels.append([raw[base + so.value // 4 + lv * lstride.value + ro[i] * D:
                               base + so.value // 4 + lv * lstride.value + ro[i + 1] * D].reshape(-1, D).copy()
                           for lv in range(SB[i].num_passes + 1)])
            finals.append(levels[-1][-1])
            assert np.isfinite(finals[-1]).all()
        gen = between(finals, levels)
        q = np.ascontiguousarray(next(gen), np.float32)            # query embeddings [graphs, D], batch order
        assert q.shape == (Gtot, D)
        put_at(ws, base + qo.value // 4, q)
        if backward == 0:
            call(_capi.STEP_PHASE_SCORES_ONLY, 0)
            continue
        call(_capi.STEP_PHASE_SCORES, 0)
        raw = np.asarray(be.get(ws))
        gq = raw[base + gqo.value // 4: base + gqo.value // 4 + Gtot * D].reshape(Gtot, D).copy()
        # d loss / d final states per batch, or {level: d loss / d states} (a readout over every level: ADD_STATE_GRADS)
        gfinal = gen.send(gq)
        for i in range(nb):
            per = gfinal[i] if isinstance(gfinal[i], dict) else {SB[i].num_passes: gfinal[i]}
            if isinstance(gfinal[i], dict):
                P.flags |= _capi.STEP_ADD_STATE_GRADS
                assert sorted(per) == list(range(1, SB[i].num_passes + 1))
            for lv, g in per.items():
                put_at(ws, base + go.value // 4 + lv * lstride.value + ro[i] * D, np.ascontiguousarray(g, np.float32))
        call(_capi.STEP_PHASE_FROM_STATES, 0)
        P.flags &= ~_capi.STEP_ADD_STATE_GRADS
    extra = None
    if dev_weights or query_out is not None or xcd_shift or readout_norms:
        # include/mpqe_amd.h: mpqe_step_extra_t -- batch weights as DEVICE scalars (host weight 2 x device weight w / 2 = w:
        # the same step), the query embeddings out
        extra = _capi.StepExtra()
        if dev_weights:
            dw = be.put(np.array([0.5 * b['weight'] for b in batches], np.float32))
            keep.append(dw)
            for i in range(nb):
                extra.batch_weight[i] = be.ptr(dw) + 4 * i
        if query_out is not None:
            dq = be.empty((Gtot, D))
            extra.query_out = be.ptr(dq)
        note = be.zeros((2,), np.int32)        # (mpqe_step_extra_t.notify: the call's number + the error word, by its last launch)
        extra.notify, extra.notify_value = be.ptr(note), 4711
        extra.xcd_shift = xcd_shift
        if readout_norms:       # (mpqe_step_extra_t.readout_norms: the regulariser's four norms formed once, ahead of the calls)
            dnorm = be.zeros((1,), np.float32)
            keep.append(dnorm)
            be.check(be.lib.mpqe_step_readout_norms(ctypes.byref(P), be.ptr(dnorm), be.stream), 'readout norms')
            extra.readout_norms = be.ptr(dnorm)
    kinds = list(backward) if isinstance(backward, (list, tuple)) else [backward] * repeat      # (a list: one call of each kind, in order)
    repeat = len(kinds)
    for rep in range(0 if between is not None else repeat):       # (repeat > 1: the same packed step again -- its hand-off epochs / counters carry on)
        args = (ctypes.byref(P), SB, nb, be.ptr(d_anchor), be.ptr(d_tg), be.ptr(d_ng), margin, ctypes.byref(G), kinds[rep],
                be.ptr(loss), be.ptr(sp), be.ptr(sn), dptr, dsb, 1 if rep == 0 else 0, wptr, wsb, be.ptr(err), lanes, None, 0, tptr,
                be.stream)
        if extra is None or (then_plain and rep == repeat - 1):
            be.check(be.lib.mpqe_step_forward_backward(*args), 'step')
        else:
            be.check(be.lib.mpqe_step_forward_backward_ex(*(args + (ctypes.byref(extra),))), 'step')
    if query_out is not None:
        query_out.append(be.get(dq))
    if extra is not None and not then_plain:
        got = np.asarray(be.get(note))
        assert int(got[0]) == 4711 and int(got[1]) == int(be.get(err)[0]), got
    if recover and (int(be.get(err)[0]) & _capi.FLAG_TOUCH_RETRY):
        # the step could not build its own touch plan: everything but the entity-table gradients is complete. Rebuild the plan
        # with the library sort and sum the table rows again from the per-entry rows the step left in its workspace.
        if before_recovery is not None:
            before_recovery({m: np.asarray(be.get(g)).copy() for m, g in zip(modes, gtabs)})
        P.flags |= _capi.STEP_TOUCH_LIBRARY_SORT
        be.check(be.lib.mpqe_step_touch_build(ctypes.byref(P), SB, nb, be.ptr(d_anchor), be.ptr(d_tg), be.ptr(d_ng), tptr, tb,
                                              (be.ptr(twbuf) + 255) // 256 * 256, twb, be.stream), 'touch rebuild')
        P.flags &= ~_capi.STEP_TOUCH_LIBRARY_SORT
        be.check(be.lib.mpqe_step_table_rows(ctypes.byref(P), SB, nb, ctypes.byref(G), dptr, wptr, wsb, tptr, be.stream), 'table rows')
        e = np.asarray(be.get(err)).copy()
        e[0] &= ~_capi.FLAG_TOUCH_RETRY
        if be.name == 'emu':
            err[:] = e
        else:
            err.copy_(be.put(e))
    if plan_out is not None and touch:
        raw = np.asarray(be.get(tbuf)).view(np.uint8)
        off = tptr - be.ptr(tbuf)
        plan_out.append(raw[off:off + tb].copy())
    grads = {'mode_embeddings.weight': be.get(gmode)}
    for key, g in gro.items():
        grads[key] = be.get(g)
    for m, g in zip(modes, gtabs):
        grads['enc.feat-%s.weight' % m] = be.get(g)
    for l in range(L):
        for k, g in zip(('basis', 'root', 'bias'), glay[l]):
            grads['layers.%d.%s' % (l, k)] = be.get(g)
    return be.get(loss), be.get(sp), be.get(sn), grads, int(be.get(err)[0])


MIXES = {
    'all7': [('1-chain', 9, 1.0), ('2-chain', 70, 0.01), ('3-chain', 33, 0.01), ('2-inter', 5, 0.005),
             ('3-inter', 65, 0.005), ('3-inter_chain', 12, 0.005), ('3-chain_inter', 40, 0.005)],
    'dup': [('3-inter', 20, 1.0), ('3-inter', 20, 0.5), ('2-chain', 17, 2.0)],
    # batch sizes that are whole K-steps with D % 64 == 0: the unpredicated LD_FAST kernels
    'fast': [('3-chain', 64, 1.0), ('3-inter_chain', 32, 0.5), ('1-chain', 96, 0.25)],
    # same dims, ragged batch sizes: LD_FAST layer tiles with clamped rows, predicated weight gradient
    'fastragged': [('3-chain_inter', 70, 1.0), ('2-inter', 33, 0.5)],
}


@pytest.mark.parametrize('readout,adaptive,shared,L', [('mp', True, False, 3), ('sum', False, False, 2),
                                                       ('max', False, True, 3), ('mp', True, True, 3)])
@pytest.mark.parametrize('mix', ['all7', 'dup', 'fast', 'fastragged'])
def test_fused_step_matches_oracle(be, readout, adaptive, shared, L, mix):
    D, margin = (64 if mix.startswith('fast') else 32), 1.0
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(
        11, D, L, shared, MIXES[mix], readout, adaptive)
    ref_loss, ref_per, ref_sp, ref_sn = oracle_step(params, cfg, node_map, batches, margin)
    loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin)
    assert err == 0
    np.testing.assert_allclose(sp, ref_sp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sn, ref_sn, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss[1:], ref_per, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss[0], ref_loss, rtol=1e-5, atol=1e-6)
    seen = set()
    for k, p in params.items():
        if id(p) in seen:
            continue                      # shared layers: one buffer, one accumulated gradient
        seen.add(id(p))
        ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
        np.testing.assert_allclose(grads[k], ref, rtol=1e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('D,readout,adaptive,L', [(64, 'mp', True, 3), (128, 'sum', False, 2), (32, 'max', False, 2),
                                                  (64, 'mlp', False, 2)])
def test_step_with_device_batch_weights_and_query_out(be, D, readout, adaptive, L):
    """mpqe_step_forward_backward_ex (the drop-in margin_loss / forward, mpqe_amd/dropin.py): per-batch loss weights read
    from DEVICE scalars give the gradients of the same step with host weights (reference train_helpers.py:81-119:
    loss = l_0 + w_1 l_1 + ...; backward); the query embeddings it writes are the oracle's readout rows (model.py:447-449);
    and a later call WITHOUT extras on the same resident descriptor table sees the host weights again."""
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(
        31, D, L, False, MIXES['all7'] if D != 128 else MIXES['dup'], readout, adaptive)
    if readout == 'mlp':
        cfg['weight_decay'] = 0.01
    ref_loss, ref_per, ref_sp, ref_sn = oracle_step(params, cfg, node_map, batches, 1.0)
    chain = D in (64, 128, 256)
    qs = [] if chain else None
    loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, dev_weights=True,
                                        query_out=qs, flags=_capi.STEP_ZERO_GRADS)
    assert err == 0
    np.testing.assert_allclose(sp, ref_sp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss[1:], ref_per, rtol=1e-5, atol=1e-6)
    seen = set()
    for k, p in params.items():
        if id(p) in seen:
            continue
        seen.add(id(p))
        ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
        if readout == 'mlp' and k.startswith('readout.'):
            continue            # (oracle_step leaves the regulariser out: compared below against the host-weight run)
        np.testing.assert_allclose(grads[k], ref, rtol=1e-4, atol=2e-6, err_msg=k)
    if chain:
        q_ref = np.concatenate([ref_cpu.encode_queries(params, cfg, node_map, b['formula'], b['col']).detach().numpy()
                                for b in batches])
        np.testing.assert_allclose(qs[0], q_ref, rtol=1e-5, atol=1e-6)
    # the same step with HOST weights: identical gradients (regulariser included), and once more with extras first and a
    # plain call last on the same resident table (the library writes the host weights back)
    for p in params.values():
        p.grad = None
    for b in batches:
        b['weight'] = float(np.float32(b['weight']))
    l_h, _, _, g_h, _ = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_ZERO_GRADS)
    for k in grads:
        np.testing.assert_allclose(grads[k], g_h[k], rtol=1e-5, atol=1e-7, err_msg=k)
    saved = [b['weight'] for b in batches]
    l2, _, _, g2, _ = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_ZERO_GRADS,
                               dev_weights=True, repeat=2, then_plain=True)
    # (then_plain's last call ran with the HOST weights of that run: 2.0 per batch)
    for b in batches:
        b['weight'] = 2.0
    l3, _, _, g3, _ = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_ZERO_GRADS)
    for b, w in zip(batches, saved):
        b['weight'] = w
    for k in g2:
        np.testing.assert_allclose(g2[k], g3[k], rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.mark.parametrize('readout,adaptive,D', [('mp', True, 64), ('mlp', False, 64), ('sum', False, 128)])
def test_forward_only_step_moved_round_the_chip(be, readout, adaptive, D):
    """mpqe_step_extra_t.xcd_shift: a forward-only step whose workgroups sit 1 .. 7 XCDs further round the chip (idle
    workgroups in front of them; forward-only steps on several streams then run side by side) returns what it returns at
    home, bit for bit -- loss, scores, error word, and the notification by the launch's last workgroup."""
    _gpu_only_when_heavy(be, D >= 128)
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(5, D, 3, False, MIXES['dup'], readout, adaptive)
    home = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, backward=0, repeat=2)
    for shift in (1, 3, 7):
        moved = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, backward=0, repeat=2, xcd_shift=shift)
        assert moved[4] == home[4] == 0
        for a, b in zip(moved[:3], home[:3]):
            np.testing.assert_array_equal(a, b)
    # ... and a whole step ignores it
    full = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_ZERO_GRADS)
    moved = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_ZERO_GRADS, xcd_shift=5)
    np.testing.assert_array_equal(moved[0], full[0])
    for k in full[3]:
        np.testing.assert_array_equal(moved[3][k], full[3][k], err_msg=k)


@pytest.mark.parametrize('D', [8, 16, 20])
@pytest.mark.parametrize('readout,adaptive', [('mp', True), ('sum', False)])
def test_fused_step_small_dimensions(be, D, readout, adaptive):
    """The level form at the goldens' dimensions (tests/golden/enc_*: D = 16 / 32). D = 16 is the one dimension whose
    reduction grid row is a single workgroup: the vector groups' column slices must not be used there (round 5: bias and
    mode-vector gradients beyond the first four columns were never summed)."""
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(5, D, 3, False, MIXES['all7'], readout, adaptive)
    ref_loss, ref_per, ref_sp, ref_sn = oracle_step(params, cfg, node_map, batches, 1.0)
    loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0)
    assert err == 0
    np.testing.assert_allclose(loss[0], ref_loss, rtol=1e-5, atol=1e-6)
    for k, p in params.items():
        ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
        np.testing.assert_allclose(grads[k], ref, rtol=1e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('D,readout,adaptive', [(128, 'mp', True), (64, 'mp', True), (128, 'sum', False)])
def test_post_pass_closures_match_oracle(be, request, D, readout, adaptive):
    """The backward post-pass of the batch-uniform node states as one closure workgroup per batch (split tail launch,
    csrc/step_closure.h; mpqe_debug_option CLOSURE) against the oracle and against the vector-op form:
    batches of more than 8 x 16 graphs, so that a column sum has more rows than the closure has row groups (its order
    of additions then differs from the vector ops': same values within rounding), every chain depth, ragged sizes."""
    if not be.lib.mpqe_debug_has_experiments():
        pytest.skip('measured slower and taken out of the shipped library: builds with -DMPQE_EXPERIMENTS only '
                    '(MPQE_EMU_EXPERIMENTS=1 / tools/build_variant.sh)')
    mix = [('3-chain', 200, 1.0), ('2-chain', 150, 0.5), ('3-inter_chain', 40, 0.25), ('3-chain_inter', 33, 2.0),
           ('3-chain', 17, 0.3)]
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(29, D, 3, False, mix, readout, adaptive)
    ref_loss, ref_per, ref_sp, ref_sn = oracle_step(params, cfg, node_map, batches, 1.0)
    be.lib.mpqe_debug_option(b'CLOSURE', 1, 1)               # (built on request: not the default form)
    request.addfinalizer(lambda: be.lib.mpqe_debug_option(b'CLOSURE', 0, 0))
    for zero in (True, False):
        flags = _capi.STEP_SPLIT_TAIL | (_capi.STEP_ZERO_GRADS if zero else 0)
        got = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=flags, repeat=2 if zero else 1)
        assert got[4] == 0
        np.testing.assert_allclose(got[0][0], ref_loss, rtol=1e-5, atol=1e-6)
        for k, p in params.items():
            ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
            np.testing.assert_allclose(got[3][k], ref, rtol=1e-4, atol=2e-6, err_msg=k)
    be.lib.mpqe_debug_option(b'CLOSURE', 0, 0)
    old = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_SPLIT_TAIL)
    for k in got[3]:
        np.testing.assert_allclose(got[3][k], old[3][k], rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.mark.parametrize('zero', [True, False])
def test_failed_in_step_touch_plan_is_recovered(be, request, zero):
    """The step builds the touch plan of its ids by workgroups that must all be resident at once (step_touch.h); on a GPU
    shared with other work that can fail -- a failure mode the reference does not have (its embedding backward is plain
    autograd, encoders.py:40-43). Forced here (mpqe_debug_option TSORT_FAIL): the step must flag MPQE_FLAG_TOUCH_RETRY and
    nothing else, leave the entity-table gradients unaccumulated (never garbage), deliver every other gradient, and after
    mpqe_step_touch_build (library sort) + mpqe_step_table_rows every gradient must be the oracle's."""
    D, margin = 64, 1.0
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(23, D, 3, False, MIXES['all7'], 'mp', True)
    ref_loss, ref_per, ref_sp, ref_sn = oracle_step(params, cfg, node_map, batches, margin)
    be.lib.mpqe_debug_option(b'TSORT_FAIL', 1, 1)
    request.addfinalizer(lambda: be.lib.mpqe_debug_option(b'TSORT_FAIL', 0, 0))
    flags = _capi.STEP_ZERO_GRADS if zero else 0
    # without recovery: the flag, and table gradients that are exactly what the caller / the zero fill left there
    loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, flags=flags)
    assert err == _capi.FLAG_TOUCH_RETRY
    np.testing.assert_allclose(loss[0], ref_loss, rtol=1e-5, atol=1e-6)
    for k, p in params.items():
        ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
        if k.startswith('enc.'):
            assert not grads[k].any(), k          # (run_step's buffers start from zero / are zero-filled by the step)
        else:
            np.testing.assert_allclose(grads[k], ref, rtol=1e-4, atol=2e-6, err_msg=k)
    seen = {}
    loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, flags=flags, recover=True,
                                        before_recovery=lambda t: seen.update(t))
    assert err == 0 and seen
    np.testing.assert_allclose(sp, ref_sp, rtol=1e-5, atol=1e-6)
    for k, p in params.items():
        ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
        np.testing.assert_allclose(grads[k], ref, rtol=1e-4, atol=2e-6, err_msg=k)
    # the recovered gradients are the ones an undisturbed step delivers, bit for bit (same plan, same order of additions)
    be.lib.mpqe_debug_option(b'TSORT_FAIL', 0, 0)
    _, _, _, good, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, flags=flags)
    assert err == 0
    for k in grads:
        np.testing.assert_array_equal(grads[k], good[k], err_msg=k)


@pytest.mark.parametrize('readout,adaptive,shared,zero', [('mlp', True, False, True), ('targetmlp', False, True, False),
                                                          ('concat', False, False, True)])
def test_step_in_three_calls_around_a_callers_readout(be, readout, adaptive, shared, zero):
    """The learned readouts (reference model.py:497-553) are the CALLER's: the step runs as three calls -- node states
    out; query embeddings in, their gradients out; state gradients in -- with the caller's readout in between, here the
    oracle's own readout code under autograd. Loss, scores and every gradient must be the oracle's for the whole model."""
    D, margin = 32, 1.0
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(17, D, 3, shared, MIXES['all7'], readout, adaptive)
    ref_loss, ref_per, ref_sp, ref_sn = oracle_step(params, cfg, node_map, batches, margin)
    seen = {}

    def between(finals, levels):
        local = {k: v.detach().clone().requires_grad_(True) for k, v in params.items() if k.startswith('readout.')}
        qs, leaves = [], []
        for b, lv in zip(batches, levels):
            col = b['col']
            if readout == 'concat':             # reference model.py:441-446: every layer's output, side by side
                hs = [torch.from_numpy(x).requires_grad_(True) for x in lv[1:]]
                h = torch.cat(hs, dim=1)
            else:
                hs = [torch.from_numpy(lv[-1]).requires_grad_(True)]
                h = hs[0]
            leaves.append(hs)
            qs.append(ref_cpu.readout(cfg['readout'], cfg['scatter_op'], local, h, torch.as_tensor(col['batch']), col['B'],
                                      col['N'], col['A']))
        q = torch.cat(qs, dim=0)
        gq = yield q.detach().numpy()
        q.backward(torch.from_numpy(gq))
        seen['readout'] = {k: v.grad.numpy() for k, v in local.items()}
        if readout == 'concat':
            yield [{1 + i: h.grad.numpy() for i, h in enumerate(hs)} for hs in leaves]
        else:
            yield [hs[0].grad.numpy() for hs in leaves]

    loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin,
                                        flags=_capi.STEP_ZERO_GRADS if zero else 0, between=between, repeat=2 if zero else 1)
    assert err == 0
    np.testing.assert_allclose(sp, ref_sp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sn, ref_sn, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss[1:], ref_per, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss[0], ref_loss, rtol=1e-5, atol=1e-6)
    grads.update(seen['readout'])
    done = set()
    for k, p in params.items():
        if id(p) in done:
            continue
        done.add(id(p))
        ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
        np.testing.assert_allclose(grads[k], ref, rtol=1e-4, atol=2e-6, err_msg=k)
    # forward only: scores and loss from the caller's embeddings, no gradient touched
    def forward_only(finals, levels):
        with torch.no_grad():
            hs = [torch.cat([torch.from_numpy(x) for x in lv[1:]], dim=1) if readout == 'concat' else torch.from_numpy(lv[-1])
                  for lv in levels]
            yield torch.cat([ref_cpu.readout(cfg['readout'], cfg['scatter_op'], params, h,
                                             torch.as_tensor(b['col']['batch']), b['col']['B'], b['col']['N'], b['col']['A'])
                             for b, h in zip(batches, hs)], dim=0).numpy()
    loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, backward=0,
                                        between=forward_only)
    assert err == 0
    np.testing.assert_allclose(sp, ref_sp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss[0], ref_loss, rtol=1e-5, atol=1e-6)
    assert all(not g.any() for g in grads.values())


@pytest.mark.parametrize('readout,scatter_op,adaptive,shared,wd', [
    ('mlp', 'add', True, False, 1e-3), ('mlp', 'max', False, True, 0), ('targetmlp', 'add', True, True, 0),
    ('targetmlp', 'mean', False, False, 1e-3), ('concat', 'add', False, False, 0), ('concat', 'max', False, True, 1e-3)])
def test_fused_step_with_learned_readout_matches_oracle(be, readout, scatter_op, adaptive, shared, wd):
    """MLPReadout / TargetMLPReadout / the concat input (reference model.py:441-446, 497-553) INSIDE the one-call step:
    gather, Linear - ReLU - Linear on the library's dense-layer kernels, the reduction over each graph's rows, scores,
    and all of it backward, with the readout's regulariser (model.py:486-490). Against the oracle's whole model."""
    D, margin = 32, 1.0
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(23, D, 3, shared, MIXES['all7'], readout, adaptive)
    cfg['scatter_op'], cfg['weight_decay'] = scatter_op, wd
    total, per, sp_ref, sn_ref = 0, [], [], []
    for b in batches:             # margin_loss per batch (regulariser included), weighted: train_helpers.py:76-120
        q = ref_cpu.encode_queries(params, cfg, node_map, b['formula'], b['col'])
        pos = ref_cpu.score(params, node_map, b['formula'], q, b['targets'])
        neg = ref_cpu.score(params, node_map, b['formula'], q, b['negs'])
        l = torch.clamp(margin - (pos - neg), min=0).mean()
        per.append(l.item())
        if wd > 0:
            l = l + wd * sum(torch.norm(v) for k, v in params.items() if k.startswith('readout.'))
        total = total + b['weight'] * l
        sp_ref.append(pos.detach().numpy())
        sn_ref.append(neg.detach().numpy())
    total.backward()
    for zero, repeat in ((True, 2), (False, 1)):
        loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin,
                                            flags=_capi.STEP_ZERO_GRADS if zero else 0, repeat=repeat)
        assert err == 0
        np.testing.assert_allclose(sp, np.concatenate(sp_ref), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sn, np.concatenate(sn_ref), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(loss[1:], per, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(loss[0], total.item(), rtol=1e-5, atol=1e-6)
        done = set()
        for k, p in params.items():
            if id(p) in done:
                continue
            done.add(id(p))
            ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
            np.testing.assert_allclose(grads[k], ref, rtol=1e-4, atol=2e-6, err_msg=k)
    loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, backward=0)
    np.testing.assert_allclose(loss[0], total.item(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sp, np.concatenate(sp_ref), rtol=1e-5, atol=1e-6)
    assert err == 0 and all(not g.any() for g in grads.values())


@pytest.mark.parametrize('readout,D,mix,scatter_op,adaptive,shared,wd,touch', [
    ('mlp', 64, 'all7', 'add', True, False, 1e-3, 'step'), ('mlp', 64, 'all7', 'max', False, True, 0, False),
    ('mlp', 64, 'many', 'mean', True, False, 1e-3, 'pack'), ('mlp', 128, 'tiny', 'add', True, False, 0, 'step'),
    ('mlp', 128, 'dup', 'max', False, False, 1e-3, 'step'),
    ('targetmlp', 64, 'all7', 'add', True, False, 1e-3, 'step'), ('targetmlp', 64, 'all7', 'max', False, True, 0, False),
    ('targetmlp', 64, 'many', 'mean', True, False, 1e-3, 'pack'), ('targetmlp', 128, 'tiny', 'mean', True, False, 0, 'step'),
    ('targetmlp', 128, 'dup', 'max', False, False, 1e-3, 'step'),
    ('concat', 128, 'all7', 'add', False, False, 1e-3, 'step'), ('concat', 128, 'dup', 'max', False, True, 0, 'pack'),
    ('concat', 128, 'tiny', 'mean', False, False, 0, False), ('concat', 64, 'many', 'add', False, False, 1e-3, 'step')])
def test_mlp_readout_on_the_chain_matches_oracle(be, capfd, readout, D, mix, scatter_op, adaptive, shared, wd, touch):
    _gpu_only_when_heavy(be, (readout == 'concat' and (mix, D) in (('all7', 128), ('many', 64), ('dup', 128))) or
                         (D == 128 and mix == 'dup' and readout == 'targetmlp'))
    """The learned readouts inside the chain launch against the oracle's whole model (see _readout_on_the_chain)."""
    _readout_on_the_chain(be, capfd, readout, D, mix, scatter_op, adaptive, shared, wd, touch, 3)


@pytest.mark.parametrize('readout,D,mix,L', [('concat', 128, 'd1', 1), ('concat', 64, 'dup', 2), ('targetmlp', 64, 'd1', 1),
                                             ('mlp', 128, 'dup', 2)])
def test_learned_readout_on_the_chain_with_fewer_layers(be, capfd, readout, D, mix, L):
    """The same with one and two message-passing layers (concat: one and two column blocks of its first Linear layer; the
    readout's virtual layers sit right behind the model's)."""
    _readout_on_the_chain(be, capfd, readout, D, mix, 'add', False, False, 1e-3, 'step', L)


@pytest.mark.parametrize('readout', ['mlp', 'concat'])
def test_learned_readout_with_fewer_passes_than_the_diameter(be, capfd, readout):
    """Fewer passes than a query's diameter leave node states batch-uniform after the last pass; a learned readout's weight
    gradient reads every node's rows of H[L], so the planner keeps every state of such a step per graph (the pre-pass off,
    as concat always has it) and the step stays on the chain form: same loss, scores and gradients as the oracle's."""
    D, margin = 64, 1.0
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(37, D, 2, False, EDGE_MIXES['tiny'], readout, False)
    cfg['weight_decay'] = 1e-3
    total, sp_ref = 0, []
    for b in batches:
        q = ref_cpu.encode_queries(params, cfg, node_map, b['formula'], b['col'])
        pos = ref_cpu.score(params, node_map, b['formula'], q, b['targets'])
        neg = ref_cpu.score(params, node_map, b['formula'], q, b['negs'])
        l = torch.clamp(margin - (pos - neg), min=0).mean()
        l = l + 1e-3 * sum(torch.norm(v) for k, v in params.items() if k.startswith('readout.'))
        total = total + b['weight'] * l
        sp_ref.append(pos.detach().numpy())
    total.backward()
    be.lib.mpqe_debug_option(b'DUMP_PLAN', 1, 1)
    try:
        capfd.readouterr()
        loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin)
        assert 'plan: chain 1 uniform 0' in capfd.readouterr().err
    finally:
        be.lib.mpqe_debug_option(b'DUMP_PLAN', 0, 0)
    assert err == 0
    np.testing.assert_allclose(sp, np.concatenate(sp_ref), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss[0], total.item(), rtol=1e-5, atol=1e-6)
    done = set()
    for k, p in params.items():
        if id(p) in done:
            continue
        done.add(id(p))
        ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
        np.testing.assert_allclose(grads[k], ref, rtol=1e-4, atol=2e-6, err_msg=k)


def _readout_on_the_chain(be, capfd, readout, D, mix, scatter_op, adaptive, shared, wd, touch, L):
    """MLPReadout / TargetMLPReadout (reference model.py:497-553) on the CHAIN form: Linear - ReLU - Linear are two more
    levels of every graph block's programme (the node's own row times W^T -- targetmlp: the target's row times the first
    column block of W_0 plus the node's times the second --, ReLU bits in LDS), the reduction over a graph's rows is the
    score phase's sum / max over the slots that have a row, the weight gradients are tiles with the operands swapped
    (nn.Linear stores [out, in]; targetmlp's [D, 2 D] as two column blocks), the bias gradients column sums like the
    layers'. Loss, scores and every gradient against the oracle's whole model; the level form (MPQE_STEP_NO_CHAIN) gives
    the same."""
    _gpu_only_when_heavy(be, D >= 128 and mix == 'dup')          # (D = 128 on the emulator: the 'tiny' / 'd1' mixes)
    margin = 1.0
    mixes = dict(MIXES, d1=[('1-chain', 20, 1.0), ('2-inter', 37, 0.5)], **EDGE_MIXES)      # (d1: diameter 1)
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(31, D, L, shared, mixes[mix], readout, adaptive)
    cfg['scatter_op'], cfg['weight_decay'] = scatter_op, wd
    total, per, sp_ref, sn_ref = 0, [], [], []
    for b in batches:
        q = ref_cpu.encode_queries(params, cfg, node_map, b['formula'], b['col'])
        pos = ref_cpu.score(params, node_map, b['formula'], q, b['targets'])
        neg = ref_cpu.score(params, node_map, b['formula'], q, b['negs'])
        l = torch.clamp(margin - (pos - neg), min=0).mean()
        per.append(l.item())
        if wd > 0:
            l = l + wd * sum(torch.norm(v) for k, v in params.items() if k.startswith('readout.'))
        total = total + b['weight'] * l
        sp_ref.append(pos.detach().numpy())
        sn_ref.append(neg.detach().numpy())
    total.backward()
    be.lib.mpqe_debug_option(b'DUMP_PLAN', 1, 1)
    try:
        capfd.readouterr()
        first = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, flags=_capi.STEP_ZERO_GRADS, touch=touch,
                         repeat=2)
        assert 'plan: chain 1' in capfd.readouterr().err
    finally:
        be.lib.mpqe_debug_option(b'DUMP_PLAN', 0, 0)
    # the prologue's items behind the chain workgroups, or in front (a launch order the library picks by the step's size; the
    # host emulator runs workgroups in order and keeps the prologue in front)
    for order in (0, 1):
        be.lib.mpqe_debug_option(b'PROLOGUE_LAST', order, 1)
        try:
            other = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, flags=_capi.STEP_ZERO_GRADS, touch=touch)
        finally:
            be.lib.mpqe_debug_option(b'PROLOGUE_LAST', 0, 0)
        assert other[4] == 0
        for a, b in zip(other[:3], first[:3]):
            np.testing.assert_array_equal(a, b)
        for k in first[3]:
            if touch or not k.startswith('enc.'):      # (table gradients by atomics: their order is the hardware's)
                np.testing.assert_array_equal(other[3][k], first[3][k], err_msg=k)
    for what, (loss, sp, sn, grads, err) in (
            ('zeroed', first),
            ('accumulate', run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, touch=touch)),
            ('level form', run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, flags=_capi.STEP_NO_CHAIN))):
        assert err == 0, what
        np.testing.assert_allclose(sp, np.concatenate(sp_ref), rtol=1e-5, atol=1e-6, err_msg=what)
        np.testing.assert_allclose(sn, np.concatenate(sn_ref), rtol=1e-5, atol=1e-6, err_msg=what)
        np.testing.assert_allclose(loss[1:], per, rtol=1e-5, atol=1e-6, err_msg=what)
        np.testing.assert_allclose(loss[0], total.item(), rtol=1e-5, atol=1e-6, err_msg=what)
        done = set()
        for k, p in params.items():
            if id(p) in done:
                continue
            done.add(id(p))
            ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
            np.testing.assert_allclose(grads[k], ref, rtol=1e-4, atol=2e-6, err_msg='%s %s' % (what, k))
    loss, sp, sn, grads, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, backward=0, repeat=2)
    np.testing.assert_allclose(loss[0], total.item(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sp, np.concatenate(sp_ref), rtol=1e-5, atol=1e-6)
    assert err == 0 and all(not g.any() for g in grads.values())
    # the regulariser from norms formed once (mpqe_step_readout_norms) instead of a launch per forward-only call: the same bits
    lossn, spn, snn, _g, errn = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin, backward=0, repeat=2,
                                         readout_norms=True)
    assert errn == 0
    np.testing.assert_array_equal(lossn, loss)
    np.testing.assert_array_equal(spn, sp)
    # forward-only and whole steps on ONE packed step, in turn: a forward-only call makes only the transposed copies its
    # readout's forward multiplies by, and counts the others' workgroups in (the chain workgroups' wait target advances alike
    # in every launch); mpqe_debug_option FWD_ALL_COPIES = 1: every copy, as until round 5
    # (the CPU emulator: the short sequence at D = 64 only -- the same code, a fifth of the time)
    if be.name == 'emu' and D > 64:
        return
    for every in ((0,) if be.name == 'emu' else (0, 1)):
        be.lib.mpqe_debug_option(b'FWD_ALL_COPIES', every, 1)
        try:
            mixed = run_step(be, schema, mode_ids, params, node_map, cfg, batches, margin,
                             backward=[0, 1, 0] if be.name == 'emu' else [0, 1, 0, 0, 1, 0],
                             flags=_capi.STEP_ZERO_GRADS, touch=touch)
        finally:
            be.lib.mpqe_debug_option(b'FWD_ALL_COPIES', 0, 0)
        assert mixed[4] == 0
        np.testing.assert_array_equal(mixed[0], loss)
        np.testing.assert_array_equal(mixed[1], sp)
        for k in first[3]:
            if touch or not k.startswith('enc.'):
                np.testing.assert_array_equal(mixed[3][k], first[3][k], err_msg=k)


def test_callers_readout_refuses_what_it_does_not_cover(be):
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(5, 64, 2, False, MIXES['dup'], 'sum', False)
    with pytest.raises(_capi.MpqeError):        # a phase of the three-call step with a readout of the library's own
        run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, backward=_capi.STEP_PHASE_STATES, touch=False)
    cfg2 = dict(cfg, readout=_capi.READOUT_CALLER)
    with pytest.raises(_capi.MpqeError):        # the caller's readout in the one-call step
        run_step(be, schema, mode_ids, params, node_map, cfg2, batches, 1.0, touch=False)


@pytest.mark.parametrize('readout,adaptive,L', [('mp', True, 3), ('mp', False, 2), ('max', False, 3)])
def test_fused_step_pruning_changes_nothing(be, readout, adaptive, L):
    """Node states that cannot reach the readout are skipped by default (MPQE_STEP_NO_PRUNE computes them
    all, as the reference does): same scores bit for bit, same gradients (the skipped terms are exact
    zeros; only the weight-gradient chunking differs)."""
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(
        13, 32, L, False, MIXES['all7'], readout, adaptive)
    ref_loss, ref_per, ref_sp, ref_sn = oracle_step(params, cfg, node_map, batches, 1.0)
    full = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_NO_PRUNE)
    got = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0)
    np.testing.assert_array_equal(got[0], full[0])
    np.testing.assert_array_equal(got[1], full[1])
    np.testing.assert_array_equal(got[2], full[2])
    for k in full[3]:
        np.testing.assert_allclose(got[3][k], full[3][k], rtol=1e-5, atol=1e-7, err_msg=k)
        ref = np.zeros_like(got[3][k]) if params[k].grad is None else params[k].grad.numpy()
        np.testing.assert_allclose(got[3][k], ref, rtol=1e-4, atol=2e-6, err_msg=k)
    assert got[4] == 0 and full[4] == 0


CHAIN_MIX = [('3-chain', 21, 1.0), ('3-inter_chain', 16, 0.5), ('1-chain', 35, 0.25), ('3-chain_inter', 7, 2.0),
             ('2-inter', 48, 0.1)]


@pytest.mark.parametrize('D,readout,adaptive,shared,L', [(64, 'mp', True, False, 3), (64, 'sum', False, True, 2),
                                                         (128, 'mp', True, False, 3), (128, 'max', False, False, 3),
                                                         (256, 'mp', True, True, 3)])
def test_fused_step_chain_kernels(be, D, readout, adaptive, shared, L):
    _gpu_only_when_heavy(be, (D, readout) in ((128, 'max'), (256, 'mp')))
    """The graph-block chain kernels (all levels of 16 graphs in one workgroup; D = 64 / 128 / 256) against
    the oracle and against the one-launch-per-level form (MPQE_STEP_NO_CHAIN): ragged batch sizes, every
    readout family, pruned and unpruned."""
    mix = CHAIN_MIX if D < 256 else CHAIN_MIX[:2]
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(17, D, L, shared, mix, readout, adaptive)
    ref_loss, ref_per, ref_sp, ref_sn = oracle_step(params, cfg, node_map, batches, 1.0)
    got = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0)
    lev = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_NO_CHAIN)
    runs = [got, lev]
    # every node state as per-graph rows (the default treats states no anchor has reached yet as one vector per batch)
    runs.append(run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_NO_UNIFORM))
    every = True
    if every or readout == 'mp':
        # Where the weight-gradient tiles + post-pass run. A step this small takes the MERGED launch by default (workgroups
        # of the chain launch: include/mpqe_amd.h MPQE_STEP_MERGE_TAIL); the benchmarked step is larger and takes the
        # SPLIT form (a launch of their own, the post-pass as closures: step_closure.h) -- forced here: the same gradients
        # (other kernels for the batch-uniform part: equal within rounding, not bit for bit). Then the merged form forced, three
        # runs on one descriptor buffer -- the later runs' counters start from the earlier ones' (targets are epoch x
        # count) -- with the call's own zero fill (the whole-root rule).
        split = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_SPLIT_TAIL)
        runs.append(split)
        for k in got[3]:
            np.testing.assert_allclose(split[3][k], got[3][k], rtol=1e-5, atol=1e-7, err_msg=k)
        # where the loss and the entity-table rows of the split form run: in the reduction launch (default), or as trailing
        # workgroups of the weight-gradient launch (EARLY_ROWS: table_sum_multi, a range of sorted positions per workgroup;
        # measured slower, kept as a switch) -- the same additions in the same order, bit for bit
        # ... and the reduction's table workgroups taking a range of positions each (ROWS_MULTI; default: one run each)
        # ... and the post-pass alone riding in the chain launch (POST_IN_CHAIN: the tiles stay a launch of their own)
        # ... and the table workgroups taking every sorted position (NO_RUNS) instead of the compacted run starts (default)
        exps = bool(be.lib.mpqe_debug_has_experiments())      # (the forms taken out of the shipped library: -DMPQE_EXPERIMENTS)
        for opt in ((b'EARLY_ROWS', b'ROWS_MULTI', b'POST_IN_CHAIN', b'NO_RUNS') if exps else (b'NO_RUNS',)):
            be.lib.mpqe_debug_option(opt, 1, 1)
            try:
                other = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_SPLIT_TAIL)
            finally:
                be.lib.mpqe_debug_option(opt, 0, 0)
            np.testing.assert_array_equal(split[0], other[0])
            for k in got[3]:
                np.testing.assert_array_equal(split[3][k], other[3][k], err_msg='%s %s' % (opt, k))
        runs.append(run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0,
                             flags=_capi.STEP_SPLIT_TAIL | _capi.STEP_ZERO_GRADS, repeat=2))
        if exps:
            be.lib.mpqe_debug_option(b'POST_IN_CHAIN', 1, 1)
            try:        # (three runs on one descriptor buffer: the counters' targets are epoch x count; the whole-root rule)
                runs.append(run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0,
                                     flags=_capi.STEP_SPLIT_TAIL | _capi.STEP_ZERO_GRADS, repeat=3))
            finally:
                be.lib.mpqe_debug_option(b'POST_IN_CHAIN', 0, 0)
        runs.append(run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0,
                             flags=_capi.STEP_MERGE_TAIL | _capi.STEP_ZERO_GRADS, repeat=3))
        # the reduction as trailing workgroups of the weight-gradient launch (two launches per step in the split form):
        # accumulate and zero-fill modes, three runs on one descriptor buffer (the arrival counter is re-armed per step)
        if exps:
            be.lib.mpqe_debug_option(b'FUSE_TAIL', 1, 1)
            try:
                fz = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0,
                              flags=_capi.STEP_SPLIT_TAIL | _capi.STEP_ZERO_GRADS, repeat=3)
                fa = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_SPLIT_TAIL)
            finally:
                be.lib.mpqe_debug_option(b'FUSE_TAIL', 0, 0)
            runs += [fz, fa]
            for k in fz[3]:          # (same tiles, same order of additions: zero fill + store == accumulate into zeros)
                np.testing.assert_array_equal(fz[3][k], fa[3][k], err_msg=k)
    if every:
        # entity-table gradients by fp32 atomics instead of the per-row sums of the touch plan
        runs.append(run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, touch=False))
        again = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0)
        for k in got[3]:          # with the touch plan EVERY gradient is bit-reproducible, the entity tables included
            np.testing.assert_array_equal(again[3][k], got[3][k], err_msg=k)
        # the plan built at pack time (mpqe_step_touch_build) instead of inside the step: the same plan, the same bits
        plans = [[], []]
        packed = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, touch='pack', plan_out=plans[0])
        run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, touch='step', plan_out=plans[1], repeat=2)
        runs.append(packed)
        for k in got[3]:
            np.testing.assert_array_equal(packed[3][k], got[3][k], err_msg=k)
        M = sum(b['B'] * (len(b['formula'].anchor_modes) + 2) for b in batches)
        al = lambda n: (n + 255) // 256 * 256
        np.testing.assert_array_equal(plans[0][0][:64], plans[1][0][:64])                       # header
        np.testing.assert_array_equal(plans[0][0][256:256 + 8 * M], plans[1][0][256:256 + 8 * M])   # sorted keys
        o = 256 + al(8 * M)
        np.testing.assert_array_equal(plans[0][0][o:o + 4 * M], plans[1][0][o:o + 4 * M])       # perm
    if D == 64:
        runs.append(run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_NO_PRUNE))
    if D == 128:      # the form whose waves own 32 columns and all of K (the default splits K between wave pairs)
        runs.append(run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_NO_KSPLIT))
        # eight waves per workgroup (the row-major phases stay with the first four)
        if be.name == 'hip' or readout == 'mp':
            runs.append(run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_EIGHT_WAVES))
    for r in runs:
        assert r[4] == 0
        np.testing.assert_allclose(r[1], ref_sp, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r[2], ref_sn, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r[0][0], ref_loss, rtol=1e-5, atol=1e-6)
        seen = set()
        for k, p in params.items():
            if id(p) in seen:
                continue
            seen.add(id(p))
            ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
            np.testing.assert_allclose(r[3][k], ref, rtol=1e-4, atol=2e-6, err_msg=k)
    # forward only
    fwd = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, backward=0)
    np.testing.assert_array_equal(fwd[0], got[0])


@pytest.mark.parametrize('D,shared', [(32, False), (64, True)])
def test_fused_step_zero_grads_flag(be, D, shared):
    """MPQE_STEP_ZERO_GRADS: the call clears the gradient buffers itself (level path and chain path)."""
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(19, D, 3, shared, MIXES['dup'], 'mp', True)
    ref = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0)
    got = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_ZERO_GRADS)
    np.testing.assert_array_equal(got[0], ref[0])
    for k in ref[3]:
        if k.startswith('enc.'):
            np.testing.assert_allclose(got[3][k], ref[3][k], rtol=1e-5, atol=1e-7, err_msg=k)    # atomics
        else:
            np.testing.assert_array_equal(got[3][k], ref[3][k], err_msg=k)


@pytest.mark.parametrize('splits', [[0, 3, 7], [0, 1, 2, 4, 7], [0, 6, 7]])
def test_fused_step_stream_lanes(be, splits):
    """Lanes (batches split over streams) change scheduling only: same loss, scores and gradients."""
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(
        11, 32, 3, False, MIXES['all7'], 'mp', True)
    ref = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0)
    got = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, lanes=splits)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])
    for k in ref[3]:
        if k.startswith('layers') or k.startswith('mode'):
            np.testing.assert_array_equal(got[3][k], ref[3][k], err_msg=k)      # fixed-order reductions
        else:
            np.testing.assert_allclose(got[3][k], ref[3][k], rtol=1e-5, atol=1e-7, err_msg=k)   # atomics
    assert got[4] == 0
    # forward-only with lanes
    fwd = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, backward=0, lanes=splits)
    np.testing.assert_array_equal(fwd[0], ref[0])


EDGE_MIXES = {
    'tiny': [('3-chain', 1, 1.0), ('2-inter', 3, 0.5)],                       # one partial block per batch
    'one': [('3-inter_chain', 15, 1.0)],                                      # a single batch, 15 of 16 rows
    'many': [(qt, 17 + i, 0.1 * (i + 1)) for i, qt in enumerate(
        ['1-chain', '2-chain', '3-chain', '2-inter', '3-inter', '3-inter_chain', '3-chain_inter'] * 2)],   # 14 batches
}


@pytest.mark.parametrize('mix,D,readout,adaptive,L', [('tiny', 128, 'mp', True, 3), ('one', 64, 'max', False, 3),
                                                      ('many', 64, 'mp', True, 3), ('tiny', 64, 'mp', False, 2),
                                                      ('many', 128, 'mp', False, 2)])
def test_fused_step_chain_edge_shapes(be, mix, D, readout, adaptive, L):
    """Chain form at the edges: partial blocks (B < 16), one batch, 14 batches that share relation matrices
    (reduction groups next to directly written matrices), and TM with fewer passes than the diameter (anchors
    that cannot reach the target: dead at level 0, zero table gradient). The gradient buffers start as garbage
    (MPQE_STEP_ZERO_GRADS), so every untouched matrix must come back zero."""
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(23, D, L, False, EDGE_MIXES[mix], readout,
                                                                             adaptive)
    ref_loss, ref_per, ref_sp, ref_sn = oracle_step(params, cfg, node_map, batches, 1.0)
    got = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, flags=_capi.STEP_ZERO_GRADS)
    assert got[4] == 0
    np.testing.assert_allclose(got[1], ref_sp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[2], ref_sn, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[0][0], ref_loss, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[0][1:], ref_per, rtol=1e-5, atol=1e-6)
    for k, p in params.items():
        ref = np.zeros(tuple(p.shape), np.float32) if p.grad is None else p.grad.numpy()
        np.testing.assert_allclose(got[3][k], ref, rtol=1e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('splits', [[0, 2, 5], [0, 1, 4, 5]])
def test_fused_step_chain_with_stream_lanes(be, splits):
    """Chain form with lanes: every lane launches its own chain kernel and weight-gradient kernel; the result
    does not depend on the split."""
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(17, 64, 3, False, CHAIN_MIX, 'mp', True)
    ref = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0)
    got = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, lanes=splits)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])
    for k in ref[3]:
        if k.startswith('layers') or k.startswith('mode'):
            np.testing.assert_array_equal(got[3][k], ref[3][k], err_msg=k)
        else:
            np.testing.assert_allclose(got[3][k], ref[3][k], rtol=1e-5, atol=1e-7, err_msg=k)
    fwd = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, backward=0, lanes=splits)
    np.testing.assert_array_equal(fwd[0], ref[0])


def test_fused_step_rejects_bad_lanes(be):
    P = _capi.StepParams()
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(
        7, 16, 2, False, [('2-inter', 6, 1.0), ('1-chain', 6, 1.0)], 'sum', False)
    for bad in ([0, 0, 2], [0, 2, 2], [1, 1, 2], [0, 1, 3]):
        with pytest.raises(_capi.MpqeError):
            run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, lanes=bad)


def test_fused_step_equals_reference_two_pass_loss(be):
    """margin_loss as the reference runs it (two full encoder passes, model.py:478-482) gives the
    loss the fused single-pass step reports."""
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(
        5, 16, 3, False, [('3-chain_inter', 21, 1.0)], 'mp', True)
    b = batches[0]
    ref = ref_cpu.margin_loss(params, cfg, node_map, b['formula'], b['col'], b['targets'], b['negs'],
                              encode_twice=True)
    loss, _, _, _, _ = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0, backward=0)
    np.testing.assert_allclose(loss[0], ref.item(), rtol=1e-5, atol=1e-6)


def test_fused_step_flags_bad_entity(be):
    schema, mode_ids, rel_ids, params, node_map, cfg, batches = make_problem(
        7, 16, 2, False, [('2-inter', 6, 1.0)], 'sum', False)
    batches[0]['targets'][2] = schema.num_entities      # maps to -1 in the LUT
    _, _, _, _, err = run_step(be, schema, mode_ids, params, node_map, cfg, batches, 1.0)
    assert err & _capi.FLAG_BAD_NODE_ID


def test_fused_step_rejects_bad_descriptors(be):
    P = _capi.StepParams()
    SB = (_capi.StepBatch * 1)()
    assert be.lib.mpqe_step_workspace_bytes(ctypes.byref(P), SB, 1, None) == 0          # dim 0
    assert be.lib.mpqe_step_workspace_bytes(ctypes.byref(P), SB, 17, None) == 0         # > MAX_BATCHES


@pytest.mark.parametrize('sizes', [(9, 5), (700, 333), (3000, 1111), (11000, 2500), (45000, 10000), (92000, 21000)])
def test_touch_plan_one_launch_equals_library_sort(be, sizes):
    """The touch plan built in ONE launch (keys, a grid-synchronised stable 8-bit LSD radix sort with one entry per thread,
    inverse permutation: csrc/step_touch.h touch_sort_kernel; plans of up to 65 536 looked-up ids) is byte for byte the plan
    of the keys kernel + the library's multi-launch radix sort (csrc/radix_sort.h; mpqe_debug_option TOUCH_MULTI_LAUNCH) -- a stable sort has one answer.
    1 to 256 workgroups of 2048 ids, bad ids included. (The host emulator runs workgroups one after the other and has no grid barrier:
    there both builds take the library-sort stand-in and the test only pins the layout.)"""
    rng = np.random.RandomState(sizes[0])
    nmodes, rows_per = 3, [70, 40000, 130]
    node_map = np.full(sum(rows_per) + 1, -1, np.int64)
    ids_of, o = [], 0
    perm = rng.permutation(sum(rows_per))
    for m, r in enumerate(rows_per):
        ids = np.sort(perm[o:o + r])
        node_map[ids] = np.arange(r)
        ids_of.append(ids)
        o += r
    B1, B2 = sizes
    SB = (_capi.StepBatch * 2)()
    SB[0] = _capi.make_step_batch('3-inter', 1, B1, [0, 0, 0], [0], [0, 1, 1], 2, 1.0)
    SB[1] = _capi.make_step_batch('1-chain', 1, B2, [0], [0], [1], 1, 1.0)
    pick = lambda m, n: ids_of[m][rng.randint(len(ids_of[m]), size=n)]
    anchors = np.concatenate([pick(0, B1), pick(1, B1), pick(1, B1), pick(1, B2)])
    targets = np.concatenate([pick(2, B1), pick(1, B2)])
    negs = np.concatenate([pick(2, B1), pick(1, B2)])
    negs[3] = sum(rows_per)                  # an id of no mode
    anchors[1] = -7                          # out of range
    P = _capi.make_step_params(64, 4, 'mp', [0] * nmodes, [r + 1 for r in rows_per], 0, len(node_map), 0, [0], [0], [0])
    d_nm = be.put(node_map)
    P.node_map = be.ptr(d_nm)
    d_a, d_t, d_n = be.put(anchors), be.put(targets), be.put(negs)
    tb = be.lib.mpqe_step_touch_bytes(ctypes.byref(P), SB, 2)
    twb = be.lib.mpqe_step_touch_workspace_bytes(ctypes.byref(P), SB, 2)
    got = []
    for force_library in (False, True, False):
        be.lib.mpqe_debug_option(b'TOUCH_MULTI_LAUNCH', 1, 1 if force_library else 0)
        try:
            tbuf, twbuf = be.nbytes(tb + 256), be.nbytes(twb + 256)
            if be.name == 'emu':
                tbuf.fill(0)
                twbuf.fill(0xa5)
            else:
                tbuf.zero_()
                twbuf.fill_(0xa5)             # (the workspace arrives as garbage)
            tptr = (be.ptr(tbuf) + 255) // 256 * 256
            be.check(be.lib.mpqe_step_touch_build(ctypes.byref(P), SB, 2, be.ptr(d_a), be.ptr(d_t), be.ptr(d_n), tptr, tb,
                                                  (be.ptr(twbuf) + 255) // 256 * 256, twb, be.stream), 'touch')
            raw = np.asarray(be.get(tbuf)).view(np.uint8)
            off = tptr - be.ptr(tbuf)
            got.append(raw[off:off + tb].copy())
        finally:
            be.lib.mpqe_debug_option(b'TOUCH_MULTI_LAUNCH', 0, 0)
    M = int(be.lib.mpqe_step_touch_entries(SB, 2))
    assert M == 5 * B1 + 3 * B2
    # header | keys [M] u64 | perm [M] i32 | erow [M] i32, each region 256-byte aligned: compare the used bytes
    al = lambda n: (n + 255) // 256 * 256
    o_keys, o_pos = 256, 256 + al(8 * M)
    o_erow = o_pos + al(4 * M)
    for other in (got[1], got[2]):
        np.testing.assert_array_equal(got[0][:64], other[:64])
        for name, lo, n in (('keys', o_keys, 8 * M), ('perm', o_pos, 4 * M), ('erow', o_erow, 4 * M)):
            np.testing.assert_array_equal(got[0][lo:lo + n], other[lo:lo + n], err_msg=name)
    keys = got[0][o_keys:o_keys + 8 * M].view(np.uint64)
    assert (np.diff(keys.astype(np.float64)) >= 0).all()
    assert int((keys == np.uint64(2 ** 64 - 1)).sum()) == 2          # the two bad ids, sorted to the end
    perm_got = got[0][o_pos:o_pos + 4 * M].view(np.int32)
    # ... and against numpy: key of every entry (table << row_bits | row; all ones for a bad id), stable argsort
    tabs = np.concatenate([np.full(B1, 0), np.full(B1, 1), np.full(B1, 1), np.full(B2, 1), np.full(B1, 2), np.full(B2, 1),
                           np.full(B1, 2), np.full(B2, 1)])
    ids_all = np.concatenate([anchors, targets, negs])
    ok = (ids_all >= 0) & (ids_all < len(node_map))
    rows = np.where(ok, node_map[np.clip(ids_all, 0, len(node_map) - 1)], -1)
    row_bits = int(np.frombuffer(got[0][8:12].tobytes(), np.int32)[0])
    want_keys = np.where(rows >= 0, (tabs.astype(np.uint64) << np.uint64(row_bits)) | rows.astype(np.uint64),
                         np.uint64(2 ** 64 - 1))
    order = np.argsort(want_keys, kind='stable')
    np.testing.assert_array_equal(perm_got, order.astype(np.int32))
    np.testing.assert_array_equal(keys, want_keys[order])


def test_adam_rows_step_equals_torch_sparse_adam(be):
    """mpqe_adam_rows_step on the rows of a touch plan == torch.optim.SparseAdam fed the same per-row gradients (three
    steps, fresh gradients each), and rows outside the plan are not touched at all."""
    rng = np.random.RandomState(5)
    D, nmodes = 64, 3
    rows_per = [7, 40, 13]
    node_map = np.full(sum(rows_per) + 1, -1, np.int64)
    ids_of = []
    perm = rng.permutation(sum(rows_per))
    o = 0
    for m, r in enumerate(rows_per):
        ids = np.sort(perm[o:o + r])
        node_map[ids] = np.arange(r)
        ids_of.append(ids)
        o += r
    # one 2-inter batch (anchors of modes 0 and 1, targets of mode 2) and one 1-chain batch (anchor mode 1, target mode 1)
    B1, B2 = 9, 5
    SB = (_capi.StepBatch * 2)()
    SB[0] = _capi.make_step_batch('2-inter', 1, B1, [0, 0], [0], [0, 1], 2, 1.0)
    SB[1] = _capi.make_step_batch('1-chain', 1, B2, [0], [0], [1], 1, 1.0)
    pick = lambda m, n: ids_of[m][rng.randint(len(ids_of[m]), size=n)]
    anchors = np.concatenate([pick(0, B1), pick(1, B1), pick(1, B2)])
    targets = np.concatenate([pick(2, B1), pick(1, B2)])
    negs = np.concatenate([pick(2, B1), pick(1, B2)])
    negs[3] = sum(rows_per)                  # an id of no mode: not a row
    tabs = [rng.randn(r + 1, D).astype(np.float32) for r in rows_per]
    P = _capi.make_step_params(D, 4, 'mp', [0] * nmodes, [r + 1 for r in rows_per], 0, len(node_map), 0, [0], [0], [0])
    d_nm = be.put(node_map)
    P.node_map = be.ptr(d_nm)
    d_a, d_t, d_n = be.put(anchors), be.put(targets), be.put(negs)
    tb = be.lib.mpqe_step_touch_bytes(ctypes.byref(P), SB, 2)
    twb = be.lib.mpqe_step_touch_workspace_bytes(ctypes.byref(P), SB, 2)
    tbuf, twbuf = be.nbytes(tb + 256), be.nbytes(twb + 256)
    tptr = (be.ptr(tbuf) + 255) // 256 * 256
    be.check(be.lib.mpqe_step_touch_build(ctypes.byref(P), SB, 2, be.ptr(d_a), be.ptr(d_t), be.ptr(d_n), tptr, tb,
                                          (be.ptr(twbuf) + 255) // 256 * 256, twb, be.stream), 'touch')
    touched = [set() for _ in range(nmodes)]
    for m, idl in ((0, anchors[:B1]), (1, anchors[B1:]), (2, targets[:B1]), (1, targets[B1:]), (2, negs[:B1]), (1, negs[B1:])):
        for i in idl:
            if 0 <= i < len(node_map) and node_map[i] >= 0:
                touched[m].add(int(node_map[i]))
    ref_p = [torch.nn.Parameter(torch.from_numpy(t.copy())) for t in tabs]
    ref_opt = torch.optim.SparseAdam(ref_p, lr=0.05, betas=(0.9, 0.99), eps=1e-6)
    d_p = [be.put(t) for t in tabs]
    d_m = [be.zeros(t.shape) for t in tabs]
    d_v = [be.zeros(t.shape) for t in tabs]
    arr = ctypes.c_void_p * nmodes
    for step in range(1, 4):
        grads = [rng.randn(*t.shape).astype(np.float32) for t in tabs]
        d_g = [be.put(g) for g in grads]
        be.check(be.lib.mpqe_adam_rows_step(tptr, be.lib.mpqe_step_touch_entries(SB, 2), arr(*[be.ptr(x) for x in d_p]), arr(*[be.ptr(x) for x in d_g]),
                                            arr(*[be.ptr(x) for x in d_m]), arr(*[be.ptr(x) for x in d_v]), nmodes, D,
                                            0.05, 0.9, 0.99, 1e-6, step, be.stream), 'adam rows')
        for m in range(nmodes):
            rows = torch.tensor(sorted(touched[m]), dtype=torch.long)
            ref_p[m].grad = torch.sparse_coo_tensor(rows[None], torch.from_numpy(grads[m])[rows], tabs[m].shape)
        ref_opt.step()
    for m in range(nmodes):
        got = be.get(d_p[m])
        np.testing.assert_allclose(got, ref_p[m].detach().numpy(), rtol=1e-5, atol=1e-6, err_msg='table %d' % m)
        untouched = sorted(set(range(tabs[m].shape[0])) - touched[m])
        np.testing.assert_array_equal(got[untouched], tabs[m][untouched])
        np.testing.assert_array_equal(be.get(d_m[m])[untouched], 0)
        state = ref_opt.state[ref_p[m]]
        np.testing.assert_allclose(be.get(d_m[m]), state['exp_avg'].numpy(), rtol=2e-5, atol=1e-7)       # (cancellation in g - m)
        np.testing.assert_allclose(be.get(d_v[m]), state['exp_avg_sq'].numpy(), rtol=2e-5, atol=1e-8)

"""Host-side logic of the mirror modules that needs no GPU: state_dict key parity with the
reference, id orderings, the integer outputs of get_query_graph, error behaviour, and that
the C-ABI library loads and exports every declared symbol."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from mpqe_amd import _capi, _lib
from tests.conftest import ROOT, build_model


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'mpqe_amd.h')).read()
    declared = set(re.findall(r'\b(mpqe_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_capi.PROTOTYPES), declared ^ set(_capi.PROTOTYPES)
    lib = _lib.load()                      # raises ImportError when the .so is missing/stale
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mpqe_abi_version() == 6
    assert lib.mpqe_status_string(-3) == b'workspace too small'


def test_size_queries_and_argument_validation_without_gpu():
    lib = _lib.load()
    assert lib.mpqe_rgcn_template_bwd_workspace_bytes(2, 512, 128, 128) > 0
    assert lib.mpqe_rgcn_plan_bytes(10, 20, 3) > 0
    assert lib.mpqe_scatter_workspace_bytes(10, 4) >= 16
    info = _capi.TemplateInfo()
    assert lib.mpqe_template_info(7, ctypes.byref(info)) == -1
    assert lib.mpqe_template_info(5, ctypes.byref(info)) == 0 and info.num_nodes == 4
    # null operands / bad sizes are refused before any launch
    assert lib.mpqe_readout_fwd(0, None, 4, 2, 0, 8, None, None, None) == -1
    assert lib.mpqe_hinge_fwd(None, None, 0, 1.0, None, None) == -1
    assert lib.mpqe_cosine_fwd(None, None, None, -1, 8, 1e-8, None, None) == -1


def test_state_dict_and_ids_match_reference(enc_case):
    c = enc_case
    model = build_model(c, 'cpu')          # strict load inside: same keys, same shapes
    assert set(model.state_dict().keys()) == set(c.params().keys())
    assert model.mode_ids == c.mode_ids
    assert model.rel_ids == c.rel_ids
    for i in range(1, c.cfg['num_layers']):
        assert (model.layers[i] is model.layers[0]) == bool(c.cfg['shared_layers'])


def test_get_query_graph_host_outputs(enc_case):
    from mpqe_amd.data_utils import RGCNQueryDataset
    c = enc_case
    anchor_ids, var_ids, g = RGCNQueryDataset.get_query_graph(c.formula, c.queries, c.rel_ids, c.mode_ids)
    assert anchor_ids.dtype == torch.int64 and var_ids.dtype == torch.int64
    np.testing.assert_array_equal(anchor_ids.numpy(), c.arrays['anchor_ids'])
    np.testing.assert_array_equal(var_ids.numpy(), c.arrays['var_ids'])
    E = g.template.E
    assert list(g.template.edge_type) == c.arrays['edge_type'][:E].tolist()
    assert g.num_nodes == c.arrays['batch'].shape[0]
    with pytest.raises(RuntimeError):
        g.edge_index                       # device tensors only exist after .to('cuda')


def test_constructor_errors_like_reference(enc_case):
    from mpqe_amd.model import RGCNConv, RGCNEncoderDecoder
    from tests.conftest import CaseGraph
    g = CaseGraph(enc_case)
    with pytest.raises(ValueError, match='Unknown readout function'):
        RGCNEncoderDecoder(g, None, readout='nope')
    with pytest.raises(ValueError, match='Unknown scatter op'):
        RGCNEncoderDecoder(g, None, scatter_op='nope')
    with pytest.raises(NotImplementedError):
        RGCNConv(8, 8, 3, num_bases=2)


def test_reset_parameters_bound():
    from mpqe_amd.model import RGCNConv
    conv = RGCNConv(16, 16, 5, 0)
    b = 1.0 / np.sqrt(5 * 16)
    for p in (conv.basis, conv.root, conv.bias):
        assert float(p.abs().max()) <= b
    assert tuple(conv.basis.shape) == (5, 16, 16) and conv.att is None


def test_ops_refuse_cpu_tensors():
    from mpqe_amd import ops
    with pytest.raises(RuntimeError, match='no CPU path'):
        ops.cosine(torch.zeros(2, 4), torch.zeros(2, 4))
    with pytest.raises(RuntimeError, match='no CPU path'):
        ops.readout('sum', torch.zeros(4, 4), 2, 2, 1)


def test_hard_negatives_only_for_intersections(enc_case):
    c = enc_case
    if 'inter' in c.query_type:
        pytest.skip('chain types only')
    model = build_model(c, 'cpu')
    with pytest.raises(Exception, match='Hard negative examples'):
        model.margin_loss(c.formula, c.queries, hard_negatives=True)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` starts its N ranks itself -- and must REFUSE (non-zero exit, no JSON line) when fewer
    than N GPUs are visible, instead of running one rank and claiming more. No GPU here: --gpus 2 is already too many.
    (Counting devices does not initialise the GPU runtime; the parent process never touches a GPU.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('two or more GPUs visible')
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert b'n_gpus' not in p.stdout
    assert b'GPU(s) visible' in p.stderr
    # a launcher's world size that disagrees with --gpus is refused too
    env2 = dict(env, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    q = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--steps', '1', '--warmup', '0'],
                       env=env2, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert q.returncode != 0 and b'WORLD_SIZE' in q.stderr


def test_graft_entry_expects_this_abi():
    """__graft_entry__.build() asserts the library's ABI version: keep the number there in step with the library's."""
    import os
    import re
    from mpqe_amd import _lib
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '__graft_entry__.py')).read()
    m = re.search(r'mpqe_abi_version\(\) == (\d+)', src)
    assert m and int(m.group(1)) == _lib.load().mpqe_abi_version()


def _choice_lists(rng, B):
    return [[int(v) for v in rng.randint(0, 10 ** 6, size=int(rng.choice([1, 2, 3, 5, 8, 17, 32, 64, 100, 1000, 1025])))]
            for _ in range(B)]


def test_host_random_choice_replays_pythons_stream():
    """mpqe_host_random_choice (include/mpqe_amd.h) over raw outputs of the interpreter's generator = the reference's
    `[random.choice(q.neg_samples) for q in queries]` (model.py:470-476): the same draws AND the same generator state
    afterwards, for ragged lists (powers of two: the rejection loop's worst case), one shared list (1-chain:
    graph.full_lists) and an empty list (random.choice raises IndexError)."""
    import random
    lib = _lib.load()
    host = _lib.load_pyhost()
    rng = np.random.RandomState(0)
    for trial in range(30):
        B = int(rng.randint(1, 700))
        lists = _choice_lists(rng, B)
        lens = np.array([len(l) for l in lists], dtype=np.int64)
        off = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        flat = np.array([v for l in lists for v in l], dtype=np.int64)
        random.seed(trial)
        ref = [random.choice(l) for l in lists]
        state = random.getstate()
        # (a) the CPython extension: rounds of getrandbits + the library's replay inside one call
        out = np.full(B, -1, dtype=np.int64)
        random.seed(trial)
        words = host.choice(random.getrandbits, lens.ctypes.data, 0, off.ctypes.data, flat.ctypes.data, B, out.ctypes.data)
        assert out.tolist() == ref and random.getstate() == state and words >= B
        # (a') the same with the generator's state read in place (csrc/host/pyhost.c: choice_mt), where the interpreter passed
        # the self test
        if host.mt_ok:
            out1 = np.full(B, -1, dtype=np.int64)
            random.seed(trial)
            w1 = host.choice_mt(random.getrandbits.__self__, lens.ctypes.data, 0, off.ctypes.data, flat.ctypes.data, B,
                                out1.ctypes.data)
            assert out1.tolist() == ref and random.getstate() == state and w1 == words
        # (b) the library routine alone, driven round by round through ctypes
        out2 = np.full(B, -1, dtype=np.int64)
        cur = np.zeros(2, dtype=np.int64)
        random.seed(trial)
        while cur[0] < B:
            n = B - int(cur[0])
            w = random.getrandbits(32 * n).to_bytes(4 * n, 'little')
            assert lib.mpqe_host_random_choice(w, n, lens.ctypes.data, 0, off.ctypes.data, flat.ctypes.data, B, cur.ctypes.data,
                                               out2.ctypes.data) == 0
        assert out2.tolist() == ref and random.getstate() == state
    # one list for every query (reference model.py:473-474)
    full = [int(v) for v in rng.randint(0, 10 ** 6, size=433)]
    arr = np.array(full, dtype=np.int64)
    random.seed(99)
    ref = [random.choice(full) for _ in range(512)]
    state = random.getstate()
    out = np.empty(512, dtype=np.int64)
    random.seed(99)
    host.choice(random.getrandbits, 0, len(full), 0, arr.ctypes.data, 512, out.ctypes.data)
    assert out.tolist() == ref and random.getstate() == state
    if host.mt_ok:
        out[:] = -1
        random.seed(99)
        host.choice_mt(random.getrandbits.__self__, 0, len(full), 0, arr.ctypes.data, 512, out.ctypes.data)
        assert out.tolist() == ref and random.getstate() == state
        with pytest.raises(RuntimeError):          # (not a random.Random: refused, nothing read)
            host.choice_mt(object(), 0, len(full), 0, arr.ctypes.data, 512, out.ctypes.data)
    # an empty list: IndexError, as random.choice([]) raises
    lens = np.array([3, 0, 2], dtype=np.int64)
    off = np.array([0, 3, 3, 5], dtype=np.int64)
    with pytest.raises(IndexError):
        host.choice(random.getrandbits, lens.ctypes.data, 0, off.ctypes.data, arr.ctypes.data, 3, out.ctypes.data)


def test_cpp_autograd_node_defers_to_one_callback_per_pass():
    """csrc/host/autograd_node.cpp on CPU tensors: the loss tensors are tensors of their own (the reference's in-place
    `loss += w * margin_loss(...)` works on the first one), a backward pass calls flush ONCE with every reached call and its
    upstream gradient, and destroyed nodes are reported."""
    import gc
    ext = _lib.load_autograd_node()
    assert ext is not None, 'python -m mpqe_amd.build builds mpqe_amd/lib/_autograd_node*.so'
    got = []
    ps = ext.Pass(lambda ids, grads: got.append(sorted(zip(ids, [float(g) for g in grads]))))
    bufs = [torch.tensor([float(i + 1), 0.0]) for i in range(3)]
    losses = [ext.make_loss(ps, 10 + i, bufs[i]) for i in range(3)]
    assert all(l.requires_grad and l.dim() == 0 and not l._is_view() for l in losses)
    loss = losses[0]
    loss += 0.5 * losses[1]
    loss += 0.25 * losses[2]
    assert abs(loss.item() - 2.75) < 1e-6
    loss.backward()
    assert got == [[(10, 1.0), (11, 0.5), (12, 0.25)]]
    del loss, losses
    gc.collect()
    assert sorted(ps.take_dead()) == [10, 11, 12] and ps.take_dead() == []
    with torch.no_grad():
        l = ext.make_loss(ps, 7, bufs[0])           # (a node is made all the same; nothing reaches it)
    del l
    gc.collect()
    assert ps.take_dead() == [7]


def test_collate_fn_ids_are_the_query_objects_ids():
    """RGCNQueryDataset.collate_fn (reference data_utils.py:293-311, 369-375) takes a window of the formula's id arrays
    (FormulaIds) instead of walking the Query objects: same anchors / targets / candidate lists as the objects hold, same
    np.random draws and windows as the reference's arithmetic."""
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import RGCNQueryDataset, get_queries_iterator
    schema = synthetic.make_schema(*synthetic.KG_SHAPES['tiny'], seed=1)
    rng = np.random.RandomState(2)
    queries = {}
    for qt in ('3-inter_chain', '2-chain'):
        f = synthetic.sample_formula(schema, qt, rng)
        queries[f] = synthetic.sample_queries(schema, f, 37 if qt == '2-chain' else 23, rng, n_neg=5, n_hard=2)

    class M(object):
        mode_ids = {m: i for i, m in enumerate(schema.modes)}
        rel_ids = {}
    for m in schema.relations:
        for (to, name) in schema.relations[m]:
            M.rel_ids[(m, name, to)] = len(M.rel_ids)
            M.rel_ids.setdefault((to, name, m), len(M.rel_ids))
    np.random.seed(5)
    it = get_queries_iterator(queries, 8, M)
    seen = 0
    for _ in range(12):
        formula, qs, anchor_ids, var_ids, g = next(it)
        ids = g.ids
        assert ids.end - ids.start == len(qs) == anchor_ids.shape[0] and ids.anchor_ref is anchor_ids
        ref_a, ref_v, ref_g = RGCNQueryDataset.get_query_graph(formula, qs, M.rel_ids, M.mode_ids)
        np.testing.assert_array_equal(anchor_ids.numpy(), ref_a.numpy())
        np.testing.assert_array_equal(var_ids.numpy(), ref_v.numpy())
        assert g.template.edge_type == ref_g.template.edge_type and g.template.B == len(qs)
        fi = ids.fi
        assert fi.targets[ids.start:ids.end].tolist() == [q.target_node for q in qs]
        np.testing.assert_array_equal(fi.anchors_sm[:, ids.start:ids.end], ref_a.numpy().T)
        flat, off, lens = fi.neg[:3]
        for k, q in enumerate(qs):
            j = ids.start + k
            assert flat[off[j]:off[j + 1]].tolist() == list(q.neg_samples) and lens[j] == len(q.neg_samples)
            if q.hard_neg_samples is not None:
                hf, ho = fi.hard[0], fi.hard[1]
                assert hf[ho[j]:ho[j + 1]].tolist() == list(q.hard_neg_samples)
        seen += len(qs)
    assert seen > 0


def test_queries_iterator_equals_the_dataloader_form():
    """get_queries_iterator (reference data_utils.py:412-426) without torch's DataLoader: the same batches (formula draws,
    windows across epoch ends, Query objects, id tensors) and the same numpy AND torch random streams as
    make_data_iterator(DataLoader(dataset, batch_size, shuffle=False, collate_fn=dataset.collate_fn))."""
    from torch.utils.data import DataLoader
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import RGCNQueryDataset, get_queries_iterator, make_data_iterator
    schema = synthetic.make_schema(*synthetic.KG_SHAPES['tiny'], seed=4)
    rng = np.random.RandomState(6)
    queries = {}
    for n in (37, 90, 64):
        f = synthetic.sample_formula(schema, '3-chain_inter', rng)
        while f in queries:
            f = synthetic.sample_formula(schema, '3-chain_inter', rng)
        queries[f] = synthetic.sample_queries(schema, f, n, rng, n_neg=4, n_hard=2)

    class M(object):
        mode_ids = {m: i for i, m in enumerate(schema.modes)}
        rel_ids = {}
    for m in schema.relations:
        for (to, name) in schema.relations[m]:
            M.rel_ids.setdefault((m, name, to), len(M.rel_ids))
            M.rel_ids.setdefault((to, name, m), len(M.rel_ids))

    def run(lean):
        np.random.seed(3)
        torch.manual_seed(5)
        if lean:
            it = get_queries_iterator(queries, 32, M)
        else:
            ds = RGCNQueryDataset(queries, M)
            it = make_data_iterator(DataLoader(ds, 32, shuffle=False, collate_fn=ds.collate_fn))
        out = []
        for _ in range(11):                 # (three epochs of the longest formula's 90 queries: 3 batches each + wrap-around)
            f, qs, a, v, g = next(it)
            out.append((f, [id(q) for q in qs], a.numpy().copy(), v.numpy().copy(), g.template.edge_type, g.template.B))
        return out, np.random.get_state()[1].tolist(), torch.get_rng_state().tolist()
    (a, na, ta), (b, nb, tb) = run(False), run(True)
    assert na == nb and ta == tb
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[1] == y[1] and x[4:] == y[4:]
        np.testing.assert_array_equal(x[2], y[2])
        np.testing.assert_array_equal(x[3], y[3])


def test_optim_constructors_fall_back_to_torch_off_the_fused_step(enc_case):
    """mpqe_amd.optim.Adam / SGD (torch.optim's constructors as reference train.py:83-88 calls them): a model that is not on
    the fused step -- here: on the CPU -- gets the torch optimiser itself, with torch.optim's calls; a deep copy of the model
    is found as its own parameters' owner."""
    import copy
    from mpqe_amd import optim
    model = build_model(enc_case, 'cpu')
    params = [p for p in model.parameters() if p.requires_grad]
    opt = optim.Adam(params, lr=0.01)
    assert not opt.flat and isinstance(opt._impl, torch.optim.Adam) and opt.param_groups[0]['lr'] == 0.01
    before = params[0].detach().clone()
    params[0].grad = torch.ones_like(params[0])
    opt.step()
    assert not torch.equal(before, params[0].detach())
    opt.zero_grad()
    assert params[0].grad is None
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    assert optim._OWNERS.get(id(params[0])) is model
    twin = copy.deepcopy(model)
    assert optim._OWNERS.get(id(next(twin.parameters()))) is twin
    with pytest.raises(ValueError):
        optim.SGD([dict(params=params)], lr=0.1)

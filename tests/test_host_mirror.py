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

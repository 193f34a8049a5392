import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def _tuplify(o):
    if isinstance(o, list):
        return tuple(_tuplify(x) for x in o)
    return o


class GoldenCase(object):
    """One fixture written by oracle/gen_golden.py, with the formula / queries
    rebuilt as this package's harness types."""

    def __init__(self, path):
        from collections import OrderedDict
        from mpqe_amd.graph import Formula, Query
        z = np.load(path)
        self.arrays = {k: z[k] for k in z.files if k != 'meta'}
        self.meta = json.loads(bytes(z['meta']).decode())
        m = self.meta
        self.name = m['name']
        self.cfg = m['cfg']
        self.D, self.B = m['D'], m['B']
        self.query_type = m['query_type']
        self.relations = OrderedDict(
            (mode, [tuple(x) for x in m['schema']['relations'][mode]])
            for mode in m['schema']['modes'])
        self.modes = m['schema']['modes']
        self.ids = {k: np.array(v, dtype=np.int64) for k, v in m['schema']['ids'].items()}
        self.num_entities = m['schema']['num_entities']
        self.mode_weights_order = m['mode_weights_order']
        self.mode_ids = m['mode_ids']
        self.rel_ids = {tuple(k): v for k, v in m['rel_ids']}
        self.formula = Formula(self.query_type, _tuplify(m['formula_rels']))
        self.queries = [Query(_tuplify(q['graph']), q['neg'], q['hard'], keep_graph=True)
                        for q in m['queries']]
        self.hard_negatives = m['hard_negatives']

    def params(self):
        import torch
        return {k[len('param/'):]: torch.from_numpy(v.copy())
                for k, v in self.arrays.items() if k.startswith('param/')}

    def grads(self):
        return {k[len('grad/'):]: v for k, v in self.arrays.items() if k.startswith('grad/')}

    def layer_outs(self):
        return [self.arrays['layer_out/%d' % i] for i in range(self.meta['n_layer_calls'])]


def golden_paths(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


def pytest_generate_tests(metafunc):
    if 'enc_case' in metafunc.fixturenames:
        paths = golden_paths('enc_')
        metafunc.parametrize('enc_case', paths, ids=[os.path.basename(p)[:-4] for p in paths],
                             indirect=True)
    if 'conv_case' in metafunc.fixturenames:
        paths = golden_paths('conv_')
        metafunc.parametrize('conv_case', paths, ids=[os.path.basename(p)[:-4] for p in paths],
                             indirect=True)


@pytest.fixture
def enc_case(request):
    return GoldenCase(request.param)


@pytest.fixture
def conv_case(request):
    z = np.load(request.param)
    return {k: z[k] for k in z.files}


class CaseGraph(object):
    """The attributes RGCNEncoderDecoder reads from the KG container, rebuilt from a fixture."""

    def __init__(self, case):
        from collections import OrderedDict
        self.relations = case.relations
        self.feature_dims = {m: case.D for m in case.modes}
        self.rel_edges = OrderedDict()
        for m in case.relations:
            for (to, name) in case.relations[m]:
                self.rel_edges[(m, name, to)] = 1.0
        self.mode_weights = OrderedDict((m, 1.0) for m in case.mode_weights_order)
        self.full_lists = {case.formula.target_mode: list(case.meta['full_list_target_mode'])}
        self.features = None
        self.adj_lists = None


def build_model(case, device):
    """This package's RGCNEncoderDecoder wired like the reference's start-up code, with the
    fixture's parameters loaded through load_state_dict (strict: key parity with the reference)."""
    import torch
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.model import RGCNEncoderDecoder
    feature_modules, node_maps = make_feature_modules(case.ids, case.D, case.num_entities)
    assert torch.equal(node_maps, torch.from_numpy(case.arrays['node_map']))
    enc = DirectEncoder(None, feature_modules, node_maps)
    cfg = case.cfg
    model = RGCNEncoderDecoder(CaseGraph(case), enc, cfg['readout'], cfg['scatter_op'], 0, cfg['weight_decay'],
                               cfg['num_layers'], cfg['shared_layers'], cfg['adaptive'])
    model.load_state_dict(case.params(), strict=True)
    return model.to(device)

"""End-to-end parity of the mirror modules on the GPU with the golden vectors produced by the
reference itself (tests/golden/, oracle/gen_golden.py): integer tensors bit-exact, forward floats
rtol 1e-5 / atol 1e-6, loss rtol 1e-5, gradients rtol 1e-4 / atol 1e-6; and with the CPU oracle at
the BASELINE batch shape."""
import random

import numpy as np
import pytest
import torch

from tests.conftest import build_model

pytestmark = pytest.mark.gpu
FWD = dict(rtol=1e-5, atol=1e-6)
BWD = dict(rtol=1e-4, atol=1e-6)


def _np(t):
    return t.detach().cpu().numpy()


def test_encoder_forward_matches_reference(enc_case):
    from mpqe_amd.data_utils import RGCNQueryDataset
    c = enc_case
    dev = torch.device('cuda:0')
    model = build_model(c, dev)
    anchor_ids, var_ids, g = RGCNQueryDataset.get_query_graph(c.formula, c.queries, model.rel_ids,
                                                              model.mode_ids)
    outs = []
    hooks = [l.register_forward_hook(lambda m, i, o: outs.append(o.detach().clone())) for l in set(model.layers)]
    targets = c.arrays['targets'].tolist()
    scores = model.forward(c.formula, c.queries, targets, anchor_ids, var_ids, g)
    for h in hooks:
        h.remove()
    # (a1) collation, expanded on the device: bit-exact
    np.testing.assert_array_equal(_np(g.edge_index), c.arrays['edge_index'])
    np.testing.assert_array_equal(_np(g.edge_type), c.arrays['edge_type'])
    np.testing.assert_array_equal(_np(g.batch), c.arrays['batch'])
    # (a2, a3) features
    np.testing.assert_allclose(_np(g.x), c.arrays['x0'], **FWD)
    # (a4) every layer; hidden layers carry the fused ReLU
    ref_layers = c.layer_outs()
    assert len(outs) == len(ref_layers)
    for i, (mine, ref) in enumerate(zip(outs, ref_layers)):
        if i < len(ref_layers) - 1:
            ref = np.maximum(ref, 0)
        np.testing.assert_allclose(_np(mine), ref, err_msg='layer %d' % i, **FWD)
    # (a6) scores, train form
    np.testing.assert_allclose(_np(scores), c.arrays['scores_pos'], **FWD)
    s_neg = model.forward(c.formula, c.queries, c.arrays['neg_nodes'].tolist(), anchor_ids, var_ids, g)
    np.testing.assert_allclose(_np(s_neg), c.arrays['scores_neg'], **FWD)
    # eval form: ragged negatives, graph rebuilt inside forward (model.py:404-408, 454-460)
    with torch.no_grad():
        s_eval = model.forward(c.formula, c.queries, targets, neg_nodes=c.arrays['eval_negs'].tolist(),
                               neg_lengths=c.arrays['neg_lengths'].tolist())
    np.testing.assert_allclose(_np(s_eval), c.arrays['eval_scores'], **FWD)


@pytest.mark.parametrize('encode_twice', [False, True])
def test_margin_loss_and_gradients_match_reference(enc_case, encode_twice):
    c = enc_case
    dev = torch.device('cuda:0')
    model = build_model(c, dev)
    model.encode_twice = encode_twice
    random.seed(4242 + c.meta['seed'])      # the generator's seed: same python `random` draws
    loss = model.margin_loss(c.formula, c.queries, hard_negatives=c.hard_negatives)
    np.testing.assert_allclose(loss.item(), float(c.arrays['loss']), rtol=1e-5, atol=1e-6)
    loss.backward()
    got = dict(model.named_parameters())
    for k, g in c.grads().items():
        mine = got[k].grad
        mine = np.zeros_like(g) if mine is None else _np(mine)
        np.testing.assert_allclose(mine, g, err_msg=k, **BWD)


def test_readout_value_matches_reference(enc_case):
    from mpqe_amd.data_utils import RGCNQueryDataset
    c = enc_case
    model = build_model(c, torch.device('cuda:0'))
    q = model.encode(c.formula, c.queries)
    np.testing.assert_allclose(_np(q), c.arrays['readout'], **FWD)


def test_conv_on_arbitrary_graph_matches_reference(conv_case):
    from mpqe_amd.model import RGCNConv
    z = conv_case
    dev = torch.device('cuda:0')
    R, Din, Dout = z['basis'].shape
    conv = RGCNConv(Din, Dout, R, 0).to(dev)
    conv.load_state_dict({k: torch.from_numpy(z[k]) for k in ('basis', 'root', 'bias')})
    x = torch.from_numpy(z['x']).to(dev).requires_grad_(True)
    ei, et = torch.from_numpy(z['edge_index']).to(dev), torch.from_numpy(z['edge_type']).to(dev)
    out = conv(x, ei, et)
    np.testing.assert_allclose(_np(out), z['out'], rtol=1e-5, atol=1e-5)
    out.backward(torch.from_numpy(z['grad_out']).to(dev))
    np.testing.assert_allclose(_np(x.grad), z['grad_x'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(_np(conv.basis.grad), z['grad_basis'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(_np(conv.root.grad), z['grad_root'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(_np(conv.bias.grad), z['grad_bias'], rtol=1e-4, atol=1e-5)
    # second call re-uses the cached plan
    out2 = conv(x, ei, et)
    assert torch.equal(out, out2)


def test_bad_entity_id_raises_index_error(enc_case):
    c = enc_case
    model = build_model(c, torch.device('cuda:0'))
    targets = c.arrays['targets'].tolist()
    targets[0] = c.num_entities + 50
    with pytest.raises(IndexError):
        model.forward(c.formula, c.queries, targets)
    # an id that belongs to no mode maps to -1 in the LUT -> IndexError, like nn.Embedding(-1).
    # (An id of ANOTHER mode has a valid row number in its own table, so -- exactly as in the
    # reference, data_utils.py:23-35 -- it silently reads that row of the wrong table.)
    targets[0] = c.num_entities
    with pytest.raises(IndexError):
        model.forward(c.formula, c.queries, targets)


def test_general_path_bad_edge_raises():
    from mpqe_amd.model import RGCNConv
    dev = torch.device('cuda:0')
    conv = RGCNConv(8, 8, 3, 0).to(dev)
    x = torch.zeros(4, 8, device=dev)
    with pytest.raises(IndexError):
        conv(x, torch.tensor([[0, 9], [1, 2]], device=dev), torch.tensor([0, 1], device=dev))
    with pytest.raises(IndexError):
        conv(x, torch.tensor([[0, 1], [1, 2]], device=dev), torch.tensor([0, 5], device=dev))


@pytest.mark.parametrize('qt', ['1-chain', '3-chain', '3-inter', '3-inter_chain'])
@pytest.mark.parametrize('readout', ['mp', 'sum', 'max'])
def test_baseline_shape_against_oracle(qt, readout):
    """B = 512, D = 128 on the AIFB-shaped synthetic KG: loss and every gradient against the CPU
    oracle run in the reference's op sequence (two encoder passes)."""
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import RGCNQueryDataset, make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.model import RGCNEncoderDecoder
    from oracle import ref_cpu
    torch.manual_seed(0)
    D, B = 128, 512
    schema = synthetic.make_schema(*synthetic.KG_SHAPES['aifb'], seed=0)
    graph = synthetic.SchemaGraph(schema, D)
    fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
    model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=readout, num_layers=3,
                               shared_layers=False, adaptive=(readout == 'mp'), weight_decay=0)
    with torch.no_grad():
        for p in model.layers.parameters():
            p.mul_(6.0)
    rng = np.random.RandomState(5)
    formula = synthetic.sample_formula(schema, qt, rng)
    queries = synthetic.sample_queries(schema, formula, B, rng)
    cfg = dict(readout=readout, scatter_op='add', num_layers=3, adaptive=(readout == 'mp'), weight_decay=0)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    col = ref_cpu.collate(formula, queries, model.rel_ids, model.mode_ids)
    targets = [q.target_node for q in queries]
    negs = [q.neg_samples[0] for q in queries]
    ref_loss = ref_cpu.margin_loss(params, cfg, node_maps, formula, col, targets, negs)
    ref_loss.backward()

    model = model.to('cuda:0')
    out = model.encode(formula, queries)
    loss = __import__('mpqe_amd').ops.hinge(model.score(formula, out, targets), model.score(formula, out, negs))
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=1e-5, atol=1e-6)
    for k, p in model.named_parameters():
        ref = params[k].grad
        ref = torch.zeros_like(params[k]) if ref is None else ref
        mine = torch.zeros_like(p) if p.grad is None else p.grad
        np.testing.assert_allclose(_np(mine), ref.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('name', ['sage_depth1_plain', 'sage_depth1_ln'])
def test_sampled_neighbour_encoder_matches_reference(name):
    """the reference's depth-1 Encoder + MeanAggregator (sampled-neighbour mean per
    relation -> concat -> compress -> [LayerNorm] -> ReLU), same python `random` draws."""
    import json
    import os
    from collections import OrderedDict
    from mpqe_amd.aggregators import MeanAggregator
    from mpqe_amd.encoders import Encoder
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(bytes(z['meta']).decode())
    dev = torch.device('cuda:0')
    D = meta['D']
    modes = meta['schema']['modes']
    relations = OrderedDict((m, [tuple(x) for x in meta['schema']['relations'][m]]) for m in modes)
    adj = {tuple(rel): {int(n): list(nb) for n, nb in items} for rel, items in meta['adj']}
    node_map = torch.from_numpy(z['node_map']).to(dev)
    fm = {m: torch.nn.Embedding(len(meta['schema']['ids'][m]) + 1, D) for m in modes}
    features = lambda nodes, mode: fm[mode](node_map[torch.as_tensor(nodes, dtype=torch.long, device=dev)])
    dims = {m: D for m in modes}
    enc = Encoder(features, dims, dims, relations, adj, feature_modules=fm, cuda=True,
                  aggregator=MeanAggregator(features), layer_norm=meta['layer_norm'])
    state = {k[len('param/'):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('param/')}
    enc.load_state_dict(state, strict=True)          # same keys as the reference
    enc = enc.to(dev)
    random.seed(meta['random_seed'])
    out = enc.forward(z['nodes'].tolist(), meta['mode'], keep_prob=meta['keep_prob'], max_keep=meta['max_keep'])
    np.testing.assert_allclose(_np(out), z['out'], **FWD)
    out.backward(torch.from_numpy(z['grad_out']).to(dev))
    for k, p in enc.named_parameters():
        g = z['grad/' + k]
        mine = np.zeros_like(g) if p.grad is None else _np(p.grad)
        np.testing.assert_allclose(mine, g, err_msg=k, **BWD)

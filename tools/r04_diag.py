"""Diagnostics: per-parameter difference between the fused step and the module path (learned readouts)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpqe_amd import synthetic
from mpqe_amd.data_utils import make_feature_modules
from mpqe_amd.encoders import DirectEncoder
from mpqe_amd.fused import FusedTrainStep
from mpqe_amd.model import RGCNEncoderDecoder
readout, B, D = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kw = {}
for a in sys.argv[4:]:
    k, v = a.split('=')
    kw[k] = {'True': True, 'False': False, 'None': None}.get(v, v)
device = torch.device('cuda:0')
schema = synthetic.make_schema(*synthetic.KG_SHAPES['aifb'], seed=0)
torch.manual_seed(0)
graph = synthetic.SchemaGraph(schema, D)
fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=readout, num_layers=3, shared_layers=False,
                           adaptive=False, weight_decay=0).to(device)
model.validate = False
data = bench.StepData(schema, model, B, np.random.RandomState(1000), device)
bench.step_modules(model, data)
ref = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
step = FusedTrainStep(model, **kw)
packed = bench.pack_for_fused(step, data, resident=True)
print('chain', step.uses_chain(packed), 'merged', step.merged(packed))
step.run(packed)
step.check()
for k, p in model.named_parameters():
    if k in ref:
        d = float((p.grad - ref[k]).abs().max() / (ref[k].abs().max() + 1e-12))
        if d > 1e-5 or 'readout' in k:
            bad = ((p.grad - ref[k]).abs() > 1e-5 * ref[k].abs().max()).nonzero()
            print('%-40s rel %.2e  shape %s  bad %d first %s' % (k, d, tuple(p.shape), len(bad), bad[:3].tolist()))

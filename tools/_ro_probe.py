"""One learned-readout fused step, eager, N times (for rocprofv3 --stats)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpqe_amd import synthetic
from mpqe_amd.data_utils import make_feature_modules
from mpqe_amd.encoders import DirectEncoder
from mpqe_amd.fused import FusedTrainStep
from mpqe_amd.model import RGCNEncoderDecoder
readout = sys.argv[1]
dev = torch.device('cuda:0')
schema = synthetic.make_schema(*synthetic.KG_SHAPES['aifb'], seed=0)
graph = synthetic.SchemaGraph(schema, 128)
fm, node_maps = make_feature_modules(schema.ids, 128, schema.num_entities)
model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=readout, num_layers=3, shared_layers=False,
                           adaptive=False, weight_decay=0).to(dev)
model.validate = False
data = bench.StepData(schema, model, 512, np.random.RandomState(1000), dev)
step = FusedTrainStep(model)
packed = bench.pack_for_fused(step, data, resident=True)
for _ in range(50):
    step.run(packed)
torch.cuda.synchronize()

import time, ctypes, torch, numpy as np
from mpqe_amd import synthetic, _capi
from mpqe_amd import ops
import bench
# build the bench model on CPU just to get descriptors
from mpqe_amd.data_utils import make_feature_modules
from mpqe_amd.encoders import DirectEncoder
from mpqe_amd.model import RGCNEncoderDecoder
schema = synthetic.make_schema(*synthetic.KG_SHAPES['aifb'], seed=0)
graph = synthetic.SchemaGraph(schema, 128)
fm, nm = make_feature_modules(schema.ids, 128, schema.num_entities)
model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, nm), readout='mp', num_layers=3, shared_layers=False, adaptive=True, weight_decay=0)
from mpqe_amd.fused import FusedTrainStep
fs = FusedTrainStep.__new__(FusedTrainStep)
fs = FusedTrainStep(model)
rng = np.random.RandomState(1000)
d = bench.StepData(schema, model, 512, rng, torch.device('cpu'))
pk = bench.pack_for_fused(fs, d)
lib = ops.lib()
t=time.perf_counter()
for _ in range(200): lib.mpqe_step_workspace_bytes(ctypes.byref(fs.P), pk.batches, pk.nb)
print('plan us', (time.perf_counter()-t)/200*1e6)

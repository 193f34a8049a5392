"""Probe: can the host-to-device copy of the ids + the touch-plan build be replayed as ONE hipGraph launch?"""
import sys, time, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from mpqe_amd import synthetic, ops, _capi
from mpqe_amd.data_utils import make_feature_modules
from mpqe_amd.encoders import DirectEncoder
from mpqe_amd.fused import FusedTrainStep
from mpqe_amd.model import RGCNEncoderDecoder
torch.manual_seed(0)
dev = torch.device('cuda:0')
schema = synthetic.make_schema(*synthetic.KG_SHAPES['aifb'], seed=0)
graph = synthetic.SchemaGraph(schema, 128)
fm, node_maps = make_feature_modules(schema.ids, 128, schema.num_entities)
model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout='mp', num_layers=3, shared_layers=False, adaptive=True, weight_decay=0).to(dev)
data = bench.StepData(schema, model, 512, np.random.RandomState(1), dev)
step = FusedTrainStep(model)
pk = bench.pack_for_fused(step, data)
step.run(pk)
torch.cuda.synchronize()
L = ops.lib()
n = pk.anchor_ids.numel() + pk.targets.numel() + pk.negs.numel()
stage = torch.empty(n, dtype=torch.long, pin_memory=True)
stage[:pk.anchor_ids.numel()] = pk.anchor_ids.cpu()
stage[pk.anchor_ids.numel():pk.anchor_ids.numel() + pk.targets.numel()] = pk.targets.cpu()
stage[pk.anchor_ids.numel() + pk.targets.numel():] = pk.negs.cpu()
ids = torch.empty(n, dtype=torch.long, device=dev)
nbytes, wbytes = pk.touch_sizes
touch = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
ws = torch.empty(wbytes + 256, dtype=torch.uint8, device=dev)
tp = (touch.data_ptr() + 255) // 256 * 256
wp = (ws.data_ptr() + 255) // 256 * 256
na, ng = pk.anchor_ids.numel(), pk.targets.numel()
def work(stream):
    ids.copy_(stage, non_blocking=True)
    st = L.mpqe_step_touch_build(ctypes.byref(step.P), pk.batches, pk.nb, ids.data_ptr(), ids.data_ptr() + 8 * na,
                                 ids.data_ptr() + 8 * (na + ng), tp, nbytes, wp, wbytes, stream)
    assert st == 0, st
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    work(s.cuda_stream)
torch.cuda.synchronize()
ref = touch.clone()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        work(torch.cuda.current_stream().cuda_stream)
    touch.zero_()
    g.replay()
    torch.cuda.synchronize()
    print('replay reproduces the plan:', bool(torch.equal(touch, ref)))
    t = time.perf_counter()
    for _ in range(200):
        g.replay()
    host = (time.perf_counter() - t) / 200
    torch.cuda.synchronize()
    print('graph replay: host %.1f us, with device %.1f us' % (host * 1e6, (time.perf_counter() - t) / 200 * 1e6))
except Exception as e:
    print('capture failed:', repr(e)[:300])
t = time.perf_counter()
for _ in range(200):
    work(torch.cuda.current_stream().cuda_stream)
host = (time.perf_counter() - t) / 200
torch.cuda.synchronize()
print('eager: host %.1f us, with device %.1f us' % (host * 1e6, (time.perf_counter() - t) / 200 * 1e6))

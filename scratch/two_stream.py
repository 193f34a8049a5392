import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from mpqe_amd import synthetic
from mpqe_amd.data_utils import make_feature_modules
from mpqe_amd.encoders import DirectEncoder
from mpqe_amd.model import RGCNEncoderDecoder
from mpqe_amd.fused import FusedTrainStep
torch.manual_seed(0)
D=128
schema = synthetic.make_schema(*synthetic.KG_SHAPES['aifb'], seed=0)
graph = synthetic.SchemaGraph(schema, D)
fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout='mp', num_layers=3, shared_layers=False, adaptive=True, weight_decay=0).to('cuda:0')
rng = np.random.RandomState(1)
data = bench.StepData(schema, model, 512, rng, torch.device('cuda:0'))
bs = [dict(formula=b['formula'], anchor_ids=b['anchor_np'], targets=b['targets_np'], negs=b['negs_np'], weight=b['weight']) for b in data.batches]
qt = [b['formula'].query_type for b in bs]
print(qt)
A = [i for i,q in enumerate(qt) if q in ('3-chain','2-chain','3-inter_chain')]
B = [i for i in range(len(bs)) if i not in A]
s_all = FusedTrainStep(model); p_all = s_all.pack(bs)
sA = FusedTrainStep(model); pA = sA.pack([bs[i] for i in A])
sB = FusedTrainStep(model); pB = sB.pack([bs[i] for i in B])
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
print('single step all      %.1f us' % t(lambda: s_all.run(p_all)))
print('A alone              %.1f us' % t(lambda: sA.run(pA, zero_grad=False)))
print('B alone              %.1f us' % t(lambda: sB.run(pB, zero_grad=False)))
def seq():
    sA.run(pA, zero_grad=False); sB.run(pB, zero_grad=False)
print('A then B sequential  %.1f us' % t(seq))
st1, st2 = torch.cuda.Stream(), torch.cuda.Stream()
def par():
    with torch.cuda.stream(st1): sA.run(pA, zero_grad=False)
    with torch.cuda.stream(st2): sB.run(pB, zero_grad=False)
print('A || B two streams   %.1f us' % t(par))

import sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from mpqe_amd import synthetic
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
for _ in range(3):
    pk = bench.pack_for_fused(step, data); step.run(pk)
torch.cuda.synchronize()
step._prof = {}
N = 200
t = time.perf_counter()
for _ in range(N):
    pk = bench.pack_for_fused(step, data)
tot = time.perf_counter() - t
torch.cuda.synchronize()
print('pack: %.1f us per call; sections (us):' % (tot / N * 1e6), {k: round(v / N * 1e6, 1) for k, v in step._prof.items()})
step._prof = None
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    pk = bench.pack_for_fused(step, data)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])

# the three loops, no synchronisation inside: pack only, run only (one packed step), pack + run
def loop(fn, n=200):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return host * 1e6, (time.perf_counter() - t) / n * 1e6
pk = bench.pack_for_fused(step, data)
step.run(pk)
print('pack only: host %.1f us, with device %.1f us per iteration' % loop(lambda: bench.pack_for_fused(step, data)))
print('run only:  host %.1f us, with device %.1f us per iteration' % loop(lambda: step.run(pk)))
print('pack+run:  host %.1f us, with device %.1f us per iteration' % loop(lambda: step.run(bench.pack_for_fused(step, data))))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step.run(bench.pack_for_fused(step, data))
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
print(s.getvalue()[:3500])

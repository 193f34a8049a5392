import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_fused_gpu import _setup
from mpqe_amd import ops
from mpqe_amd.fused import FusedTrainStep
model, batches = _setup('sum', False, False)
model.zero_grad(set_to_none=True)
total = None
for b in batches:
    out = model.encode(b['formula'], b['queries'])
    pos = model.score(b['formula'], out, b['targets'].tolist())
    neg = model.score(b['formula'], out, b['negs'].tolist())
    l = ops.hinge(pos, neg, 1.0) * b['weight']
    total = l if total is None else total + l
total.backward()
ref = {k: (torch.zeros_like(p) if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}
step = FusedTrainStep(model)
packed = step.pack(batches)
step.run(packed); torch.cuda.synchronize()
g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
step.run(packed); torch.cuda.synchronize()
g2 = {k: p.grad.clone() for k, p in model.named_parameters()}
for k in ref:
    d = (g1[k]-ref[k]).abs()
    d2 = (g1[k]-g2[k]).abs()
    print(k, 'max|fused-ref| %.3e  max|run1-run2| %.3e  max|ref| %.3e' % (d.max().item(), d2.max().item(), ref[k].abs().max().item()))
    if k.startswith('enc.feat') and d.max() > 1e-5:
        rows = d.reshape(d.shape[0], -1).max(1).values
        bad = (rows > 1e-5).nonzero().flatten().tolist()
        print('   bad rows', bad, [rows[r].item() for r in bad])
mode_of = {v: k for k, v in model.mode_ids.items()}
for i, b in enumerate(batches):
    print(i, b['formula'].query_type, 'target mode', b['formula'].target_mode, 'anchor modes', b['formula'].anchor_modes)

"""Where an iteration of the reference's training loop goes on the host when it runs through the drop-in entry points
(mpqe_amd/dropin.py): perf_counter around the loop's own statements -- the eleven margin_loss calls, the `loss += w * l`
arithmetic, loss.item(), loss.backward(), the optimiser --, microseconds per iteration (DESIGN.md 1a quotes it).

    python tools/dropin_sections.py [readout]
"""
import os, sys, time, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dropin_loop_bench as b
from mpqe_amd import dropin as dmod
torch.cuda.set_device(0)
model, tq = b.build(sys.argv[1] if len(sys.argv) > 1 else 'mp')
model = model.to('cuda:0')
np.random.seed(0); random.seed(0)
steps = b.precollate(model, tq, 512, 64)
opt = b._FlatAdapter(model, 0.001)
b.loop(model, opt, steps, 30, 10)
acc = dict(ml=0.0, arith=0.0, item=0.0, bwd=0.0, opt=0.0, zero=0.0)
pc = time.perf_counter
def body(step):
    t0 = pc(); opt.zero_grad(); t1 = pc(); acc['zero'] += t1 - t0
    loss = None
    for batch, hard, w in step:
        t = pc(); l = model.margin_loss(*batch, hard_negatives=hard); t2 = pc(); acc['ml'] += t2 - t
        if loss is None: loss = l
        else: loss += w * l
        acc['arith'] += pc() - t2
    t = pc(); v = loss.item(); t2 = pc(); acc['item'] += t2 - t
    loss.backward(); t3 = pc(); acc['bwd'] += t3 - t2
    opt.step(); acc['opt'] += pc() - t3
N = 300
torch.cuda.synchronize(); T = pc()
for i in range(N): body(steps[i % 64])
torch.cuda.synchronize(); T = pc() - T
print('total us/iter %.1f' % (T / N * 1e6))
for k, v in acc.items(): print('%-6s %.1f us/iter' % (k, v / N * 1e6))


"""Soak run of the reference's training loop through the drop-in entry points (mpqe_amd/dropin.py): per readout, two runs of
--iters iterations (collation included, flat Adam) from the same seeds must give the same loss values to the last bit, stay
finite and leave no error flag behind (a lost hand-off, a bad id, a side stream that read parameters too early would show
as one of the three).

    python tools/dropin_soak.py [--iters 10000] [--readouts mp,mlp,targetmlp,concat,sum,max]
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dropin_loop_bench as b      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10000)
    ap.add_argument('--readouts', default='mp,mlp,targetmlp,concat,sum,max')
    args = ap.parse_args()
    torch.cuda.set_device(0)
    for readout in args.readouts.split(','):
        runs, dt, d = [], 0.0, None
        for rep in range(2):
            model, tq = b.build(readout)
            model = model.to('cuda:0')
            np.random.seed(0)
            random.seed(0)
            live = b._Live(model, tq, 512)
            opt = b._FlatAdapter(model, 0.001)
            d = model.dropin()
            vals = []
            t0 = time.perf_counter()
            for i in range(args.iters):
                opt.zero_grad()
                loss = None
                for batch, hard, w in live[0]:
                    l = model.margin_loss(*batch, hard_negatives=hard)
                    if loss is None:
                        loss = l
                    else:
                        loss += w * l
                if i % 50 == 0:
                    vals.append(loss.item())
                loss.backward()
                opt.step()
                if i % 100 == 99:
                    d._check_mirror()
            torch.cuda.synchronize()
            d._check_mirror()
            dt = time.perf_counter() - t0
            runs.append(vals)
        same, finite = runs[0] == runs[1], bool(np.isfinite(runs[0]).all())
        print('%-10s %d iterations x 2: %.2f ms per iteration, loss %.4f -> %.4f, finite %s, the two runs identical %s, side '
              'streams %d, fused steps %d' % (readout, args.iters, dt / args.iters * 1e3, runs[0][0], runs[0][-1], finite, same,
                                              len(d.lanes), d.steps), flush=True)
        if not (same and finite):
            raise SystemExit('soak failed for readout %s' % readout)


if __name__ == '__main__':
    main()

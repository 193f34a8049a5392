"""Training step with a learned readout (reference model.py:497-553: mlp / targetmlp), AIFB-shaped full query mix:
the module path (one autograd graph over the per-batch ops) against the fused step (the readout inside the same library
call: csrc/step_readout.h). Prints one JSON line per readout.

    python tools/readout_step_bench.py [--readouts mlp,targetmlp] [--steps 50] [--batch-size 512] [--embed-dim 128]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / steps)
    return float(np.median(best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--readouts', default='mlp,targetmlp')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch-size', type=int, default=512)
    ap.add_argument('--embed-dim', type=int, default=128)
    ap.add_argument('--kg', default='aifb')
    args = ap.parse_args()
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.fused import FusedTrainStep
    from mpqe_amd.model import RGCNEncoderDecoder
    device = torch.device('cuda:0')
    D = args.embed_dim
    schema = synthetic.make_schema(*synthetic.KG_SHAPES[args.kg], seed=0)
    for readout in args.readouts.split(','):
        torch.manual_seed(0)
        graph = synthetic.SchemaGraph(schema, D)
        fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
        model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=readout, num_layers=3,
                                   shared_layers=False, adaptive=False, weight_decay=0).to(device)
        model.validate = False
        rng = np.random.RandomState(1000)
        data = bench.StepData(schema, model, args.batch_size, rng, device)
        t_mod = timed(lambda: bench.step_modules(model, data), args.steps, args.warmup)
        ref_loss = bench.step_modules(model, data).item()
        ref = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        step = FusedTrainStep(model)
        packed = bench.pack_for_fused(step, data, resident=True)
        loss = step.run(packed)
        worst = 0.0
        for k, p in model.named_parameters():
            if k in ref:
                worst = max(worst, float((p.grad - ref[k]).abs().max() / (ref[k].abs().max() + 1e-12)))
        t_fused = timed(lambda: step.run(packed), args.steps, args.warmup)
        descs = [dict(formula=b['formula'], weight=b['weight'], batch_size=len(b['targets_np'])) for b in data.batches]
        fresh = bench.draw_ids_device(schema, data, 64, 7, device)
        k = [0]

        def fresh_step():
            k[0] += 1
            return step.run(step.pack(descs, ids=fresh[k[0] % fresh.shape[0]]))
        t_fresh = timed(fresh_step, args.steps, args.warmup)
        step.check()
        cap = step.capture(bench.pack_for_fused(step, data, resident=True))
        t_graph = timed(cap.replay, args.steps, args.warmup)
        print(json.dumps(dict(readout=readout, query_graphs=data.num_graphs, embed_dim=D,
                              module_path_ms=round(1e3 * t_mod, 4), fused_step_ms=round(1e3 * t_fused, 4),
                              fused_fresh_ids_ms=round(1e3 * t_fresh, 4), fused_graph_replay_ms=round(1e3 * t_graph, 4),
                              fused_q_graphs_per_s=round(data.num_graphs / t_fresh, 1),
                              speedup=round(t_mod / t_fresh, 2), loss_module=ref_loss, loss_fused=loss[0].item(),
                              max_rel_grad_diff=worst)), flush=True)


if __name__ == '__main__':
    main()

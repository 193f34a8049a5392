"""Child process of tests/test_parallel_gpu.py: one data-parallel rank (FusedTrainStep + StepExchange). Started before
it touches the GPU; writes its reduced flat gradient (and, rank 0, the single-process gradient of all ranks' batches)
to the directory given on the command line."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def batches_of(schema, rank, B, scale):
    from mpqe_amd import synthetic
    rng = np.random.RandomState(1000 + rank)
    out = []
    for qt in ('1-chain', '3-chain', '2-inter', '3-inter_chain', '3-chain_inter'):
        f = synthetic.sample_formula(schema, qt, rng)
        anchors = np.stack([synthetic._pick(schema, m, rng, size=B) for m in f.anchor_modes], axis=1)
        out.append(dict(formula=f, anchor_ids=anchors, targets=synthetic._pick(schema, f.target_mode, rng, size=B),
                        negs=synthetic._pick(schema, f.target_mode, rng, size=B), weight=scale * (1.0 if qt == '1-chain' else 0.1)))
    return out


def main():
    outdir, sparse, tables = sys.argv[1], sys.argv[2] == '1', sys.argv[3]
    readout = sys.argv[4] if len(sys.argv) > 4 else 'mp'
    touch = sys.argv[5] if len(sys.argv) > 5 else ('pack' if tables == 'rows' else 'step')
    transport = sys.argv[6] if len(sys.argv) > 6 else 'rccl'
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    # MPQE_DP_BACKEND=nccl: one rank per GPU over RCCL (needs `world` visible devices); default: gloo through the host,
    # every rank on cuda:0 (what a one-GPU box can run)
    backend = os.environ.get('MPQE_DP_BACKEND', 'gloo')
    dev = 'cuda:%d' % (rank if backend == 'nccl' else 0)
    torch.cuda.set_device(torch.device(dev))
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from mpqe_amd import synthetic
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.fused import FusedTrainStep
    from mpqe_amd.model import RGCNEncoderDecoder
    from mpqe_amd.optim import FlatOptimizer
    from mpqe_amd.parallel import StepExchange
    torch.manual_seed(0)
    D, B = 64, 40
    schema = synthetic.make_schema(*synthetic.KG_SHAPES['tiny'], seed=3)
    graph = synthetic.SchemaGraph(schema, D)
    fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
    model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=readout, num_layers=3, shared_layers=False,
                               adaptive=readout != 'concat', weight_decay=1e-3 if readout != 'mp' else 0).to(dev)
    with torch.no_grad():
        for p in model.layers.parameters():
            p.mul_(4.0)
    # tables = 'rows': the row exchange (plans with the keys of a touch plan built at pack time); 'dense': the tables ride in
    # the all-reduce and the step builds its touch plan itself (the default)
    step = FusedTrainStep(model, sparse_tables=sparse, touch=touch)
    packed = step.pack(batches_of(schema, rank, B, 1.0 / world))
    ex = StepExchange(step, tables=tables, transport=transport)
    if transport == 'p2p':
        assert ex.transport == 'p2p', ex.transport_note
    plan = ex.plan(packed, key='set0')
    if tables == 'dense' or touch == 'step':
        assert ex.plan(packed, key='set0', verify=True) is plan          # a recurring key: the cached plan (verified across ranks)
        # a key re-used for ANOTHER formula set must be refused, not reduce the wrong matrices
        other = step.pack(batches_of(schema, rank + 7, B, 1.0 / world))
        try:
            ex.plan(other, key='set0')
            raise SystemExit('a plan key re-used for another descriptor set was accepted')
        except ValueError:
            pass
    for p in model.parameters():
        p.grad.fill_(3.0)
    fail_rank = int(os.environ.get('MPQE_DP_FAIL_SORT_RANK', '-1'))
    if rank == fail_rank:
        # this rank's in-step sort gives up (as if its workgroups were not co-resident): run(checked=True) rebuilds the plan
        # with the library sort and sums the table rows again BEFORE the exchange -- the peers never see a short bucket
        from mpqe_amd import ops
        ops.lib().mpqe_debug_option(b'TSORT_FAIL', 1, 1)
    step.run(packed, checked=True)
    if rank == fail_rank:
        ops.lib().mpqe_debug_option(b'TSORT_FAIL', 0, 0)
        assert step.touch_retries == 1, step.touch_retries
    ex.reduce(plan, packed=packed)
    ex.check()                    # (collective: every rank learns what any rank met -- a peer that never arrived included)
    torch.cuda.synchronize()
    out = dict(flat=step.flat_grad.cpu().numpy(), wire=np.array([plan.wire_bytes]), dense=np.array([step.flat_grad.numel() * 4]))
    out['form'] = np.array([plan.form])
    if tables == 'rows':
        # which table rows hold the reduced gradient (sparse mode leaves the others alone)
        base = plan.plan_ptr - plan.plan.data_ptr()
        keys = plan.plan[base + 256: base + 256 + 8 * plan.entries].view(torch.int64)
        out['union_keys'] = torch.unique(keys[keys != -1]).cpu().numpy()
    out['row_bits'] = np.array([ex.row_bits])
    if rank == 0:   # single process, all ranks' batches in one step (dense tables)
        ref_step = FusedTrainStep(model)
        allb = []
        for r in range(world):
            allb += batches_of(schema, r, B, 1.0 / world)
        ref_step.run(ref_step.pack(allb))
        torch.cuda.synchronize()
        out['ref'] = ref_step.flat_grad.cpu().numpy()
        out['table_floats'] = np.array([sum(t.numel() for t in ex.tables)])
    if sparse:      # one optimiser step on the union rows: replicas must stay identical
        step.bind_grads()           # (the reference step above re-bound p.grad to its own buffer)
        opt = FlatOptimizer(step, lr=0.01, sparse_tables=True)
        opt.step(packed, rows_plan=ex.rows_plan(plan))
        torch.cuda.synchronize()
        out['params'] = opt.flat_param.cpu().numpy()
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **out)
    if ex.peer is not None:
        ex.peer.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""Throughput of the MPQE R-GCN query-graph encoder hot path (forward + backward) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the reference's post-burn-in training step on an
AIFB-shaped synthetic KG -- 11 batches of B=512 query graphs (1x 1/2/3-chain, 2x each of
2-inter / 3-inter / 3-inter_chain / 3-chain_inter; reference train_helpers.py:81, 97-112),
embed_dim 128, readout 'mp' (TM), adaptive, num_layers 3, unshared layers. A step = feature
assembly -> L layers -> readout -> cosine scores vs positive and negative targets -> hinge,
forward and backward to every parameter gradient, with all inputs already resident in HBM.
Collation from python objects and the optimiser are outside the metric.
EVERY timed step runs on ids it has not seen before (pre-drawn, resident in HBM, one set per step):
the id -> table-row lookups and the ordering of the embedding-gradient accumulation happen inside the
step (reference train_helpers.py:76-120 draws a new batch per call; encoders.py:40-43 and the embedding
backward resolve ids inside forward / backward). Formulas come from a pool of 4 formula sets.
Each rank runs the same-sized workload (weak scaling); N > 1 adds the gradient exchange (RCCL) per step.
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(a child `python -m torch.distributed.run`, before this process touches a GPU).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PATH_WEIGHT, INTER_WEIGHT = 0.01, 0.005     # reference train_helpers.py:60-61
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3                # MI355X_MICROARCH.md: fp32 MFMA dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--repeats', type=int, default=0,
                    help='the timed block of --steps steps is repeated this many times; the median block is reported. '
                         '0 = as many blocks as make ~0.6 s of timed steps (a few steps are timed first), at least 25 '
                         '(5 when a block takes more than 50 ms), at most 500')
    ap.add_argument('--kg', default='aifb')
    ap.add_argument('--embed-dim', type=int, default=128)
    ap.add_argument('--batch-size', type=int, default=512)
    ap.add_argument('--readout', default='mp')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dropin-loop', action='store_true',
                    help='skip the secondary key dropin_loop (the reference loop body through model.margin_loss)')
    ap.add_argument('--no-pack-ms', action='store_true',
                    help='skip the pack_ms record (host-side loops with ids in host memory): profiler runs, so that the per-kernel averages are those of the timed loop')
    ap.add_argument('--no-scatter', action='store_true', help='skip the roofline_scatter record (general-graph scatter-aggregate)')
    ap.add_argument('--no-self-check', action='store_true',
                    help='skip the pre-timing comparison with the module path (timing experiments with builds that are wrong on purpose)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--path', default='auto', choices=['auto', 'modules', 'fused'])
    ap.add_argument('--lanes', type=int, default=1, help='HIP streams one fused step is spread over (level form only)')
    ap.add_argument('--graph', type=int, default=0,
                    help='1: replay each step from a captured hipGraph')
    ap.add_argument('--no-prune', action='store_true', help='compute node states that cannot reach the readout too')
    ap.add_argument('--no-chain', action='store_true', help='one launch per message-passing level')
    ap.add_argument('--merge-tail', type=int, default=-1,
                    help='chain form: weight-gradient tiles and backward post-pass as workgroups of the chain launch (1), '
                         'as a launch of their own (0), or the library\'s choice by step size (-1, default; '
                         'MPQE_STEP_MERGE_TAIL / MPQE_STEP_SPLIT_TAIL)')
    ap.add_argument('--no-uniform', action='store_true',
                    help='chain form: node states no anchor has reached yet as per-graph rows instead of one vector per batch')
    ap.add_argument('--no-ksplit', action='store_true', help='dim 128: chain waves own 32 columns and all of K')
    ap.add_argument('--eight-waves', action='store_true', help='dim 128: chain workgroups of eight waves (experimental)')
    ap.add_argument('--sparse-tables', action='store_true',
                    help='row-sparse entity-table gradients (MPQE_STEP_SPARSE_TABLES): only the rows a step touches are '
                         'written, no zero fill of the tables (for row-sparse consumers: mpqe_adam_rows_step, the row exchange)')
    ap.add_argument('--exchange', default='rccl', choices=['rccl', 'p2p'],
                    help='N > 1: how the gradient bucket is summed -- rccl: torch.distributed all-reduce; p2p: the '
                         "library's one-hop reduce-scatter + all-gather over IPC-mapped peer buffers (csrc/p2p.hip), "
                         'self-tested against the all-reduce at start-up and falling back to it (exchange_note) on any failure')
    ap.add_argument('--dense-allreduce', action='store_true',
                    help='N > 1: all-reduce the whole flat gradient buffer (what a literal port would do) instead of the '
                         'touched-matrix bucket + row exchange')
    ap.add_argument('--debug-opt', action='append', default=[], metavar='NAME=VALUE',
                    help='timing experiments: a diagnostics switch of the library (mpqe_debug_option), e.g. TAIL_UX=0')
    ap.add_argument('--host-ids', default='direct', choices=['direct', 'copy'],
                    help='pack_ms loops: ids in host memory read in place from pinned memory (default) or copied per pack')
    ap.add_argument('--replay', action='store_true',
                    help='time the replay of 4 pre-packed steps (round 2\'s headline) instead of fresh ids per step')
    ap.add_argument('--touch', default='step', choices=['step', 'pack', 'atomics'],
                    help="entity-table gradients: per-row sums with the plan built inside the step (default), built by "
                         "pack(), or fp32 atomics")
    ap.add_argument('--fresh-sets', type=int, default=0,
                    help='distinct pre-drawn id sets (176 KB each at the default shape); the timed steps walk through them, '
                         'wrapping around only when warmup + steps x repeats exceeds it. 0 = one per step, at most 16 384 (2.9 GB)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='nccl = RCCL over xGMI (the real path); gloo only to exercise the multi-rank code on one GPU')
    return ap.parse_args()


class StepData(object):
    """One training step's 11 batches, pre-collated and resident in HBM."""

    def __init__(self, schema, model, B, rng, device):
        from mpqe_amd import synthetic
        from mpqe_amd.data_utils import RGCNQueryDataset
        self.batches = []
        for qt, hard in synthetic.FULL_MIX:
            formula = synthetic.sample_formula(schema, qt, rng)
            A = len(formula.anchor_modes)
            anchors = np.stack([synthetic._pick(schema, m, rng, size=B) for m in formula.anchor_modes], axis=1)
            targets = synthetic._pick(schema, formula.target_mode, rng, size=B)
            negs = synthetic._pick(schema, formula.target_mode, rng, size=B)
            # light-weight query records: get_query_graph only reads anchor_nodes
            queries = [_Q(tuple(int(v) for v in anchors[b])) for b in range(B)]
            anchor_ids, var_ids, graph = RGCNQueryDataset.get_query_graph(formula, queries, model.rel_ids,
                                                                          model.mode_ids)
            weight = 1.0 if qt == '1-chain' else (INTER_WEIGHT if 'inter' in qt else PATH_WEIGHT)
            self.batches.append(dict(
                formula=formula, queries=queries, weight=weight, A=A,
                anchor_ids=anchor_ids.to(device), var_ids=var_ids.to(device), graph=graph.to(device),
                targets=torch.from_numpy(targets).to(device), negs=torch.from_numpy(negs).to(device),
                anchor_np=anchors, targets_np=targets, negs_np=negs))
        self.num_graphs = B * len(self.batches)


def draw_ids_device(schema, data, nsets, seed, device):
    """nsets fresh id sets for the formulas of `data`, drawn ON the device: an int64 tensor [nsets, n_ids] whose rows have
    the layout of FusedTrainStep.flatten_ids -- [anchors of batch 0, slot-major | ... | targets | negatives]; every id
    uniform over the ids of its slot's mode (the same distribution as StepData's host draws)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    ids_of = {m: torch.from_numpy(np.asarray(v)).to(device) for m, v in schema.ids.items()}

    def pick(mode, B):
        pool = ids_of[mode]
        return pool[torch.randint(len(pool), (nsets, B), device=device, generator=gen)]
    cols = []
    for b in data.batches:
        B = len(b['targets_np'])
        for m in b['formula'].anchor_modes:
            cols.append(pick(m, B))
    for _ in range(2):                                   # targets, then negatives
        for b in data.batches:
            cols.append(pick(b['formula'].target_mode, len(b['targets_np'])))
    return torch.cat(cols, dim=1).contiguous()


class _Q(object):
    __slots__ = ('anchor_nodes',)

    def __init__(self, anchor_nodes):
        self.anchor_nodes = anchor_nodes


def step_modules(model, data):
    """The step through the drop-in modules (one autograd graph, one backward)."""
    from mpqe_amd import ops
    model.zero_grad(set_to_none=True)
    loss = None
    for b in data.batches:
        out = model.encode(b['formula'], b['queries'], b['anchor_ids'], b['var_ids'], b['graph'])
        pos = model.score(b['formula'], out, b['targets'])
        neg = model.score(b['formula'], out, b['negs'])
        l = ops.hinge(pos, neg, 1.0) * b['weight']
        loss = l if loss is None else loss + l
    loss.backward()
    return loss


def self_check(model, fstep, packed, data, world):
    """Before anything is timed: the fused step on the first packed step against the drop-in module path (itself
    pinned to the oracle and the reference-generated goldens by tests/) on the same batches -- loss and every
    parameter gradient. A mismatch ends the run with a non-zero status: no throughput line for wrong results."""
    loss = fstep.run(packed)
    fstep.check()
    got = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    got_loss = float(loss[0].item())
    for p in model.parameters():
        p.grad = None
    ref_loss = float(step_modules(model, data).item()) / world       # (the 1 / world of the mean is in the packed weights)
    bad = []
    if abs(got_loss - ref_loss) > 1e-5 * abs(ref_loss) + 1e-6:
        bad.append('loss %.8g vs %.8g' % (got_loss, ref_loss))
    for k, p in model.named_parameters():
        ref = (torch.zeros_like(p) if p.grad is None else p.grad) / world
        err = (got[k] - ref).abs()
        tol = 1e-4 * ref.abs() + 2e-6
        if bool((err > tol).any()):
            bad.append('%s: max abs err %.3g' % (k, float(err.max())))
    for p in model.parameters():
        p.grad = None
    fstep.bind_grads()
    if bad:
        sys.stderr.write('bench self-check FAILED (fused step vs module path): ' + '; '.join(bad) + '\n')
        raise SystemExit(3)


def fresh_check(fstep, desc, ids, data, exact=True):
    """The timed loop's form of a step -- descriptors + a flat id tensor already on the device, the touch plan built
    inside the step -- against pack() from host arrays of the same ids: same loss and gradients, bit for bit (exact;
    the level form under the learned readouts adds table gradients with fp32 atomics: gradient tolerance there)."""
    loss_a = fstep.run(fstep.pack(desc, ids=ids)).clone()
    ga = fstep.flat_grad.clone()
    h = ids.cpu().numpy()
    batches, oa = [], 0
    na = sum(len(b['targets_np']) * b['A'] for b in data.batches)
    ng = sum(len(b['targets_np']) for b in data.batches)
    og = 0
    for d, b in zip(desc, data.batches):
        B, A = len(b['targets_np']), b['A']
        batches.append(dict(formula=d['formula'], weight=d['weight'], anchor_ids=h[oa:oa + A * B].reshape(A, B).T.copy(),
                            targets=h[na + og:na + og + B].copy(), negs=h[na + ng + og:na + ng + og + B].copy()))
        oa += A * B
        og += B
    loss_b = fstep.run(fstep.pack(batches))
    fstep.check()
    same = (torch.equal(loss_a, loss_b) and torch.equal(ga, fstep.flat_grad)) if exact else \
        (torch.allclose(loss_a, loss_b, rtol=1e-5, atol=1e-6) and torch.allclose(ga, fstep.flat_grad, rtol=1e-4, atol=2e-6))
    if not same:
        sys.stderr.write('bench self-check FAILED: device-resident ids vs host ids differ\n')
        raise SystemExit(3)


def exchange_check(fstep, exchange, xplan, packed):
    """One step reduced both ways: StepExchange.reduce against a dense all-reduce of the whole flat gradient. Returns None
    when they agree, else the reason the dense form is used."""
    import torch.distributed as dist
    try:
        fstep.run(packed)
        ref = fstep.flat_grad.clone()
        if exchange.backend == 'gloo':
            h = ref.cpu()
            dist.all_reduce(h)
            ref.copy_(h)
        else:
            dist.all_reduce(ref)
        exchange.reduce(xplan, packed=packed)
        torch.cuda.synchronize()
        err = float((fstep.flat_grad - ref).abs().max())
        tol = 1e-6 * max(1.0, float(ref.abs().max()))
        bad = torch.tensor([1.0 if not (err <= tol) else 0.0], device='cpu' if exchange.backend == 'gloo' else ref.device)
        dist.all_reduce(bad)                              # (every rank takes the same branch)
        return None if float(bad) == 0.0 else 'exchange differed from the dense all-reduce (max abs %.3g): dense form used' % err
    except Exception as e:                                # noqa: BLE001 -- any failure of the untested N-rank path: say so
        return 'exchange failed (%s: %s): dense form used' % (type(e).__name__, e)


def pack_for_fused(step, data, scale=1.0, resident=False):
    """scale: 1 / world_size under data parallelism -- the mean over ranks is folded into the batch weights,
    so the all-reduce (a sum) leaves the averaged gradient with no extra pass over the bucket.
    resident: the ids go to HBM first (one tensor in the library's layout) and the packed step reads them there, as the
    timed loop's steps do -- for the kernel timings and the replay figure; otherwise the ids stay in host memory (the
    `pack_ms` loops: read in place from the packed step's pinned buffer)."""
    batches = [dict(formula=b['formula'], anchor_ids=b['anchor_np'], targets=b['targets_np'],
                    negs=b['negs_np'], weight=b['weight'] * scale) for b in data.batches]
    if resident:
        ids = torch.from_numpy(step.flatten_ids(batches)).to(step.device)
        return step.pack([dict(formula=b['formula'], weight=b['weight'], batch_size=len(b['targets'])) for b in batches],
                         ids=ids)
    return step.pack(batches)


def time_fused_kernels(step, packed, data, model, readout, reps=20):
    """Per-launch durations of the MFMA kernels of the fused step, from HIP events the library records
    around single launches on the stream of each launch (order: mpqe_amd.h). Flops are the EXECUTED
    ones: with the TM readout node states that cannot reach the target row are skipped
    (mpqe_amd.fused.live_units), so they are fewer than SURVEY 8d's 2 D^2 (E + N) per graph and pass."""
    import ctypes
    from mpqe_amd.data_utils import RGCNQueryDataset
    from mpqe_amd.fused import live_units
    D = model.emb_dim
    prune = not (step.flags & 1)
    # library batch i = bench batch packed.order[i]; lane l owns library batches [lane_begin[l], lane_begin[l+1])
    tmpl = [data.batches[j]['graph'].template for j in packed.order]
    Ls = [RGCNQueryDataset.query_diameters[t.query_type] if model.adaptive else model.num_layers for t in tmpl]
    units = [live_units(t.query_type, L, readout, prune, step.uniform and step.uses_chain(packed))
             for t, L in zip(tmpl, Ls)]
    Lmax = max(Ls)

    def flops(lo, hi, p):
        return sum(2.0 * t.B * D * D * u[p] for t, L, u in zip(tmpl[lo:hi], Ls[lo:hi], units[lo:hi]) if L > p)
    total = sum(flops(0, len(tmpl), p) for p in range(Lmax))
    if step.learned and step.uses_chain(packed):
        # the MLP readout on the chain form: two Linear layers on every node row -- 2 N more [B, D] x [D, D] products per
        # batch, forward, backward-x and weight gradient alike
        # (targetmlp: rows [target | node] of the N - 1 other nodes, 2 + 1 products each; concat: L + 1 per node)
        per_graph = {'mlp': lambda t, L: 2 * t.N, 'targetmlp': lambda t, L: 3 * (t.N - 1),
                     'concat': lambda t, L: (L + 1) * t.N}[readout]
        total += sum(2.0 * t.B * D * D * per_graph(t, L) for t, L in zip(tmpl, Ls))
    plan = []                                            # (kernel, flops) per event pair, in library order
    lanes = [(packed.lane_begin[l], packed.lane_begin[l + 1]) for l in range(len(packed.lane_begin) - 1)]
    lane_total = [sum(flops(lo, hi, p) for p in range(Lmax)) for lo, hi in lanes]
    if step.uses_chain(packed):
        # forward + backward-x levels (+ gather, scores) in one launch; then the weight gradients (the chain form
        # runs on the caller's stream whatever the lane split)
        if step.merged(packed):
            # merged launch: the weight-gradient tiles are workgroups of the chain launch (second pair: empty)
            plan += [('step_chain_kernel', 3.0 * total), ('(empty pair)', 0.0)]
        else:
            plan += [('step_chain_kernel', 2.0 * total)]
            plan += [('step_tail_kernel', total)]
        plan += [('step_reduce_kernel', 0.0)]           # (the step's last launch: no MFMA work)
    else:
        for p in range(Lmax):
            plan += [('step_layer_fwd_kernel', flops(lo, hi, p)) for lo, hi in lanes if max(Ls[lo:hi]) > p]
        for p in range(Lmax - 1, -1, -1):
            plan += [('step_layer_bwd_x_kernel', flops(lo, hi, p)) for lo, hi in lanes if max(Ls[lo:hi]) > p]
        plan.append(('step_tail_kernel', total))
        plan.append(('step_reduce_kernel', 0.0))
    n_ev = 2 * len(plan)
    fam = {}
    for name, _ in plan:
        fam.setdefault(name, [0.0, 0, 0.0])             # ms, launches, flops
    per_launch = [0.0] * len(plan)
    for _ in range(reps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)]
        for e in evs:
            e.record()                                   # creates the underlying hipEvent_t
        arr = (ctypes.c_void_p * n_ev)(*[e.cuda_event for e in evs])
        step.run(packed, events=arr)
        torch.cuda.synchronize()
        for k, (name, fl) in enumerate(plan):
            f = fam[name]
            f[0] += evs[2 * k].elapsed_time(evs[2 * k + 1])
            per_launch[k] += evs[2 * k].elapsed_time(evs[2 * k + 1])
            f[1] += 1
            f[2] += fl
    out = []
    for name, (ms, n, fl) in fam.items():
        if name.startswith('('):
            continue
        out.append(dict(kernel=name, launches_per_step=n // reps, avg_launch_us=ms / n * 1e3,
                        algorithmic_flops_per_launch=fl / n, achieved=fl / (ms * 1e-3) / 1e12,
                        total_us_per_step=ms / reps * 1e3,
                        launches=[dict(us=per_launch[k] / reps * 1e3, gflop=fl / 1e9)
                                  for k, (nm, fl) in enumerate(plan) if nm == name]))
    return out, 3.0 * total


def pmc_traffic(kernel):
    """(bytes, source): HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    that has it (profiles/r*_chain_pmc.json, made by tools/pmc_summary.py: separate FETCH_SIZE and WRITE_SIZE passes,
    FETCH_SIZE doubled as the MI355X guide prescribes for gfx950), and the file it came from. PMC counters need
    the rocprofv3 wrapper, so they are NOT collected by the process that prints the line: the value is the one
    measured when that profile was taken (same command, same code when the file's round tag is current);
    (None, None) when no summary is committed for the kernel."""
    import glob
    # (r*_chain_pmc.json: the default workload's summaries; the other configurations' -- r*_mutag_pmc.json ... -- list the same
    # kernel names at other shapes)
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_chain_pmc.json')), reverse=True):
        try:
            table = json.load(open(path))
        except (OSError, ValueError):
            continue
        for name, rec in table.items():
            if name.split('<')[0] == kernel and isinstance(rec, dict) and rec.get('hbm_bytes') is not None:
                return rec.get('hbm_bytes'), 'profiles/' + os.path.basename(path)
    return None, None


def layer_work(data, model):
    """Algorithmic work of one step's R-GCN layers: per query graph and executed
    layer flops fwd = 2 D^2 (E+N), fwd+bwd = 6 D^2 (E+N); scatter-aggregate bytes fwd+bwd =
    12 D (E+2N)."""
    from mpqe_amd.data_utils import RGCNQueryDataset
    D = model.emb_dim
    flops_fwd = flops_all = bytes_all = 0
    launches = 0
    for b in data.batches:
        t = b['graph'].template
        L = RGCNQueryDataset.query_diameters[t.query_type] if model.adaptive else model.num_layers
        flops_fwd += L * t.B * 2 * D * D * (t.E + t.N)
        flops_all += L * t.B * 6 * D * D * (t.E + t.N)
        bytes_all += L * t.B * 12 * D * (t.E + 2 * t.N)
        launches += L
    return flops_fwd, flops_all, bytes_all, launches


def time_layer_forward(model, data, reps=20):
    """Average duration of the dominant kernel (the fused template layer forward,
    rgcn_tmpl_fwd_kernel: exactly one launch per mpqe_rgcn_template_fwd call), measured with
    HIP events on the stream the kernel is launched on, over the layer calls of one step."""
    from mpqe_amd import ops
    from mpqe_amd.data_utils import RGCNQueryDataset
    D = model.emb_dim
    calls = []
    for b in data.batches:
        t = b['graph'].template
        L = RGCNQueryDataset.query_diameters[t.query_type] if model.adaptive else model.num_layers
        x = torch.randn(t.B * t.N, D, device='cuda')
        for i in range(L):
            layer = model.layers[i] if i < L - 1 else model.layers[-1]
            calls.append((t, x, layer, i < L - 1))
    lib = ops.lib()
    stream = torch.cuda.current_stream()
    outs = [torch.empty_like(c[1]) for c in calls]

    def launch(c, out):
        t, x, layer, relu = c
        st = lib.mpqe_rgcn_template_fwd(t.qid, t.B, t.et_ptr, x.data_ptr(), layer.basis.data_ptr(),
                                        layer.num_relations, layer.root.data_ptr(), layer.bias.data_ptr(), D, D,
                                        int(relu), out.data_ptr(), stream.cuda_stream)
        assert st == 0
    for c, o in zip(calls, outs):
        launch(c, o)
    torch.cuda.synchronize()
    total_ms, n = 0.0, 0
    for _ in range(reps):
        evs = []
        for c, o in zip(calls, outs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            launch(c, o)
            e1.record(stream)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        for e0, e1 in evs:
            total_ms += e0.elapsed_time(e1)
            n += 1
    return total_ms / n * 1e-3, len(calls)


def time_scatter_aggregate(D=256, reps=20):
    """The scatter-aggregate of the general-graph path -- the destination-sorted segmented sum `segment_sum_kernel`
    (mpqe_rgcn_general_aggregate: out[i] = relu(bias + msg[E + i] + sum of the message rows of the edges into i)) --
    timed alone, in this run, with events on the stream it is launched on, on 3-inter query graphs given as a plain
    edge list: at BASELINE.json configs[4]'s shape (B = 8192, D = 256: 92 MB of rows, which the 256 MB Infinity Cache
    holds) and at B = 65536 (738 MB: beyond it, so that one is an HBM rate). Algorithmic bytes per launch
    4 D (E + 2 Nn); peak 8 TB/s."""
    import ctypes
    from mpqe_amd import ops
    dev = torch.device('cuda', torch.cuda.current_device())
    lib = ops.lib()
    src, dst, N = np.array([0, 1, 2]), np.array([3, 3, 3]), 4          # 3-inter template (reference data_utils.py:346-350)
    recs = []
    for B in (8192, 65536):
        offs = (np.arange(B, dtype=np.int64) * N)[:, None]
        ei = torch.from_numpy(np.stack([(src[None] + offs).reshape(-1), (dst[None] + offs).reshape(-1)])).to(dev)
        E, Nn, R = int(ei.shape[1]), B * N, 128
        et = torch.randint(0, R, (E,), device=dev, dtype=torch.long)
        plan = ops.GraphPlan(ei, et, Nn, R)
        msg = torch.randn(E + Nn, D, device=dev)
        bias = torch.randn(D, device=dev)
        out = torch.empty(Nn, D, device=dev)
        stream = torch.cuda.current_stream()

        def launch():
            st = lib.mpqe_rgcn_general_aggregate(plan.buf.data_ptr(), Nn, E, R, msg.data_ptr(), bias.data_ptr(), D, 1,
                                                 out.data_ptr(), stream.cuda_stream)
            assert st == 0
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            launch()
            e1.record(stream)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        us = float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
        nbytes = 4 * D * (E + 2 * Nn)
        # sanity (the kernel is parity-tested in tests/test_kernels.py): every message row is added exactly once --
        # without the ReLU, the column sums of the output are the column sums of all message rows + Nn x bias
        st = lib.mpqe_rgcn_general_aggregate(plan.buf.data_ptr(), Nn, E, R, msg.data_ptr(), bias.data_ptr(), D, 0,
                                             out.data_ptr(), stream.cuda_stream)
        assert st == 0
        got, want = out.double().sum(0), msg.double().sum(0) + Nn * bias.double()
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-2), float((got - want).abs().max())
        recs.append({'graphs': B, 'nodes': Nn, 'edges': E, 'dim': D, 'algorithmic_bytes_per_launch': nbytes,
                     'avg_launch_us': us, 'achieved': nbytes / (us * 1e-6) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     'working_set_vs_infinity_cache': 'inside (a fabric rate)' if nbytes < 256e6 else 'beyond (an HBM rate)'})
        del plan, msg, out, ei, et
    return {'bound': 'hbm', 'kernel': 'segment_sum_kernel', 'shapes': recs, 'frac': recs[-1]['frac'],
            'achieved': recs[-1]['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'note': 'top-level achieved / frac = the shape beyond the Infinity Cache'}


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(args, schema, model_state, node_maps, rel_ids, mode_ids, cfg, seconds):
    """The oracle in the reference's op sequence (per-edge weight copy + bmm + index_add, two
    encoder passes per loss: reference model.py:280-305, 478-482) timed on the host cores for a
    bounded sample of the same workload: whole formula batches of the full mix, forward + backward,
    drawn in mix order until the time budget is spent. The torch thread count is calibrated on one
    3-chain batch (1, 8, 16, 32, 64 threads at most) and the fastest is used and reported."""
    from mpqe_amd import synthetic
    from oracle import ref_cpu
    B = args.batch_size
    rng = np.random.RandomState(777)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model_state.items()}

    def make(qt):
        formula = synthetic.sample_formula(schema, qt, rng)
        anchors = np.stack([synthetic._pick(schema, m, rng, size=B) for m in formula.anchor_modes], axis=1)
        queries = [_Q(tuple(int(v) for v in anchors[b])) for b in range(B)]
        col = ref_cpu.collate(formula, queries, rel_ids, mode_ids)
        return formula, col, synthetic._pick(schema, formula.target_mode, rng, size=B), \
            synthetic._pick(schema, formula.target_mode, rng, size=B)

    def run(item, w):
        formula, col, tg, ng = item
        for p in params.values():
            p.grad = None
        loss = w * ref_cpu.margin_loss(params, cfg, node_maps, formula, col, tg, ng)
        loss.backward()

    avail = usable_cores()
    cand = sorted(set(c for c in (1, 8, 16, 32, min(avail, 64)) if c <= avail))     # (256 threads: 50 s per batch)
    probe = make('3-chain')
    timing = {}
    for c in cand:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        run(probe, 1.0)                                  # warm-up; already decisive when a setting is pathological
        warm = time.perf_counter() - t0
        if timing and warm > 4 * min(timing.values()) + 1.0:
            timing[c] = warm                             # (e.g. 256 threads on this box: ~50 s per batch)
            break
        t0 = time.perf_counter()
        run(probe, 1.0)
        timing[c] = time.perf_counter() - t0
    best = min(timing, key=timing.get)
    torch.set_num_threads(best)
    done, t0, nbatch = 0, time.perf_counter(), 0
    items = [(make(qt), 1.0 if qt == '1-chain' else (INTER_WEIGHT if 'inter' in qt else PATH_WEIGHT))
             for qt, _ in synthetic.FULL_MIX]
    t0 = time.perf_counter()
    while True:
        item, w = items[nbatch % len(items)]
        run(item, w)
        done += B
        nbatch += 1
        if time.perf_counter() - t0 >= seconds and nbatch >= len(items):
            break
        if time.perf_counter() - t0 >= 4 * seconds:
            break
    el = time.perf_counter() - t0
    # SURVEY 8d also asks for the 1-thread figure: the calibration's 3-chain batch (B query graphs, forward + backward)
    threads_1 = dict(value=B / timing[1], unit='query-graphs/s', sample='one 3-chain batch of B=%d, 1 torch thread' % B) if 1 in timing else None
    return dict(value=done / el, unit='query-graphs/s', cores=best, kind='port', threads_1=threads_1,
                sample='%d formula batches of B=%d in full-mix order (%d query graphs, %.1f s), oracle in the '
                       'reference op sequence (weight copy + bmm + index_add, two encoder passes), torch %s; '
                       '%d usable host cores; 3-chain batch seconds by thread count: %s'
                       % (nbatch, B, done, el, torch.__version__, avail,
                          ', '.join('%d: %.2f' % (c, t) for c, t in sorted(timing.items()))))


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as a child `torch.distributed.run` -- this
    process has not touched a GPU (counting devices does not) and only relays the child's exit status."""
    ndev = torch.cuda.device_count()
    if args.backend == 'nccl' and ndev < args.gpus:
        sys.stderr.write('bench.py --gpus %d: only %d GPU(s) visible (one process per GPU)\n' % (args.gpus, ndev))
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU)' % (args.gpus, world))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        ndev = torch.cuda.device_count()
        if args.backend == 'nccl' and local >= ndev:
            raise SystemExit('rank %d has no GPU (%d visible): one process per GPU' % (local, ndev))
        torch.cuda.set_device(local % ndev)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local % ndev))
        else:
            dist.init_process_group('gloo')
        assert dist.get_world_size() == args.gpus
    else:
        torch.cuda.set_device(0)
    device = torch.device('cuda', torch.cuda.current_device())

    from mpqe_amd import ops, synthetic
    from mpqe_amd.data_utils import make_feature_modules
    from mpqe_amd.encoders import DirectEncoder
    from mpqe_amd.model import RGCNEncoderDecoder
    for kv in args.debug_opt:
        name, _, val = kv.partition('=')
        ops.lib().mpqe_debug_option(name.encode(), int(val or 1), 1)
    torch.manual_seed(0)                                  # identical replicas on every rank
    D = args.embed_dim
    schema = synthetic.make_schema(*synthetic.KG_SHAPES[args.kg], seed=0)
    graph = synthetic.SchemaGraph(schema, D)
    fm, node_maps = make_feature_modules(schema.ids, D, schema.num_entities)
    adaptive = args.readout == 'mp'
    model = RGCNEncoderDecoder(graph, DirectEncoder(None, fm, node_maps), readout=args.readout, num_layers=3,
                               shared_layers=False, adaptive=adaptive, weight_decay=0)
    cpu_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    model.validate = False                                # no per-call D2H flag read in the timed loop
    rng = np.random.RandomState(1000 + rank)
    pool = [StepData(schema, model, args.batch_size, rng, device) for _ in range(4)]      # 4 formula sets

    use_fused = args.path in ('auto', 'fused') and args.readout in ('mp', 'sum', 'max', 'mlp', 'targetmlp', 'concat')
    learned = args.readout in ('mlp', 'targetmlp', 'concat')        # (fused step on the level form, the readout inside the call: csrc/step_readout.h)
    fresh = use_fused and not args.replay and not args.graph
    reducer = fstep = packed = captured = exchange = xplans = fresh_ids = descs = None

    def auto_sizes(run_once):
        """--repeats 0 / --fresh-sets 0: sized from a few steps timed here (every rank takes the slowest rank's estimate)."""
        if args.repeats <= 0:
            for _ in range(3):
                run_once()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                run_once()
            torch.cuda.synchronize()
            est = (time.perf_counter() - t0) / 5 * args.steps
            if world > 1:
                import torch.distributed as dist
                t = torch.tensor([est], device=device if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                est = float(t.item())
            args.repeats = int(max(25 if est < 0.05 else 5, min(500, -(-0.6 // max(est, 1e-6)))))
        if args.fresh_sets <= 0:
            args.fresh_sets = min(16384, args.warmup + args.steps * args.repeats)
        return args.warmup + args.steps * max(1, args.repeats)
    if use_fused:
        from mpqe_amd.fused import FusedTrainStep
        touch = {'step': 'step', 'pack': 'pack', 'atomics': False}[args.touch]
        fstep = FusedTrainStep(model, lanes=args.lanes, prune=not args.no_prune, chain=not args.no_chain,
                               ksplit=not args.no_ksplit, eight_waves=args.eight_waves, uniform=not args.no_uniform,
                               touch=touch, sparse_tables=args.sparse_tables, host_ids=args.host_ids,
                               merge_tail=None if args.merge_tail < 0 else bool(args.merge_tail))
        packed = [pack_for_fused(fstep, d, 1.0 / world, resident=True) for d in pool]
        captured = [fstep.capture(p) for p in packed] if args.graph else None
        n_total = auto_sizes(lambda: fstep.run(packed[0]))
        if fresh:
            # the timed steps' inputs: per formula set its descriptors (formula, weight, size) and, resident in HBM before
            # the timed region starts, one NEVER-SEEN id set per step
            descs = [[dict(formula=b['formula'], weight=b['weight'] / world, batch_size=len(b['targets_np']))
                      for b in d.batches] for d in pool]
            per_set = (min(n_total, max(args.fresh_sets, len(pool))) + len(pool) - 1) // len(pool)
            fresh_ids = [draw_ids_device(schema, d, per_set, 77000 + 16 * rank + j, device) for j, d in enumerate(pool)]
        if world > 1 and not args.dense_allreduce:
            # gradient exchange of the steps (collective, once per formula set): which relation matrices ANY rank touches
            from mpqe_amd.parallel import StepExchange
            exchange = StepExchange(fstep, transport=args.exchange)
            xplans = [exchange.plan(p, key=j) for j, p in enumerate(packed)]
    else:
        n_total = auto_sizes(lambda: step_modules(model, pool[0]))
        if world > 1:
            from mpqe_amd.parallel import GradReducer
            reducer = GradReducer(model)

    def one_step(i):
        j = i % len(pool)
        if use_fused:
            if captured is not None:
                loss = captured[j].replay()
            elif fresh:
                ids = fresh_ids[j][(i // len(pool)) % fresh_ids[j].shape[0]]
                pk = fstep.pack(descs[j], ids=ids)
                loss = fstep.run(pk)
            else:
                pk = packed[j]
                loss = fstep.run(pk)
            if exchange is not None:
                # (the 1 / world of the mean is in the batch weights; the row exchange of a step that builds its own touch
                # plan takes its keys from the step that has just run)
                exchange.reduce(xplans[j], packed=pk)
            elif world > 1:
                import torch.distributed as dist
                dist.all_reduce(fstep.flat_grad)          # the literal form: every parameter's dense gradient
            return loss
        loss = step_modules(model, pool[j])
        if reducer is not None:
            reducer.all_reduce()
        return loss

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # (learned readouts: mlp rides on the chain form -- csrc/step_chain.h --, targetmlp / concat on the level form)
    level_form = bool(learned and use_fused and (fstep.flags & 2))      # (MPQE_STEP_NO_CHAIN)
    if use_fused and not args.no_self_check and not args.sparse_tables:      # (the check compares DENSE gradients)
        self_check(model, fstep, packed[0], pool[0], world)
        if fresh:
            fresh_check(fstep, descs[0], fresh_ids[0][0], pool[0], exact=not level_form)
    xnote = None
    if exchange is not None and exchange.transport_note:
        xnote = exchange.transport_note
    if exchange is not None:
        # the exchange against the literal dense all-reduce on one step (N-rank hardware is not available to the tests):
        # a mismatch falls back to the dense form, and says so
        xcheck = exchange_check(fstep, exchange, xplans[0], packed[0])
        if xcheck is not None:
            xnote = xcheck
            exchange = None
    step_no = 0
    for i in range(args.warmup):
        one_step(step_no)
        step_no += 1
    # The timed region is EXACTLY `steps` steps between barrier + synchronize on both sides. At ~0.07 ms per step the
    # driver's --steps 20 is a ~1.4 ms region, within launch / clock noise of a single sample: the same block is
    # repeated `--repeats` times (each one bracketed the same way, each step on the next fresh id set) and the MEDIAN
    # block is the reported one, with min and max beside it. Per block the time is the max over ranks.
    blocks = []
    for r in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(step_no + i)
        barrier()
        blocks.append(time.perf_counter() - t0)
        step_no += args.steps
    if use_fused:
        # an invalid id / a failed hand-off / a p2p peer that never arrived inside any timed step -- agreed between the ranks
        if exchange is not None:
            exchange.check()
        else:
            fstep.check()
    split_us = None
    if world > 1 and use_fused:
        # where an N-rank step's time goes (outside the timed region): events around the step's own launches and around the
        # gradient exchange, on the stream both run on, over the same fresh-id steps
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(min(20, max(4, args.steps)))]
        barrier()
        for k, (e0, e1, e2) in enumerate(evs):
            j = (step_no + k) % len(pool)
            e0.record()
            if fresh:
                pk = fstep.pack(descs[j], ids=fresh_ids[j][((step_no + k) // len(pool)) % fresh_ids[j].shape[0]])
            else:
                pk = packed[j]
            fstep.run(pk)
            e1.record()
            if exchange is not None:
                exchange.reduce(xplans[j], packed=pk)
            else:
                import torch.distributed as dist
                dist.all_reduce(fstep.flat_grad)
            e2.record()
        barrier()
        comp = float(np.median([a.elapsed_time(b) for a, b, _ in evs])) * 1e3
        exch = float(np.median([b.elapsed_time(c) for _, b, c in evs])) * 1e3
        t = torch.tensor([comp, exch], device=device if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        split_us = [float(v) for v in t.tolist()]
        if exchange is not None:
            exchange.check()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor(blocks, device=device if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blocks = [float(v) for v in t.tolist()]
    elapsed = float(np.median(blocks))
    replay_ms = None
    if fresh and rank == 0 and world == 1:
        # secondary figure: the replay of 4 pre-packed steps (round 2's headline; nothing id-dependent left in the loop)
        rb = []
        for r in range(max(1, min(args.repeats, 9))):
            barrier()
            t0 = time.perf_counter()
            for i in range(args.steps):
                fstep.run(packed[i % len(pool)])
            barrier()
            rb.append(time.perf_counter() - t0)
        replay_ms = float(np.median(rb)) / args.steps * 1e3

    graphs_per_step = pool[0].num_graphs * world
    value = graphs_per_step * args.steps / elapsed
    out = {
        'metric': 'query-graphs/sec (encoder fwd+bwd)', 'value': value, 'unit': 'query-graphs/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'timed_blocks': {'repeats': len(blocks), 'reported': 'median block of --steps steps',
                         'ms_per_step_min': min(blocks) / args.steps * 1e3,
                         'ms_per_step_max': max(blocks) / args.steps * 1e3},
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'AIFB-shaped full query mix: 11 batches x B=%d per step (1/2/3-chain, 2x each of '
                               '2-inter, 3-inter, 3-inter_chain, 3-chain_inter), embed_dim=%d, readout=%s, '
                               'adaptive=%s, num_layers=3 unshared, KG %s (%d entities, %d typed relations)'
                               % (args.batch_size, D, args.readout, adaptive, args.kg, schema.num_entities,
                                  len(graph.rel_edges)),
                   'global_query_graphs_per_step': graphs_per_step,
                   'parallelism': 'dp%d (graphs sharded by rank, RCCL all-reduce of gradients)' % world
                                  if world > 1 else 'single GPU',
                   'host_path': ('fused step: one C-ABI call per step%s, %d stream lane(s)%s'
                                 % ((' (level form, learned readout inside)' if level_form else
                                     ' (chain form, the readout\'s Linear layers inside the chain launch)') if learned else '',
                                    args.lanes, ', replayed from a hipGraph' if args.graph else '')) if use_fused
                                else 'drop-in modules (one autograd graph per step)',
                   'ids': ('fresh every step: %d pre-drawn id sets resident in HBM, 4 formula sets; id -> row lookups and the '
                           '%s inside the timed step' % (sum(int(t.shape[0]) for t in fresh_ids),
                                                         'table gradients by fp32 atomics' if level_form else 'touch plan (%s)' % args.touch))
                          if fresh else 'replay of 4 pre-packed steps'},
    }
    if replay_ms is not None:
        out['replay'] = {'ms_per_step': replay_ms, 'value': graphs_per_step / (replay_ms * 1e-3),
                         'note': '4 pre-packed steps replayed (no fresh ids): secondary figure'}
    if xnote is None and exchange is not None and exchange.transport_note:
        xnote = exchange.transport_note          # (a fall-back decided during the run: StepExchange.check)
    if xnote is not None:
        out['exchange_note'] = xnote
    if split_us is not None:
        out['compute'] = {'us_per_step': split_us[0], 'note': 'events around the step\'s own launches (max over ranks, median step)'}
    if world > 1 and use_fused:
        dense_bytes = fstep.flat_grad.numel() * 4
        out['exchange'] = ({'form': 'dense all-reduce of the flat gradient buffer', 'bytes_per_rank_per_step': int(2 * (world - 1) / world * dense_bytes)}
                           if exchange is None else
                           {'form': 'bucket all-reduce of the relation matrices some rank touches (+ root, bias, mode rows) and an '
                                    'all-gather of touched entity-table rows',
                            'bytes_per_rank_per_step': int(np.mean([x.wire_bytes for x in xplans])),
                            'bucket_bytes': int(np.mean([x.bucket_bytes for x in xplans])),
                            'table_rows_gathered': int(np.mean([x.entries for x in xplans])),
                            'forms': sorted(set(x.form for x in xplans)), 'transport': exchange.transport,
                            'dense_gradient_bytes': dense_bytes})
        out['exchange']['transport_requested'] = args.exchange
        if args.exchange == 'p2p':
            out['exchange']['p2p_self_test'] = ('passed (one exchange against torch.distributed\'s all-reduce at start-up)'
                                                if exchange is not None and exchange.transport == 'p2p' else
                                                'not in use: %s' % (xnote or 'fell back'))
        if split_us is not None:
            out['exchange']['us_per_step'] = split_us[1]
    if rank == 0 and use_fused and world == 1 and not args.no_pack_ms:
        # host side of a step whose ids arrive from the HOST (a data loader's numpy arrays): packing = descriptors (cached
        # per formula) + ids into the packed step's pinned buffer, which the kernels read in place (--host-ids copy: one host-to-device copy per
        # pack, in stream order). Outside `value` (SURVEY.md 8d excludes collation;
        # `value` itself runs on fresh ids that are already resident), reported so that it can be held against ms_per_step.
        host = StepData(schema, model, args.batch_size, np.random.RandomState(4242), device)
        torch.cuda.synchronize()
        tp = []
        for _ in range(5):
            t0 = time.perf_counter()
            pk = pack_for_fused(fstep, host)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            fstep.run(pk)
            torch.cuda.synchronize()
            tp.append((t1 - t0, t2 - t0, time.perf_counter() - t2))
        for _ in range(8):
            pk = pack_for_fused(fstep, host)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            pk = pack_for_fused(fstep, host)
        t_pack = (time.perf_counter() - t0) / 100
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fstep.run(pack_for_fused(fstep, host))
        torch.cuda.synchronize()
        t_loop = (time.perf_counter() - t0) / 200
        t_dev = t_flat = None
        if fresh:
            # ... and ids the collate function wrote in the library's layout (one int64 array per step, pinned or not):
            # descriptors + ONE staging copy + the host-to-device copy; a different id set every step
            hids = [fresh_ids[0][k].cpu().numpy() for k in range(min(64, fresh_ids[0].shape[0]))]
            for k in range(8):
                fstep.run(fstep.pack(descs[0], ids=hids[k % len(hids)]))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(200):
                fstep.run(fstep.pack(descs[0], ids=hids[k % len(hids)]))
            torch.cuda.synchronize()
            t_flat = (time.perf_counter() - t0) / 200
            d0, i0 = descs[0], fresh_ids[0]
            for k in range(8):
                fstep.pack(d0, ids=i0[k])
            t0 = time.perf_counter()
            for k in range(200):
                fstep.pack(d0, ids=i0[k % i0.shape[0]])
            t_dev = (time.perf_counter() - t0) / 200
        out['pack_ms'] = {'host_call': float(np.median([a for a, _, _ in tp])) * 1e3,
                          'until_device_idle': float(np.median([b for _, b, _ in tp])) * 1e3,
                          'first_run_with_descriptor_upload': float(np.median([c for _, _, c in tp])) * 1e3,
                          'host_call_back_to_back': t_pack * 1e3,
                          'host_call_device_resident_ids': None if t_dev is None else t_dev * 1e3,
                          'pack_and_run_loop_per_step': t_loop * 1e3,
                          'flat_host_ids_loop_per_step': None if t_flat is None else t_flat * 1e3,
                          'note': 'pack_and_run_loop_per_step: per-batch host arrays [B, A] / [B] -> layout conversion + copy '
                                  '+ step; flat_host_ids_loop_per_step: host ids already in the library layout -> copy + step'}
    if rank == 0:
        flops_fwd, flops_all, bytes_all, launches = layer_work(pool[0], model)
        executed = None
        default_workload = args.readout == 'mp' and args.batch_size == 512 and D == 128 and args.kg == 'aifb'
        if use_fused:
            fams, executed = time_fused_kernels(fstep, packed[0], pool[0], model, args.readout)
            dom = max(fams, key=lambda f: f['total_us_per_step'])
            # frac = EXECUTED flops (after liveness pruning and the batch-uniform node states) over the kernel's measured time;
            # frac_nominal_8d = SURVEY 8d's nominal count for the same launches (every node state of every graph: products
            # the kernel never forms) over the same time -- beside it so that the line explains itself
            share = dom['algorithmic_flops_per_launch'] * dom['launches_per_step'] / max(executed, 1.0)
            nominal = flops_all * share / (dom['total_us_per_step'] * 1e-6) / 1e12
            out['roofline'] = {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': dom['achieved'],
                               'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': dom['achieved'] / MFMA_F32_PEAK_TFLOPS,
                               'frac_nominal_8d': nominal / MFMA_F32_PEAK_TFLOPS,
                               # (the committed PMC summary is of the default workload: no figure for another one)
                               'traffic': pmc_traffic(dom['kernel'])[0] if default_workload else None,
                               'traffic_source': pmc_traffic(dom['kernel'])[1] if default_workload else None,
                               'avg_launch_us': dom['avg_launch_us'],
                               'algorithmic_flops_per_launch': dom['algorithmic_flops_per_launch'],
                               'nominal_8d_flops_per_launch': flops_all * share / max(dom['launches_per_step'], 1),
                               'launches_per_step': dom['launches_per_step']}
            out['kernels'] = fams
            out['kernels_note'] = ('every launch of the step, timed with HIP event pairs recorded INSIDE the library call on the '
                                   'stream of the launch (a separate pass of 20 steps): each pair adds ~2 - 3 us, so the entries '
                                   'sum to more than ms_per_step, which the timed loop measures without them; rocprofv3 '
                                   'averages: profiles/*_kernel_stats.csv')
            # the fused kernel also IS the path's scatter-aggregate (gather of source rows + neighbour sum + write-back
            # happen inside it, on states that live in LDS): SURVEY 8d's bytes, 12 D (E + 2N) per graph and executed layer,
            # over the same kernel time. An ACCOUNTING ratio -- those bytes never travel (`traffic` is what does) -- hence
            # the key's name; the HBM roofline evidence is `roofline_scatter`.
            if dom['kernel'] == 'step_chain_kernel':
                gbs = bytes_all / (dom['avg_launch_us'] * 1e-6) / 1e9
                out['accounting_scatter_bytes_over_chain_time'] = {
                    'kernel': dom['kernel'], 'gbs': gbs, 'over_hbm_peak': gbs / HBM_PEAK_GBS,
                    'algorithmic_bytes_per_launch': bytes_all, 'fabric_bytes_per_launch': pmc_traffic(dom['kernel'])[0],
                    'note': 'SURVEY 8d scatter-aggregate bytes / chain-kernel time: the states live in LDS, these bytes never '
                            'travel; not a bandwidth measurement (see roofline_scatter)'}
        else:
            dur, ncalls = time_layer_forward(model, pool[0])
            per_launch_flops = flops_fwd / launches
            achieved = per_launch_flops / dur / 1e12
            out['roofline'] = {'bound': 'mfma', 'kernel': 'rgcn_tmpl_fwd_kernel', 'achieved': achieved,
                               'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': achieved / MFMA_F32_PEAK_TFLOPS, 'traffic': None,
                               'avg_launch_us': dur * 1e6, 'algorithmic_flops_per_launch': per_launch_flops,
                               'launches_per_step': launches}
        if world == 1 and not args.no_scatter:
            out['roofline_scatter'] = time_scatter_aggregate()
        out['step_work'] = {'layer_flops_fwd_bwd': flops_all, 'scatter_aggregate_bytes_fwd_bwd': bytes_all,
                            'layer_tflops_over_whole_step': flops_all / (elapsed / args.steps) / 1e12}
        if executed is not None:       # flops_all is SURVEY 8d's count (every node state); executed = after pruning
            out['step_work']['layer_flops_executed'] = executed
            out['step_work']['executed_tflops_over_whole_step'] = executed / (elapsed / args.steps) / 1e12
        if world == 1 and default_workload and not args.no_dropin_loop:
            # secondary: the reference's OWN loop body (train_helpers.py:76-120: 11 x model.margin_loss with python's
            # negatives, `loss += w * ...`, loss.item(), loss.backward(), optimizer step) over pre-collated batches --
            # what a maintainer who changed only the imports gets (mpqe_amd/dropin.py), next to the per-op module path
            # the same calls took before round 5. Not `value`: that is the step alone, ids resident.
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
            import dropin_loop_bench
            out['dropin_loop'] = dropin_loop_bench.run(readout=args.readout, D=D, B=args.batch_size, iters=300,
                                                       module_iters=8)
        if world == 1 and not args.no_cpu_baseline:
            cfg = dict(readout=args.readout, scatter_op='add', num_layers=3, adaptive=adaptive, weight_decay=0)
            out['cpu_baseline'] = cpu_baseline(args, schema, cpu_state, node_maps, model.rel_ids, model.mode_ids,
                                               cfg, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

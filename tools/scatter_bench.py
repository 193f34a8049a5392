"""The GENERAL-graph R-GCN path (`RGCNConv.forward(x, edge_index, edge_type)` on a plain edge list: reference
model.py:269-305) at the BASELINE.json stress shape: B = 8192 query graphs of a 3-inter / 3-chain template, D = 256,
128 relations. Per-kernel durations come from running it under rocprofv3:

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/scatter_bench.py

Prints per template the algorithmic work of the kernels (forward sum: reads (E + Nn) message rows, writes Nn rows ->
4 D (E + 2 Nn) bytes; gather-GEMM 2 D^2 (E + Nn) flops) and the wall time of forward + backward per iteration, each
iteration timed on its own (events) so that a stall of a single iteration shows as such. Both orders of the two
templates are run: round 1's figure for 3-inter (7.97 ms against 0.68 ms) was the FIRST template of the process paying
for the caching allocator's growth during its timed iterations (hipMalloc of the 33 MB gradient / 117 MB slab
buffers), not a property of the template.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(qt, D, B, R, dev, iters=20):
    from mpqe_amd.model import RGCNConv
    from mpqe_amd.fused import _TEMPLATES            # query type -> (anchors, nodes, [(src, dst)])
    _, N, edges = _TEMPLATES[qt]
    src, dst = np.array([e[0] for e in edges]), np.array([e[1] for e in edges])
    offs = (np.arange(B, dtype=np.int64) * N)[:, None]
    ei = np.stack([(src[None] + offs).reshape(-1), (dst[None] + offs).reshape(-1)])
    et = np.random.RandomState(0).randint(0, R, size=ei.shape[1]).astype(np.int64)
    conv = RGCNConv(D, D, R, 0).to(dev)
    x = torch.randn(B * N, D, device=dev, requires_grad=True)
    ei_t, et_t = torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev)
    for _ in range(5):
        y = conv(x, ei_t, et_t, relu=True)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    ms, ms_glue = [], []
    gy = torch.ones_like(y)
    leaves = [x] + list(conv.parameters())
    for _ in range(iters):
        # the layer's own launches: the output gradient exists already (the next op's backward made it) and the
        # gradients are taken as they come (a loop that clears grads with set_to_none, PyTorch's default)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = conv(x, ei_t, et_t, relu=True)
        torch.autograd.grad([y], leaves, [gy])
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    for _ in range(iters):
        # round 1 / 2's form, for comparison: + a 33 MB ones_like fill and autograd's accumulation into existing .grad
        # tensors (two 33 MB adds) inside the timed region
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = conv(x, ei_t, et_t, relu=True)
        y.backward(torch.ones_like(y))
        e1.record()
        torch.cuda.synchronize()
        ms_glue.append(e0.elapsed_time(e1))
    E, Nn = ei.shape[1], B * N
    return dict(nodes=Nn, edges=E, D=D, fwd_sum_bytes=4 * D * (E + 2 * Nn), bwd_sum_bytes=4 * D * (E + 2 * Nn),
                gemm_flops_fwd=2 * D * D * (E + Nn), ms_fwd_bwd_median=float(np.median(ms)), ms_fwd_bwd_max=float(max(ms)),
                ms_fwd_bwd_with_autograd_glue_median=float(np.median(ms_glue)),
                ms_fwd_bwd_first=float(ms[0]), allocator_mb=torch.cuda.memory_reserved() / 2 ** 20)


def main():
    dev = torch.device('cuda:0')
    D, B, R = int(os.environ.get('SCATTER_BENCH_D', 256)), int(os.environ.get('SCATTER_BENCH_B', 8192)), 128
    # timing experiments: SCATTER_BENCH_OPTS="NAME=VALUE,..." -> the library's diagnostics switches (mpqe_debug_option)
    from mpqe_amd import ops
    for kv in filter(None, os.environ.get('SCATTER_BENCH_OPTS', '').split(',')):
        name, _, val = kv.partition('=')
        ops.lib().mpqe_debug_option(name.encode(), int(val or 1), 1)
    out = {}
    for k, qt in enumerate(('3-inter', '3-chain', '3-inter')):
        out['%d:%s' % (k, qt)] = run(qt, D, B, R, dev)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

"""Scatter-aggregate (destination-sorted segmented sum, `segment_sum_kernel`) and the grouped gather-GEMM of
the GENERAL-graph R-GCN path at the BASELINE.json stress shape (B = 8192 query graphs of a 3-inter / 3-chain
template given as a plain edge list, D = 256). Run under rocprofv3 to get the per-kernel durations:

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/scatter_bench.py

Prints the algorithmic bytes per launch so that GB/s = bytes / avg duration:
  forward sum: reads (E + Nn) message rows, writes Nn rows  -> 4*D*(E + 2*Nn) bytes
"""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from mpqe_amd import ops
    from mpqe_amd.model import RGCNConv
    from oracle import ref_cpu  # template tables only (host ints)
    dev = torch.device('cuda:0')
    D, B, R = 256, 8192, 128
    out = {}
    for qt in ('3-inter', '3-chain'):
        t = ref_cpu.TEMPLATES[qt]
        N = 1 + max(t['src'] + t['dst'])
        offs = (np.arange(B, dtype=np.int64) * N)[:, None]
        ei = np.stack([(np.array(t['src'])[None] + offs).reshape(-1), (np.array(t['dst'])[None] + offs).reshape(-1)])
        et = np.random.RandomState(0).randint(0, R, size=ei.shape[1]).astype(np.int64)
        conv = RGCNConv(D, D, R, 0).to(dev)
        x = torch.randn(B * N, D, device=dev, requires_grad=True)
        ei_t, et_t = torch.from_numpy(ei).to(dev), torch.from_numpy(et).to(dev)
        for _ in range(3):
            y = conv(x, ei_t, et_t, relu=True)
            y.sum().backward()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = conv(x, ei_t, et_t, relu=True)
            y.backward(torch.ones_like(y))
        e1.record()
        torch.cuda.synchronize()
        E, Nn = ei.shape[1], B * N
        out[qt] = dict(nodes=Nn, edges=E, D=D, fwd_sum_bytes=4 * D * (E + 2 * Nn), bwd_sum_bytes=4 * D * (E + 2 * Nn),
                       gemm_flops_fwd=2 * D * D * (E + Nn), ms_fwd_bwd=e0.elapsed_time(e1) / 20)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

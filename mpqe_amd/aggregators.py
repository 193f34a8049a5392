"""Neighbour aggregators with the reference's interface (mpqe/aggregators.py).

Only MeanAggregator is ever instantiated by the reference (utils.py:104-122, `--depth >= 1`); the
Fast / Pool variants are never constructed and are not mirrored. The mean over the sampled
neighbours' feature rows runs in the scatter kernel (mpqe_scatter_fwd/bwd, mean mode) instead of the
reference's dense [B, U] mask GEMM (aggregators.py:56-67).
"""
import math
import random

import torch
import torch.nn as nn

from . import ops


class MeanAggregator(nn.Module):
    """reference: aggregators.py:17-68. `features(nodes, mode)` is the caller's lookup closure."""

    def __init__(self, features, cuda=False):
        super(MeanAggregator, self).__init__()
        self.features = features
        self.cuda = cuda

    def forward(self, to_neighs, rel, keep_prob=0.5, max_keep=10):
        """to_neighs: one neighbour collection per node of the batch. Samples
        min(ceil(len * keep_prob), max_keep) neighbours without replacement with python `random`
        (same calls in the same order as the reference, so the same draws under the same seed) and
        returns their mean feature [B, D]."""
        samp = [set(random.sample(list(nb) if isinstance(nb, (set, frozenset)) else nb,
                                  min(int(math.ceil(len(nb) * keep_prob)), max_keep))) for nb in to_neighs]
        flat = [n for s in samp for n in s]
        rows = [i for i, s in enumerate(samp) for _ in s]
        embed = self.features(flat, rel[-1])
        if embed.dim() == 1:
            embed = embed.unsqueeze(0)
        index = torch.as_tensor(rows, dtype=torch.long, device=embed.device)
        return ops.scatter_mean(embed, index, dim=0, dim_size=len(samp))

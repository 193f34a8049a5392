"""Negative sampling on the device (reference model.py:466-476).

The reference draws one negative per query per step with python's `random.choice` -- 512 interpreter calls
per batch. Here the candidate lists of ALL queries of a formula are flattened once into CSR arrays in HBM;
a step then draws its negatives with one launch of `mpqe_sample_negatives` (a counter-based hash of the seed
and the batch position: stateless, reproduced on the CPU by oracle/ref_cpu.py) straight into the id buffer the
fused step reads. Same distribution as the reference (uniform over the list), different numbers than python's
Mersenne Twister.

    sampler = NegativeSampler(queries, device)                     # once per formula
    negs = sampler.sample(idx, seed)                               # idx: positions of the batch's queries
    sampler.sample(idx, seed, out=ids[lo:hi])      # or straight into the device id tensor handed to FusedTrainStep.pack(ids=...)
"""
import numpy as np
import torch

from . import _capi, ops


class NegativeSampler(object):
    def __init__(self, queries, device, full_list=None):
        """queries: the formula's Query objects (neg_samples / hard_neg_samples lists). full_list: for 1-chain
        formulas the reference samples from graph.full_lists[target_mode] instead (model.py:473-474)."""
        self.device = torch.device(device)
        self.n = len(queries)

        def csr(lists):
            lens = np.fromiter((len(l) for l in lists), dtype=np.int64, count=len(lists))
            off = np.zeros(len(lists) + 1, dtype=np.int64)
            np.cumsum(lens, out=off[1:])
            flat = np.fromiter((v for l in lists for v in l), dtype=np.int64, count=int(off[-1]))
            return torch.from_numpy(flat).to(self.device), torch.from_numpy(off).to(self.device)
        self.shared = None
        if full_list is not None:
            self.shared = torch.as_tensor(np.asarray(list(full_list), dtype=np.int64)).to(self.device)
        self.neg = csr([q.neg_samples if q.neg_samples is not None else () for q in queries])
        hard = [getattr(q, 'hard_neg_samples', None) for q in queries]
        self.hard = csr([h if h is not None else () for h in hard]) if any(h is not None for h in hard) else None
        self.err = ops.new_error_word(self.device)

    def sample(self, idx, seed, hard_negatives=False, out=None):
        """idx: int64 tensor / array of query positions (the batch). Returns the device tensor of negatives."""
        idx = torch.as_tensor(idx, dtype=torch.long).to(self.device)
        nq = idx.shape[0]
        if out is None:
            out = torch.empty(nq, dtype=torch.long, device=self.device)
        if out.shape[0] != nq or out.dtype != torch.long or not out.is_contiguous():
            raise ValueError('out must be a contiguous int64 tensor with one slot per query')
        lib = ops.lib()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            if hard_negatives:
                if self.hard is None:
                    raise Exception("Hard negative examples can only be used with "
                                    "intersection queries")                   # reference model.py:467-469
                cand, off = self.hard
                st = lib.mpqe_sample_negatives(cand.data_ptr(), cand.shape[0], off.data_ptr(), self.n, idx.data_ptr(),
                                               nq, seed, out.data_ptr(), self.err.data_ptr(), stream)
            elif self.shared is not None:
                st = lib.mpqe_sample_negatives(self.shared.data_ptr(), self.shared.shape[0], None, 0, None, nq, seed,
                                               out.data_ptr(), self.err.data_ptr(), stream)
            else:
                cand, off = self.neg
                st = lib.mpqe_sample_negatives(cand.data_ptr(), cand.shape[0], off.data_ptr(), self.n, idx.data_ptr(),
                                               nq, seed, out.data_ptr(), self.err.data_ptr(), stream)
        _capi.check(lib, st, 'mpqe_sample_negatives')
        return out

    def check(self):
        """IndexError if a previous draw met an empty candidate list (random.choice([]) in the reference)."""
        ops.raise_on_flags(self.err)

"""The R-GCN query encoder with the reference's module surface (mpqe/model.py:206-553):
same class names, constructor arguments, forward()/margin_loss() signatures, attributes
and state_dict keys -- every float computed by the gfx950 kernels behind include/mpqe_amd.h.
"""
import math
import random

import torch
import torch.nn as nn

from . import ops
from .data_utils import RGCNQueryDataset
from .ops import scatter_add, scatter_max, scatter_mean  # noqa: F401  (re-exported like the reference)


class RGCNConv(nn.Module):
    """reference: RGCNConv, model.py:206-310 (a vendored PyG <= 1.4 layer).

        out_i = sum_{(j -> i, r)} x_j . basis[r]  +  x_i . root  +  bias

    'add' aggregation, no edge normalisation (the reference always passes edge_norm=None,
    model.py:436, 441). num_bases must be 0 as in the reference's only construction site
    (model.py:346); the basis-decomposition branch is not built.
    """

    def __init__(self, in_channels, out_channels, num_relations, num_bases, bias=True):
        super(RGCNConv, self).__init__()
        if num_bases != 0:
            raise NotImplementedError('basis decomposition (num_bases > 0) is never reached by the '
                                      'reference (model.py:346 hard-codes 0) and is not built')
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_relations = num_relations
        self.num_bases = num_bases
        self.basis = nn.Parameter(torch.Tensor(num_relations, in_channels, out_channels))
        self.att = None
        self.root = nn.Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        # reference model.py:258-267: every tensor ~ U(-b, b), b = 1/sqrt(num_relations*in_channels)
        bound = 1.0 / math.sqrt(self.num_relations * self.in_channels)
        for p in (self.basis, self.root, self.bias):
            if p is not None:
                p.data.uniform_(-bound, bound)

    def forward(self, x, edge_index, edge_type, edge_norm=None, relu=False):
        """x [N_total, in] -> [N_total, out]. `relu=True` fuses the F.relu the encoder applies
        after all but the last layer (model.py:437) into the kernel epilogue."""
        if edge_norm is not None:
            raise NotImplementedError('edge_norm is always None on the reference path (model.py:436, 441)')
        if x is None or x.dtype == torch.long:
            raise NotImplementedError('featureless (x=None / long x) RGCNConv is never used by the reference')
        graph = getattr(edge_index, '_mpqe_graph', None)
        if graph is None or (isinstance(graph, ops.GraphPlan) and
                             (graph.Nn != x.shape[0] or getattr(edge_index, '_mpqe_et', None) is not edge_type)):
            graph = ops.GraphPlan(edge_index, edge_type, x.shape[0], self.num_relations)
            edge_index._mpqe_graph = graph           # sorted once, reused by later layers / backward
            edge_index._mpqe_et = edge_type
        return ops.rgcn_layer(x, self.basis, self.root, self.bias, graph, relu)

    def __repr__(self):
        return '{}({}, {}, num_relations={})'.format(self.__class__.__name__, self.in_channels,
                                                     self.out_channels, self.num_relations)


class RGCNEncoderDecoder(nn.Module):
    """reference: RGCNEncoderDecoder, model.py:313-494."""

    def __init__(self, graph, enc, readout='mp', scatter_op='add', dropout=0, weight_decay=1e-3,
                 num_layers=3, shared_layers=True, adaptive=True):
        super(RGCNEncoderDecoder, self).__init__()
        self.enc = enc
        self.graph = graph
        self.emb_dim = graph.feature_dims[next(iter(graph.feature_dims))]
        self.mode_embeddings = nn.Embedding(len(graph.mode_weights), self.emb_dim)
        self.num_layers = num_layers
        self.adaptive = adaptive

        self.mode_ids = {mode: i for i, mode in enumerate(graph.mode_weights)}
        self.rel_ids = {}
        for r1 in graph.relations:
            for r2 in graph.relations[r1]:
                self.rel_ids[(r1, r2[1], r2[0])] = len(self.rel_ids)

        self.layers = nn.ModuleList()
        for i in range(num_layers):
            if len(self.layers) == 0 or not shared_layers:
                rgcn = RGCNConv(in_channels=self.emb_dim, out_channels=self.emb_dim,
                                num_relations=len(graph.rel_edges), num_bases=0)
            self.layers.append(rgcn)

        if scatter_op == 'add':
            scatter_fn = scatter_add
        elif scatter_op == 'max':
            scatter_fn = scatter_max
        elif scatter_op == 'mean':
            scatter_fn = scatter_mean
        else:
            raise ValueError(f'Unknown scatter op {scatter_op}')

        self.readout_str = readout
        if readout == 'sum':
            self.readout = self.sum_readout
        elif readout == 'max':
            self.readout = self.max_readout
        elif readout == 'mlp':
            self.readout = MLPReadout(self.emb_dim, self.emb_dim, scatter_fn)
        elif readout == 'targetmlp':
            self.readout = TargetMLPReadout(self.emb_dim, scatter_fn)
        elif readout == 'concat':
            self.readout = MLPReadout(self.emb_dim * num_layers, self.emb_dim, scatter_fn)
        elif readout == 'mp':
            self.readout = self.target_message_readout
        else:
            raise ValueError(f'Unknown readout function {readout}')

        self.dropout = nn.Dropout(dropout)      # built and never applied, as in the reference (model.py:377)
        self.weight_decay = weight_decay
        # the reference encodes the same query graphs twice per margin_loss (model.py:478-482);
        # the query embedding does not depend on the target, so it is computed once here and
        # scored twice. Set True for the reference's literal op sequence.
        self.encode_twice = False
        # one 4-byte D2H read per call to turn a bad id into IndexError (see ops.raise_on_flags)
        self.validate = True
        self._err = None
        # margin_loss / forward-without-autograd on the fused step (mpqe_amd/dropin.py): one library call per margin_loss for
        # the loss value, ONE fused step per backward pass for all of them. False: the per-op module path below (also taken
        # by foreign encoders, by encode_twice and by configurations the fused step does not cover).
        self.fused = True
        self._dropin_state = None
        from .optim import register_model
        register_model(self)            # (mpqe_amd.optim.Adam / SGD find the model of the parameters they are handed)

    # ------------------------------------------------------------------ readouts (model.py:380-398)
    def sum_readout(self, embs, batch_idx, batch_size=None, num_nodes=None, **kwargs):
        if batch_size is not None and num_nodes is not None and embs.shape[0] == batch_size * num_nodes:
            return ops.readout('sum', embs, batch_size, num_nodes, kwargs.get('num_anchors', 0))
        return scatter_add(embs, batch_idx, dim=0)

    def max_readout(self, embs, batch_idx, batch_size=None, num_nodes=None, **kwargs):
        if batch_size is not None and num_nodes is not None and embs.shape[0] == batch_size * num_nodes:
            return ops.readout('max', embs, batch_size, num_nodes, kwargs.get('num_anchors', 0))
        out, argmax = scatter_max(embs, batch_idx, dim=0)
        return out

    def target_message_readout(self, embs, batch_size, num_nodes, num_anchors, **kwargs):
        return ops.readout('mp', embs, batch_size, num_nodes, num_anchors)

    # ------------------------------------------------------------------ helpers
    def _device(self):
        return next(self.parameters()).device

    def _error_word(self, device):
        if self._err is None or self._err.device != device:
            self._err = ops.new_error_word(device)
        return self._err

    def _node_features(self, formula, anchor_ids, var_ids, device):
        """x [B*N, D]: anchors = normalised entity embeddings, variables = mode embeddings
        (model.py:418-422)."""
        enc = self.enc
        if hasattr(enc, 'table') and getattr(enc, 'node_maps', None) is not None:
            modes = list(formula.anchor_modes)
            uniq = []
            for m in modes:
                if m not in uniq:
                    uniq.append(m)
            ids_t = anchor_ids.to(device).t().contiguous()
            return ops.assemble_x(self.mode_embeddings.weight, enc.node_maps, ids_t, var_ids,
                                  [uniq.index(m) for m in modes], [enc.table(m) for m in uniq],
                                  self._error_word(device))
        # a foreign encoder: keep its call protocol (enc(ids, mode) -> [D, B])
        cols = [self.enc(anchor_ids[:, i], mode).t() for i, mode in enumerate(formula.anchor_modes)]
        var = self.mode_embeddings.weight[var_ids]
        B = anchor_ids.shape[0]
        x = torch.cat([c[:, None, :] for c in cols] + [var[None].expand(B, -1, -1)], dim=1)
        return x.reshape(-1, self.emb_dim)

    def encode(self, formula, queries, anchor_ids=None, var_ids=None, q_graphs=None):
        """Query embeddings [B, D] (everything in model.py:404-449 that precedes the scoring)."""
        if anchor_ids is None or var_ids is None or q_graphs is None:
            anchor_ids, var_ids, q_graphs = RGCNQueryDataset.get_query_graph(formula, queries, self.rel_ids,
                                                                             self.mode_ids)
        device = self._device()
        var_ids = var_ids.to(device)
        q_graphs = q_graphs.to(device)
        batch_size, num_anchors = anchor_ids.shape
        n_nodes = num_anchors + var_ids.shape[0]

        x = self._node_features(formula, anchor_ids, var_ids, device)
        q_graphs.x = x

        if self.adaptive:
            num_passes = RGCNQueryDataset.query_diameters[formula.query_type]
            if num_passes > len(self.layers):
                raise ValueError(f'RGCN is adaptive with {len(self.layers)}'
                                 f' layers, but query requires {num_passes}.')
        else:
            num_passes = self.num_layers

        h1 = x
        h_layers = []
        for i in range(num_passes - 1):
            h1 = self.layers[i](h1, q_graphs.edge_index, q_graphs.edge_type, relu=True)
            if self.readout_str == 'concat':
                h_layers.append(h1)
        h1 = self.layers[-1](h1, q_graphs.edge_index, q_graphs.edge_type)
        if self.readout_str == 'concat':
            h_layers.append(h1)
            h1 = torch.cat(h_layers, dim=1)

        return self.readout(embs=h1, batch_idx=q_graphs.batch, batch_size=batch_size, num_nodes=n_nodes,
                            num_anchors=num_anchors)

    def _target_embeds(self, nodes, mode, device):
        enc = self.enc
        if hasattr(enc, 'table') and getattr(enc, 'node_maps', None) is not None:
            ids = nodes if torch.is_tensor(nodes) else torch.as_tensor(nodes, dtype=torch.long)
            return ops.embed_l2norm(enc.table(mode), enc.node_maps, ids.to(device=device, dtype=torch.long),
                                    self._error_word(device))
        return self.enc(nodes, mode).t()

    def score(self, formula, out, target_nodes, neg_nodes=None, neg_lengths=None):
        """reference: model.py:451-462."""
        device = out.device
        scores = ops.cosine(out, self._target_embeds(target_nodes, formula.target_mode, device))
        if neg_nodes is not None:
            neg_embeds = self._target_embeds(neg_nodes, formula.target_mode, device)
            lengths = torch.as_tensor(neg_lengths, dtype=torch.long)
            q_row = torch.repeat_interleave(torch.arange(lengths.shape[0]), lengths).to(device)
            neg_scores = ops.cosine(out, neg_embeds, q_row=q_row)
            scores = torch.cat((scores, neg_scores), dim=0)
        return scores

    def _check(self):
        if self.validate:
            for err in (self._err, getattr(self.enc, '_err', None)):
                if err is not None:
                    ops.raise_on_flags(err)

    # ------------------------------------------------------------------ the fused step behind the entry points
    def __getstate__(self):
        # (copy.deepcopy / torch.save of the whole module: the fused step's bookkeeping -- device buffers, argument blocks,
        # the C++ pass object -- belongs to THIS object and is rebuilt by the copy at its first call)
        state = self.__dict__.copy()
        state['_dropin_state'] = None
        state['_dropin_checked'] = False
        state['_err'] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        from .optim import register_model
        register_model(self)

    def _apply(self, fn, *args, **kwargs):
        # (.to() / .cuda() / .float(): the parameters move -- the fused step's addresses are taken again at the next call)
        out = super(RGCNEncoderDecoder, self)._apply(fn, *args, **kwargs)
        self.__dict__['_dropin_checked'] = False
        return out

    def dropin(self):
        """The model's DropIn (mpqe_amd/dropin.py), or None when this model takes the module path."""
        if not self.fused or self.encode_twice:
            return None
        d = self._dropin_state
        if d is False:
            return None
        if d is not None and self.__dict__.get('_dropin_checked'):
            return d                # (nothing moved the parameters since the addresses were taken: _apply resets the flag;
                                    # FlatOptimizer, which re-homes them, refreshes the step's pointers itself)
        self.__dict__['_dropin_checked'] = True
        if d is not None and d.stale() and d.refresh():
            return d
        if d is None or d.stale():
            from .dropin import DropIn
            enc = self.enc
            try:
                if not (hasattr(enc, 'table') and getattr(enc, 'node_maps', None) is not None):
                    raise ValueError('foreign encoder')
                if self._device().type != 'cuda':
                    raise ValueError('not on the GPU')          # (the module path raises the package's usual error)
                d = DropIn(self)
            except (ValueError, NotImplementedError):
                d = False
            self.__dict__['_dropin_state'] = d
        return d or None

    def _fused_covers(self, d, formula, n_queries):
        # (concat reads one block per layer: the reference's own Linear fails on fewer passes; huge batches: the in-step
        # touch plan's limit -- both stay on the module path)
        if d.step.learned and self.readout_str == 'concat' and d._passes(formula) != self.num_layers:
            return False
        from .dropin import MAX_IDS
        return 0 < n_queries and 5 * n_queries <= MAX_IDS

    # ------------------------------------------------------------------ reference entry points
    def forward(self, formula, queries, target_nodes, anchor_ids=None, var_ids=None, q_graphs=None,
                neg_nodes=None, neg_lengths=None):
        if not torch.is_grad_enabled():
            d = self.dropin()
            if d is not None and self._fused_covers(d, formula, len(queries)) and (
                    neg_nodes is None or d.step.uses_chain_dims()):
                return d.forward(formula, queries, target_nodes, anchor_ids, var_ids, q_graphs, neg_nodes, neg_lengths)
        out = self.encode(formula, queries, anchor_ids, var_ids, q_graphs)
        scores = self.score(formula, out, target_nodes, neg_nodes, neg_lengths)
        self._check()
        return scores

    def sample_negatives(self, formula, queries, hard_negatives=False):
        """reference: model.py:466-476 (same python `random` stream, so the same draws)."""
        if "inter" not in formula.query_type and hard_negatives:
            raise Exception("Hard negative examples can only be used with "
                            "intersection queries")
        elif hard_negatives:
            return [random.choice(query.hard_neg_samples) for query in queries]
        elif formula.query_type == "1-chain":
            return [random.choice(self.graph.full_lists[formula.target_mode]) for _ in queries]
        return [random.choice(query.neg_samples) for query in queries]

    def margin_loss(self, formula, queries, anchor_ids=None, var_ids=None, q_graphs=None,
                    hard_negatives=False, margin=1):
        d = self.dropin()
        if d is not None and self._fused_covers(d, formula, len(queries)):
            return d.margin_loss(formula, queries, anchor_ids, var_ids, q_graphs, hard_negatives, margin)
        neg_nodes = self.sample_negatives(formula, queries, hard_negatives)
        targets = [query.target_node for query in queries]
        if self.encode_twice:
            affs = self.forward(formula, queries, targets, anchor_ids, var_ids, q_graphs)
            neg_affs = self.forward(formula, queries, neg_nodes, anchor_ids, var_ids, q_graphs)
        else:
            out = self.encode(formula, queries, anchor_ids, var_ids, q_graphs)
            affs = self.score(formula, out, targets)
            neg_affs = self.score(formula, out, neg_nodes)
        loss = ops.hinge(affs, neg_affs, margin)

        if isinstance(self.readout, nn.Module) and self.weight_decay > 0:
            # (reference model.py:486-490: l2_reg = sum of torch.norm(param) -- on the library's kernel, the one the fused step uses)
            loss = loss + self.weight_decay * ops.l2_norms(list(self.readout.parameters()))
        self._check()
        return loss


def _mlp(layers, x):
    """nn.Sequential(Linear, ReLU, Linear) of the readouts through ops.linear."""
    h = ops.linear(x.contiguous(), layers[0].weight, layers[0].bias, relu=True)
    return ops.linear(h, layers[2].weight, layers[2].bias)


class MLPReadout(nn.Module):
    """reference: model.py:497-515. Linear-ReLU-Linear per node, then the scatter reduction kernel. The nn.Linear
    modules hold the parameters (state_dict keys layers.0 / layers.2, the reference's init); the arithmetic runs on the
    library's own MFMA tiles (ops.linear: mpqe_linear_fwd / bwd, the ReLU fused into the first product's epilogue)."""

    def __init__(self, input_dim, output_dim, scatter_fn):
        super(MLPReadout, self).__init__()
        self.layers = nn.Sequential(nn.Linear(in_features=input_dim, out_features=output_dim),
                                    nn.ReLU(),
                                    nn.Linear(in_features=output_dim, out_features=output_dim))
        self.scatter_fn = scatter_fn

    def forward(self, embs, batch_idx, batch_size=None, **kwargs):
        x = _mlp(self.layers, embs)
        x = self.scatter_fn(x, batch_idx, dim=0, dim_size=batch_size)
        if isinstance(x, tuple):
            x = x[0]
        return x


class TargetMLPReadout(nn.Module):
    """reference: model.py:518-553."""

    def __init__(self, dim, scatter_fn):
        super(TargetMLPReadout, self).__init__()
        self.layers = nn.Sequential(nn.Linear(in_features=2 * dim, out_features=dim),
                                    nn.ReLU(),
                                    nn.Linear(in_features=dim, out_features=dim))
        self.scatter_fn = scatter_fn

    def forward(self, embs, batch_idx, batch_size, num_nodes, num_anchors, **kwargs):
        keep = [n for n in range(num_nodes) if n != num_anchors]
        batch_idx = batch_idx.reshape(batch_size, -1)[:, keep].reshape(-1)
        embs = embs.reshape(batch_size, num_nodes, -1)
        non_targets = embs[:, keep]
        targets = embs[:, num_anchors:num_anchors + 1].expand_as(non_targets)
        x = torch.cat((targets, non_targets), dim=-1)
        x = x.reshape(batch_size * (num_nodes - 1), -1).contiguous()
        x = _mlp(self.layers, x)
        x = self.scatter_fn(x, batch_idx, dim=0, dim_size=batch_size)
        if isinstance(x, tuple):
            x = x[0]
        return x

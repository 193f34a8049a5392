"""Entity encoders with the reference's interface (mpqe/encoders.py).

DirectEncoder (reference encoders.py:11-45) is the depth-0 encoder every
BASELINE config uses: embedding lookup + L2 normalisation, here one fused HIP
kernel (gather through the global-id -> per-mode-row LUT, normalise, write).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class DirectEncoder(nn.Module):
    """DirectEncoder(features, feature_modules[, node_maps])

    features         -- the reference's closure (nodes, mode) -> [B, D] embeddings. Only used when
                        `node_maps` is not given (then the lookup stays the caller's and just the
                        normalisation runs in the HIP kernel).
    feature_modules  -- {mode: nn.Embedding}; registered as "feat-<mode>" like the reference
                        (encoders.py:25-26) so state_dict keys match.
    node_maps        -- int64 LUT global entity id -> row of its mode's table, -1 for foreign ids
                        (what load_graph builds, data_utils.py:23-29). With it the whole
                        features(...) + normalise sequence is one kernel.
    """

    def __init__(self, features, feature_modules, node_maps=None):
        super(DirectEncoder, self).__init__()
        for name, module in feature_modules.items():
            self.add_module('feat-' + name, module)
        self.features = features
        self.feature_modules = feature_modules
        if node_maps is not None:
            self.register_buffer('node_maps', torch.as_tensor(node_maps, dtype=torch.long), persistent=False)
        else:
            self.node_maps = None
        self._err = None

    def table(self, mode):
        return self.feature_modules[mode].weight

    def error_word(self, device):
        if self._err is None or self._err.device != device:
            self._err = ops.new_error_word(device)
        return self._err

    def _ids(self, nodes, device):
        if not torch.is_tensor(nodes):
            nodes = torch.as_tensor(nodes, dtype=torch.long)
        return nodes.to(device=device, dtype=torch.long)

    def forward(self, nodes, mode, offset=None, **kwargs):
        """[D, B] unit-norm columns, like the reference (encoders.py:40-43)."""
        if offset is not None:
            raise NotImplementedError('EmbeddingBag offsets are never used by the R-GCN path')
        table = self.table(mode)
        if self.node_maps is not None:
            out = ops.embed_l2norm(table, self.node_maps, self._ids(nodes, table.device),
                                   self.error_word(table.device))
        else:
            emb = self.features(nodes, mode)
            rows = torch.arange(emb.shape[0], device=emb.device)
            out = ops.embed_l2norm(emb, None, rows, self.error_word(emb.device))
        return out.t()


class LayerNorm(nn.Module):
    """reference: encoders.py:132-146 (unbiased std, eps added to the std, not the variance)."""

    def __init__(self, feature_dim, eps=1e-6):
        super(LayerNorm, self).__init__()
        self.gamma = nn.Parameter(torch.ones((feature_dim,)))
        self.beta = nn.Parameter(torch.zeros((feature_dim,)))
        self.eps = eps

    def forward(self, x):
        mean = x.mean(-1, keepdim=True)
        std = x.std(-1, keepdim=True)
        return self.gamma * (x - mean) / (std + self.eps) + self.beta


class Encoder(nn.Module):
    """GraphSAGE-style entity encoder inherited from GQE (reference encoders.py:47-129, `--depth >= 1`):
    per relation the sampled-neighbour mean (aggregator), concatenated with the node's own feature,
    compressed by a per-mode matrix, optional LayerNorm, ReLU. Returns [out_dim, B] like the
    reference. The neighbour mean runs in the scatter kernel; the compress GEMM is a plain library GEMM.

    Note: the reference's RGCNEncoderDecoder passes a tensor column as `nodes`,
    which its own Encoder cannot use as dict keys; like the reference this class expects a python list
    of entity ids (the GQE call protocol)."""

    def __init__(self, features, feature_dims, out_dims, relations, adj_lists, aggregator, base_model=None,
                 cuda=False, layer_norm=False, feature_modules={}):
        super(Encoder, self).__init__()
        self.features = features
        self.feat_dims = feature_dims
        self.adj_lists = adj_lists
        self.relations = relations
        self.aggregator = aggregator
        for name, module in feature_modules.items():
            self.add_module('feat-' + name, module)
        if base_model is not None:
            self.base_model = base_model
        self.out_dims = out_dims
        self.cuda = cuda
        self.aggregator.cuda = cuda
        self.layer_norm = layer_norm
        self.compress_dims = {}
        for source_mode in relations:
            self.compress_dims[source_mode] = self.feat_dims[source_mode]
            for (to_mode, _) in relations[source_mode]:
                self.compress_dims[source_mode] += self.feat_dims[to_mode]
        self.compress_params = {}
        self.lns = {}
        for mode in self.feat_dims:
            if self.layer_norm:
                self.lns[mode] = LayerNorm(out_dims[mode])
                self.add_module(mode + '_ln', self.lns[mode])
            self.compress_params[mode] = nn.Parameter(torch.FloatTensor(out_dims[mode], self.compress_dims[mode]))
            nn.init.xavier_uniform_(self.compress_params[mode])
            self.register_parameter(mode + '_compress', self.compress_params[mode])

    def forward(self, nodes, mode, keep_prob=0.5, max_keep=10):
        self_feat = self.features(nodes, mode).t()
        neigh_feats = []
        for to_r in self.relations[mode]:
            rel = (mode, to_r[1], to_r[0])
            to_neighs = [[-1] if node == -1 else self.adj_lists[rel][node] for node in nodes]
            to_neighs = [[-1] if len(l) == 0 else l for l in to_neighs]     # null neighbour, as the reference
            neigh_feats.append(self.aggregator.forward(to_neighs, rel, keep_prob, max_keep).t())
        neigh_feats.append(self_feat)
        combined = torch.cat(neigh_feats, dim=0)
        combined = self.compress_params[mode].mm(combined)
        if self.layer_norm:
            combined = self.lns[mode](combined.t()).t()
        return F.relu(combined)

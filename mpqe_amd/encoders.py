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
    """The Encoder's normalisation (reference encoders.py:132-146): per row, UNBIASED standard deviation, eps added to
    the standard deviation. Parameters `gamma` / `beta` as in the reference (state_dict keys `<mode>_ln.gamma|beta`).
    One HIP kernel (mpqe_layernorm_relu_fwd / bwd); `relu=True` folds the Encoder's activation into it."""

    def __init__(self, feature_dim, eps=1e-6):
        super(LayerNorm, self).__init__()
        self.gamma = nn.Parameter(torch.ones(feature_dim))
        self.beta = nn.Parameter(torch.zeros(feature_dim))
        self.eps = eps

    def forward(self, x, relu=False):
        lead = x.shape[:-1]
        y = ops.layernorm_relu(x.reshape(-1, x.shape[-1]), self.gamma, self.beta, self.eps, relu)
        return y.reshape(*lead, x.shape[-1])


class Encoder(nn.Module):
    """GraphSAGE-style entity encoder the reference inherits from GQE (encoders.py:47-129, `--depth >= 1`; SURVEY.md 8 a9,
    secondary row: unreachable from the R-GCN model in the reference itself). For a node of mode m:

        out = ReLU([LayerNorm](compress_m . [mean_r1 | ... | mean_rk | self]))        -> [out_dim, B]

    with mean_r = mean feature of the neighbours SAMPLED for relation r (python `random`, the reference's draws in the
    reference's order: aggregators.py). Constructor arguments, attribute names and state_dict keys (`<mode>_compress`,
    `<mode>_ln.gamma|beta`, `feat-<name>.*`) are the reference's; the data path is this package's:
      * ONE flat list of sampled neighbours per relation goes through the features closure and the scatter kernel
        (mean mode) -- no dense [B, U] mask matrix, no per-node python set arithmetic on the device side;
      * the concatenation is never materialised: compress_m is applied block by block (one mpqe_linear_fwd per relation
        block and one for the self block, accumulating), which is the same sum;
      * LayerNorm and the ReLU are one kernel (mpqe_layernorm_relu_fwd).
    `nodes` is a python list of entity ids (the GQE call protocol)."""

    def __init__(self, features, feature_dims, out_dims, relations, adj_lists, aggregator, base_model=None,
                 cuda=False, layer_norm=False, feature_modules={}):
        super(Encoder, self).__init__()
        self.features, self.feat_dims, self.out_dims = features, feature_dims, out_dims
        self.relations, self.adj_lists, self.aggregator = relations, adj_lists, aggregator
        self.cuda, self.layer_norm = cuda, layer_norm
        self.aggregator.cuda = cuda
        if base_model is not None:
            self.base_model = base_model
        for name, module in feature_modules.items():
            self.add_module('feat-' + name, module)
        # per mode: the column blocks of its compress matrix -- one per outgoing relation (width = feature dim of the
        # relation's target mode), then the node's own features
        self.blocks, self.compress_dims, self.compress_params, self.lns = {}, {}, {}, {}
        for mode in relations:
            widths = [feature_dims[to_mode] for (to_mode, _) in relations[mode]] + [feature_dims[mode]]
            self.blocks[mode] = [(sum(widths[:k]), w) for k, w in enumerate(widths)]
            self.compress_dims[mode] = sum(widths)
        for mode in feature_dims:
            if layer_norm:
                self.lns[mode] = LayerNorm(out_dims[mode])
                self.add_module(mode + '_ln', self.lns[mode])
            w = nn.Parameter(torch.empty(out_dims[mode], self.compress_dims[mode]))
            nn.init.xavier_uniform_(w)
            self.compress_params[mode] = w
            self.register_parameter(mode + '_compress', w)

    def forward(self, nodes, mode, keep_prob=0.5, max_keep=10):
        W = self.compress_params[mode]
        blocks = self.blocks[mode]
        feats = []
        for (to_mode, name), (off, width) in zip(self.relations[mode], blocks[:-1]):
            rel = (mode, name, to_mode)
            adj = self.adj_lists[rel]
            # the null neighbour -1 stands in for a padding node and for a node without edges of this relation
            neigh = [[-1] if (n == -1 or len(adj[n]) == 0) else adj[n] for n in nodes]
            feats.append(self.aggregator.forward(neigh, rel, keep_prob, max_keep).contiguous())    # [B, width]
        feats.append(self.features(nodes, mode).contiguous())
        # compress_m applied block by block on the library's own MFMA tiles (mpqe_linear_fwd, accumulating): the column
        # blocks of W are read in place, the concatenation of the reference (encoders.py:120) is never built
        out = ops.blocks_linear(feats, W, blocks)
        if self.layer_norm:
            out = self.lns[mode](out, relu=True)
        else:
            out = F.relu(out)
        return out.t()

"""Entity encoders with the reference's interface (mpqe/encoders.py).

DirectEncoder (reference encoders.py:11-45) is the depth-0 encoder every
BASELINE config uses: embedding lookup + L2 normalisation, here one fused HIP
kernel (gather through the global-id -> per-mode-row LUT, normalise, write).
"""
import torch
import torch.nn as nn

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

"""mpqe_amd -- MI355X-native (gfx950) implementation of MPQE's R-GCN query-graph encoder
hot path behind the reference's own module interface. See DESIGN.md."""
from .graph import Formula, Graph, Query  # noqa: F401

__all__ = ['Formula', 'Graph', 'Query', 'RGCNConv', 'RGCNEncoderDecoder', 'DirectEncoder',
           'RGCNQueryDataset', 'MLPReadout', 'TargetMLPReadout']


def __getattr__(name):
    # torch-dependent modules are imported on first use so that `import mpqe_amd.graph`
    # (pure python) stays cheap
    if name in ('RGCNConv', 'RGCNEncoderDecoder', 'MLPReadout', 'TargetMLPReadout'):
        from . import model
        return getattr(model, name)
    if name == 'DirectEncoder':
        from .encoders import DirectEncoder
        return DirectEncoder
    if name == 'RGCNQueryDataset':
        from .data_utils import RGCNQueryDataset
        return RGCNQueryDataset
    raise AttributeError(name)

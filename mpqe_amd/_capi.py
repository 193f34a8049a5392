"""ctypes prototypes of every entry point declared in include/mpqe_amd.h.

`bind(cdll)` attaches argtypes/restype and fails if a symbol is missing, so a
stale or partial shared object is caught at import time, not at first use.
Pointers are passed as integers (tensor.data_ptr()); see mpqe_amd/ops.py.
"""
import ctypes
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

P = c_void_p      # device pointer (or host pointer where the header says *_host)
I = c_int
L = c_int64
Z = c_size_t
F = c_float


class TemplateInfo(ctypes.Structure):
    _fields_ = [('num_anchors', ctypes.c_int32), ('num_vars', ctypes.c_int32),
                ('num_nodes', ctypes.c_int32), ('num_edges', ctypes.c_int32),
                ('diameter', ctypes.c_int32),
                ('src', ctypes.c_int32 * 3), ('dst', ctypes.c_int32 * 3),
                ('rel_label', ctypes.c_int32 * 3), ('var_node', ctypes.c_int32 * 4)]


# name: (restype, [argtypes])
PROTOTYPES = {
    'mpqe_status_string': (c_char_p, [I]),
    'mpqe_abi_version': (I, []),
    'mpqe_template_info': (I, [I, ctypes.POINTER(TemplateInfo)]),
    'mpqe_collate_template': (I, [I, L, P, P, P, P, P]),
    'mpqe_embed_l2norm_fwd': (I, [P, L, L, P, L, P, L, P, L, P, P, P]),
    'mpqe_embed_l2norm_bwd': (I, [P, L, P, L, L, P, L, P, L, P, P, P]),
    'mpqe_var_rows_fwd': (I, [P, L, L, P, L, L, L, L, P, P, P]),
    'mpqe_var_rows_bwd': (I, [P, L, L, P, L, L, L, L, P, P, P]),
    'mpqe_rgcn_template_fwd': (I, [I, L, P, P, P, L, P, P, L, L, I, P, P]),
    'mpqe_rgcn_template_bwd_workspace_bytes': (Z, [I, L, L, L]),
    'mpqe_rgcn_template_bwd': (I, [I, L, P, P, P, P, P, L, P, L, L, I, P, P, P, P, P, Z, P]),
    'mpqe_rgcn_plan_bytes': (Z, [L, L, L]),
    'mpqe_rgcn_plan_workspace_bytes': (Z, [L, L, L]),
    'mpqe_rgcn_plan_build': (I, [P, P, L, L, L, P, Z, P, Z, P, P]),
    'mpqe_rgcn_general_workspace_bytes': (Z, [L, L, L, L, L, I]),
    'mpqe_rgcn_general_fwd': (I, [P, L, L, L, P, P, P, P, L, L, I, P, P, Z, P]),
    'mpqe_rgcn_general_bwd': (I, [P, L, L, L, P, P, P, P, P, L, L, I, P, P, P, P, P, Z, P]),
    'mpqe_readout_fwd': (I, [I, P, L, L, L, L, P, P, P]),
    'mpqe_readout_bwd': (I, [I, P, P, L, L, L, L, P, P]),
    'mpqe_scatter_workspace_bytes': (Z, [L, L]),
    'mpqe_scatter_fwd': (I, [I, P, P, L, L, L, P, P, P, Z, P, P]),
    'mpqe_scatter_bwd': (I, [I, P, P, P, L, L, L, P, P, Z, P]),
    'mpqe_cosine_fwd': (I, [P, P, P, L, L, F, P, P]),
    'mpqe_cosine_bwd': (I, [P, P, P, P, L, L, F, P, P, P]),
    'mpqe_hinge_fwd': (I, [P, P, L, F, P, P]),
    'mpqe_hinge_bwd': (I, [P, P, L, F, P, P, P, P]),
}

QUERY_TYPE_IDS = {'1-chain': 0, '2-chain': 1, '3-chain': 2, '2-inter': 3, '3-inter': 4,
                  '3-inter_chain': 5, '3-chain_inter': 6}
READOUT_IDS = {'sum': 0, 'max': 1, 'mp': 2}
SCATTER_IDS = {'add': 0, 'max': 1, 'mean': 2}

FLAG_BAD_NODE_ID, FLAG_BAD_EDGE, FLAG_BAD_RELATION, FLAG_BAD_INDEX = 1, 2, 4, 8


def bind(cdll):
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(cdll, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise ImportError('shared library lacks C-ABI symbols: %s' % ', '.join(missing))
    return cdll


class MpqeError(RuntimeError):
    pass


def check(cdll, status, what):
    if status != 0:
        msg = cdll.mpqe_status_string(status)
        raise MpqeError('%s failed: %s (%d)' % (what, msg.decode() if msg else '?', status))

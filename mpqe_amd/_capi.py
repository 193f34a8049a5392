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
DBL = ctypes.c_double


class TemplateInfo(ctypes.Structure):
    _fields_ = [('num_anchors', ctypes.c_int32), ('num_vars', ctypes.c_int32),
                ('num_nodes', ctypes.c_int32), ('num_edges', ctypes.c_int32),
                ('diameter', ctypes.c_int32),
                ('src', ctypes.c_int32 * 3), ('dst', ctypes.c_int32 * 3),
                ('rel_label', ctypes.c_int32 * 3), ('var_node', ctypes.c_int32 * 4)]


STEP_MAX_BATCHES, STEP_MAX_LAYERS, STEP_MAX_MODES = 16, 8, 16


class StepBatch(ctypes.Structure):
    _fields_ = [('query_type', ctypes.c_int32), ('num_passes', ctypes.c_int32),
                ('batch_size', ctypes.c_int32), ('target_mode', ctypes.c_int32),
                ('edge_type', ctypes.c_int64 * 3), ('var_ids', ctypes.c_int64 * 3),
                ('anchor_mode', ctypes.c_int32 * 3), ('weight', ctypes.c_float)]


class StepParams(ctypes.Structure):
    _fields_ = [('dim', ctypes.c_int32), ('num_layers', ctypes.c_int32),
                ('num_relations', ctypes.c_int32), ('num_modes', ctypes.c_int32),
                ('readout', ctypes.c_int32), ('flags', ctypes.c_int32),
                ('tables', c_void_p * STEP_MAX_MODES), ('table_rows', ctypes.c_int64 * STEP_MAX_MODES),
                ('node_map', c_void_p), ('node_map_len', ctypes.c_int64), ('mode_emb', c_void_p),
                ('basis', c_void_p * STEP_MAX_LAYERS), ('root', c_void_p * STEP_MAX_LAYERS),
                ('bias', c_void_p * STEP_MAX_LAYERS),
                ('readout_w0', c_void_p), ('readout_b0', c_void_p), ('readout_w2', c_void_p), ('readout_b2', c_void_p),
                ('readout_scatter', ctypes.c_int32), ('readout_weight_decay', ctypes.c_float)]


class StepGrads(ctypes.Structure):
    _fields_ = [('tables', c_void_p * STEP_MAX_MODES), ('mode_emb', c_void_p),
                ('basis', c_void_p * STEP_MAX_LAYERS), ('root', c_void_p * STEP_MAX_LAYERS),
                ('bias', c_void_p * STEP_MAX_LAYERS),
                ('readout_w0', c_void_p), ('readout_b0', c_void_p), ('readout_w2', c_void_p), ('readout_b2', c_void_p)]


STEP_MAX_LANES = 4


class StepExtra(ctypes.Structure):
    # include/mpqe_amd.h: mpqe_step_extra_t
    _fields_ = [('batch_weight', c_void_p * STEP_MAX_BATCHES), ('query_out', c_void_p), ('notify', c_void_p),
                ('notify_value', ctypes.c_uint32), ('xcd_shift', ctypes.c_int32), ('join_event', c_void_p), ('join_stream', c_void_p),
                ('readout_norms', c_void_p)]


class StepLanes(ctypes.Structure):
    _fields_ = [('num_lanes', ctypes.c_int32), ('batch_begin', ctypes.c_int32 * (STEP_MAX_LANES + 1)),
                ('aux_stream', c_void_p * STEP_MAX_LANES), ('fork_event', c_void_p),
                ('join_event', c_void_p * STEP_MAX_LANES)]


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
    'mpqe_rgcn_general_mask_bytes': (Z, [L, L]),
    'mpqe_rgcn_general_fwd': (I, [P, L, L, L, P, P, P, P, L, L, I, P, P, P, Z, P]),
    'mpqe_rgcn_general_aggregate': (I, [P, L, L, L, P, P, L, I, P, P]),
    'mpqe_rgcn_general_bwd': (I, [P, L, L, L, P, P, P, P, P, P, L, L, I, I, P, P, P, P, P, Z, P]),
    'mpqe_linear_fwd': (I, [P, L, P, L, P, L, L, I, I, P, P]),
    'mpqe_linear_bwd_workspace_bytes': (Z, [L, L, L]),
    'mpqe_linear_bwd': (I, [P, L, P, L, P, P, L, L, I, I, P, P, L, P, P, Z, P]),
    'mpqe_readout_fwd': (I, [I, P, L, L, L, L, P, P, P]),
    'mpqe_readout_bwd': (I, [I, P, P, L, L, L, L, P, P]),
    'mpqe_scatter_workspace_bytes': (Z, [L, L]),
    'mpqe_scatter_fwd': (I, [I, P, P, L, L, L, P, P, P, Z, P, P]),
    'mpqe_scatter_bwd': (I, [I, P, P, P, L, L, L, P, P, Z, P]),
    'mpqe_layernorm_relu_fwd': (I, [P, L, L, P, P, F, I, P, P, P]),
    'mpqe_layernorm_relu_bwd_workspace_bytes': (Z, [L, L]),
    'mpqe_layernorm_relu_bwd': (I, [P, P, P, L, L, P, P, F, I, P, P, P, P, Z, P]),
    'mpqe_cosine_fwd': (I, [P, P, P, L, L, F, P, P]),
    'mpqe_cosine_bwd': (I, [P, P, P, P, L, L, F, P, P, P]),
    'mpqe_hinge_fwd': (I, [P, P, L, F, P, P]),
    'mpqe_hinge_bwd': (I, [P, P, L, F, P, P, P, P]),
    'mpqe_debug_chain_stamps': (None, [P, Z]),
    'mpqe_debug_tail_stamps': (None, [P, Z]),
    'mpqe_debug_option': (None, [c_char_p, I, I]),
    'mpqe_debug_has_experiments': (I, []),
    'mpqe_copy_to_device': (I, [P, P, Z, P]),
    'mpqe_sample_negatives': (I, [P, L, P, L, P, L, ctypes.c_uint64, P, P, P]),
    'mpqe_adam_step': (I, [P, P, P, P, L, DBL, DBL, DBL, DBL, DBL, L, P]),
    'mpqe_sgd_step': (I, [P, P, L, DBL, DBL, P]),
    'mpqe_adam_rows_step': (I, [P, L, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                ctypes.POINTER(c_void_p), I, L, DBL, DBL, DBL, DBL, L, P]),
    'mpqe_step_workspace_bytes': (Z, [ctypes.POINTER(StepParams), ctypes.POINTER(StepBatch), I, ctypes.POINTER(StepLanes)]),
    'mpqe_step_desc_bytes': (Z, [ctypes.POINTER(StepParams), ctypes.POINTER(StepBatch), I, ctypes.POINTER(StepLanes)]),
    'mpqe_step_states_layout': (I, [ctypes.POINTER(StepParams), ctypes.POINTER(StepBatch), I, ctypes.POINTER(StepLanes),
                                    ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                    ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                    ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'mpqe_step_forward_backward': (I, [ctypes.POINTER(StepParams), ctypes.POINTER(StepBatch), I, P, P, P, F,
                                       ctypes.POINTER(StepGrads), I, P, P, P, P, Z, I, P, Z, P, ctypes.POINTER(StepLanes),
                                       P, I, P, P]),
    'mpqe_step_forward_backward_ex': (I, [ctypes.POINTER(StepParams), ctypes.POINTER(StepBatch), I, P, P, P, F,
                                          ctypes.POINTER(StepGrads), I, P, P, P, P, Z, I, P, Z, P, ctypes.POINTER(StepLanes),
                                          P, I, P, P, ctypes.POINTER(StepExtra)]),
    'mpqe_step_readout_norms': (I, [ctypes.POINTER(StepParams), P, P]),
    'mpqe_host_random_choice': (I, [P, L, P, L, P, P, L, P, P]),
    'mpqe_step_touch_bytes': (Z, [ctypes.POINTER(StepParams), ctypes.POINTER(StepBatch), I]),
    'mpqe_step_touch_workspace_bytes': (Z, [ctypes.POINTER(StepParams), ctypes.POINTER(StepBatch), I]),
    'mpqe_step_touch_entries': (L, [ctypes.POINTER(StepBatch), I]),
    'mpqe_rows_plan_bytes': (Z, [L]),
    'mpqe_rows_plan_workspace_bytes': (Z, [L, I]),
    'mpqe_rows_plan_build': (I, [P, L, I, I, P, Z, P, Z, P]),
    'mpqe_table_rows_sum': (I, [P, L, P, L, ctypes.POINTER(c_void_p), I, I, P]),
    'mpqe_step_touch_build': (I, [ctypes.POINTER(StepParams), ctypes.POINTER(StepBatch), I, P, P, P, P, Z, P, Z, P]),
    'mpqe_p2p_handle_bytes': (Z, []),
    'mpqe_p2p_buffer_bytes': (Z, [L, I, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'mpqe_p2p_alloc': (I, [Z, ctypes.POINTER(c_void_p), P]),
    'mpqe_p2p_free': (I, [P]),
    'mpqe_p2p_open': (I, [P, ctypes.POINTER(c_void_p)]),
    'mpqe_p2p_close': (I, [P]),
    'mpqe_p2p_allreduce': (I, [ctypes.POINTER(c_void_p), I, I, L, L, ctypes.c_uint32, I, P, P]),
    'mpqe_l2_norms': (I, [ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int64), I, P, P, ctypes.POINTER(c_void_p), P]),
    'mpqe_spans_copy': (I, [P, P, P, I, L, P]),
    'mpqe_rows_prepare': (I, [P, L, L, I, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), I, P, P, P]),
    'mpqe_rows_gather': (I, [P, P, L, L, P, P]),
    'mpqe_step_table_rows': (I, [ctypes.POINTER(StepParams), ctypes.POINTER(StepBatch), I, ctypes.POINTER(StepGrads), P, P, Z, P, P]),
}

QUERY_TYPE_IDS = {'1-chain': 0, '2-chain': 1, '3-chain': 2, '2-inter': 3, '3-inter': 4,
                  '3-inter_chain': 5, '3-chain_inter': 6}
QUERY_NAMES = {v: k for k, v in QUERY_TYPE_IDS.items()}
READOUT_IDS = {'sum': 0, 'max': 1, 'mp': 2}
READOUT_CALLER = 3          # fused step only: the readout is the caller's (STEP_PHASE_*)
LEARNED_READOUT_IDS = {'mlp': 4, 'targetmlp': 5, 'concat': 6}      # fused step only (StepParams.readout_*)
SCATTER_IDS = {'add': 0, 'max': 1, 'mean': 2}

FLAG_BAD_NODE_ID, FLAG_BAD_EDGE, FLAG_BAD_RELATION, FLAG_BAD_INDEX = 1, 2, 4, 8
FLAG_INTERNAL, FLAG_TOUCH_RETRY = 16, 32


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


STEP_NO_PRUNE, STEP_NO_CHAIN, STEP_ZERO_GRADS, STEP_NO_KSPLIT, STEP_EIGHT_WAVES, STEP_NO_UNIFORM = 1, 2, 4, 8, 16, 32
STEP_SPARSE_TABLES = 64
STEP_MERGE_TAIL = 128
STEP_SPLIT_TAIL = 256
STEP_BUILD_TOUCH = 512
STEP_ADD_STATE_GRADS = 1024
STEP_TOUCH_LIBRARY_SORT = 2048
# `backward` values of the step in three calls around the caller's readout (learned readouts; include/mpqe_amd.h)
STEP_PHASE_STATES, STEP_PHASE_SCORES, STEP_PHASE_FROM_STATES, STEP_PHASE_SCORES_ONLY = 2, 3, 4, 5
TSORT_MAX_ENTRIES = 256 * 2048        # csrc/step_touch.h: the in-step touch plan covers this many looked-up ids


def make_step_params(dim, num_relations, readout, table_ptrs, table_rows, node_map_ptr, node_map_len,
                     mode_emb_ptr, basis_ptrs, root_ptrs, bias_ptrs, flags=0):
    """StepParams from raw addresses (ints); lists are per mode / per layer."""
    p = StepParams()
    p.flags = flags
    p.dim, p.num_layers, p.num_relations, p.num_modes = dim, len(basis_ptrs), num_relations, len(table_ptrs)
    p.readout = READOUT_IDS[readout] if isinstance(readout, str) else readout
    for m, (ptr, rows) in enumerate(zip(table_ptrs, table_rows)):
        p.tables[m], p.table_rows[m] = ptr, rows
    p.node_map, p.node_map_len, p.mode_emb = node_map_ptr, node_map_len, mode_emb_ptr
    for l, (b, r, bi) in enumerate(zip(basis_ptrs, root_ptrs, bias_ptrs)):
        p.basis[l], p.root[l], p.bias[l] = b, r, bi
    return p


def make_step_grads(table_ptrs, mode_emb_ptr, basis_ptrs, root_ptrs, bias_ptrs):
    g = StepGrads()
    for m, ptr in enumerate(table_ptrs):
        g.tables[m] = ptr
    g.mode_emb = mode_emb_ptr
    for l, (b, r, bi) in enumerate(zip(basis_ptrs, root_ptrs, bias_ptrs)):
        g.basis[l], g.root[l], g.bias[l] = b, r, bi
    return g


def make_step_batch(query_type, num_passes, batch_size, edge_type, var_ids, anchor_modes, target_mode, weight):
    b = StepBatch()
    b.query_type = QUERY_TYPE_IDS[query_type] if isinstance(query_type, str) else query_type
    b.num_passes, b.batch_size, b.target_mode, b.weight = num_passes, batch_size, target_mode, weight
    for i, e in enumerate(edge_type):
        b.edge_type[i] = int(e)
    for i, v in enumerate(var_ids):
        b.var_ids[i] = int(v)
    for i, m in enumerate(anchor_modes):
        b.anchor_mode[i] = int(m)
    return b

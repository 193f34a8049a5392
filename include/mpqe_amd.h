/* mpqe_amd.h -- C ABI of the MI355X-native MPQE R-GCN query-graph encoder.
 *
 * The reference (dfdazac/mpqe) is pure Python and has no FFI layer: its
 * boundary for this path is the nn.Module surface of mpqe/model.py plus the
 * torch_scatter / PyTorch Geometric calls it makes. Each entry point below
 * names the reference lines whose arithmetic it replaces. The Python mirror
 * (mpqe_amd/model.py, encoders.py, data_utils.py) binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - floats are fp32, contiguous row-major; indices are int64 like the
 *     reference's LongTensors; sizes are int64_t;
 *   - the caller owns every buffer, including workspaces (query the size
 *     first); the library never allocates device memory and is re-entrant per
 *     stream. Host-side state, all behind one mutex: a cache of launch plans
 *     (pure functions of the descriptors, keyed by the caller's `desc`
 *     buffer). The library never reads the environment. The three
 *     mpqe_debug_* entry points at the end of this file are DIAGNOSTICS:
 *     process-global, off by default, to be set from one thread while no
 *     call is in flight; nothing in the data path depends on them;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); the fused
 *     step (mpqe_step_forward_backward) hands data from workgroup to workgroup
 *     INSIDE its launches and expects the GPU to itself: one process per GPU
 *     (several processes oversubscribing one GPU can exhaust its bounded spins
 *     -> MPQE_FLAG_INTERNAL, never a hang);
 *   - every function returns MPQE_OK or a negative MPQE_ERR_* code and never
 *     synchronises. Data-dependent faults (an index outside its table) cannot
 *     be seen from the host without a sync: kernels then skip the access and
 *     OR a MPQE_FLAG_* bit into the caller's `err` word (int32 in HBM, may be
 *     NULL); the host mirror raises IndexError from it like the reference's
 *     index_select would.
 */
#ifndef MPQE_AMD_H
#define MPQE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPQE_OK 0
#define MPQE_ERR_INVALID_ARG (-1)    /* null pointer, negative size, bad enum          */
#define MPQE_ERR_UNSUPPORTED (-2)    /* shape outside what the kernels cover           */
#define MPQE_ERR_WORKSPACE (-3)      /* workspace smaller than the *_workspace_bytes() */
#define MPQE_ERR_LAUNCH (-4)         /* hipGetLastError() != hipSuccess after launch   */

#define MPQE_FLAG_BAD_NODE_ID 1      /* entity id outside node_map / maps to -1        */
#define MPQE_FLAG_BAD_EDGE 2         /* edge endpoint outside [0, num_nodes)           */
#define MPQE_FLAG_BAD_RELATION 4     /* edge type outside [0, num_relations)           */
#define MPQE_FLAG_BAD_INDEX 8        /* scatter index outside [0, dim_size)            */
#define MPQE_FLAG_TOUCH_RETRY 32     /* the step could not build its own touch plan (MPQE_STEP_BUILD_TOUCH: the sort's
                                        workgroups were not all resident at once, e.g. on a shared GPU). Everything but
                                        the entity-table gradients is complete; those were NOT accumulated (dense: the
                                        step's zero fill stands; SPARSE_TABLES: the rows are untouched). Recover with
                                        mpqe_step_touch_build (MPQE_STEP_TOUCH_LIBRARY_SORT) + mpqe_step_table_rows.  */
#define MPQE_FLAG_INTERNAL 16        /* a hand-off between workgroups inside a launch did
                                        not arrive within its spin bound (library fault);
                                        bits 8.. say which (diagnostics: 0x100 a pre-pass
                                        vector, 0x200 the transposed copies, 0x400 a
                                        completion counter, 0x800 a vector op's inputs,
                                        0x1000 the fused tail's arrivals, 0x2000 the
                                        touch plan's sort)                                 */

/* query templates, reference data_utils.py:325-362 */
enum {
    MPQE_Q_1CHAIN = 0, MPQE_Q_2CHAIN = 1, MPQE_Q_3CHAIN = 2, MPQE_Q_2INTER = 3,
    MPQE_Q_3INTER = 4, MPQE_Q_3INTER_CHAIN = 5, MPQE_Q_3CHAIN_INTER = 6, MPQE_Q_COUNT = 7
};
enum { MPQE_READOUT_SUM = 0, MPQE_READOUT_MAX = 1, MPQE_READOUT_TM = 2,
       MPQE_READOUT_CALLER = 3 /* fused step only: the readout is the caller's (MPQE_STEP_PHASE_*) */,
       /* fused step only: the reference's learned readouts, Linear(in, dim) - ReLU - Linear(dim, dim) per row, then a
        * reduction over each graph's rows (mpqe_step_params_t.readout_*): MLP (model.py:497-515; in = dim, a row per node),
        * TARGETMLP (model.py:518-553; in = 2 dim, a row [target | node] per non-target node), CONCAT (model.py:441-446;
        * in = num_layers dim, a node's states after every layer side by side -- every batch runs num_layers passes).
        * They ride on the chain form when it applies (dim 64 / 128 / 256, at most 3 passes per batch, two free layer
        * slots): their Linear layers are two more levels of every graph block's programme; otherwise the level form (one
        * launch per level + dense-layer launches) */
       MPQE_READOUT_MLP = 4, MPQE_READOUT_TARGETMLP = 5, MPQE_READOUT_CONCAT = 6 };
enum { MPQE_SCATTER_ADD = 0, MPQE_SCATTER_MAX = 1, MPQE_SCATTER_MEAN = 2 };
#define MPQE_MAX_TEMPLATE_EDGES 3
#define MPQE_MAX_TEMPLATE_NODES 4

const char *mpqe_status_string(int status);
/* 5: MPQE_FLAG_TOUCH_RETRY, MPQE_STEP_TOUCH_LIBRARY_SORT, mpqe_step_table_rows (a failed in-step touch plan is recovered, not
 * fatal). 4: mpqe_step_params_t / mpqe_step_grads_t carry the learned readouts' Linear layers (readout_*); mpqe_step_states_layout,
 * MPQE_READOUT_CALLER and the MPQE_STEP_PHASE_* values of `backward`. 3: mpqe_linear_*, mpqe_debug_option, touch = OUT. */
int mpqe_abi_version(void);

/* Static shape of one query template (host side, no GPU needed).
 * reference: RGCNQueryDataset.query_edge_indices / query_diameters /
 * query_edge_label_idx / variable_node_idx, data_utils.py:325-362. */
typedef struct {
    int32_t num_anchors, num_vars, num_nodes, num_edges, diameter;
    int32_t src[MPQE_MAX_TEMPLATE_EDGES];      /* edge source row inside a graph     */
    int32_t dst[MPQE_MAX_TEMPLATE_EDGES];      /* edge destination row               */
    int32_t rel_label[MPQE_MAX_TEMPLATE_EDGES];/* index into Formula.get_rels()      */
    int32_t var_node[MPQE_MAX_TEMPLATE_NODES]; /* index into Formula.get_nodes()     */
} mpqe_template_t;
int mpqe_template_info(int query_type, mpqe_template_t *out_host);

/* ---- (a1) collation ------------------------------------------------------------------
 * reference: RGCNQueryDataset.get_query_graph data_utils.py:394-405 and PyG
 * Batch.from_data_list: B replicas of one template.
 *   edge_index[0, b*E+e] = src[e] + b*N     edge_index[1, b*E+e] = dst[e] + b*N
 *   edge_type[b*E+e] = edge_type_host[e]    batch[b*N+n] = b                        */
int mpqe_collate_template(int query_type, int64_t batch_size,
                          const int64_t *edge_type_host /*[E]*/,
                          int64_t *edge_index /*[2, B*E]*/, int64_t *edge_type /*[B*E]*/,
                          int64_t *batch /*[B*N]*/, void *stream);

/* ---- (a2, a3) entity embedding gather + L2 normalise -------------------------------------
 * reference: DirectEncoder.forward encoders.py:40-43 with the features closure
 * data_utils.py:35 (row = node_map[id]; v = table[row]; y = v / ||v||_2, no eps), written
 * where RGCNEncoderDecoder.forward puts it (model.py:418-420):
 *   out[i*out_row_stride + 0..dim) = y_i.   inv_norm[i] = 1/||v_i|| (optional, for bwd).
 * node_map == NULL means ids are table rows already (identity map).                        */
int mpqe_embed_l2norm_fwd(const float *table, int64_t table_rows, int64_t dim,
                          const int64_t *node_map, int64_t node_map_len,
                          const int64_t *ids, int64_t n,
                          float *out, int64_t out_row_stride, float *inv_norm /*[n] or NULL*/,
                          int32_t *err, void *stream);
/* grad_table[row_i] += (g_i - y_i (y_i . g_i)) / ||v_i||   (fp32 atomics on duplicates) */
int mpqe_embed_l2norm_bwd(const float *grad_out, int64_t grad_row_stride,
                          const float *table, int64_t table_rows, int64_t dim,
                          const int64_t *node_map, int64_t node_map_len,
                          const int64_t *ids, int64_t n,
                          float *grad_table /*[table_rows, dim], accumulated into*/,
                          int32_t *err, void *stream);
/* variable rows: out[(b*N + A + k)*dim ..] = mode_emb[var_ids[k]] for every b
 * reference: model.py:421. */
int mpqe_var_rows_fwd(const float *mode_emb, int64_t num_modes, int64_t dim,
                      const int64_t *var_ids /*[V]*/, int64_t num_vars,
                      int64_t batch_size, int64_t num_nodes, int64_t num_anchors,
                      float *x /*[B*N, dim]*/, int32_t *err, void *stream);
/* grad_mode_emb[var_ids[k]] += sum_b grad_x[(b*N + A + k)] (deterministic order) */
int mpqe_var_rows_bwd(const float *grad_x, int64_t num_modes, int64_t dim,
                      const int64_t *var_ids, int64_t num_vars,
                      int64_t batch_size, int64_t num_nodes, int64_t num_anchors,
                      float *grad_mode_emb, int32_t *err, void *stream);

/* ---- (a4) R-GCN layer -------------------------------------------------------------------
 * reference: RGCNConv.forward/message/update model.py:269-305 with PyG propagate +
 * torch_scatter.scatter_add ('add' aggregation, edge_norm None):
 *   out[i] = sum_{e: dst_e = i} x[src_e] . basis[type_e]  +  x[i] . root  +  bias
 * relu != 0 additionally applies the F.relu the caller puts after all but the last layer
 * (model.py:437).
 *
 * TEMPLATE form: the batch is B replicas of one template (row = b*N + n). The neighbour sum
 * runs inside the MFMA K loop: for node slot n with in-edges e1..ek the tile computes
 * [x[:,src_e1] | ... | x[:,n]] . [basis[r_e1]; ...; root], so nothing is scattered and no
 * [B*E, D, D] weight copy (model.py:292-293) exists.                                        */
int mpqe_rgcn_template_fwd(int query_type, int64_t batch_size,
                           const int64_t *edge_type_host /*[E] relation id per template edge*/,
                           const float *x /*[B*N, dim_in]*/,
                           const float *basis /*[R, dim_in, dim_out]*/, int64_t num_relations,
                           const float *root /*[dim_in, dim_out]*/, const float *bias /*[dim_out] or NULL*/,
                           int64_t dim_in, int64_t dim_out, int relu,
                           float *out /*[B*N, dim_out]*/, void *stream);
size_t mpqe_rgcn_template_bwd_workspace_bytes(int query_type, int64_t batch_size,
                                              int64_t dim_in, int64_t dim_out);
/* grad_out is d loss / d out (post-ReLU when relu != 0; `out` is then needed for the mask).
 * grad_x is overwritten; grad_basis / grad_root / grad_bias are ACCUMULATED into (dense, like
 * the reference's autograd). Any of the three may be NULL to skip it. Deterministic.        */
int mpqe_rgcn_template_bwd(int query_type, int64_t batch_size, const int64_t *edge_type_host,
                           const float *x, const float *out, const float *grad_out,
                           const float *basis, int64_t num_relations, const float *root,
                           int64_t dim_in, int64_t dim_out, int relu,
                           float *grad_x, float *grad_basis, float *grad_root, float *grad_bias,
                           void *workspace, size_t workspace_bytes, void *stream);

/* GENERAL form: arbitrary edge_index / edge_type (duplicates, self loops, isolated nodes,
 * unused relations). A plan sorts the edges once per graph: by relation for the grouped
 * MFMA GEMM, by destination (forward) and by source (backward) for the segmented sums.    */
size_t mpqe_rgcn_plan_bytes(int64_t num_nodes, int64_t num_edges, int64_t num_relations);
size_t mpqe_rgcn_plan_workspace_bytes(int64_t num_nodes, int64_t num_edges, int64_t num_relations);
int mpqe_rgcn_plan_build(const int64_t *edge_index /*[2, E]*/, const int64_t *edge_type /*[E]*/,
                         int64_t num_nodes, int64_t num_edges, int64_t num_relations,
                         void *plan, size_t plan_bytes, void *workspace, size_t workspace_bytes,
                         int32_t *err, void *stream);
size_t mpqe_rgcn_general_workspace_bytes(int64_t num_nodes, int64_t num_edges, int64_t num_relations,
                                         int64_t dim_in, int64_t dim_out, int backward);
/* relu_mask (may be NULL; needs relu, dim_out % 64 == 0): mpqe_rgcn_general_mask_bytes(num_nodes, dim_out) bytes, 8-byte
 * aligned; the forward leaves the ReLU mask of `out` there as bit words (bit c % 64 of word [node][c / 64] = out > 0) and
 * the backward, given the same buffer, masks with them and never reads `out` (its gathered reads of `out` rows were a
 * third of the backward kernels' row traffic).                                                            */
size_t mpqe_rgcn_general_mask_bytes(int64_t num_nodes, int64_t dim_out);
int mpqe_rgcn_general_fwd(const void *plan, int64_t num_nodes, int64_t num_edges, int64_t num_relations,
                          const float *x, const float *basis, const float *root, const float *bias,
                          int64_t dim_in, int64_t dim_out, int relu, float *out, uint64_t *relu_mask,
                          void *workspace, size_t workspace_bytes, void *stream);
/* The scatter-aggregate of the forward alone (the destination-sorted segmented sum): out[i] = act(bias + msg[E + i] +
 * sum over the edges into i, in edge order, of their message rows); msg [E + Nn, dim] in the plan's slot order, as the
 * gather-GEMM of mpqe_rgcn_general_fwd leaves its messages (an edge's row = its position in the relation-sorted order,
 * the self term of node i at row E + i). Algorithmic bytes: 4 dim (E + 2 Nn).                                    */
int mpqe_rgcn_general_aggregate(const void *plan, int64_t num_nodes, int64_t num_edges, int64_t num_relations,
                                const float *msg, const float *bias, int64_t dim, int relu, float *out, void *stream);
/* overwrite = 0: grad_basis / grad_root / grad_bias are ACCUMULATED into (the caller zero-fills them); 1: every one of
 * them is WRITTEN whole, once -- the gradient, or zeros for a relation without an edge -- so the caller neither fills
 * [R, dim_in, dim_out] with zeros nor pays a read-modify-write of it (33 MB each at the stress shape). grad_x is always
 * written.                                                                                              */
int mpqe_rgcn_general_bwd(const void *plan, int64_t num_nodes, int64_t num_edges, int64_t num_relations,
                          const float *x, const float *out /* may be NULL with relu_mask */,
                          const uint64_t *relu_mask /* the forward's, or NULL */, const float *grad_out,
                          const float *basis, const float *root,
                          int64_t dim_in, int64_t dim_out, int relu, int overwrite,
                          float *grad_x, float *grad_basis, float *grad_root, float *grad_bias,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---- dense layer (the MLP readouts' nn.Linear, model.py:497-553; Encoder's compress product, encoders.py:120-124) ----
 * y [rows, dim_out] = [relu]( (accumulate ? y : 0) + x [rows, dim_in] . W^T + bias ), W [dim_out, dim_in] as nn.Linear
 * stores it, with row stride ldw >= dim_in (a column block of a wider matrix: the compress matrix applied to one
 * neighbour block at a time, the products accumulated, so the reference's concatenation is never materialised).
 * x, y, grad_y, grad_x contiguous. Backward: y = the layer's post-activation output (read only when relu != 0);
 * grad_x is written; grad_W [dim_out, .] (row stride ldgw) and grad_bias are accumulated into, or written
 * (overwrite = 1). Fixed-order reductions (no float atomics).                                              */
int mpqe_linear_fwd(const float *x, int64_t rows, const float *W, int64_t ldw, const float *bias, int64_t dim_in,
                    int64_t dim_out, int relu, int accumulate, float *y, void *stream);
size_t mpqe_linear_bwd_workspace_bytes(int64_t rows, int64_t dim_in, int64_t dim_out);
int mpqe_linear_bwd(const float *x, int64_t rows, const float *W, int64_t ldw, const float *y, const float *grad_y,
                    int64_t dim_in, int64_t dim_out, int relu, int overwrite, float *grad_x, float *grad_W, int64_t ldgw,
                    float *grad_bias, void *workspace, size_t workspace_bytes, void *stream);

/* ---- (a5) readouts ----------------------------------------------------------------------
 * Regular form for template batches (batch_idx = b repeated N):
 *   SUM  reference sum_readout model.py:380-381   out[b] = sum_n h[b*N+n]
 *   MAX  reference max_readout model.py:383-385   out[b] = max_n h[b*N+n]; argmax[b,d] = lowest n
 *   TM   reference target_message_readout 387-398 out[b] = h[b*N + A]                        */
int mpqe_readout_fwd(int kind, const float *h, int64_t batch_size, int64_t num_nodes,
                     int64_t num_anchors, int64_t dim, float *out /*[B, dim]*/,
                     int32_t *argmax /*[B, dim], MAX only, may be NULL*/, void *stream);
int mpqe_readout_bwd(int kind, const float *grad_out /*[B, dim]*/, const int32_t *argmax,
                     int64_t batch_size, int64_t num_nodes, int64_t num_anchors, int64_t dim,
                     float *grad_h /*[B*N, dim], overwritten*/, void *stream);
/* torch_scatter.scatter_add / scatter_max / scatter_mean along dim 0 (reference call sites
 * model.py:351-355, 381, 384, 509, 547). index may come in any order.
 * out[i] = reduce_{j: index[j] = i} src[j]; empty rows are 0; arg = lowest j on ties, -1 empty.
 * add / mean accumulate with fp32 atomics (order-dependent last bits when rows collide).     */
size_t mpqe_scatter_workspace_bytes(int64_t n_src, int64_t dim_size);
int mpqe_scatter_fwd(int op, const float *src, const int64_t *index, int64_t n_src, int64_t dim,
                     int64_t dim_size, float *out /*[dim_size, dim]*/, int64_t *arg /*MAX only*/,
                     void *workspace, size_t workspace_bytes, int32_t *err, void *stream);
int mpqe_scatter_bwd(int op, const float *grad_out, const int64_t *index, const int64_t *arg,
                     int64_t n_src, int64_t dim, int64_t dim_size, float *grad_src,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ---- (a9) LayerNorm + ReLU of the GraphSAGE-style Encoder ---------------------------------------
 * reference encoders.py:132-146 (unbiased std, eps added to the std) + the Encoder's ReLU (encoders.py:127-128):
 *   y = act(gamma * (x - mean) / (std + eps) + beta), per row of x [rows, dim]; stats [rows, 2] = {mean, 1/(std+eps)}.
 * bwd: grad_x overwritten; grad_gamma / grad_beta ACCUMULATED (fixed-order column sums).                */
int mpqe_layernorm_relu_fwd(const float *x, int64_t rows, int64_t dim, const float *gamma, const float *beta, float eps,
                            int relu, float *y, float *stats, void *stream);
size_t mpqe_layernorm_relu_bwd_workspace_bytes(int64_t rows, int64_t dim);
int mpqe_layernorm_relu_bwd(const float *grad_y, const float *x, const float *y, int64_t rows, int64_t dim,
                            const float *gamma, const float *stats, float eps, int relu, float *grad_x,
                            float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream);

/* ---- (a6, a7) scoring and loss ------------------------------------------------------------
 * reference: F.cosine_similarity(q, t, dim=1) model.py:452, 458 (eps 1e-8 clamps each norm):
 *   scores[i] = q[qrow(i)] . t[i] / (max(||q||, eps) * max(||t||, eps))
 * q_row (may be NULL = identity) is the repeat_interleave(out, neg_lengths) map of model.py:456. */
int mpqe_cosine_fwd(const float *q, const int64_t *q_row, const float *t, int64_t n, int64_t dim,
                    float eps, float *scores, void *stream);
/* grad_q is ACCUMULATED into when q_row != NULL (several i share a row; fp32 atomics),
 * overwritten otherwise; grad_t overwritten. Either may be NULL.                          */
int mpqe_cosine_bwd(const float *grad_scores, const float *q, const int64_t *q_row, const float *t,
                    int64_t n, int64_t dim, float eps, float *grad_q, float *grad_t, void *stream);
/* reference margin_loss model.py:483-485: loss = mean(clamp(margin - (pos - neg), min=0)).
 * Deterministic single-block reduction. */
int mpqe_hinge_fwd(const float *pos, const float *neg, int64_t n, float margin, float *loss, void *stream);
int mpqe_hinge_bwd(const float *pos, const float *neg, int64_t n, float margin, const float *grad_loss /*[1]*/,
                   float *grad_pos, float *grad_neg, void *stream);

/* ---- fused training step -------------------------------------------------------------------
 * Forward + backward of RGCNEncoderDecoder.margin_loss (reference model.py:464-494) for ALL the
 * formula batches of one training step (reference train_helpers.py:76-120 draws 11 after
 * burn-in), each with explicit positive and negative targets, in 2 - 3 launches (dim 64 / 128 / 256: the
 * graph-block chain form; ~15 launches in the level form, MPQE_STEP_NO_CHAIN or other dims):
 *     loss[0] = sum_b weight_b * mean_i clamp(margin - (pos_bi - neg_bi), 0);  loss[1+b] = that mean
 * Gradients of every parameter are ACCUMULATED into `grads` (dense, like the reference's
 * autograd). Readouts: sum / max / mp(TM). The query embedding is computed once per batch and
 * scored against both targets (the reference encodes twice; same maths).
 * Descriptors are HOST structs; every pointer inside them is a device pointer.               */
#define MPQE_STEP_MAX_BATCHES 16
#define MPQE_STEP_MAX_LAYERS 8
#define MPQE_STEP_MAX_MODES 16
/* mpqe_step_params_t.flags -- speed switches; loss, scores and gradients are the same either way:
 * NO_PRUNE  also compute node states that cannot reach the readout (the reference computes all of them;
 *           with the TM readout only the target row is read, model.py:391-398, so e.g. 9 of the 21
 *           node updates of a 3-chain feed nothing and get an exactly-zero gradient). Default: skip them.
 * NO_CHAIN  one launch per message-passing level instead of the graph-block chain kernels.      */
#define MPQE_STEP_NO_PRUNE 1
#define MPQE_STEP_NO_CHAIN 2
/* ZERO_GRADS  the call zero-fills every buffer of `grads` before accumulating into it (one launch shared
 *             with the step's other prologue work, instead of a memset per buffer by the caller).    */
#define MPQE_STEP_ZERO_GRADS 4
/* NO_KSPLIT   dim 128 only: the chain kernel's waves each own 32 columns and the whole K range (the first form)
 *             instead of 64 columns and half of K.                                                     */
#define MPQE_STEP_NO_KSPLIT 8
/* EIGHT_WAVES dim 128 only, experimental: chain workgroups of eight waves (a wave of each K half on every SIMD, 32
 *             columns each) instead of four. Same results; measured 1 % slower on the AIFB mix.  */
#define MPQE_STEP_EIGHT_WAVES 16
/* NO_UNIFORM  chain form: treat every node state as per-graph rows. Default: node states no anchor has reached yet
 *             are ONE vector per batch (the variable rows of x0 are the same mode_embeddings row for every graph of
 *             a batch, reference model.py:421), so their updates are matrix-VECTOR products done once per batch by a
 *             pre-pass, their backward runs on column sums, and their weight gradients are rank-1 updates -- the
 *             AIFB mix keeps 34 of its 66 live [B, D] x [D, D] products per direction.                       */
#define MPQE_STEP_NO_UNIFORM 32
/* SPARSE_TABLES  (with a touch plan) the entity-table gradients are delivered ROW-SPARSE: the rows the step's ids
 *             touched are written (not accumulated) into the dense `grads->tables` buffers and every other row is
 *             left as it is -- no zero fill and no pass over tables of 10^5 .. 10^6 rows per step. For consumers that
 *             read only the touched rows: mpqe_adam_rows_step, the row exchange of the data-parallel path.      */
#define MPQE_STEP_SPARSE_TABLES 64
/* MERGE_TAIL / SPLIT_TAIL  (chain form) where the weight-gradient tiles and the backward post-pass run. Merged: as
 *             workgroups of the chain launch itself, queued behind the chain workgroups and started per batch as soon
 *             as that batch's chain workgroups have published their rows (two launches per step: chain, reduction).
 *             Split: as a launch of their own behind the chain launch (three launches). Same results either way.
 *             Default: merged while the step has at most 9/8 x 256 chain workgroups (16 query graphs each) -- there
 *             it is 4 .. 12 % faster --, split beyond (the AIFB step of 11 x 512 graphs: merged is 4 % slower).
 *             MERGE_TAIL forces the merged form, SPLIT_TAIL the split one.                                    */
#define MPQE_STEP_MERGE_TAIL 128
#define MPQE_STEP_SPLIT_TAIL 256
/* BUILD_TOUCH  (chain form, backward) the step builds the touch plan of the ids it is called with ITSELF, by workgroups
 *             that lead its first launch and run beside the forward / backward chains (step_touch.h): `touch` is then an
 *             OUTPUT buffer (mpqe_step_touch_bytes; valid once the call has run: mpqe_adam_rows_step and the row exchange
 *             read its keys), and a step with fresh ids costs no more than a replayed one -- nothing id-dependent is left
 *             for collation time (the reference resolves ids and accumulates embedding gradients inside forward /
 *             backward: encoders.py:40-43, data_utils.py:35). The level form has no use for a plan and leaves the buffer
 *             alone. Steps of more than 524 288 looked-up ids, or split over stream lanes, return MPQE_ERR_UNSUPPORTED:
 *             build the plan with mpqe_step_touch_build instead.                                              */
/*             The plan buffer must be ZERO-FILLED once, before its first use as an output: a build that gives up
 *             (MPQE_FLAG_TOUCH_RETRY) leaves it as it was, and the step's last launch reads plan entries before it knows. */
#define MPQE_STEP_BUILD_TOUCH 512
/* ADD_STATE_GRADS  (MPQE_READOUT_CALLER, PHASE_FROM_STATES) the caller's readout read the states of EVERY level 1 .. L_b
 *             (the reference's `concat` readout, model.py:441-446) and has written its d loss / d state into every row of
 *             those gradient levels, not only the final one: the backward adds the gradient it propagates to them.  */
#define MPQE_STEP_ADD_STATE_GRADS 1024
/* TOUCH_LIBRARY_SORT  (mpqe_step_touch_build only) build the plan with the library's multi-launch radix sort instead of
 *             the one-launch sort whose workgroups synchronise among themselves: slower, and independent of how many
 *             workgroups the device can hold at once -- the recovery path behind MPQE_FLAG_TOUCH_RETRY.          */
#define MPQE_STEP_TOUCH_LIBRARY_SORT 2048

typedef struct {
    int32_t query_type;        /* MPQE_Q_*                                                     */
    int32_t num_passes;        /* message-passing passes: diameter if adaptive else num_layers */
    int32_t batch_size;        /* B: query graphs of this formula                              */
    int32_t target_mode;       /* entity-table index of the targets / negatives                */
    int64_t edge_type[MPQE_MAX_TEMPLATE_EDGES];  /* relation id per template edge              */
    int64_t var_ids[MPQE_MAX_TEMPLATE_NODES - 1];/* mode id per variable node                  */
    int32_t anchor_mode[MPQE_MAX_TEMPLATE_EDGES];/* entity-table index per anchor slot         */
    float weight;              /* weight of this batch's loss in the step loss                 */
} mpqe_step_batch_t;

typedef struct {
    int32_t dim, num_layers, num_relations, num_modes;
    int32_t readout;           /* MPQE_READOUT_*                                               */
    int32_t flags;             /* MPQE_STEP_* bits below; 0 = everything on                    */
    const float *tables[MPQE_STEP_MAX_MODES];    /* per-mode entity table [rows, dim]          */
    int64_t table_rows[MPQE_STEP_MAX_MODES];
    const int64_t *node_map;   /* global entity id -> row of its mode's table (-1: none)       */
    int64_t node_map_len;
    const float *mode_emb;     /* [num_modes, dim]                                             */
    const float *basis[MPQE_STEP_MAX_LAYERS];    /* layers[i].basis [R, dim, dim]; shared      */
    const float *root[MPQE_STEP_MAX_LAYERS];     /* layers repeat the same pointers            */
    const float *bias[MPQE_STEP_MAX_LAYERS];
    /* learned readouts (MPQE_READOUT_MLP / _TARGETMLP / _CONCAT): the two Linear layers as nn.Linear stores them
     * (weight [out, in] row-major, bias [out]); readout_scatter = MPQE_SCATTER_* of the reduction (the reference's
     * --scatter_op); readout_weight_decay: model.py:486-490, loss += weight_decay * (sum of the four parameters'
     * 2-norms) per margin_loss call, i.e. times the sum of the batch weights (0 = off). dim % 4 == 0; chain form: 16-byte
     * aligned parameters.                                                                                             */
    const float *readout_w0, *readout_b0, *readout_w2, *readout_b2;
    int32_t readout_scatter;
    float readout_weight_decay;
} mpqe_step_params_t;

typedef struct {
    float *tables[MPQE_STEP_MAX_MODES];          /* NULL = not wanted                          */
    float *mode_emb;
    float *basis[MPQE_STEP_MAX_LAYERS];
    float *root[MPQE_STEP_MAX_LAYERS];
    float *bias[MPQE_STEP_MAX_LAYERS];
    float *readout_w0, *readout_b0, *readout_w2, *readout_b2;      /* learned readouts                  */
} mpqe_step_grads_t;

/* Stream lanes (optional). At the reference's batch size every launch of the step is short enough that
 * its fixed cost (~8 us: dispatch, descriptor + first-tile latency, drain) dominates, and one step is a
 * chain of ~13 dependent launches. With lanes, lane l runs the whole chain (assemble -> message-passing
 * levels -> score -> levels back) of batches [batch_begin[l], batch_begin[l+1]) on its own stream; lane 0
 * is `stream`. The lanes fork after the descriptor upload and join before the weight-gradient launch (chain
 * form: every lane also launches the weight gradients of its own batches, the lanes join before the reduction).
 * The caller owns the streams and events (no allocation, no synchronisation in the library).          */
#define MPQE_STEP_MAX_LANES 4
typedef struct {
    int32_t num_lanes;                              /* 1 .. MPQE_STEP_MAX_LANES                      */
    int32_t batch_begin[MPQE_STEP_MAX_LANES + 1];   /* ascending, [0] = 0, [num_lanes] = num_batches */
    void *aux_stream[MPQE_STEP_MAX_LANES];          /* hipStream_t of lanes 1.. ([0] unused)         */
    void *fork_event;                               /* hipEvent_t                                    */
    void *join_event[MPQE_STEP_MAX_LANES];          /* hipEvent_t of lanes 1.. ([0] unused)          */
} mpqe_step_lanes_t;

size_t mpqe_step_workspace_bytes(const mpqe_step_params_t *params_host, const mpqe_step_batch_t *batches_host,
                                 int num_batches, const mpqe_step_lanes_t *lanes /* NULL = one lane */);
/* The kernels read a small descriptor table (templates, relations, offsets, reduction groups). It
 * lives in a caller-owned device buffer `desc` (256-byte aligned): with upload_desc != 0 the call
 * first writes it (by kernel arguments, no host memory is read asynchronously); a packed step
 * that is run again unchanged passes upload_desc = 0 and re-uses it -- building it is part of
 * collation, like the reference's collate_fn building edge_index / edge_type.                  */
size_t mpqe_step_desc_bytes(const mpqe_step_params_t *params_host, const mpqe_step_batch_t *batches_host,
                            int num_batches, const mpqe_step_lanes_t *lanes /* the split the step will run with */);
/* Entity-table gradients without float atomics ("touch plan"). The reference's autograd scatters the gradient rows
 * of the looked-up entities into the dense tables with index_add (deterministic on the CPU). Which looked-up ids share
 * a table row is a function of the ids alone, so it is found once per packed step, at collation time:
 * mpqe_step_touch_build gathers (table, row) keys through node_map and sorts them (stable) into `touch`, a caller-owned
 * device buffer (256-byte aligned, mpqe_step_touch_bytes; scratch: mpqe_step_touch_workspace_bytes, free after the
 * call has run). Handed to mpqe_step_forward_backward (chain form), the step stores one gradient row per looked-up id
 * and adds the rows of each table row in that fixed order: bit-reproducible, and no atomic traffic. touch = NULL keeps
 * the fp32-atomic form (results equal up to the order of the additions). The ids must be the ones the step is run with. */
size_t mpqe_step_touch_bytes(const mpqe_step_params_t *params_host, const mpqe_step_batch_t *batches_host,
                             int num_batches);
size_t mpqe_step_touch_workspace_bytes(const mpqe_step_params_t *params_host, const mpqe_step_batch_t *batches_host,
                                       int num_batches);
int64_t mpqe_step_touch_entries(const mpqe_step_batch_t *batches_host, int num_batches);   /* looked-up ids of a step */
int mpqe_step_touch_build(const mpqe_step_params_t *params_host, const mpqe_step_batch_t *batches_host, int num_batches,
                          const int64_t *anchor_ids, const int64_t *targets, const int64_t *negs, void *touch,
                          size_t touch_bytes, void *workspace, size_t workspace_bytes, void *stream);
/* anchor_ids: per batch b a block of [A_b, B_b] ids (slot-major), blocks concatenated in batch
 * order; targets / negs: [sum_b B_b]. backward = 0 stops after the loss (grads may be NULL).
 * scores_pos / scores_neg: [sum_b B_b] or NULL. workspace must be 256-byte aligned.
 * lanes (may be NULL = one lane): see mpqe_step_lanes_t.
 * events (may be NULL): hipEvent_t handles recorded in pairs around single launches on the stream of the
 * launch, in this order: for level 0..Lmax-1, for each lane that has the level: layer forward; for level
 * Lmax-1..0, for each such lane: backward-x; then the weight-gradient launch. When the graph-block chain
 * kernel runs (dim 64 / 128 / 256, MPQE_STEP_NO_CHAIN clear, every batch at most 5 passes) the order is: for each lane its chain launch (assemble, levels forward, scores,
 * levels backward), then for each lane its weight-gradient launch; in both forms last the step's reduction launch. Fewer
 * are filled as far as they go.
 * For roofline accounting only.                                                                     */
/* readout = MPQE_READOUT_CALLER: the step in THREE calls around a readout the caller computes itself -- the learned
 * readouts of the reference (MLPReadout / TargetMLPReadout, model.py:497-553), whose Linear layers are the caller's
 * (mpqe_linear_fwd / bwd); everything else of the step stays in the library. `backward` then names the call; the
 * level form runs (node states in HBM, every state live), one lane. Same params / batches / ids / desc / workspace in all
 * calls; upload_desc as usual in the first, 0 afterwards. (readout CALLER with backward 0 / 1, or a phase with another
 * readout: MPQE_ERR_INVALID_ARG.)
 *   PHASE_STATES       zero fill (MPQE_STEP_ZERO_GRADS), gather, every level forward. Afterwards the node states of
 *                      level p are rows [rows_total, dim] at workspace + states_offset + 4 * p * level_stride
 *                      (mpqe_step_states_layout); batch b's graphs own rows row_offset[b] + g * N_b + n, and its final
 *                      states are those of level num_passes_b.
 *   PHASE_SCORES       the caller has written the query embeddings [graphs_total, dim] (batch order) at workspace +
 *                      queries_offset: cosine scores against the targets / negatives, hinge terms, d loss / d embedding
 *                      to workspace + query_grads_offset, the targets' / negatives' entity-table gradients into grads.
 *                      scores_pos / scores_neg are written.
 *   PHASE_FROM_STATES  the caller has put d loss / d (final state) into the same rows of the gradient levels
 *                      (workspace + grads_offset + ...) -- every row of a batch's final level --: levels backward,
 *                      weight / bias / mode-vector gradients, the anchors' entity-table gradients, the reduction;
 *                      `loss` is written.
 *   PHASE_SCORES_ONLY  PHASE_SCORES without gradients, and `loss` written: the forward-only step (grads may be NULL). */
#define MPQE_STEP_PHASE_STATES 2
#define MPQE_STEP_PHASE_SCORES 3
#define MPQE_STEP_PHASE_FROM_STATES 4
#define MPQE_STEP_PHASE_SCORES_ONLY 5
int mpqe_step_states_layout(const mpqe_step_params_t *params_host, const mpqe_step_batch_t *batches_host,
                            int num_batches, const mpqe_step_lanes_t *lanes, int64_t *states_offset /* bytes */,
                            int64_t *grads_offset /* bytes */, int64_t *level_stride /* floats */,
                            int64_t *row_offset /* [num_batches + 1], rows */, int64_t *queries_offset /* bytes */,
                            int64_t *query_grads_offset /* bytes */);
int mpqe_step_forward_backward(const mpqe_step_params_t *params_host, const mpqe_step_batch_t *batches_host,
                               int num_batches, const int64_t *anchor_ids, const int64_t *targets,
                               const int64_t *negs, float margin, const mpqe_step_grads_t *grads_host,
                               int backward, float *loss /*[1 + num_batches]*/, float *scores_pos,
                               float *scores_neg, void *desc, size_t desc_bytes, int upload_desc,
                               void *workspace, size_t workspace_bytes, int32_t *err,
                               const mpqe_step_lanes_t *lanes, void *const *events, int num_events,
                               void *touch /* mpqe_step_touch_build's buffer (read), the plan buffer to fill (MPQE_STEP_BUILD_TOUCH), or
                                              NULL */, void *stream);

/* The same call with per-call extras (extra = NULL: exactly mpqe_step_forward_backward). For the host mirror's drop-in
 * entry points -- RGCNEncoderDecoder.margin_loss / .forward (reference model.py:400-494) routed through the fused step:
 *   batch_weight[i]  a DEVICE scalar (or NULL = 1): batch i's loss weight is batches_host[i].weight * *batch_weight[i],
 *                    read when the step runs. `loss = l_0 + w_1 * l_1 + ...; loss.backward()` (reference
 *                    train_helpers.py:81-119) hands every margin_loss node its upstream gradient as a 0-dim device tensor:
 *                    the backward of ALL nodes is then ONE step whose weights are those tensors, with no device-to-host read.
 *                    The weights scale the gradients (and the readout regulariser's, model.py:486-490); loss[0] is formed
 *                    with the host weights alone. A launch of one workgroup writes them into the resident descriptor table
 *                    in front of the step's launches (a later call without extras writes the host weights back).
 *   query_out        [sum_b B_b, dim] or NULL: the query embeddings (the readout's output rows, model.py:447-449) in batch
 *                    order -- what the evaluation form scores against ragged negative lists (model.py:454-460,
 *                    mpqe_cosine_fwd with q_row). Chain form only (MPQE_ERR_UNSUPPORTED otherwise).
 *   notify           two 32-bit words the DEVICE can write and the host can read without a call (pinned host memory), or
 *                    NULL: the workgroup that forms the loss (behind every launch and workgroup of the call that reads the
 *                    ids) stores notify[1] = the error word as it
 *                    stands, then notify[0] = notify_value. Once the host reads notify_value there, every launch of the call
 *                    that reads anchor_ids / targets / negs has run (the id arrays may be refilled), and notify[1] tells
 *                    whether a bad id was met -- the reference raises IndexError inside forward (encoders.py:40-43); a host
 *                    mirror that must not synchronise per call polls this word instead.
 *   xcd_shift        0 .. 7: a FORWARD-ONLY step in the chain form places its workgroups `xcd_shift` XCDs further round the
 *                    chip (a batch's graph blocks are dealt to one XCD per 32, so that its matrices stay in one L2; workgroup
 *                    b of a launch runs on XCD b % 8). Forward-only steps of SEVERAL packed steps issued on different streams
 *                    at once (each with a workspace of its own) then run side by side instead of sharing the CUs of XCD 0.
 *                    Results do not depend on it.
 *   readout_norms    a DEVICE scalar holding sum_i ||readout parameter i||_2 as mpqe_step_readout_norms wrote it, or NULL. A
 *                    FORWARD-ONLY step in the chain form with a learned readout and readout_weight_decay > 0 then adds the
 *                    regulariser (model.py:486-490) to loss[0] from it instead of re-forming the four norms in a launch of
 *                    its own per call: a caller that issues many forward-only calls between two parameter updates (the
 *                    reference's loop: eleven margin_loss calls per optimiser step) computes them once. The caller answers
 *                    for the scalar being current. Same arithmetic, same value.
 *   join_event,      join_event != NULL (a hipEvent_t; join_stream a hipStream_t, NULL = the null stream): behind the call's
 *   join_stream      last launch the library records join_event on `stream` and makes join_stream wait for it -- the call ran
 *                    on a side stream, its consumer is enqueued on join_stream. (The other direction -- `stream` waiting for
 *                    the parameters' last writer -- is the caller's.)                                                    */
typedef struct {
    const float *batch_weight[MPQE_STEP_MAX_BATCHES];
    float *query_out;
    uint32_t *notify;
    uint32_t notify_value;
    int32_t xcd_shift;
    void *join_event;
    void *join_stream;
    const float *readout_norms;
} mpqe_step_extra_t;
/* out[0] = sum_i ||p_i||_2 over the learned readout's four parameters (readout_w0, readout_b0, readout_w2, readout_b2 of
 * params_host; reference model.py:486-490), in the order and arithmetic of the step's own regulariser launch. One launch. */
int mpqe_step_readout_norms(const mpqe_step_params_t *params_host, float *out, void *stream);
int mpqe_step_forward_backward_ex(const mpqe_step_params_t *params_host, const mpqe_step_batch_t *batches_host,
                                  int num_batches, const int64_t *anchor_ids, const int64_t *targets,
                                  const int64_t *negs, float margin, const mpqe_step_grads_t *grads_host,
                                  int backward, float *loss, float *scores_pos, float *scores_neg, void *desc,
                                  size_t desc_bytes, int upload_desc, void *workspace, size_t workspace_bytes,
                                  int32_t *err, const mpqe_step_lanes_t *lanes, void *const *events, int num_events,
                                  void *touch, void *stream, const mpqe_step_extra_t *extra_host);

/* The entity-table part of a step's reduction again, from a plan built AFTER the step ran: sums the per-entry gradient
 * rows the step left in `workspace` (its chain launch writes them whatever happens to the plan) per destination row in
 * plan order into grads->tables -- written where the step would have written them (MPQE_STEP_ZERO_GRADS or
 * MPQE_STEP_SPARSE_TABLES in params->flags), added otherwise. For a step that reported MPQE_FLAG_TOUCH_RETRY: same
 * params / batches / desc / workspace as that call (nothing else may have run in the workspace since), `touch` from
 * mpqe_step_touch_build on the same ids. Chain form only. The reference's embedding backward cannot fail
 * (encoders.py:40-43 + autograd); this is what makes the in-step plan equally safe.                            */
int mpqe_step_table_rows(const mpqe_step_params_t *params_host, const mpqe_step_batch_t *batches_host, int num_batches,
                         const mpqe_step_grads_t *grads_host, const void *desc, void *workspace, size_t workspace_bytes,
                         const void *touch, void *stream);

/* margin_loss's regulariser on the readout's parameters (reference model.py:486-490: sum_i ||p_i||_2, unsquared) for the
 * drop-in modules: out[0] += sum_i ||params[i]||_2 (out != NULL; the caller zero-fills it) and / or grads[i] += *grad_out *
 * params[i] / ||params[i]|| (grads != NULL; grad_out: the upstream gradient, one float on the device, NULL = 1). Up to four
 * tensors per call, one workgroup, fixed order of every sum. The fused step computes the same inside its own call.   */
int mpqe_l2_norms(const float *const *params /* host array of device pointers */, const int64_t *sizes /* host */, int count,
                  const float *grad_out, float *out, float *const *grads, void *stream);

/* Ids of the next step from pinned host memory to the device on `stream` (hipMemcpyAsync; stream-ordered, returns at
 * once): the reference moves its index tensors with .to(device) per call (utils.py:17-23). For host mirrors without a
 * HIP binding of their own.                                                                          */
int mpqe_copy_to_device(void *dst, const void *src_host, size_t bytes, void *stream);

/* ---- optimiser step ---------------------------------------------------------
 * reference train.py:83-88: optim.Adam(params, lr) / optim.SGD(params, lr, momentum=0) over every
 * parameter, dense entity tables included. One launch over flat fp32 buffers (the fused step keeps the
 * gradients flat already). Same update rule as torch.optim.Adam (amsgrad off) / torch.optim.SGD:
 *   g += wd p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * step = t >= 1 (the caller counts). All four arrays [n], updated in place. Hyper-parameters are doubles,
 * as torch holds them: 1 - beta and the bias corrections are formed in double and rounded once.       */
int mpqe_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr,
                   double beta1, double beta2, double eps, double weight_decay, int64_t step, void *stream);
int mpqe_sgd_step(float *param, const float *grad, int64_t n, double lr, double weight_decay, void *stream);
/* Adam over ONLY the entity-table rows a packed step touched ("row-sparse Adam for entity tables"):
 * the distinct (table, row) keys of `touch` (mpqe_step_touch_build). Update rule = torch.optim.SparseAdam's, applied
 * to the per-row gradient sums the step left in grads[t][row]:
 *   m += (1-b1)(g - m);  v += (1-b2)(g g - v);  p -= lr sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
 * Deviation from the reference's dense Adam (train.py:87), stated: rows that a step does not touch keep their moments
 * undecayed and do not move (dense Adam decays m, v of every row every step and moves rows whose m is non-zero). In
 * exchange a step costs O(touched rows), not 16 bytes x every table element (AM: 191 MB tables, 1M entities: 1 GB).
 * params / grads / exp_avg / exp_avg_sq: [num_modes] host arrays of device pointers, tables [rows_t, dim]; step >= 1. */
int mpqe_adam_rows_step(const void *touch, int64_t num_entries /* of the plan: mpqe_step_touch_entries, or n of
                        mpqe_rows_plan_build */, float *const *params, const float *const *grads, float *const *exp_avg,
                        float *const *exp_avg_sq, int num_modes, int64_t dim, double lr, double beta1, double beta2,
                        double eps, int64_t step, void *stream);

/* ---- data-parallel exchange of entity-table gradient rows ("exchange touched rows only") ---------
 * Every rank all-gathers the (table, row) keys its packed step touches (key = table << row_bits | row, ~0 = none: the
 * touch plan's sorted keys without repeats, padded) ONCE at pack time and builds one plan over all ranks' keys
 * (mpqe_rows_plan_build: stable sort, so equal keys stay in rank order). Per step each rank all-gathers its gradient
 * rows in its own key order and mpqe_table_rows_sum adds, per key, the gathered rows in plan order into the dense table
 * gradients: the same additions in the same order on every rank -- equal to the dense all-reduce (a sum), bit-identical
 * replicas, and bytes on the wire proportional to the touched rows, not to the tables.                          */
size_t mpqe_rows_plan_bytes(int64_t n);
size_t mpqe_rows_plan_workspace_bytes(int64_t n, int key_bits);
int mpqe_rows_plan_build(const uint64_t *keys /* [n], device */, int64_t n, int row_bits, int key_bits, void *plan,
                         size_t plan_bytes, void *workspace, size_t workspace_bytes, void *stream);
int mpqe_table_rows_sum(const void *plan, int64_t n, const float *rows /* [n, dim] in the order of `keys` */, int64_t dim,
                        float *const *table_grads /* [num_modes] host array of device pointers */, int num_modes,
                        int store /* 1: rows are written, 0: added to */, void *stream);

/* ---- gradient-bucket exchange over peer-mapped buffers (SURVEY.md 5 last row, 8e: "one-hop reduce-scatter + all-gather
 * using all 7 links concurrently ... a custom P2P two-shot over IPC-mapped buffers") ------------------------------------
 * No reference counterpart (the reference is single-process). Every rank of ONE node owns a communication buffer
 * [bucket n floats | world staging slots of one shard | flags] (mpqe_p2p_buffer_bytes; fine-grained device memory from
 * mpqe_p2p_alloc, which also returns its IPC handle) and maps its peers' buffers (mpqe_p2p_open on the handles, exchanged
 * by the host side). mpqe_p2p_allreduce, stream-ordered, per rank: push my contribution to every shard into its owner's
 * slot -> sum the slots of MY shard in rank order and write the sum into that shard of every rank's bucket -> wait until
 * every shard of my bucket has landed. Each byte crosses one link once per direction; the sum of a shard is computed once,
 * so all replicas hold the same bits. epoch: != 0, a new value per call (flags are epoch-stamped, never cleared). phases:
 * 1 push | 2 reduce | 4 wait (7 = the whole exchange; separate phases are for single-process tests). Every poll is
 * bounded: a peer that never arrives ORs MPQE_FLAG_INTERNAL into err (the caller falls back to another collective). */
size_t mpqe_p2p_handle_bytes(void);
size_t mpqe_p2p_buffer_bytes(int64_t n, int world, int64_t *stage_offset, int64_t *flags_offset);
int mpqe_p2p_alloc(size_t bytes, void **ptr, void *handle_out);
int mpqe_p2p_free(void *ptr);
int mpqe_p2p_open(const void *handle, void **mapped);
int mpqe_p2p_close(void *mapped);
int mpqe_p2p_allreduce(void *const *buffers /* [world] host array: buffer of rank p as THIS process addresses it */, int rank,
                       int world, int64_t capacity /* the n the buffers were sized for */, int64_t n /* floats to reduce, <= capacity */,
                       uint32_t epoch, int phases, int32_t *err, void *stream);

/* Glue of the per-step exchange as library launches (the host mirror did it with torch._foreach_copy_ / index_select /
 * where): mpqe_spans_copy: dst[d .. d + n) <- src[s .. s + n) for every span of a DEVICE table of {int64 dst offset, src
 * offset, floats, first workgroup} records (4 096 floats per workgroup; total_blocks = sum of ceil(n / 4096)) -- the
 * touched matrices into the contiguous bucket and back. mpqe_rows_prepare: from the SORTED keys of a touch plan the step
 * built itself, the first key of every run (other slots ~0: fixed size `cap`) and the row of the flat [rows, dim]
 * table-gradient view it names (row_base[table] + row). mpqe_rows_gather: out[i] = rows[gidx[i]].              */
int mpqe_spans_copy(float *dst, const float *src, const void *spans_device, int nspans, int64_t total_blocks, void *stream);
int mpqe_rows_prepare(const uint64_t *sorted_keys, int64_t M, int64_t cap, int row_bits, const int64_t *table_rows /* host */,
                      const int64_t *row_base /* host */, int num_tables, uint64_t *send_keys, int64_t *gidx, void *stream);
int mpqe_rows_gather(const float *rows, const int64_t *gidx, int64_t n, int64_t dim, float *out, void *stream);

/* ---- negative sampling with python's own stream (reference model.py:466-476) -- HOST function, no device work ----------
 * margin_loss draws one negative per query with random.choice(list): CPython's Random.choice is
 * list[_randbelow(len)], _randbelow(n): k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k), and
 * getrandbits(k <= 32) is ONE 32-bit Mersenne-Twister output shifted right by 32 - k. The drop-in keeps that stream: the
 * host mirror takes raw outputs from the interpreter's generator (random.getrandbits(32 * n) = n consecutive outputs, the
 * first in the low bits) and this routine replays the rejection loop over them for queries cursor[0] .. nq - 1 in order:
 *   words [nwords] raw outputs; lens_host [nq] list length per query (or NULL: every list has `len_all` entries);
 *   base_host [nq] (or NULL = 0): offset of query i's list in `cand_host`; out_host[i] = cand_host[base + r] (cand_host
 *   NULL: out_host[i] = r, the drawn position).
 * cursor_host[0] (in/out) = next query to serve, cursor_host[1] (out) = words consumed by this call (all of them unless the
 * last query was served first). Returns MPQE_OK, or MPQE_ERR_INVALID_ARG for an empty list (random.choice raises
 * IndexError) or one of 2^32 or more entries. Every pointer is a HOST pointer.                                        */
int mpqe_host_random_choice(const uint32_t *words_host, int64_t nwords, const int64_t *lens_host, int64_t len_all,
                            const int64_t *base_host, const int64_t *cand_host, int64_t nq, int64_t *cursor_host,
                            int64_t *out_host);

/* ---- negative sampling on the device -------------------------------------------
 * reference model.py:466-476: one negative per query, random.choice over query.neg_samples /
 * query.hard_neg_samples (ragged per query) or graph.full_lists[target_mode] (1-chain: one list for all).
 * cand [n_cand]: candidate entity ids. offsets [n_lists + 1] (CSR over cand) or NULL = every query draws
 * from the whole of cand. qidx [nq] (or NULL = identity): list of batch position i. out[i] = a uniform
 * draw from the list, chosen by a counter-based hash of (seed, i) -- stateless, reproduced on the CPU by
 * oracle/ref_cpu.py; NOT python's random stream (same distribution, other numbers). An empty or invalid
 * list ORs MPQE_FLAG_BAD_INDEX into err and writes -1 (the reference raises IndexError).             */
int mpqe_sample_negatives(const int64_t *cand, int64_t n_cand, const int64_t *offsets, int64_t n_lists,
                          const int64_t *qidx, int64_t nq, uint64_t seed, int64_t *out, int32_t *err,
                          void *stream);

/* Diagnostics, not part of the data path: while `device_buffer` (8 int64 per workgroup, num_blocks
 * workgroups) is set, every chain-kernel launch with at most num_blocks workgroups writes per workgroup the
 * device wall clock (100 MHz) at its phase boundaries [0..6] and HW_ID | XCC_ID << 32 in [7]; workgroup g of a
 * launch of G workgroups also writes its shader-clock ticks at the first / last stamp to words 0 / 1 of entry
 * G + g, so 2 G <= num_blocks is required. NULL turns it off. Process-global; tools/chain_timeline.py only. */
void mpqe_debug_chain_stamps(void *device_buffer, size_t num_blocks);
/* The same for the weight-gradient launch: 8 int64 per workgroup: wall clock at start [0] and end [1], shader-clock
 * ticks start -> end [2], HW_ID | XCC_ID << 32 in [3], wall clock when the tile's record is read [4], when its first
 * K-step has landed [5] and when its K loop ends [6].                                                */
void mpqe_debug_tail_stamps(void *device_buffer, size_t num_blocks);
/* Named diagnostics switches (timing experiments; tests that force a rarely taken path, e.g. "TOUCH_MULTI_LAUNCH" = the
 * multi-launch sort (csrc/radix_sort.h) instead of the one-launch sort, "TSORT_FAIL" = the in-step sort gives up as if its workgroups were not
 * co-resident, "GEN_SLOTS" = grid size of the persistent gather-GEMMs, "PROLOGUE_LAST" = 1 / 0: the chain launch's prologue
 * items behind / in front of its chain workgroups whatever their number, "POST_IN_CHAIN" = the backward post-pass as roles of
 * the chain launch with the tiles a launch of their own, "NO_RUNS" = the reduction's table workgroups take every sorted
 * position of the touch plan instead of its compacted run starts, "DUMP_PLAN" = the step's plan and launch shape on stderr). set != 0
 * stores `value` under `name`, set == 0 removes it. Process-global (see Conventions).                    */
void mpqe_debug_option(const char *name, int value, int set);
/* The launch forms of the fused step that were built, proven bit-equal and measured SLOWER (the post-pass as closures
 * "CLOSURE", the loss + table rows as roles of the weight-gradient launch "EARLY_ROWS", the reduction fused into it
 * "FUSE_TAIL", range-per-workgroup table sums "ROWS_MULTI", the post-pass alone in the chain launch "POST_IN_CHAIN") are NOT
 * in the shipped library: they compile with -DMPQE_EXPERIMENTS only (tools/build_variant.sh). 1 = this build has them. */
int mpqe_debug_has_experiments(void);

#ifdef __cplusplus
}
#endif
#endif /* MPQE_AMD_H */

// Part of csrc/step.hip (one translation unit; included there after the kernels): the HOST side of the fused step -- the
// launch plan of a descriptor set (a pure function of mpqe_step_params_t + mpqe_step_batch_t[]: liveness of node states,
// batch-uniform states and their vector ops, chain programmes per 16-graph block, XCD placement, weight-gradient tile
// sources and reduction groups, workspace / descriptor-table layout) and the cache that keeps it next to the caller's
// descriptor buffer. reference: the static structure of RGCNEncoderDecoder.forward (model.py:404-449) for one formula.
#pragma once

// ------------------------------------------------------------------------------------ host side
namespace {

#define STEP_CUS 256
#define STEP_XCDS 8
#define STEP_RESIDENT 4
struct HostPlan {
    StepDev sd;
    int Lmax;
    // per (lane, level): tile tables of the forward / backward-x launches
    int nlanes, lane_begin[MPQE_STEP_MAX_LANES + 1], lane_Lmax[MPQE_STEP_MAX_LANES];
    std::vector<TileRef> tfwd[MPQE_STEP_MAX_LANES][STEP_MAX_LEVELS], tbwd[MPQE_STEP_MAX_LANES][STEP_MAX_LEVELS];
    size_t o_tf[MPQE_STEP_MAX_LANES][STEP_MAX_LEVELS], o_tb[MPQE_STEP_MAX_LANES][STEP_MAX_LEVELS];
    std::vector<int> wref;            // per weight-gradient workgroup: source index
    std::vector<WSource> wsrc;
    std::vector<WBlock> wblock;       // per weight-gradient workgroup: (source, block), the lanes' blocks in lane order
    int wblock_begin[MPQE_STEP_MAX_LANES + 1];     // lane l: wblock[wblock_begin[l] .. wblock_begin[l+1])
    int wblocks_total, vblocks_total;
    std::vector<VSource> vsrc;
    std::vector<int> vblock;
    std::vector<RGroup> groups;
    std::vector<int> anchor_off;      // nb + 1 (rows of the anchor backward)
    int total_slabs, total_parts;
    bool whole_ksteps;                // every batch size is a multiple of the K-step (weight-gradient LD_FAST)
    // graph-block chain kernels (step_chain.h): one entry per workgroup, heaviest blocks first
    std::vector<ChainRef> crefs;      // the lanes' grids one after the other
    int cref_begin[MPQE_STEP_MAX_LANES + 1];
    std::vector<ChainOp> cops;
    std::vector<WtSlot> wt_slots;     // matrices with a transposed copy (those of the backward programmes)
    // batch-uniform node states: vector ops of the forward pre-pass / backward post-pass, rank-1 weight-gradient terms
    bool chain, uniform;
    std::vector<UOp> uops_f, uops_b;
    // split tail launch: the backward post-pass as one closure workgroup per batch (step_closure.h); empty: the vector-op form
    std::vector<ClBlock> closures;
    size_t o_closures;
    std::vector<Rank1> rank1;
    int nvec, ngran;
    size_t o_uopf, o_uopb, o_rank1, o_epoch, o_gran, o_VT, o_DG, o_runs;
    std::vector<char> image;      // the descriptor table as uploaded ([0, o_epoch) of the desc buffer)
    long long touch_M;
    // touch plan built inside the step (MPQE_STEP_BUILD_TOUCH; step_touch.h: tsort_block): sort workgroups, key widths,
    // the batch table in the descriptor image, the sort's buffers in the workspace; ts_blocks = 0: not in this plan
    int ts_blocks, ts_key_bits, ts_row_bits;
    int sort_na, sort_rank[STEP_XCDS_MAX];       // the XCDs the sort's workgroups are dealt to (rank, or -1)
    int pl_na, pl_rank[STEP_XCDS_MAX];           // the XCDs with at most one chain workgroup per CU (PrepArgs.plast)
    size_t o_tmeta, o_tsort;
    int blk_off[MPQE_STEP_MAX_BATCHES + 1];        // chain blocks before batch i (slots of block_terms)
    std::vector<ZMat> zmats;                       // relation matrices of the gradient that no source touches
    size_t o_zmats;
    size_t o_done_inc, o_done;        // merged launch: chain workgroups per `done` counter (table), the counters (hand-off state)
    std::vector<int> done_inc;
    DoneMeta dm;
    int tile_n;                       // columns per weight-gradient tile (64; chain form: 32 when tiles would be few)
    int post_na, post_rank[STEP_XCDS_MAX];      // merged launch: the XCDs the post roles are dealt to (rank, or -1)
    std::vector<int> whole_roots;     // layers whose ROOT gradient matrix is written whole inside the chain launch (direct
                                      // tiles / a rank-1-only op): the launch's zero fill must leave them alone
    size_t o_bterms;
    size_t o_cref, o_cops, o_wtslots, o_WT;
    // workspace offsets (bytes)
    size_t o_sd, o_wsrc, o_wblock, o_vsrc, o_vblock, o_groups, o_anchor, desc_total;     // descriptor buffer
    size_t o_H, o_GH, o_tpos, o_tneg, o_spos, o_sneg, o_terms, o_slabs, o_parts, o_Q, o_GQ, total;  // workspace
    // learned readouts (step_readout.h): input rows, hidden, output and their gradients, argmax, dense-layer workspace
    size_t o_rx, o_rh, o_ry, o_rgy, o_rgh, o_rgx, o_rlin, rlin_bytes;
    long long ro_rows;
    int ro_kin;
    bool ro_direct;
    // the learned readouts (MPQE_READOUT_MLP / _TARGETMLP / _CONCAT) on the chain form: the readout's two Linear layers are
    // levels L + 1, L + 2 of every batch, their parameters the `root` / `bias` of the virtual layers ro_layer, ro_layer + 1
    // (= num_layers, + 1; stored [out, in]: the transposed form of a root matrix; the first may be [D, n D]: column blocks)
    bool ro_chain;
    int ro_layer;
    long long level_stride;
};

void pick_chunks(long long count, int max_chunks, int *nch, int *ch, int rows = 512) {
    // ~512 rows (16 K-steps) per workgroup: long enough to amortise the pipeline fill and the
    // 16 KB slab store, short enough that the AIFB-sized step still yields ~500 workgroups
    // (`rows`: the planner halves it while the step's tiles would leave most CUs without one)
    long long n = (count + rows - 1) / rows;
    if (n < 1) n = 1;
    if (n > max_chunks) n = max_chunks;
    long long c = (count + n - 1) / n;
    c = (c + GT_BK - 1) / GT_BK * GT_BK;
    if (c < GT_BK) c = GT_BK;
    n = (count + c - 1) / c;
    if (n < 1) n = 1;
    *nch = (int)n;
    *ch = (int)c;
}

// `in` is sorted by descending K length. The first STEP_CUS * STEP_RESIDENT tiles start at once, block
// b on CU b % STEP_CUS: give each to the least-loaded CU that still has a free position; the rest
// follow in descending order and are picked up by whichever CU drains first.
void place_tiles(const std::vector<TileRef> &in, const std::vector<int> &steps, std::vector<TileRef> &out) {
    const size_t n = in.size();
    const size_t first = n < (size_t)STEP_CUS * STEP_RESIDENT ? n : (size_t)STEP_CUS * STEP_RESIDENT;
    out.assign(in.begin(), in.end());
    int load[STEP_CUS] = {0}, used[STEP_CUS] = {0}, cap[STEP_CUS];
    for (int c = 0; c < STEP_CUS; ++c) cap[c] = (int)(first / STEP_CUS) + ((size_t)c < first % STEP_CUS);
    for (size_t t = 0; t < first; ++t) {
        int best = -1;
        for (int c = 0; c < STEP_CUS; ++c)
            if (used[c] < cap[c] && (best < 0 || load[c] < load[best])) best = c;
        out[(size_t)used[best] * STEP_CUS + best] = in[t];
        used[best]++;
        load[best] += steps[t];
    }
}

int make_plan(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb, const mpqe_step_lanes_t *lanes,
              bool chain, HostPlan *hp) {
    if (!P || !B || nb <= 0 || nb > MPQE_STEP_MAX_BATCHES) return MPQE_ERR_INVALID_ARG;
    hp->chain = chain;
    hp->nlanes = 1;
    hp->lane_begin[0] = 0;
    hp->lane_begin[1] = nb;
    // (the chain form is ONE launch per step on the caller's stream: a lane split only re-orders the batches)
    if (!chain && lanes && lanes->num_lanes > 1) {
        if (lanes->num_lanes > MPQE_STEP_MAX_LANES) return MPQE_ERR_INVALID_ARG;
        hp->nlanes = lanes->num_lanes;
        for (int l = 0; l <= hp->nlanes; ++l) hp->lane_begin[l] = lanes->batch_begin[l];
        if (hp->lane_begin[0] != 0 || hp->lane_begin[hp->nlanes] != nb) return MPQE_ERR_INVALID_ARG;
        for (int l = 0; l < hp->nlanes; ++l)
            if (hp->lane_begin[l + 1] <= hp->lane_begin[l]) return MPQE_ERR_INVALID_ARG;     // no empty lane
        if (!lanes->fork_event) return MPQE_ERR_INVALID_ARG;
        for (int l = 1; l < hp->nlanes; ++l)
            if (!lanes->aux_stream[l] || !lanes->join_event[l]) return MPQE_ERR_INVALID_ARG;
    }
    if (P->dim <= 0 || P->dim > 64 * STEP_MAX_COLS_PER_LANE) return MPQE_ERR_UNSUPPORTED;
    if (P->num_layers <= 0 || P->num_layers > MPQE_STEP_MAX_LAYERS) return MPQE_ERR_UNSUPPORTED;
    if (P->num_modes <= 0 || P->num_modes > MPQE_STEP_MAX_MODES) return MPQE_ERR_UNSUPPORTED;
    if (P->readout < 0 || P->readout > MPQE_READOUT_CONCAT) return MPQE_ERR_INVALID_ARG;
    const int D = P->dim;
    const bool ro = chain && P->readout >= MPQE_READOUT_MLP && P->readout <= MPQE_READOUT_CONCAT;
    const bool ro_pairs = ro && P->readout == MPQE_READOUT_TARGETMLP;       // rows [target | node] of the non-target nodes
    const bool ro_cat = ro && P->readout == MPQE_READOUT_CONCAT;            // rows [H_1 | .. | H_L] of every node
    const int ro_blocks = ro_pairs ? 2 : (ro_cat ? P->num_layers : 1);      // D x D column blocks of the first Linear layer
    const int VL0 = P->num_layers, ROL = ro ? 2 : 0;
    if (chain && P->readout >= MPQE_READOUT_CALLER && !ro) return MPQE_ERR_UNSUPPORTED;
    if (ro && P->num_layers + 2 > MPQE_STEP_MAX_LAYERS) return MPQE_ERR_UNSUPPORTED;
    hp->ro_chain = ro;
    hp->ro_layer = VL0;
    StepDev &sd = hp->sd;
    memset(&sd, 0, sizeof(sd));
    sd.nb = nb;
    sd.D = D;
    sd.num_layers = P->num_layers;
    sd.readout = P->readout;
    long long rows = 0, graphs = 0, anchors = 0;
    hp->Lmax = 0;
    hp->whole_ksteps = true;
    hp->anchor_off.assign(nb + 1, 0);
    for (int i = 0; i < nb; ++i) {
        const mpqe_step_batch_t &b = B[i];
        if (b.query_type < 0 || b.query_type >= MPQE_Q_COUNT || b.batch_size <= 0) return MPQE_ERR_INVALID_ARG;
        if (b.num_passes <= 0 || b.num_passes > P->num_layers) return MPQE_ERR_INVALID_ARG;
        const TemplateDesc &t = kTemplates[b.query_type];
        BatchDev &d = sd.b[i];
        d.tp.N = t.N;
        d.tp.E = t.E;
        for (int e = 0; e < 3; ++e) {
            d.tp.src[e] = e < t.E ? t.src[e] : 0;
            d.tp.dst[e] = e < t.E ? t.dst[e] : 0;
            d.tp.rel[e] = e < t.E ? b.edge_type[e] : 0;
            if (e < t.E && (b.edge_type[e] < 0 || b.edge_type[e] >= P->num_relations)) return MPQE_ERR_INVALID_ARG;
        }
        d.A = t.A;
        d.V = t.V;
        d.L = b.num_passes;
        d.B = b.batch_size;
        for (int k = 0; k < 3; ++k) d.var_id[k] = k < t.V ? b.var_ids[k] : 0;
        for (int a = 0; a < 3; ++a) {
            d.anchor_tab[a] = a < t.A ? b.anchor_mode[a] : 0;
            if (a < t.A && (b.anchor_mode[a] < 0 || b.anchor_mode[a] >= P->num_modes)) return MPQE_ERR_INVALID_ARG;
        }
        if (b.target_mode < 0 || b.target_mode >= P->num_modes) return MPQE_ERR_INVALID_ARG;
        d.target_tab = b.target_mode;
        d.row_off = rows;
        d.g_off = graphs;
        d.anchor_off = anchors;
        d.weight = b.weight;
        hp->anchor_off[i] = (int)anchors;
        rows += (long long)d.B * t.N;
        graphs += d.B;
        anchors += (long long)d.B * t.A;
        if (d.L > hp->Lmax) hp->Lmax = d.L;
        if (d.B % GT_BK != 0) hp->whole_ksteps = false;
        // liveness, from the readout backwards: H[p][n] matters iff n itself or a destination of one
        // of its out-edges matters at p+1 (reference RGCNConv: out_i = sum_j x_j W_r + x_i root)
        const unsigned all = (1u << t.N) - 1u;
        const bool prune = !(P->flags & MPQE_STEP_NO_PRUNE);
        d.live[d.L] = (prune && P->readout == MPQE_READOUT_TM) ? (1u << t.A) : all;
        for (int p = d.L - 1; p >= 0; --p) {
            unsigned m = d.live[p + 1];
            for (int e = 0; e < t.E; ++e)
                if ((d.live[p + 1] >> t.dst[e]) & 1u) m |= 1u << t.src[e];
            d.live[p] = prune ? m : all;
        }
        if (ro) {       // the readout's hidden and output rows of every node slot (ReLU bits: level L + 1 <= CH_MASK_LEVELS)
            if (d.L + 1 > CH_MASK_LEVELS) return MPQE_ERR_UNSUPPORTED;
            if (ro_cat && d.L != P->num_layers) return MPQE_ERR_INVALID_ARG;     // (model.py:441-446: one input block per layer)
            d.live[d.L + 1] = d.live[d.L + 2] = ro_pairs ? all & ~(1u << t.A) : all;    // (targetmlp: the target has no row)
        }
    }
    if (rows >= (1ll << 30)) return MPQE_ERR_UNSUPPORTED;
    hp->anchor_off[nb] = (int)anchors;
    sd.rows_total = rows;
    hp->level_stride = rows * D;
    sd.graphs_total = graphs;
    const int ct = (D + GT_BN - 1) / GT_BN;
    const int spb = (D + GT_BK - 1) / GT_BK;
    for (int l = 0; l < hp->nlanes; ++l) {
        hp->lane_Lmax[l] = 0;
        for (int i = hp->lane_begin[l]; i < hp->lane_begin[l + 1]; ++i)
            if (sd.b[i].L > hp->lane_Lmax[l]) hp->lane_Lmax[l] = sd.b[i].L;
        for (int p = 0; p < STEP_MAX_LEVELS && !chain; ++p)     // (the chain form has no per-level launches)
            for (int dir = 0; dir < 2; ++dir) {
                std::vector<TileGroup> g;
                for (int i = hp->lane_begin[l]; i < hp->lane_begin[l + 1]; ++i) {
                    if (sd.b[i].L <= p) continue;
                    const TmplArgs &tp = sd.b[i].tp;
                    const unsigned lin = sd.b[i].live[p], lout = sd.b[i].live[p + 1];
                    for (int n = 0; n < tp.N; ++n) {
                        // forward: H[p+1][n] from the edges INTO n (+ self); backward-x: gH[p][n] from the
                        // live destinations of the edges OUT of n (+ self if live)
                        if (!(((dir ? lin : lout) >> n) & 1u)) continue;
                        int blocks = dir ? (int)((lout >> n) & 1u) : 1;
                        for (int e = 0; e < tp.E; ++e) {
                            if (!dir) blocks += tp.dst[e] == n;
                            else blocks += tp.src[e] == n && ((lout >> tp.dst[e]) & 1u);
                        }
                        g.push_back(TileGroup{i, n, 0, blocks * spb});
                    }
                }
                std::stable_sort(g.begin(), g.end(),
                                 [](const TileGroup &a, const TileGroup &b) { return a.steps > b.steps; });
                std::vector<TileRef> sorted;
                std::vector<int> steps;
                for (size_t k = 0; k < g.size(); ++k) {
                    const int cnt = ((sd.b[g[k].batch].B + GT_BM - 1) / GT_BM) * ct;
                    for (int r = 0; r < cnt; ++r) {
                        sorted.push_back(TileRef{(short)g[k].batch, (short)g[k].node, r});
                        steps.push_back(g[k].steps);
                    }
                }
                place_tiles(sorted, steps, dir ? hp->tbwd[l][p] : hp->tfwd[l][p]);
            }
    }

    // unique layer buffers (shared layers alias one parameter set -> one gradient buffer)
    int uid[MPQE_STEP_MAX_LAYERS];
    for (int l = 0; l < MPQE_STEP_MAX_LAYERS; ++l) uid[l] = l;       // (the readout's virtual layers: themselves)
    for (int l = 0; l < P->num_layers; ++l) {
        uid[l] = l;
        for (int m = 0; m < l; ++m)
            if (P->basis[m] == P->basis[l]) {
                uid[l] = uid[m];
                break;
            }
    }
    // ---- batch-uniform node states (see UOp): uni[i][p] = node slots of batch i that are one vector per batch at level p
    // (concat reads every node's state after EVERY layer: no state is left to the pre-pass as a vector)
    bool uniform = chain && !(P->flags & MPQE_STEP_NO_UNIFORM) && !ro_cat;
    unsigned uni[MPQE_STEP_MAX_BATCHES][MPQE_STEP_MAX_LAYERS + 1];
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool left_over = false;
        for (int i = 0; i < nb; ++i) {
            const BatchDev &d = sd.b[i];
            const TmplArgs &tp = d.tp;
            uni[i][0] = uniform ? (((1u << tp.N) - 1u) & ~((1u << d.A) - 1u)) : 0u;
            for (int p = 0; p < d.L; ++p) {
                unsigned m = uni[i][p];
                for (int e = 0; e < tp.E; ++e)
                    if (!((uni[i][p] >> tp.src[e]) & 1u)) m &= ~(1u << tp.dst[e]);
                uni[i][p + 1] = m;
            }
            if (ro) {
                // (a node slot no anchor has reached after the last pass -- fewer passes than the query's diameter -- has no
                // rows in H[L], which the readout's weight gradient reads: such a step keeps every state per graph)
                left_over = left_over || (uni[i][d.L] & d.live[d.L]) != 0u;
                uni[i][d.L + 1] = uni[i][d.L + 2] = 0u;
            }
        }
        if (!left_over) break;
        uniform = false;
    }
    hp->uniform = uniform;
    // vector table ids: (kind, batch, level, node slot) -> row of VT; granule slots only for vectors another
    // workgroup of the producing launch reads
    enum { V_UV = 0, V_CV = 1, V_SV = 2 };
    struct VecInfo {
        int kind, batch, level, node;
    };
    std::vector<VecInfo> vinfo;
    std::vector<int> gran_of;
    std::unordered_map<long long, int> vec_of;
    auto vec = [&](int kind, int i, int p, int n) -> int {
        const long long key = (((long long)kind * MPQE_STEP_MAX_BATCHES + i) * (MPQE_STEP_MAX_LAYERS + 1) + p) * 4 + n;
        auto it = vec_of.find(key);
        if (it != vec_of.end()) return it->second;
        const int id = (int)vinfo.size();
        vinfo.push_back(VecInfo{kind, i, p, n});
        gran_of.push_back(-1);
        vec_of[key] = id;
        return id;
    };
    int ngran = 0;
    auto gran = [&](int v) -> int {
        if (gran_of[v] < 0) gran_of[v] = ngran++;
        return gran_of[v];
    };
    // reference model.py:435-441; levels L, L + 1 (chain form with a learned readout): its two Linear layers
    auto layer_of = [&](int i, int p) {
        return p < sd.b[i].L - 1 ? p : (p < sd.b[i].L ? P->num_layers - 1 : VL0 + (p - sd.b[i].L));
    };

    // weight-gradient sources, ordered by (unique layer, relation | root) so every reduction group
    // owns a contiguous slab range. A source whose input state is batch-uniform is a rank-1 term u (x) colsum
    // of the reduction instead of a K = batch tile.
    struct Key {
        int layer;
        long long rel;     // relation id, or -1 for root
        int batch, level, slot;
        int xo = -1, go = -1;      // >= 0: node slots of the x / g rows given (not derived from `slot`)
        int glev = -1;             // >= 0: level of the gH rows (else: level + 1)
    };
    struct R1Key {
        int layer;
        long long rel;
        Rank1 t;
    };
    std::vector<Key> keys;
    std::vector<R1Key> r1keys;
    std::vector<char> sv_needed;       // per vector id: somebody reads this column-sum vector
    std::vector<int> copy_vecs;        // UV vectors of level 0 (mode rows) the rank-1 terms read
    auto need_sv = [&](int v) {
        if (sv_needed.size() <= (size_t)v) sv_needed.resize(v + 1, 0);
        sv_needed[v] = 1;
    };
    for (int i = 0; i < nb; ++i)
        for (int p = 0; p < sd.b[i].L; ++p) {
            const int li = uid[layer_of(i, p)];
            const unsigned lout = sd.b[i].live[p + 1];
            const TmplArgs &tp = sd.b[i].tp;
            auto add = [&](int slot, int s, int dnode, long long rel) {
                if (!((uni[i][p] >> s) & 1u)) {
                    keys.push_back(Key{li, rel, i, p, slot});
                    return;
                }
                const size_t before = vinfo.size();
                const int u = vec(V_UV, i, p, s), v = vec(V_SV, i, p + 1, dnode);
                if (p == 0 && (size_t)u >= before) copy_vecs.push_back(u);      // (first use of this mode row's copy)
                need_sv(v);
                r1keys.push_back(R1Key{li, rel, Rank1{u, v}});
            };
            for (int z = 0; z < tp.E; ++z)
                if ((lout >> tp.dst[z]) & 1u) add(z, tp.src[z], tp.dst[z], tp.rel[z]);
            for (int n = 0; n < tp.N; ++n)          // root term: one source per live node slot
                if ((lout >> n) & 1u) add(tp.E + n, n, n, -1);
        }
    // the readout's Linear layers: a root-like source per row-bearing node slot and layer (x: the layer's input rows of
    // the slot, g: its output rows' gradients). targetmlp's first layer [D, 2 D] is two column blocks: `rel` -1 = the block
    // that multiplies the target's row (x of slot A for every node), -2 = the node's own
    for (int i = 0; i < nb && ro; ++i)
        for (int r = 0; r < ROL; ++r)
            for (int n = 0; n < sd.b[i].tp.N; ++n) {
                if (!((sd.b[i].live[sd.b[i].L + 1] >> n) & 1u)) continue;
                Key k{VL0 + r, -1, i, sd.b[i].L + r, sd.b[i].tp.E + n};
                if (ro_pairs && r == 0) {       // (operands as the tile takes them: x rows = gH of slot n, g rows = H of slot A)
                    Key a = k;
                    a.xo = n;
                    a.go = sd.b[i].A;
                    keys.push_back(a);
                    k.rel = -2;
                }
                if (ro_cat && r == 0) {         // column block l - 1: the hidden rows' gradients (gH[L + 1]) x the states H[l]
                    for (int l = 1; l <= sd.b[i].L; ++l) {
                        Key c = k;
                        c.rel = -l;
                        c.level = l;
                        c.glev = sd.b[i].L + 1;
                        keys.push_back(c);
                    }
                    continue;
                }
                keys.push_back(k);
            }
    auto key_less = [](int la, long long ra, int lb, long long rb) { return la != lb ? la < lb : ra < rb; };
    std::stable_sort(keys.begin(), keys.end(),
                     [&](const Key &a, const Key &b) { return key_less(a.layer, a.rel, b.layer, b.rel); });
    std::stable_sort(r1keys.begin(), r1keys.end(),
                     [&](const R1Key &a, const R1Key &b) { return key_less(a.layer, a.rel, b.layer, b.rel); });
    // Weight-gradient tiles of the chain form: 64 x 64 outputs per workgroup. (64 x 32 -- twice as many tiles at half the
    // MFMA time each, no K split, so no extra slab -- is built in, mpqe_debug_option TILE_N = 32, and was measured on the AIFB step:
    // the tiles end at 11.0 us instead of 14.8, but 320 of them next to the post-pass' 100 vector-op workgroups slow ITS
    // latency chain from 15 to 18.9 us, and the launch from 19.7 to 23.5.)
    // (Round 4, with the post-pass on two XCDs of its own: 64 x 32 while all of them are resident at once on the other six --
    // 272 for the AIFB step: 67.8 -> 67.2 us per step, three runs each on one box.)
    int tile_n = GT_BN;
    if (chain && D % 64 == 0) {
        long long n32 = 0;
        for (size_t k = 0; k < keys.size(); ++k) {
            int nch1, ch1;
            pick_chunks(sd.b[keys[k].batch].B, 32, &nch1, &ch1, 512);
            n32 += (long long)nch1 * (D / 64) * (D / 32);
        }
        long long blk_all = 0;
        for (int i = 0; i < nb; ++i) blk_all += (sd.b[i].B + CH_GB - 1) / CH_GB;
        // (not where the tiles ride in the chain launch -- the merged form of small steps, measured with 64 x 64 only)
        const bool rides = hp->nlanes == 1 && !(P->flags & MPQE_STEP_SPLIT_TAIL) &&
                           ((P->flags & MPQE_STEP_MERGE_TAIL) || blk_all <= STEP_CUS + STEP_CUS / 8);
        const int forced = mpqe_dbg_value("TILE_N", 0);           // (timing experiments: 32 / 64)
        if (forced == 32 || (forced != 64 && !rides && n32 <= 6 * 2 * (STEP_CUS / STEP_XCDS))) tile_n = 32;
    }
    hp->tile_n = tile_n;
    const int wct = (D + tile_n - 1) / tile_n;            // column tiles of a weight gradient
    const int tiles = wct * ((D + GT_BM - 1) / GT_BM);
    // Balance: with one K-chunk per source the step has (sources x tiles) workgroups; a few more than there are
    // CUs (264 for the AIFB mix) means a handful of CUs run two whole tiles and the launch lasts twice a tile.
    // Then the surplus is taken out of a few ROOT sources (they go through the reduction anyway), cut into
    // four K-chunks: their short workgroups ride along on CUs that also hold one whole tile.
    // K-chunk length: with few sources (the chain form's uniform node states leave 34 of the AIFB mix's 66) whole-batch
    // chunks would put a 14 us tile on half of the CUs and nothing on the rest: halve the chunks until the launch has
    // a workgroup for most CUs (the extra slabs go through the reduction)
    int chunk_rows = 512;
    {
        const int dbg = mpqe_dbg_value("CHUNK_ROWS", 0);        // (timing experiments)
        auto blocks_at = [&](int rows) {
            long long nblk = 0;
            for (size_t k = 0; k < keys.size(); ++k) {
                int nch1, ch1;
                pick_chunks(sd.b[keys[k].batch].B, 32, &nch1, &ch1, rows);
                nblk += (long long)nch1 * tiles;
            }
            return nblk;
        };
        if (dbg >= GT_BK) chunk_rows = dbg / GT_BK * GT_BK;
        (void)blocks_at;
        // (measured on the AIFB mix, 136 whole-batch tiles of 13.8 us: 272 half-batch tiles take 8.2 us each but 16 CUs
        // get two of them and the launch needs the reduction for every matrix: 22.6 us against 18.3. Kept at 512.)
    }
    std::vector<char> split4(keys.size(), 0);
    {
        long long blocks1 = 0;
        for (size_t k = 0; k < keys.size(); ++k) {
            int nch1, ch1;
            pick_chunks(sd.b[keys[k].batch].B, 32, &nch1, &ch1, chunk_rows);
            blocks1 += (long long)nch1 * tiles;
        }
        long long excess = blocks1 - STEP_CUS;
        if (excess > 0 && excess <= STEP_CUS / 4)
            for (size_t k = keys.size(); k-- > 0 && excess > 0;) {
                const int Bk = sd.b[keys[k].batch].B;
                int nch1, ch1;
                pick_chunks(Bk, 32, &nch1, &ch1, chunk_rows);
                if (keys[k].rel >= 0 || nch1 != 1 || Bk < 4 * 4 * GT_BK || Bk % (4 * GT_BK) != 0) continue;
                split4[k] = 1;
                excess -= tiles;
            }
    }
    std::vector<RGroup> r1_only;
    int slab = 0, block = 0;
    hp->wsrc.clear();
    hp->wblock.clear();
    hp->groups.clear();
    for (size_t k = 0; k < keys.size(); ++k) {
        const Key &key = keys[k];
        const BatchDev &d = sd.b[key.batch];
        WSource s;
        s.batch = key.batch;
        s.level = key.level;
        s.slot = key.slot;
        s.relu = 0;      // gH is stored as a pre-activation gradient (masked by its producer)
        pick_chunks(d.B, 32, &s.nch, &s.ch, chunk_rows);
        if (split4[k]) {
            s.nch = 4;
            s.ch = d.B / 4;
        }
        s.slab_start = slab;
        s.block_start = block;
        s.direct = -1;
        s.pad = (ro && key.layer >= VL0) ? 1 : 0;       // (nn.Linear's [out, in]: the tile's operands change places)
        if (key.xo >= 0) s.pad |= 2 | (key.xo << 4) | (key.go << 8);
        if (key.glev >= 0) s.pad |= 4 | (key.glev << 12);
        s.rel = key.rel;
        hp->wsrc.push_back(s);

        slab += s.nch;
        block += s.nch * tiles;
    }
    hp->wblocks_total = block;
    hp->total_slabs = slab;
    // reduction groups of the gradient matrices: per (unique layer, relation | root) the slabs of its tile
    // sources (contiguous: the sources are sorted) and its rank-1 terms. A matrix with ONE contribution that is a
    // single-chunk tile source (most relation matrices: a relation rarely occurs in two batches of a step) needs no
    // slab and no reduction: its tiles write straight into the gradient (deterministic: one writer per element).
    {
        hp->rank1.clear();
        hp->whole_roots.clear();
        for (size_t k = 0; k < r1keys.size(); ++k) hp->rank1.push_back(r1keys[k].t);
        r1_only.clear();
        std::vector<char> written((size_t)P->num_layers * (size_t)P->num_relations, 0);
        size_t ks = 0, kr = 0;
        while (ks < keys.size() || kr < r1keys.size()) {
            int layer;
            long long rel;
            if (kr >= r1keys.size() || (ks < keys.size() && !key_less(r1keys[kr].layer, r1keys[kr].rel, keys[ks].layer,
                                                                        keys[ks].rel))) {
                layer = keys[ks].layer;
                rel = keys[ks].rel;
            } else {
                layer = r1keys[kr].layer;
                rel = r1keys[kr].rel;
            }
            RGroup g;
            g.kind = rel < 0 ? 1 : 0;
            // (targetmlp's first Linear layer [D, 2 D]: its two column blocks are groups of their own, written with the row
            // length 2 D -- kinds 4 / 5)
            const bool wide_g = ro && layer == VL0 && ro_blocks > 1;
            if (wide_g) g.kind = 4;
            g.layer = layer;
            g.row = rel < 0 ? 0 : rel;
            if (wide_g) g.row = (-1 - rel) | ((long long)ro_blocks << 8);      // column block | blocks per row
            g.start = ks < keys.size() ? hp->wsrc[ks].slab_start : 0;
            g.count = 0;
            g.r1_start = (int)kr;
            g.r1_count = 0;
            const size_t first_src = ks;
            int nsrc = 0;
            while (ks < keys.size() && keys[ks].layer == layer && keys[ks].rel == rel) {
                g.count += hp->wsrc[ks].nch;
                ++nsrc;
                ++ks;
            }
            while (kr < r1keys.size() && r1keys[kr].layer == layer && r1keys[kr].rel == rel) {
                ++g.r1_count;
                ++kr;
            }
            if (rel >= 0) written[(size_t)layer * P->num_relations + rel] = 1;
            if (nsrc == 1 && g.count == 1 && g.r1_count == 0 && g.kind <= 1) {
                hp->wsrc[first_src].direct = layer;
                if (rel < 0) hp->whole_roots.push_back(layer);
            } else if (nsrc == 0 && g.r1_count <= UOP_MAX_TERMS) {
                r1_only.push_back(g);    // written by the post-pass (UOP_R1)
                if (rel < 0) hp->whole_roots.push_back(layer);
            } else {
                hp->groups.push_back(g);
            }
        }
        // every other relation matrix of every (unique) layer is untouched
        hp->zmats.clear();
        for (int l = 0; l < P->num_layers; ++l) {
            if (uid[l] != l) continue;
            for (long long r = 0; r < P->num_relations; ++r)
                if (!written[(size_t)l * P->num_relations + r]) hp->zmats.push_back(ZMat{l, 0, r});
        }
    }
    hp->done_inc.clear();
    for (int i = 0; i < nb; ++i) {                  // `done` counters: one per DONE_GRAPHS graphs of a batch
        hp->dm.base[i] = (int)hp->done_inc.size();
        const int nblk = (sd.b[i].B + CH_GB - 1) / CH_GB, per = DONE_GRAPHS / CH_GB;
        for (int k = 0; k < nblk; k += per) hp->done_inc.push_back(nblk - k < per ? nblk - k : per);
    }
    for (int i = nb; i <= MPQE_STEP_MAX_BATCHES; ++i) hp->dm.base[i] = (int)hp->done_inc.size();
    for (int l = 0; l < hp->nlanes; ++l) {          // block table, grouped by stream lane (a lane launches its own)
        hp->wblock_begin[l] = (int)hp->wblock.size();
        for (int pass = 0; pass < 2; ++pass)        // whole-batch chunks first, the short ride-along chunks last
            for (size_t k = 0; k < hp->wsrc.size(); ++k) {
                const WSource &ws = hp->wsrc[k];
                if (ws.batch < hp->lane_begin[l] || ws.batch >= hp->lane_begin[l + 1]) continue;
                const bool is_short = ws.nch > 1 && ws.ch < sd.b[ws.batch].B && ws.ch <= 4 * 4 * GT_BK;
                if ((int)is_short != pass) continue;
                const BatchDev &bd = sd.b[ws.batch];
                const bool is_root = ws.slot >= bd.tp.E;
                for (int q = 0; q < ws.nch * tiles; ++q) {
                    const int c = q / tiles, tile = q - c * tiles;
                    WBlock wkb;
                    wkb.x_off = (long long)ws.level * hp->level_stride + bd.row_off * D;
                    wkb.g_off = (long long)((ws.pad & 4) ? (ws.pad >> 12) & 15 : ws.level + 1) * hp->level_stride + bd.row_off * D;
                    wkb.slab_off = (long long)(ws.slab_start + c) * D * D;
                    wkb.rel = ws.rel;
                    wkb.xs = bd.tp.N;
                    wkb.xo = is_root ? ws.slot - bd.tp.E : bd.tp.src[ws.slot];
                    wkb.go = is_root ? ws.slot - bd.tp.E : bd.tp.dst[ws.slot];
                    if (ws.pad & 2) {           // (given, as the tile takes its operands)
                        wkb.xo = (ws.pad >> 4) & 15;
                        wkb.go = (ws.pad >> 8) & 15;
                    }
                    wkb.q0 = c * ws.ch;
                    wkb.q1 = wkb.q0 + ws.ch < bd.B ? wkb.q0 + ws.ch : bd.B;
                    wkb.i0 = (tile / wct) * GT_BM;
                    wkb.j0 = (tile % wct) * tile_n;
                    wkb.direct = ws.direct;
                    wkb.batch = ws.batch;
                    wkb.pad = ws.pad & 1;
                    wkb.d0 = hp->dm.base[ws.batch] + wkb.q0 / DONE_GRAPHS;
                    wkb.dn = (wkb.q1 - 1) / DONE_GRAPHS - wkb.q0 / DONE_GRAPHS + 1;
                    hp->wblock.push_back(wkb);
                }
            }
    }
    hp->wblock_begin[hp->nlanes] = (int)hp->wblock.size();

    hp->vsrc.clear();
    hp->vblock.clear();
    hp->vblocks_total = 0;
    hp->uops_f.clear();
    hp->uops_b.clear();
    // part_row[i][p][n]: first row in `parts` of the column sums of gH[p][n] of batch i (-1: none)
    int part_row[MPQE_STEP_MAX_BATCHES][MPQE_STEP_MAX_LAYERS + 1][4];
    for (int i = 0; i < MPQE_STEP_MAX_BATCHES; ++i)
        for (int q = 0; q <= MPQE_STEP_MAX_LAYERS; ++q)
            for (int n = 0; n < 4; ++n) part_row[i][q][n] = -1;
    if (!chain) {
        // vector partial sources: bias per (unique layer) and variable rows per mode id
        struct VKey {
            int kind, layer;
            long long row;
            int batch, lk;
        };
        std::vector<VKey> vk;
        for (int i = 0; i < nb; ++i) {
            for (int p = 0; p < sd.b[i].L; ++p)
                vk.push_back(VKey{0, uid[p < sd.b[i].L - 1 ? p : P->num_layers - 1], 0, i, p});
            for (int k = 0; k < sd.b[i].V; ++k)
                if ((sd.b[i].live[0] >> (sd.b[i].A + k)) & 1u) vk.push_back(VKey{1, 0, sd.b[i].var_id[k], i, k});
        }
        std::stable_sort(vk.begin(), vk.end(), [](const VKey &a, const VKey &b) {
            if (a.kind != b.kind) return a.kind < b.kind;
            if (a.layer != b.layer) return a.layer < b.layer;
            return a.row < b.row;
        });
        const int cchunks = (D + 63) / 64;
        int part = 0, vblock = 0;
        hp->vsrc.clear();
        hp->vblock.clear();
        for (size_t k = 0; k < vk.size(); ++k) {
            const VKey &key = vk[k];
            const BatchDev &d = sd.b[key.batch];
            VSource s;
            s.kind = key.kind;
            s.batch = key.batch;
            s.level_or_k = key.lk;
            s.relu = 0;
            s.nblk = (d.B + CH_GB - 1) / CH_GB;
            s.part_start = part;
            s.block_start = vblock;
            s.pad = 0;
            hp->vsrc.push_back(s);
            for (int q = 0; q < s.nblk * cchunks; ++q) hp->vblock.push_back((int)hp->vsrc.size() - 1);
            if (k == 0 || vk[k - 1].kind != key.kind || vk[k - 1].layer != key.layer || vk[k - 1].row != key.row) {
                RGroup g;
                g.kind = key.kind == 0 ? 2 : 3;
                g.layer = key.layer;
                g.row = key.row;
                g.start = part;
                g.count = 0;
                g.r1_start = g.r1_count = 0;
                hp->groups.push_back(g);
            }
            hp->groups.back().count += s.nblk;
            part += s.nblk;
            vblock += s.nblk * cchunks;
        }
        hp->vblocks_total = vblock;
        hp->total_parts = part;
    } else {
        // Chain form: one row of `parts` per (batch, level >= 1, live node slot) and chain block -- the node's gradient
        // rows summed over the block's graphs -- written by the chain kernel; a node slot that is batch-uniform below
        // level L has ONE row instead, written by the backward post-pass (its column sum IS what the post-pass
        // computes). Rows of one reduction group (bias of a unique layer; a mode_embeddings row) are contiguous.
        struct VKey {
            int kind, layer;
            long long row;
            int batch, level, node;
        };
        std::vector<VKey> vk;
        for (int i = 0; i < nb; ++i) {
            const BatchDev &d = sd.b[i];
            for (int p = 1; p <= d.L + ROL; ++p)
                for (int n = 0; n < d.tp.N; ++n)
                    if ((d.live[p] >> n) & 1u) vk.push_back(VKey{0, uid[layer_of(i, p - 1)], 0, i, p, n});
            for (int k = 0; k < d.V; ++k)
                if ((d.live[0] >> (d.A + k)) & 1u) vk.push_back(VKey{1, 0, d.var_id[k], i, 0, d.A + k});
        }
        std::stable_sort(vk.begin(), vk.end(), [](const VKey &a, const VKey &b) {
            if (a.kind != b.kind) return a.kind < b.kind;
            if (a.layer != b.layer) return a.layer < b.layer;
            return a.row < b.row;
        });
        int part = 0;
        for (size_t k = 0; k < vk.size(); ++k) {
            const VKey &key = vk[k];
            const BatchDev &d = sd.b[key.batch];
            const bool is_u = (uni[key.batch][key.level] >> key.node) & 1u;
            const int rows = (is_u && key.level < d.L) ? 1 : (d.B + CH_GB - 1) / CH_GB;
            part_row[key.batch][key.level][key.node] = part;
            if (k == 0 || vk[k - 1].kind != key.kind || vk[k - 1].layer != key.layer || vk[k - 1].row != key.row) {
                RGroup g;
                g.kind = key.kind == 0 ? 2 : 3;
                g.layer = key.layer;
                g.row = key.row;
                g.start = part;
                g.count = 0;
                g.r1_start = g.r1_count = 0;
                hp->groups.push_back(g);
            }
            hp->groups.back().count += rows;
            part += rows;
        }
        hp->total_parts = part;
        for (int i = 0; i < nb; ++i)
            for (int n = 0; n < 4; ++n) {
                BatchDev &d = sd.b[i];
                const bool liveL = n < d.tp.N && ((d.live[d.L + ROL] >> n) & 1u);
                d.lpart[n] = liveL ? part_row[i][d.L + ROL][n] : -1;      // (a learned readout: its output rows' gradients)
                d.uvL[n] = liveL && ((uni[i][d.L] >> n) & 1u) ? gran(vec(V_UV, i, d.L, n)) : -1;     // (its granule slot)
            }
        if (uniform) {
            // ---- forward pre-pass, level by level (a level's inputs are the outputs of the level before)
            for (int p = 0; p < hp->Lmax; ++p)
                for (int i = 0; i < nb; ++i) {
                    const BatchDev &d = sd.b[i];
                    if (d.L <= p) continue;
                    const TmplArgs &tp = d.tp;
                    const int li = layer_of(i, p);
                    for (int n = 0; n < tp.N; ++n) {
                        if (!((d.live[p + 1] >> n) & 1u)) continue;
                        const bool nu = !((uni[i][p + 1] >> n) & 1u);
                        UOp op;
                        memset(&op, 0, sizeof(op));
                        op.kind = UOP_FWD;
                        op.out_gran = op.out_part = op.mask_vec = -1;
                        auto add_in = [&](int src, int mat) {
                            const int t = op.nterms++;
                            op.layer[t] = li;
                            op.mat[t] = mat;
                            if (p == 0) {           // a variable row of x0 = a mode_embeddings row
                                op.in_kind[t] = 1;
                                op.in_vec[t] = (int)d.var_id[src - d.A];
                            } else {
                                op.in_kind[t] = 0;
                                op.in_vec[t] = vec(V_UV, i, p, src);
                                op.in_gran[t] = gran(op.in_vec[t]);
                            }
                        };
                        for (int e = 0; e < tp.E; ++e)
                            if (tp.dst[e] == n && ((uni[i][p] >> tp.src[e]) & 1u)) add_in(tp.src[e], (int)tp.rel[e]);
                        if ((uni[i][p] >> n) & 1u) add_in(n, -1);
                        if (nu && op.nterms == 0) continue;      // its constant is the layer's bias itself (ChainOp.aux = -1)
                        op.out_vec = vec(nu ? V_CV : V_UV, i, p + 1, n);
                        if (nu) (void)gran(op.out_vec);      // read by the chain workgroups of the same launch
                        op.bias_layer = li;
                        op.relu = (!nu && p < d.L - 1) ? 1 : 0;
                        hp->uops_f.push_back(op);
                    }
                }
            for (size_t k = 0; k < copy_vecs.size(); ++k) {      // mode rows the rank-1 weight-gradient terms read
                const VecInfo &vi = vinfo[copy_vecs[k]];
                UOp op;
                memset(&op, 0, sizeof(op));
                op.kind = UOP_COPY;
                op.out_vec = copy_vecs[k];
                op.out_gran = op.out_part = op.mask_vec = -1;
                op.mode_row = sd.b[vi.batch].var_id[vi.node - sd.b[vi.batch].A];
                hp->uops_f.push_back(op);
            }
            // ---- backward post-pass: the uniform nodes' gradient column sums, level L-1 down to 0
            std::vector<UOp> bwd;
            for (int p = hp->Lmax - 1; p >= 0; --p)
                for (int i = 0; i < nb; ++i) {
                    const BatchDev &d = sd.b[i];
                    if (d.L <= p) continue;
                    const TmplArgs &tp = d.tp;
                    const int li = layer_of(i, p);
                    for (int m = 0; m < tp.N; ++m) {
                        if (!((d.live[p] >> m) & 1u) || !((uni[i][p] >> m) & 1u)) continue;
                        UOp op;
                        memset(&op, 0, sizeof(op));
                        op.kind = UOP_BWD;
                        op.out_gran = -1;
                        op.out_vec = vec(V_SV, i, p, m);
                        op.out_part = part_row[i][p][m];
                        op.mask_vec = p >= 1 ? vec(V_UV, i, p, m) : -1;       // H[p] = ReLU(..) for 1 <= p <= L-1
                        auto add_in = [&](int dnode, int mat) {
                            const int t = op.nterms++;
                            op.layer[t] = li;
                            op.mat[t] = mat;
                            if (((uni[i][p + 1] >> dnode) & 1u) && p + 1 < d.L) {
                                op.in_kind[t] = 0;          // another op of this launch produces it: through its granules
                                op.in_vec[t] = vec(V_SV, i, p + 1, dnode);
                                op.in_gran[t] = gran(op.in_vec[t]);
                            } else {                        // a sum of the chain kernel's per-block rows: formed on the fly
                                op.in_kind[t] = 3;
                                op.in_vec[t] = part_row[i][p + 1][dnode];
                                op.in_gran[t] = (d.B + CH_GB - 1) / CH_GB;
                                op.wait_mask |= 1u << i;
                            }
                        };
                        for (int e = 0; e < tp.E; ++e)
                            if (tp.src[e] == m && ((d.live[p + 1] >> tp.dst[e]) & 1u)) add_in(tp.dst[e], (int)tp.rel[e]);
                        if ((d.live[p + 1] >> m) & 1u) add_in(m, -1);
                        bwd.push_back(op);
                    }
                }
            // the column sums somebody reads and no BWD op produces: sums of the chain kernel's per-block rows
            sv_needed.resize(vinfo.size(), 0);
            for (size_t v = 0; v < vinfo.size(); ++v) {
                const VecInfo &vi = vinfo[v];
                if (vi.kind != V_SV || !sv_needed[v]) continue;
                const bool is_u = (uni[vi.batch][vi.level] >> vi.node) & 1u;
                if (is_u && vi.level < sd.b[vi.batch].L) continue;
                UOp op;
                memset(&op, 0, sizeof(op));
                op.kind = UOP_RED;
                op.out_vec = (int)v;
                op.out_gran = op.out_part = op.mask_vec = -1;
                op.row0 = part_row[vi.batch][vi.level][vi.node];
                op.nrows = (sd.b[vi.batch].B + CH_GB - 1) / CH_GB;
                op.wait_mask = 1u << vi.batch;
                hp->uops_b.push_back(op);
            }
            hp->uops_b.insert(hp->uops_b.end(), bwd.begin(), bwd.end());
            // gradient matrices made of rank-1 terms only: u (x) v as soon as v (a column sum) exists
            for (size_t k = 0; k < r1_only.size(); ++k) {
                const RGroup &g = r1_only[k];
                UOp op;
                memset(&op, 0, sizeof(op));
                op.kind = UOP_R1;
                op.out_vec = op.out_gran = op.out_part = op.mask_vec = -1;
                op.r1_layer = g.layer;
                op.r1_rel = g.kind == 1 ? -1 : (int)g.row;
                for (int t = 0; t < g.r1_count; ++t) {
                    const Rank1 rk = hp->rank1[g.r1_start + t];
                    const VecInfo &vi = vinfo[rk.v];
                    op.u_vec[t] = rk.u;
                    if (((uni[vi.batch][vi.level] >> vi.node) & 1u) && vi.level < sd.b[vi.batch].L) {
                        op.in_kind[t] = 0;
                        op.in_vec[t] = rk.v;
                        op.in_gran[t] = gran(rk.v);
                    } else {
                        op.in_kind[t] = 3;
                        op.in_vec[t] = part_row[vi.batch][vi.level][vi.node];
                        op.in_gran[t] = (sd.b[vi.batch].B + CH_GB - 1) / CH_GB;
                        op.wait_mask |= 1u << vi.batch;
                    }
                    op.nterms++;
                }
                hp->uops_b.push_back(op);
            }
            for (size_t k = 0; k < hp->uops_f.size(); ++k)
                if (hp->uops_f[k].out_vec >= 0) hp->uops_f[k].out_gran = gran_of[hp->uops_f[k].out_vec];
            for (size_t k = 0; k < hp->uops_b.size(); ++k)          // (a rank-1 op writes a matrix, not a vector: out_vec = -1)
                if (hp->uops_b[k].out_vec >= 0) hp->uops_b[k].out_gran = gran_of[hp->uops_b[k].out_vec];
            // ---- the same post-pass as CLOSURES (step_closure.h): per batch ONE workgroup runs its ops in dependence order,
            // vectors handed on through LDS slots. Only where the post-pass is a launch's own role -- the split tail launch
            // (the merged launch and the fused tail keep the vector-op form: their ops wait for other workgroups anyway).
            hp->closures.clear();
            long long blk_total = 0;
            for (int i = 0; i < nb; ++i) blk_total += (sd.b[i].B + CH_GB - 1) / CH_GB;
            const bool will_merge = hp->nlanes == 1 && !(P->flags & MPQE_STEP_SPLIT_TAIL) &&
                                    ((P->flags & MPQE_STEP_MERGE_TAIL) || blk_total <= STEP_CUS + STEP_CUS / 8);
            // Measured on the AIFB step (profiles/r04_*): NOT faster yet -- a closure is one wave per SIMD working through
            // dependent LDS / scalar reads: ~1 us per item, the 3-chain batch's closure 21 - 30 us against 15.4 for the
            // vector-op form's last op -- so it is built only on request (mpqe_debug_option CLOSURE = 1).
#if MPQE_HAS_EXPERIMENTS
            if (!will_merge && hp->nlanes == 1 && exp_on("CLOSURE") && !exp_on("FUSE_TAIL")) {
                bool ok = true;
                std::vector<ClBlock> cls;
                std::vector<RGroup> moved;               // rank-1-only matrices whose terms span batches: reduction groups
                // the R1 ops of uops_b are its last r1_only.size() entries, in r1_only's order
                const size_t r1_first = hp->uops_b.size() - r1_only.size();
                std::vector<char> r1_taken(r1_only.size(), 0);
                for (int i = 0; i < nb && ok; ++i) {
                    ClBlock cb;
                    memset(&cb, 0, sizeof(cb));
                    cb.batch = i;
                    std::unordered_map<int, int> slot_of_vec, slot_of_part;
                    int nslots = 0;
                    auto add_pre = [&](int kind, int row, int nrows) -> int {
                        if (cb.npre >= CL_MAX_PRE) { ok = false; return 0; }
                        ClPreRec &r = cb.pre[cb.npre++];
                        r.kind = kind;
                        r.row = row;
                        r.nrows = nrows;
                        r.slot = nslots++;
                        r.out_vec = -1;
                        return r.slot;
                    };
                    auto part_slot = [&](int row0, int nrows) -> int {      // the column sum of rows [row0, row0 + nrows) of `parts`
                        auto it = slot_of_part.find(row0);
                        if (it != slot_of_part.end()) return it->second;
                        const int sl = add_pre(3, row0, nrows);
                        slot_of_part[row0] = sl;
                        return sl;
                    };
                    auto vt_slot = [&](int v) -> int {                      // a copy of VT row v (a pre-pass vector)
                        auto it = slot_of_vec.find(v);
                        if (it != slot_of_vec.end()) return it->second;
                        const int sl = add_pre(2, v, 1);
                        slot_of_vec[v] = sl;
                        return sl;
                    };
                    // the batch's column-sum vectors somebody reads (UOP_RED of the vector-op form): slot + VT row
                    for (size_t k = 0; k < r1_first && ok; ++k) {
                        const UOp &o = hp->uops_b[k];
                        if (o.kind != UOP_RED || vinfo[o.out_vec].batch != i) continue;
                        const int sl = part_slot(o.row0, o.nrows);
                        for (int q = 0; q < cb.npre; ++q)
                            if (cb.pre[q].slot == sl) cb.pre[q].out_vec = o.out_vec;
                        slot_of_vec[o.out_vec] = sl;
                    }
                    auto in_slot = [&](const UOp &op, int t) -> int {
                        if (op.in_kind[t] == 3) return part_slot(op.in_vec[t], op.in_gran[t]);
                        if (op.in_kind[t] == 0) {
                            auto it = slot_of_vec.find(op.in_vec[t]);
                            if (it != slot_of_vec.end()) return it->second;
                        }
                        ok = false;                                         // (its producer is not of this batch: cannot be)
                        return 0;
                    };
                    // BWD ops, level L-1 down to 0 (uops_b's order: the ops of one level are adjacent and independent of each
                    // other). Per level and 64-row chunk ONE item per distinct matrix, with every (op, term) that multiplies by it.
                    struct LevOp { int k, acc, out_slot, mask_slot, terms_left[4]; int ins[UOP_MAX_TERMS]; };
                    {
                        std::vector<size_t> mine;
                        for (size_t k = 0; k < r1_first; ++k)
                            if (hp->uops_b[k].kind == UOP_BWD && vinfo[hp->uops_b[k].out_vec].batch == i) mine.push_back(k);
                        size_t q0 = 0;
                        while (q0 < mine.size() && ok) {
                            const int lev = vinfo[hp->uops_b[mine[q0]].out_vec].level;
                            size_t q1 = q0;
                            while (q1 < mine.size() && vinfo[hp->uops_b[mine[q1]].out_vec].level == lev) ++q1;
                            if (q1 - q0 > CL_ACCS) { ok = false; break; }
                            std::vector<LevOp> lops;
                            for (size_t q = q0; q < q1 && ok; ++q) {
                                const UOp &op = hp->uops_b[mine[q]];
                                LevOp lo;
                                memset(&lo, 0, sizeof(lo));
                                lo.k = (int)mine[q];
                                lo.acc = (int)(q - q0);
                                for (int t = 0; t < op.nterms; ++t) lo.ins[t] = in_slot(op, t);
                                lo.mask_slot = op.mask_vec >= 0 ? vt_slot(op.mask_vec) : -1;
                                lops.push_back(lo);
                            }
                            // (outputs get their slots after every input of the level is resolved: a level never reads its own)
                            for (size_t q = 0; q < lops.size(); ++q) {
                                lops[q].out_slot = nslots++;
                                const UOp &op = hp->uops_b[lops[q].k];
                                if (op.out_vec >= 0 || op.out_part >= 0) {
                                    if (cb.nout >= CL_MAX_OUT) { ok = false; break; }
                                    ClOutRec &o = cb.out[cb.nout++];
                                    o.slot = lops[q].out_slot;
                                    o.out_vec = op.out_vec;
                                    o.out_part = op.out_part;
                                }
                            }
                            // distinct matrices of the level in first-use order, each with its (op, term) uses
                            struct MatUse { int layer, mat; std::vector<std::pair<int, int>> uses; };
                            std::vector<MatUse> mats;
                            for (size_t q = 0; q < lops.size(); ++q) {
                                const UOp &op = hp->uops_b[lops[q].k];
                                for (int t = 0; t < op.nterms; ++t) {
                                    size_t m = 0;
                                    for (; m < mats.size(); ++m)
                                        if (mats[m].layer == uid[op.layer[t]] && mats[m].mat == op.mat[t] && mats[m].uses.size() < CL_USES) break;
                                    if (m == mats.size()) mats.push_back(MatUse{uid[op.layer[t]], op.mat[t], {}});
                                    mats[m].uses.push_back(std::make_pair((int)q, t));
                                }
                            }
                            for (int ch = 0; ch < D / 64 && ok; ++ch) {
                                int seen[CL_ACCS] = {0, 0, 0};             // terms of each op already emitted in this chunk
                                for (size_t m = 0; m < mats.size(); ++m) {
                                    if (cb.nitems >= CL_MAX_ITEMS) { ok = false; break; }
                                    ClItemRec &r = cb.item[cb.nitems++];
                                    memset(&r, 0, sizeof(r));
                                    r.layer = mats[m].layer;
                                    r.mat = mats[m].mat;
                                    const bool level_end = ch == D / 64 - 1 && m + 1 == mats.size();
                                    r.meta = ch | ((int)mats[m].uses.size() << 8) | (level_end ? 1 << 16 : 0);
                                    for (size_t u = 0; u < mats[m].uses.size(); ++u) {
                                        const int q = mats[m].uses[u].first, t = mats[m].uses[u].second;
                                        const UOp &op = hp->uops_b[lops[q].k];
                                        const int fl = (seen[q] == 0 ? CLI_FIRST : 0) | (seen[q] == op.nterms - 1 ? CLI_LAST : 0);
                                        ++seen[q];
                                        if (lops[q].ins[t] > 31 || lops[q].out_slot > 31 || lops[q].mask_slot > 31) ok = false;
                                        r.use[u] = lops[q].ins[t] | (lops[q].acc << 5) | (fl << 7) | (lops[q].out_slot << 9) |
                                                   ((lops[q].mask_slot + 1) << 14);
                                    }
                                }
                            }
                            for (size_t q = 0; q < lops.size(); ++q) slot_of_vec[hp->uops_b[lops[q].k].out_vec] = lops[q].out_slot;
                            q0 = q1;
                        }
                    }
                    for (size_t k = 0; k < r1_only.size() && ok; ++k) {   // rank-1-only matrices all of whose terms are this batch's
                        const UOp &op = hp->uops_b[r1_first + k];
                        bool mine = true, any = false;
                        for (int t = 0; t < op.nterms; ++t) {
                            const int bt = vinfo[hp->rank1[r1_only[k].r1_start + t].v].batch;
                            mine = mine && bt == i;
                            any = any || bt == i;
                        }
                        if (!mine) {
                            if (any && !r1_taken[k]) {
                                r1_taken[k] = 2;
                                moved.push_back(r1_only[k]);
                            }
                            continue;
                        }
                        r1_taken[k] = 1;
                        if (cb.nr1 >= CL_MAX_R1) { ok = false; break; }
                        ClR1Rec &r = cb.r1[cb.nr1++];
                        r.layer = op.r1_layer;
                        r.rel = op.r1_rel;
                        r.nterms = op.nterms;
                        for (int t = 0; t < op.nterms && ok; ++t) {
                            r.v[t] = in_slot(op, t);
                            r.u[t] = vt_slot(op.u_vec[t]);
                        }
                    }
                    if (nslots > CL_MAX_SLOTS) ok = false;
                    while (ok && cb.nitems % 4 != 0) {                     // whole trips of the item loop: items that do nothing
                        if (cb.nitems >= CL_MAX_ITEMS) { ok = false; break; }
                        ClItemRec &r = cb.item[cb.nitems++];
                        memset(&r, 0, sizeof(r));
                        r.mat = -1;                 // (no uses, no barrier: the root matrix of layer 0 is read and dropped)
                    }
                    if (!ok || (cb.npre == 0 && cb.nitems == 0 && cb.nr1 == 0)) continue;
                    cls.push_back(cb);
                }
                for (size_t k = 0; k < r1_only.size(); ++k) ok = ok && r1_taken[k] != 0;
                if (ok && !cls.empty()) {
                    // heaviest closures first: they start first
                    std::stable_sort(cls.begin(), cls.end(), [](const ClBlock &a, const ClBlock &b) { return a.nitems > b.nitems; });
                    hp->closures.swap(cls);
                    hp->groups.insert(hp->groups.end(), moved.begin(), moved.end());
                }
            }
#else
            (void)will_merge;
#endif
        }
    }
    hp->nvec = (int)vinfo.size();
    hp->ngran = ngran;

    hp->blk_off[0] = 0;
    for (int i = 0; i < nb; ++i) hp->blk_off[i + 1] = hp->blk_off[i] + (sd.b[i].B + CH_GB - 1) / CH_GB;
    // chain programmes: per batch the K-blocks (source slot, matrix) of every live node update, forward levels
    // 0 .. L-1 then backward levels L-1 .. 0, in execution order
    hp->cops.clear();
    hp->crefs.clear();
    hp->wt_slots.clear();
    {
        struct Prog {
            int work, batch, fb, fc, bb, bc, rof;
        };
        std::vector<Prog> progs;
        for (int i = 0; i < nb; ++i) {
            const BatchDev &d = sd.b[i];
            const TmplArgs &tp = d.tp;
            Prog pr;
            pr.batch = i;
            int cv_slots = 0;
            pr.rof = 0;
            // a learned readout's Linear layers r = 0, 1 (reference model.py:497-515): per node slot one K-block, the node's own
            // row times W_r^T (forward, a transposed copy) / its gradient row times W_r (backward: the parameter itself)
            auto copy_slot = [&](int layer, int col0, int ld, int plain) -> int {      // a D x D block prepared by the prologue
                size_t k = 0;
                for (; k < hp->wt_slots.size(); ++k) {
                    const WtSlot &w = hp->wt_slots[k];
                    if (w.layer == layer && w.mat == -1 && w.col0 == col0 && w.ld == ld && w.plain == plain) break;
                }
                if (k == hp->wt_slots.size()) hp->wt_slots.push_back(WtSlot{layer, -1, col0, ld, plain});
                return (int)k;
            };
            auto readout_ops = [&](int dir) {
                const unsigned rows = d.live[d.L + 1];          // node slots with a readout row (targetmlp: not the target)
                auto op_of = [&](int src, int node, int r, int level) {
                    ChainOp op;
                    op.src = (unsigned char)src;
                    op.node = (unsigned char)node;
                    op.layer = (unsigned char)(VL0 + r);
                    op.level = (unsigned char)level;
                    op.mat = -1;
                    op.flags = 0;
                    op.wt_slot = 0;
                    op.aux = -1;
                    op.pad = 0;
                    return op;
                };
                for (int q = 0; q < ROL; ++q) {
                    const int r = dir ? ROL - 1 - q : q;
                    const size_t level_first = hp->cops.size();
                    if (!dir) {
                        // forward: row n = ReLU([target |] node n) W_0^T + b_0), then W_2^T + b_2 -- transposed copies; the
                        // hidden rows H[L + 1] feed the second layer's weight gradient, the output rows only the scores
                        for (int n = 0; n < tp.N; ++n) {
                            if (!((rows >> n) & 1u)) continue;
                            const size_t first = hp->cops.size();
                            if (ro_pairs && r == 0) {
                                ChainOp ta = op_of(d.A, n, r, d.L + 1);
                                ta.pad = 1 + copy_slot(VL0, 0, 2 * D, 0);
                                hp->cops.push_back(ta);
                            }
                            ChainOp op = op_of(n, n, r, d.L + r + 1);
                            if (ro_cat && r == 0) {      // the last level's block; the earlier levels' products come back from HBM
                                op.pad = 1 + copy_slot(VL0, (d.L - 1) * D, ro_blocks * D, 0);
                                if (d.L > 1) op.flags |= CH_ADDG;
                            } else
                                op.pad = 1 + copy_slot(VL0 + r, (ro_pairs && r == 0) ? D : 0, (ro_pairs && r == 0) ? 2 * D : D, 0);
                            hp->cops.push_back(op);
                            hp->cops[first].flags |= CH_FIRST;
                            hp->cops.back().flags |= CH_LAST | (r == 0 ? CH_RELU : CH_NOSTORE);
                            for (size_t k = first; k < hp->cops.size(); ++k) {
                                hp->cops[k].flags |= hp->cops.back().flags & (CH_RELU | CH_NOSTORE);
                                hp->cops[k].wt_slot = r;      // (its bias: constant slot r, loaded in front of the readout's K loop)
                            }
                        }
                    } else if (r == 1) {
                        // backward: gH[L + 1][n] = (gH[L + 2][n] W_2) through the hidden rows' ReLU -- the parameter itself
                        for (int n = 0; n < tp.N; ++n) {
                            if (!((rows >> n) & 1u)) continue;
                            ChainOp op = op_of(n, n, r, d.L + 1);
                            op.flags = CH_FIRST | CH_LAST | CH_MASK;
                            op.wt_slot = -1;
                            op.aux = part_row[i][d.L + 1][n];
                            hp->cops.push_back(op);
                        }
                    } else {
                        // gH[L][n] = gH[L + 1][n] W_0 (targetmlp: its node block; the target's row: the sum over the nodes of
                        // gH[L + 1][n] times the target block -- plain copies of the column blocks)
                        // (concat: first the readout's share of the state gradients of levels 1 .. L - 1 -- gH[L + 1][n] times
                        // column block l - 1 --, stored to gH[l][n] through scratch tiles; the level's own update adds it)
                        int scratch = 0;
                        for (int l = 1; ro_cat && l < d.L; ++l)
                            for (int n = 0; n < tp.N; ++n) {
                                ChainOp op = op_of(n, n, r, l);
                                op.flags = CH_FIRST | CH_LAST;
                                op.wt_slot = copy_slot(VL0, (l - 1) * D, ro_blocks * D, 1);
                                op.pad = CH_TSLOT_ON | ((3 - (scratch++ & 1)) << 16);
                                hp->cops.push_back(op);
                            }
                        for (int n = 0; n < tp.N; ++n) {
                            const size_t first = hp->cops.size();
                            if ((rows >> n) & 1u) {
                                ChainOp op = op_of(n, n, r, d.L);
                                op.wt_slot = ro_pairs ? copy_slot(VL0, D, 2 * D, 1) : -1;
                                if (ro_cat) op.wt_slot = copy_slot(VL0, (d.L - 1) * D, ro_blocks * D, 1);
                                hp->cops.push_back(op);
                            } else {
                                for (int m = 0; m < tp.N; ++m) {
                                    if (!((rows >> m) & 1u)) continue;
                                    ChainOp op = op_of(m, n, r, d.L);
                                    op.wt_slot = copy_slot(VL0, 0, 2 * D, 1);
                                    hp->cops.push_back(op);
                                }
                            }
                            if (hp->cops.size() == first) continue;
                            hp->cops[first].flags |= CH_FIRST;
                            hp->cops.back().flags |= CH_LAST;
                            hp->cops.back().aux = part_row[i][d.L][n];
                        }
                    }
                    if (hp->cops.size() > level_first) hp->cops.back().flags |= CH_LEVEL_END;
                }
            };
            for (int dir = 0; dir < 2; ++dir) {
                const int begin = (int)hp->cops.size();
                if (dir && ro) readout_ops(1);
                for (int q = 0; q < d.L; ++q) {
                    const int p = dir ? d.L - 1 - q : q;
                    const int li = p < d.L - 1 ? p : P->num_layers - 1;
                    const unsigned lin = d.live[p], lout = d.live[p + 1];
                    int lvl_flags = 0;
                    if (!dir && p < d.L - 1) lvl_flags |= CH_RELU;
                    if (dir && p >= 1) lvl_flags |= CH_MASK;
                    // the weight gradients read H[0 .. L-1] and gH[1 .. L]; H[L] feeds only the scores and gH[0]
                    // only the anchor / variable-row gradients, all inside the chain kernel
                    // (a learned readout on the chain: H[L] is the input of its first layer's weight gradient)
                    if ((!dir && p == d.L - 1 && !ro) || (dir && p == 0)) lvl_flags |= CH_NOSTORE;
                    const size_t level_first = hp->cops.size();
                    if (ro_cat && !dir && p >= 1) {
                        // concat: the first readout layer's product with THIS level's input states H[p] (column block p - 1),
                        // added to the sum so far (H[L + 1][n], through scratch tiles 3 / 2: written whole one barrier later)
                        int scratch = 0;
                        for (int n = 0; n < tp.N; ++n) {
                            ChainOp op;
                            op.src = op.node = (unsigned char)n;
                            op.layer = (unsigned char)VL0;
                            op.level = (unsigned char)(d.L + 1);
                            op.mat = -1;
                            op.flags = CH_FIRST | CH_LAST | CH_NOBIAS | (p > 1 ? CH_ADDG : 0);
                            op.wt_slot = 0;
                            op.aux = -1;
                            op.pad = (1 + copy_slot(VL0, (p - 1) * D, ro_blocks * D, 0)) | CH_TSLOT_ON | ((3 - (scratch++ & 1)) << 16);
                            hp->cops.push_back(op);
                        }
                    }
                    if (ro_cat && dir && p >= 1 && p < d.L) lvl_flags |= CH_ADDG;
                    // per-graph (NU) node slots only: a batch-uniform state is a vector of the pre-pass, its gradient
                    // a column sum of the post-pass. The sources of an NU node's K-blocks are its NU sources (the
                    // uniform ones are in the node's constant vector); backward, every destination of an NU node is NU.
                    const unsigned uin = uni[i][p], uout = uni[i][p + 1];
                    for (int n = 0; n < tp.N; ++n) {
                        if (!(((dir ? lin : lout) >> n) & 1u)) continue;
                        if (((dir ? uin : uout) >> n) & 1u) continue;
                        const size_t first = hp->cops.size();
                        auto push = [&](int src, int mat) {
                            ChainOp op;
                            op.src = (unsigned char)src;
                            op.node = (unsigned char)n;
                            op.layer = (unsigned char)li;
                            op.level = (unsigned char)(dir ? p : p + 1);
                            op.mat = mat;
                            op.flags = lvl_flags;
                            op.wt_slot = 0;
                            op.aux = -1;
                            op.pad = 0;
                            if (dir) {      // shared layers alias one parameter set: one copy per unique (layer, matrix)
                                size_t k = 0;
                                for (; k < hp->wt_slots.size(); ++k)
                                    if (hp->wt_slots[k].layer == uid[li] && hp->wt_slots[k].mat == mat) break;
                                if (k == hp->wt_slots.size()) hp->wt_slots.push_back(WtSlot{uid[li], mat, 0, D, 0});
                                op.wt_slot = (int)k;
                            }
                            hp->cops.push_back(op);
                        };
                        for (int e = 0; e < tp.E; ++e) {
                            if (!dir && tp.dst[e] == n && !((uin >> tp.src[e]) & 1u)) push(tp.src[e], (int)tp.rel[e]);
                            if (dir && tp.src[e] == n && ((lout >> tp.dst[e]) & 1u)) push(tp.dst[e], (int)tp.rel[e]);
                        }
                        if (dir ? ((lout >> n) & 1u) != 0 : !((uin >> n) & 1u)) push(n, -1);
                        if (hp->cops.size() == first) return MPQE_ERR_UNSUPPORTED;      // (cannot happen: see the liveness / uniformity rules)
                        hp->cops[first].flags |= CH_FIRST;
                        hp->cops.back().flags |= CH_LAST;
                        if (!dir) {         // the node's constant: bias + its uniform sources' products (-1: the bias itself)
                            const auto it = vec_of.find((((long long)V_CV * MPQE_STEP_MAX_BATCHES + i) *
                                                         (MPQE_STEP_MAX_LAYERS + 1) + (p + 1)) * 4 + n);
                            hp->cops.back().aux = it == vec_of.end() ? -1 : gran_of[it->second];      // (its granule slot)
                            hp->cops.back().wt_slot = cv_slots++;
                        } else {
                            hp->cops.back().aux = part_row[i][p][n];       // (anchors at level 0: -1)
                        }
                    }
                    if (hp->cops.size() > level_first) hp->cops.back().flags |= CH_LEVEL_END;
                }
                (dir ? pr.bb : pr.fb) = begin;
                (dir ? pr.bc : pr.fc) = (int)hp->cops.size() - begin;
                if (!dir && ro) {
                    const int rb = (int)hp->cops.size();
                    readout_ops(0);
                    pr.rof = (int)hp->cops.size() - rb;
                }
            }
            // (the chain kernel's LDS tables: step_chain.h. Steps beyond them take the level form.)
            if (chain && (cv_slots > CH_MAX_CV || pr.fc + pr.rof + pr.bc > CH_MAX_OPS)) return MPQE_ERR_UNSUPPORTED;
            pr.work = pr.fc + pr.rof + pr.bc;
            progs.push_back(pr);
        }
        // Placement (speed only, results never depend on it). Workgroups are dealt round-robin over the 8 XCDs
        // (block i -> XCD i % 8, measured) and each XCD has its own 4 MB L2, which cannot hold the weight
        // matrices of all batches plus their transposed copies: so every batch is given to ONE XCD (all its
        // blocks multiply by the same few matrices: one fetch per XCD, L2 hits for the other blocks), batches
        // dealt to XCDs heaviest first onto the least loaded. Inside an XCD (32 CUs; block k of the XCD shares
        // its CU with block k + 32, measured) the heaviest blocks run alone and the lightest pair up.
        // Grid = 8 x (largest XCD list); the holes are refs with batch = -1 (the workgroup exits at once).
        std::stable_sort(progs.begin(), progs.end(), [](const Prog &a, const Prog &b) { return a.work > b.work; });
        const size_t cus = STEP_CUS / STEP_XCDS;
        for (int l = 0; l < hp->nlanes; ++l) {             // one grid per stream lane
            hp->cref_begin[l] = (int)hp->crefs.size();
            std::vector<ChainRef> bins[STEP_XCDS];
            long long load[STEP_XCDS] = {0};
            for (size_t k = 0; k < progs.size(); ++k) {    // a big batch goes out in chunks of one block per CU
                if (progs[k].batch < hp->lane_begin[l] || progs[k].batch >= hp->lane_begin[l + 1]) continue;
                const int Bk = sd.b[progs[k].batch].B;
                for (int c0 = 0; c0 < Bk; c0 += (int)cus * CH_GB) {
                    int best = 0;
                    for (int x = 1; x < STEP_XCDS; ++x)
                        if (load[x] < load[best]) best = x;
                    for (int g0 = c0; g0 < Bk && g0 < c0 + (int)cus * CH_GB; g0 += CH_GB) {   // progs is sorted: bins stay sorted
                        const BatchDev &bd = sd.b[progs[k].batch];
                        const unsigned meta = (unsigned)bd.tp.N | (unsigned)bd.A << 4 |
                                              (unsigned)(bd.anchor_tab[0] & 15) << 8 | (unsigned)(bd.anchor_tab[1] & 15) << 12 |
                                              (unsigned)(bd.anchor_tab[2] & 15) << 16 | (unsigned)(bd.target_tab & 15) << 20;
                        bins[best].push_back(ChainRef{progs[k].batch, g0, progs[k].fb, progs[k].fc, progs[k].bb,
                                                      progs[k].bc, hp->blk_off[progs[k].batch] + g0 / CH_GB,
                                                      hp->dm.base[progs[k].batch] + g0 / DONE_GRAPHS,
                                                      (int)(bd.anchor_off + g0), (int)(bd.g_off + g0), bd.B, meta, progs[k].rof});
                        load[best] += progs[k].work;
                    }
                }
            }
            if (l == 0) {
                // Merged launch: where the post roles (weight-gradient tiles, post-pass) run. They wait in a slot of a CU
                // until their batch's chain workgroups are done and then compete with the chain workgroups that still
                // run there -- harmless on the XCDs of LIGHT batches (their chain workgroups are not the launch's
                // critical path), costly on the XCDs of the heaviest ones. Pick the XCDs that have a free slot per CU
                // (at most one chain workgroup per CU) and do not host a workgroup of the heaviest programme; failing
                // that, every XCD.
                int wmax = 0;
                for (size_t k = 0; k < progs.size(); ++k) wmax = std::max(wmax, progs[k].work);
                bool heavy[STEP_XCDS];
                for (int x = 0; x < STEP_XCDS; ++x) {
                    heavy[x] = false;
                    for (size_t k = 0; k < bins[x].size(); ++k)
                        heavy[x] = heavy[x] || (bins[x][k].fwd_count + bins[x][k].rof + bins[x][k].bwd_count) >= wmax;
                }
                // ... with room for all of them at once (two workgroups per CU): first the XCDs with a free slot on every CU
                // that host no workgroup of the heaviest programme, then every XCD without one, then all. (AIFB mix, D = 128,
                // B per batch 64 / 128: 50.0 / 52.5 us per step with this rule against 51.7 / 55.2 on all XCDs; B = 384:
                // the first choice is short of room -- 63.0 against 61.0.)
                const long long need = (long long)hp->wblock.size() + (long long)hp->uops_b.size() * (D / 64);
                const int pmv = mpqe_dbg_value("POST_MODE", -1);    // (timing experiments: force a choice)
                const bool pm = pmv >= 0;
                int na = 0;
                for (int mode = pm ? pmv : 0; mode < 3; ++mode) {
                    long long room = 0;
                    na = 0;
                    for (int x = 0; x < STEP_XCDS; ++x) {
                        const bool ok = mode == 2 || (!heavy[x] && (mode == 1 || bins[x].size() <= cus));
                        hp->post_rank[x] = ok ? na++ : -1;
                        if (ok) room += std::max<long long>(0, 2 * (long long)cus - (long long)bins[x].size());
                    }
                    if (na > 0 && (room >= need || mode == 2 || pm)) break;
                }
                if (na == 0) {
                    na = STEP_XCDS;
                    for (int x = 0; x < STEP_XCDS; ++x) hp->post_rank[x] = x;
                }
                hp->post_na = na;
                // The touch plan's sort (MPQE_STEP_BUILD_TOUCH) holds a slot of a CU for most of the launch: on an XCD whose
                // CUs all take two chain workgroups that slot is missing (AIFB step: 22 chain workgroups started 20 us late,
                // launch 43 -> 57 us). Same choice as above: the XCDs with a free slot per CU and no workgroup of the
                // heaviest programme, then those with a free slot, then all (a step that fills every XCD many times over).
                int sna = 0;
                const int smv = mpqe_dbg_value("SORT_MODE", 0);     // (timing experiments: force a choice)
                for (int mode = smv; mode < 3 && sna == 0; ++mode) {
                    sna = 0;
                    for (int x = 0; x < STEP_XCDS; ++x) {
                        const bool ok = mode == 2 || (bins[x].size() <= cus && (mode == 1 || !heavy[x]));
                        hp->sort_rank[x] = ok ? sna++ : -1;
                    }
                }
                hp->sort_na = sna;
                hp->pl_na = 0;
                for (int x = 0; x < STEP_XCDS; ++x) hp->pl_rank[x] = bins[x].size() <= cus ? hp->pl_na++ : -1;
            }
            size_t longest = 0;
            for (int x = 0; x < STEP_XCDS; ++x) {
                std::vector<ChainRef> &v = bins[x];
                const size_t n = v.size();
                if (n > cus && n <= 2 * cus) {
                    std::vector<ChainRef> o;
                    const size_t R = n - cus;                   // CUs that take two blocks
                    for (size_t k = 0; k < R; ++k) o.push_back(v[n - 2 * R + k]);           // heavier of a pair
                    for (size_t k = 0; k < n - 2 * R; ++k) o.push_back(v[k]);               // alone
                    for (size_t k = 0; k < R; ++k) o.push_back(v[n - 1 - k]);               // its light partner
                    v.swap(o);
                }
                if (n > longest) longest = n;
            }
            for (size_t k = 0; k < longest; ++k)
                for (int x = 0; x < STEP_XCDS; ++x)
                    hp->crefs.push_back(k < bins[x].size() ? bins[x][k] : ChainRef{-1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0u, 0});
        }
        hp->cref_begin[hp->nlanes] = (int)hp->crefs.size();
        // Merged launch: the tiles queue behind the chain workgroups of their XCD and wait for the chain workgroups of
        // their batch; the batches with the shortest programmes finish first, so their tiles go first (the `tiles`
        // workgroups of a K-chunk stay adjacent: grad_w_block puts them on one XCD).
        if (chain && hp->nlanes == 1 && !hp->wblock.empty()) {
            int work[MPQE_STEP_MAX_BATCHES] = {0};
            for (size_t k = 0; k < progs.size(); ++k) work[progs[k].batch] = progs[k].work;
            const size_t nchunks = hp->wblock.size() / tiles;
            std::vector<size_t> order(nchunks);
            for (size_t c = 0; c < nchunks; ++c) order[c] = c;
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
                return work[hp->wblock[a * tiles].batch] < work[hp->wblock[b * tiles].batch];
            });
            std::vector<WBlock> sorted;
            sorted.reserve(hp->wblock.size());
            for (size_t c = 0; c < nchunks; ++c)
                for (int t = 0; t < tiles; ++t) sorted.push_back(hp->wblock[order[c] * tiles + t]);
            hp->wblock.swap(sorted);
        }
    }

    // workspace layout
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    hp->level_stride = rows * D;
    hp->o_sd = take(sizeof(StepDev));
    hp->o_wsrc = take(hp->wsrc.size() * sizeof(WSource));
    hp->o_wblock = take(hp->wblock.size() * sizeof(WBlock));
    hp->o_vsrc = take(hp->vsrc.size() * sizeof(VSource));
    hp->o_vblock = take(hp->vblock.size() * sizeof(int));
    hp->o_groups = take(hp->groups.size() * sizeof(RGroup));
    hp->o_anchor = take(hp->anchor_off.size() * sizeof(int));
    for (int l = 0; l < hp->nlanes && !chain; ++l)
        for (int p = 0; p < hp->lane_Lmax[l]; ++p) {
            hp->o_tf[l][p] = take(hp->tfwd[l][p].size() * sizeof(TileRef));
            hp->o_tb[l][p] = take(hp->tbwd[l][p].size() * sizeof(TileRef));
        }
    hp->o_cref = take(hp->crefs.size() * sizeof(ChainRef));
    hp->o_cops = take(hp->cops.size() * sizeof(ChainOp));
    hp->o_wtslots = take(hp->wt_slots.size() * sizeof(WtSlot));
    hp->o_zmats = take(hp->zmats.size() * sizeof(ZMat));
    {   // post-pass outputs the step's reduction reads (fused tail: they travel inside one launch)
        std::vector<char> isv((size_t)std::max(hp->nvec, 1), 0);
        for (size_t k = 0; k < hp->rank1.size(); ++k)
            if (hp->rank1[k].v >= 0 && hp->rank1[k].v < hp->nvec) isv[hp->rank1[k].v] = 1;
        for (size_t k = 0; k < hp->uops_b.size(); ++k) {
            UOp &op = hp->uops_b[k];
            op.through = (op.out_part >= 0 || (op.out_vec >= 0 && op.out_vec < hp->nvec && isv[op.out_vec])) ? 1 : 0;
        }
        for (size_t k = 0; k < hp->uops_f.size(); ++k) hp->uops_f[k].through = 1;
    }
    hp->o_uopf = take(hp->uops_f.size() * sizeof(UOp));
    hp->o_uopb = take(hp->uops_b.size() * sizeof(UOp));
#if MPQE_HAS_EXPERIMENTS
    hp->o_closures = take(hp->closures.size() * (size_t)CL_BLOCK_WORDS * 4);
#else
    hp->o_closures = take(0);
#endif
    hp->o_rank1 = take(hp->rank1.size() * sizeof(Rank1));
    hp->o_done_inc = take(hp->done_inc.size() * sizeof(int));
    TouchMeta tmeta;
    memset(&tmeta, 0, sizeof(tmeta));
    hp->ts_blocks = 0;
    hp->ts_key_bits = hp->ts_row_bits = 0;
    {
        long long trows = 1;
        for (int m = 0; m < P->num_modes && m < MPQE_STEP_MAX_MODES; ++m) trows = std::max(trows, (long long)P->table_rows[m]);
        const int rb = touch_bits(trows), kb = rb + 5;
        const long long M = anchors + 2 * graphs;
        if (chain && (P->flags & MPQE_STEP_BUILD_TOUCH) && M <= TSORT_MAX_ENTRIES && kb <= 31) {
            hp->ts_blocks = tsort_blocks(M);
            hp->ts_key_bits = kb;
            hp->ts_row_bits = rb;
            tmeta.nb = nb;
            tmeta.row_bits = rb;
            for (int i = 0; i < nb; ++i) {
                const TemplateDesc &t = kTemplates[B[i].query_type];
                tmeta.B[i] = B[i].batch_size;
                tmeta.A[i] = t.A;
                tmeta.anchor_off[i] = sd.b[i].anchor_off;
                tmeta.g_off[i] = sd.b[i].g_off;
                for (int a = 0; a < 3; ++a) tmeta.anchor_tab[i][a] = a < t.A ? B[i].anchor_mode[a] : 0;
                tmeta.target_tab[i] = B[i].target_mode;
            }
            tmeta.anchor_off[nb] = anchors;
            tmeta.g_off[nb] = graphs;
            for (int m = 0; m < P->num_modes && m < MPQE_STEP_MAX_MODES; ++m) tmeta.table_rows[m] = P->table_rows[m];
        }
    }
    hp->o_tmeta = take(hp->ts_blocks ? sizeof(TouchMeta) : 0);
    // hand-off state of the packed step, zeroed when the table is uploaded: the two epoch words (forward pre-pass,
    // backward post-pass), then the granules
    hp->o_epoch = take(256);
    hp->o_gran = take((size_t)hp->ngran * D * sizeof(u64));
    hp->o_done = take(2 * hp->done_inc.size() * sizeof(unsigned));      // published | arrived
    hp->desc_total = off;
    off = 0;
    hp->o_H = take((size_t)(hp->Lmax + 1 + ROL) * rows * D * 4);
    hp->o_GH = take((size_t)(hp->Lmax + 1 + ROL) * rows * D * 4);
    hp->o_tpos = take((size_t)graphs * D * 4);
    hp->o_tneg = take((size_t)graphs * D * 4);
    hp->o_spos = take((size_t)graphs * 4);
    hp->o_sneg = take((size_t)graphs * 4);
    hp->o_terms = take((size_t)graphs * 4);
    // (the caller's readout: its query embeddings in, their gradients out)
    hp->o_Q = take(P->readout == MPQE_READOUT_CALLER ? (size_t)graphs * D * 4 : 0);
    hp->o_GQ = take(P->readout == MPQE_READOUT_CALLER ? (size_t)graphs * D * 4 : 0);
    hp->ro_rows = 0;
    hp->ro_kin = 0;
    hp->ro_direct = false;
    hp->rlin_bytes = 0;
    if (ro) {           // (on the chain: no buffers of its own -- levels L + 1, L + 2 of H / GH)
        hp->ro_rows = ro_pairs ? rows - graphs : rows;
        hp->ro_kin = ro_blocks * D;
    }
    if (P->readout >= MPQE_READOUT_MLP && !ro) {
        const bool pairs = P->readout == MPQE_READOUT_TARGETMLP;
        hp->ro_rows = pairs ? rows - graphs : rows;
        hp->ro_kin = pairs ? 2 * D : (P->readout == MPQE_READOUT_CONCAT ? P->num_layers * D : D);
        // (mlp with every batch at the same depth: the input rows ARE the final level of H, their gradient the same level of GH)
        bool same = true;
        for (int i = 1; i < nb; ++i) same = same && hp->sd.b[i].L == hp->sd.b[0].L;
        hp->ro_direct = P->readout == MPQE_READOUT_MLP && same;
        const size_t xin = (size_t)hp->ro_rows * hp->ro_kin * 4, xd = (size_t)hp->ro_rows * D * 4;
        hp->o_rx = take(hp->ro_direct ? 0 : xin);
        hp->o_rgx = take(hp->ro_direct ? 0 : xin);
        hp->o_rh = take(xd);
        hp->o_ry = take(xd);
        hp->o_rgy = take(xd);
        hp->o_rgh = take(xd);
        hp->rlin_bytes = std::max(mpqe_linear_bwd_workspace_bytes(hp->ro_rows, hp->ro_kin, D),
                                  mpqe_linear_bwd_workspace_bytes(hp->ro_rows, D, D));
        hp->o_rlin = take(hp->rlin_bytes);
    }
    hp->o_slabs = take((size_t)hp->total_slabs * D * D * 4);
    hp->o_parts = take((size_t)hp->total_parts * D * 4);
    hp->o_WT = take(hp->wt_slots.size() * (size_t)D * D * 4);
    hp->o_bterms = take((size_t)hp->blk_off[nb] * 4);
    {   // host image of the descriptor table ([0, o_epoch) of the caller's desc buffer)
        hp->image.assign(hp->o_epoch, 0);
        auto put = [&](size_t o, const void *src, size_t n) {
            if (n) memcpy(hp->image.data() + o, src, n);
        };
        put(hp->o_sd, &hp->sd, sizeof(StepDev));
        put(hp->o_wsrc, hp->wsrc.data(), hp->wsrc.size() * sizeof(WSource));
        put(hp->o_wblock, hp->wblock.data(), hp->wblock.size() * sizeof(WBlock));
        put(hp->o_vsrc, hp->vsrc.data(), hp->vsrc.size() * sizeof(VSource));
        put(hp->o_vblock, hp->vblock.data(), hp->vblock.size() * sizeof(int));
        put(hp->o_groups, hp->groups.data(), hp->groups.size() * sizeof(RGroup));
        put(hp->o_anchor, hp->anchor_off.data(), hp->anchor_off.size() * sizeof(int));
        for (int l = 0; l < hp->nlanes && !chain; ++l)
            for (int p = 0; p < hp->lane_Lmax[l]; ++p) {
                put(hp->o_tf[l][p], hp->tfwd[l][p].data(), hp->tfwd[l][p].size() * sizeof(TileRef));
                put(hp->o_tb[l][p], hp->tbwd[l][p].data(), hp->tbwd[l][p].size() * sizeof(TileRef));
            }
        put(hp->o_cref, hp->crefs.data(), hp->crefs.size() * sizeof(ChainRef));
        put(hp->o_cops, hp->cops.data(), hp->cops.size() * sizeof(ChainOp));
        put(hp->o_wtslots, hp->wt_slots.data(), hp->wt_slots.size() * sizeof(WtSlot));
        put(hp->o_zmats, hp->zmats.data(), hp->zmats.size() * sizeof(ZMat));
        put(hp->o_uopf, hp->uops_f.data(), hp->uops_f.size() * sizeof(UOp));
        put(hp->o_uopb, hp->uops_b.data(), hp->uops_b.size() * sizeof(UOp));
#if MPQE_HAS_EXPERIMENTS
        for (size_t k = 0; k < hp->closures.size(); ++k)
            put(hp->o_closures + k * (size_t)CL_BLOCK_WORDS * 4, &hp->closures[k], sizeof(ClBlock));
#endif
        put(hp->o_rank1, hp->rank1.data(), hp->rank1.size() * sizeof(Rank1));
        put(hp->o_done_inc, hp->done_inc.data(), hp->done_inc.size() * sizeof(int));
        if (hp->ts_blocks) put(hp->o_tmeta, &tmeta, sizeof(tmeta));
    }
    hp->o_VT = take((size_t)hp->nvec * D * 4);
    hp->touch_M = anchors + 2 * graphs;
    hp->o_DG = take(chain ? (size_t)hp->touch_M * D * 4 : 0);       // per-entry table-gradient rows (step_touch.h)
    hp->o_runs = take(chain ? ((size_t)hp->touch_M + 64) * sizeof(int) : 0);      // the touch plan's run starts + their number
    // in-step sort: (key, entry) ping-pong buffers [4][blocks x 1024] + digit counts [4 passes][blocks][256]
    hp->o_tsort = take(hp->ts_blocks ? (size_t)hp->ts_blocks * (4 * (size_t)TSORT_THREADS * tsort_rounds(hp->touch_M) + 4 * 256) * sizeof(unsigned) : 0);
    hp->total = off;
    if (dbg_on("DUMP_PLAN")) {        // diagnostics: what the step's launches consist of
        fprintf(stderr, "plan: chain %d uniform %d blocks %d | tile sources %zu tiles %d slabs %d | groups %zu | uops f %zu b %zu | closures %zu | rank1 %zu | zmats %zu | touch M %lld\n",
                (int)chain, (int)hp->uniform, hp->blk_off[nb], hp->wsrc.size(), hp->wblocks_total, hp->total_slabs, hp->groups.size(),
                hp->uops_f.size(), hp->uops_b.size(), hp->closures.size(), hp->rank1.size(), hp->zmats.size(), hp->touch_M);
        for (size_t k = 0; k < hp->groups.size(); ++k)
            fprintf(stderr, "  group %zu kind %d layer %d row %lld slabs/rows %d rank1 %d\n", k, hp->groups[k].kind, hp->groups[k].layer,
                    hp->groups[k].row, hp->groups[k].count, hp->groups[k].r1_count);
#if MPQE_HAS_EXPERIMENTS
        for (size_t k = 0; k < hp->closures.size(); ++k)
            fprintf(stderr, "  closure %zu batch %d pre %d items %d r1 %d\n", k, hp->closures[k].batch, hp->closures[k].npre,
                    hp->closures[k].nitems, hp->closures[k].nr1);
#endif
    }
    return MPQE_OK;
}

void upload(hipStream_t s, char *dst, const void *src, size_t n) {
    const char *p = reinterpret_cast<const char *>(src);
    for (size_t o = 0; o < n; o += UPLOAD_BYTES) {
        Blob b;
        const size_t m = n - o < UPLOAD_BYTES ? n - o : UPLOAD_BYTES;
        memcpy(b.bytes, p + o, m);
        hipLaunchKernelGGL(step_upload_kernel, dim3(1), dim3(256), 0, s, b, dst + o, (int)m);
    }
}

// chain form when the step qualifies (want_chain) and fits the chain kernel's tables, the level form otherwise
int plan_auto(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb, const mpqe_step_lanes_t *lanes,
              bool chain, HostPlan *hp) {
    if (chain) {
        const int st = make_plan(P, B, nb, lanes, true, hp);
        if (st != MPQE_ERR_UNSUPPORTED) return st;
        *hp = HostPlan();
    }
    return make_plan(P, B, nb, lanes, false, hp);
}

// Everything make_plan() reads, field by field (struct padding never takes part in the comparison).
struct PlanKey {
    int dim, num_layers, num_relations, num_modes, readout, flags, nb, nlanes, chain;
    int dbg_gen;                              // diagnostics switches may shape a plan (TILE_N, NO_CLOSURE, ...): their generation
    int lane_begin[MPQE_STEP_MAX_LANES + 1];
    int alias[MPQE_STEP_MAX_LAYERS];          // first layer with the same parameter buffers
    long long table_rows[MPQE_STEP_MAX_MODES];      // (the in-step touch plan's key widths and batch table)
    mpqe_step_batch_t b[MPQE_STEP_MAX_BATCHES];
};
struct CachedPlan {
    PlanKey key;
    HostPlan hp;
    // the batch weights in the resident descriptor table are not the plan's (a call with mpqe_step_extra_t.batch_weight wrote
    // host x device products there): the next call without extras writes the host weights back first
    mutable bool weights_patched = false;
};
std::mutex g_plan_mu;
std::unordered_map<void *, std::shared_ptr<CachedPlan>> g_plans;
// the plan the size queries of a packed step built: the step's first run takes it over instead of planning again
std::shared_ptr<CachedPlan> g_recent;

void make_key(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb, const mpqe_step_lanes_t *lanes,
              PlanKey *k) {
    memset(k, 0, sizeof(*k));
    k->dim = P->dim; k->num_layers = P->num_layers; k->num_relations = P->num_relations;
    k->num_modes = P->num_modes; k->readout = P->readout; k->flags = P->flags & ~(MPQE_STEP_ZERO_GRADS | MPQE_STEP_NO_KSPLIT | MPQE_STEP_EIGHT_WAVES | MPQE_STEP_ADD_STATE_GRADS | MPQE_STEP_TOUCH_LIBRARY_SORT); k->nb = nb;
    k->nlanes = lanes ? lanes->num_lanes : 1;
    k->dbg_gen = mpqe_dbg_generation();
    for (int m = 0; m < P->num_modes && m < MPQE_STEP_MAX_MODES; ++m) k->table_rows[m] = P->table_rows[m];
    if (lanes)
        for (int l = 0; l <= MPQE_STEP_MAX_LANES; ++l) k->lane_begin[l] = lanes->batch_begin[l];
    for (int l = 0; l < P->num_layers && l < MPQE_STEP_MAX_LAYERS; ++l) {
        k->alias[l] = l;
        for (int m = 0; m < l; ++m)
            if (P->basis[m] == P->basis[l]) {
                k->alias[l] = k->alias[m];
                break;
            }
    }
    for (int i = 0; i < nb; ++i) {
        mpqe_step_batch_t &d = k->b[i];
        d.query_type = B[i].query_type; d.num_passes = B[i].num_passes; d.batch_size = B[i].batch_size;
        d.target_mode = B[i].target_mode; d.weight = B[i].weight;
        for (int e = 0; e < MPQE_MAX_TEMPLATE_EDGES; ++e) { d.edge_type[e] = B[i].edge_type[e]; d.anchor_mode[e] = B[i].anchor_mode[e]; }
        for (int v = 0; v < MPQE_MAX_TEMPLATE_NODES - 1; ++v) d.var_ids[v] = B[i].var_ids[v];
    }
}

}  // namespace

// Part of csrc/step.hip (one translation unit; included there after the level-form kernels): the backward tail of the
// fused step -- weight-gradient tiles, the reduction (slabs + rank-1 terms -> gradient matrices, bias / mode rows, the loss,
// entity-table rows per destination), the backward post-pass' vector ops as roles of the weight-gradient launch.
// reference: the gradients of basis / root / bias of RGCNConv (model.py:292-305) and of the embedding tables
// (encoders.py:40-43) that autograd produces op by op.
#pragma once

// ------------------------------------------------------------------------------------ weight gradients
template <int MODE, int LDS_TILES = 4>
__device__ __forceinline__ void grad_w_block(const StepDev *__restrict__ sd, const WSource *__restrict__ src,
                                             int nsrc, const WBlock *__restrict__ block_start,
                                             const float *__restrict__ H, const float *__restrict__ GH,
                                             long long level_stride, float *__restrict__ slabs, int bid,
                                             int wblocks_total, float *smem, const GradPtrs &gp, bool zeroed,
                                             long long *dbg, int D, const PostArgs *po = nullptr, int tile_n = GT_BN,
                                             int nxcd = 8, bool through = false) {
    const int tiles_j = (D + tile_n - 1) / tile_n, tiles = tiles_j * ((D + GT_BM - 1) / GT_BM);
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: blocks b and b+8 share
    // one. The `tiles` output tiles of a K-chunk read the SAME rows of H and gH (different column
    // halves), so they are mapped to blocks 8 apart -> one XCD, one L2 fetch of the rows instead of
    // `tiles`. Pure placement: correctness never depends on it.
    const int nx = po ? po->na : nxcd;           // XCDs the tile workgroups are dealt to (bid % nx = the XCD's rank)
    const int span = nx * tiles;
    int vb = bid;
    if (bid < (wblocks_total / span) * span) {
        const int grp = bid / span, r = bid - grp * span;
        vb = grp * span + (r % nx) * tiles + (r / nx);
    }
    const WBlock wk = block_start[vb];       // one record, no search, no second hop
    if (po) {
        // merged launch: the rows of the K-chunk are written by chain workgroups of THIS launch; wave 0 waits for their
        // counters (agent-scope loads, bounded), then the workgroup's barrier. No acquire fence: nothing on this CU or
        // XCD has read these lines before their writers released them (rows are written once per step, and caches do not
        // survive a launch boundary).
        if (threadIdx.x < 64) {
            const unsigned ep = *po->epoch_m + 1u;
            for (int c = wk.d0 + (int)threadIdx.x; c < wk.d0 + wk.dn; c += 64)
                uop_wait_until(po->done + c, ep * (unsigned)po->done_inc[c], po->err);
        }
        __syncthreads();
    }
    const long long xs = wk.xs, xo = wk.xo, gs = wk.xs, go = wk.go;
    const long long q0 = wk.q0, q1 = wk.q1;
    const float *x = wk.pad ? GH + wk.g_off : H + wk.x_off;
    const float *out = nullptr;              // (masks are applied by the producers: relu = 0 everywhere)
    const float *g = wk.pad ? H + wk.x_off : GH + wk.g_off;
    float *dst = slabs + wk.slab_off;
    bool direct = false;
    if (wk.direct >= 0) {
        float *gm = wk.rel >= 0 ? pick_grad(gp.basis, wk.direct) : pick_grad(gp.root, wk.direct);
        if (gm) {
            dst = gm + (wk.rel >= 0 ? wk.rel * (long long)D * D : 0);
            direct = true;
        }
    }
    if constexpr (MODE == LD_T) {      // chain form (D % 64 == 0, 16-byte aligned rows): register-only K loop
        // (through: a slab of the fused tail is read by the reduction workgroups of the same launch)
        if (tile_n == 32) grad_w_tile_rows<LDS_TILES, 2>(x, g, D, xs, xo, go, q0, q1, wk.i0, wk.j0, dst, smem, direct && !zeroed, dbg, through && !direct);
        else grad_w_tile_rows<LDS_TILES, 4>(x, g, D, xs, xo, go, q0, q1, wk.i0, wk.j0, dst, smem, direct && !zeroed, dbg, through && !direct);
        (void)gs; (void)out;
    } else if constexpr (MODE == LD_FAST)      // whole K-steps, D % 64 == 0: deep LDS-DMA pipeline
        // (a form with NO LDS -- every MFMA operand one coalesced global_load_dword into its register, four
        // register buffers -- measured slower: 32 dword loads per 16 MFMAs cost more issue time than the ring's
        // four DMA pieces, a whole tile took 16.5 us against 14.5)
        grad_w_tile_dma(x, g, D, D, xs, xo, gs, go, q0, (int)((q1 - q0) / GT_BK), wk.i0, wk.j0, dst, smem,
                        direct && !zeroed, dbg);
    else
        tmpl_grad_w_tile<MODE>(x, g, out, D, D, 0, xs, xo, gs, go, q0, q1, wk.i0, wk.j0, dst, smem, direct && !zeroed);
}

// partial vectors. kind 0: column sums of gpre over 64-row blocks of (batch, level).
// kind 1: sums of gH[0] variable row k over 64-graph blocks of the batch; 4 row groups x 64 columns per
// workgroup in both kinds.
__device__ __forceinline__ void vec_partial_block(const StepDev *__restrict__ sd, const VSource *__restrict__ src,
                                                  int nsrc, const int *__restrict__ block_start,
                                                  const float *__restrict__ H, const float *__restrict__ GH,
                                                  long long level_stride, float *__restrict__ partial, int bid,
                                                  float *smem) {
    float(*part)[64] = reinterpret_cast<float(*)[64]>(smem);
    const int D = sd->D;
    const int si = block_start[bid];
    const VSource s = src[si];
    const int lb = bid - s.block_start;
    const BatchDev &b = sd->b[s.batch];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int cchunks = (D + 63) / 64;
    const int blk = lb / cchunks, c = (lb % cchunks) * 64 + cl;
    float acc = 0.f;
    if (s.kind == 0) {
        const long long rows = (long long)b.B * b.tp.N;
        const float *g = GH + (long long)(s.level_or_k + 1) * level_stride + b.row_off * D;
        const float *o = H + (long long)(s.level_or_k + 1) * level_stride + b.row_off * D;
        const long long r0 = (long long)blk * CH_GB * b.tp.N;      // one partial row per 16 graphs (a chain block)
        const long long r1 = r0 + (long long)CH_GB * b.tp.N;
        const unsigned live = b.live[s.level_or_k + 1];
        const int N = b.tp.N;
        if (c < D)
            for (long long r = r0 + rg; r < r1 && r < rows; r += 4) {
                if (!((live >> (int)(r % N)) & 1u)) continue;      // rows the step never wrote: zero gradient
                float v = g[r * D + c];
                if (s.relu && !(o[r * D + c] > 0.f)) v = 0.f;
                acc += v;
            }
    } else {
        const float *g = GH + b.row_off * D;          // level 0
        const int k = s.level_or_k;
        const long long g0 = (long long)blk * CH_GB;
        if (c < D)
            for (long long gi = g0 + rg; gi < g0 + CH_GB && gi < b.B; gi += 4)
                acc += g[(gi * b.tp.N + b.A + k) * D + c];
    }
    part[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < D)
        partial[(long long)(s.part_start + blk) * D + c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// anchor rows of gH[0] through the L2 normalisation into the entity-table gradients (fp32 atomics:
// an entity can occur in several graphs)
__device__ __forceinline__ void anchor_bwd_block(const StepDev *__restrict__ sd, const TablePtrs &tabs,
                                                 const long long *__restrict__ node_map, long long map_len,
                                                 const long long *__restrict__ anchor_ids,
                                                 const float *__restrict__ G0,
                                                 const int *__restrict__ anchor_row_off, int nb, int bid) {
    const long long w = (long long)bid * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= anchor_row_off[nb]) return;
    const int bi = find_le(anchor_row_off, nb + 1, (int)w);
    const BatchDev &b = sd->b[bi];
    const long long lr = w - anchor_row_off[bi];
    const int n = (int)(lr / b.B);
    const long long g = lr - (long long)n * b.B;
    const int D = sd->D, tab = b.anchor_tab[n];
    float *gt = tabs.grad[tab];
    if (!gt || !((b.live[0] >> n) & 1u)) return;
    const long long row = table_row(node_map, map_len, anchor_ids[b.anchor_off + lr], tabs.rows[tab], nullptr);
    if (row < 0) return;
    const float *v = tabs.table[tab] + row * D;
    const float *gi = G0 + (b.row_off + g * b.tp.N + n) * D;
    float ss = 0.f, vg = 0.f;
    for (int c = lane; c < D; c += 64) {
        ss += v[c] * v[c];
        vg += v[c] * gi[c];
    }
    ss = wave_sum(ss);
    vg = wave_sum(vg);
    const float nrm = sqrtf(ss), inv = 1.f / nrm, ydotg = vg * inv;
    for (int c = lane; c < D; c += 64) atomicAdd(gt + row * D + c, (gi[c] - (v[c] / nrm) * ydotg) * inv);
}

// A relation matrix of the gradient that nothing writes this step (with MPQE_STEP_ZERO_GRADS it must read zero
// afterwards): zero-filled by spare workgroups of the weight-gradient launch instead of the step's prologue.
// Backward tail: weight-gradient tiles, bias / variable-row partial sums and anchor-table gradients
// all depend only on H and gH and write disjoint outputs, so they share ONE launch (a role per block
// range, heavy MFMA tiles first) instead of three half-empty ones.
// out += sum of the group's slabs / partial rows. A workgroup owns 256 consecutive elements (4 per
// lane, 16-byte loads); its 4 waves each add every 4th slab (two loads in flight), the four sums
// are combined as (0+1)+(2+3): a fixed order. (A one-thread-per-16-elements variant that walked all
// slabs serially measured 2.5x slower: the 40-slab root group became the long pole.)
#ifndef VEC_SLICES
#define VEC_SLICES 4     // column slices (workgroups) per vector group of the reduction
#endif
struct ReduceArgs {
    const RGroup *groups;
    int ngroups, D;
    GradPtrs gp;
    const float *slabs, *partial;
    int vec, zeroed;
    const StepDev *sd;
    const float *terms;
    float *loss;
    LossMeta lm;
    const float *bterms;
    const Rank1 *rank1;
    const float *VT;
    unsigned *epoch_b;
    const char *touch;
    size_t touch_keys, touch_perm;
    const float *DG;
    TablePtrs tabs;
    int table_store;
    long long touch_M;
    int touch_row_bits;
    int32_t *err;
    // fused tail (the reduction as trailing workgroups of the weight-gradient launch): the groups and the loss workgroup
    // wait until `arrive` has counted the launch's `phase1` tile and vector-op workgroups; NULL: a launch of its own
    const unsigned *arrive;
    unsigned phase1;
    int rows_multi;          // 1: the entity-table workgroups take a range of sorted positions each (table_sum_multi)
    int nmat;                // >= 0: the launch's rows are packed (step_reduce_kernel): the first nmat groups are the matrix groups
    const int *runs;         // != NULL: the plan's run starts, compacted by a role of the weight-gradient launch
                             // (touch_runs_block): runs[0 .. runs[touch_M]) -- the table workgroups take those, not every position
    int early;               // 1: the loss and the entity-table rows were roles of the weight-gradient launch (TailArgs.extra0):
                             // the loss workgroup here only closes the step (epochs, the sort's barrier word, the plan's failure flag)
    unsigned *notify;        // mpqe_step_extra_t.notify (pinned host words) or NULL; written by the loss workgroup
    unsigned notify_value;
};
// workgroup (bx, by) of the reduction: by < ngroups: 256 elements of group by (gx workgroups along x); by == ngroups: the
// loss (bx 0); beyond: entity-table rows
__device__ __forceinline__ void reduce_block(const ReduceArgs &ra, int bx, int by, int gx, f32x4 (*part)[64]) {
    const RGroup *__restrict__ groups = ra.groups;
    const int ngroups = ra.ngroups, D = ra.D, vec = ra.vec, zeroed = ra.zeroed, table_store = ra.table_store;
    const GradPtrs &gp = ra.gp;
    const float *__restrict__ slabs = ra.slabs, *__restrict__ partial = ra.partial, *__restrict__ VT = ra.VT;
    const Rank1 *__restrict__ rank1 = ra.rank1;
    unsigned *epoch_b = ra.epoch_b;
    const char *__restrict__ touch = ra.touch;
    // zeroed: this call zero-filled the gradients, so `out` is known to be 0 -- a store replaces the
    // read-modify-write (whose read would be one more dependent round trip at the end of the chain)
    if (by > ngroups) {        // further rows: entity-table gradients, per destination row (step_touch.h).
        // (As workgroups of the weight-gradient launch they are throttled to two per CU by its 64 KB of LDS: 23.6 us
        // for that launch instead of 16.6; here they cost 2.6 us.)
#if MPQE_HAS_EXPERIMENTS
        if (ra.rows_multi) {
            // a RANGE of sorted positions per workgroup (step_touch.h: table_sum_multi): 344 workgroups for the AIFB step's
            // 22 016 ids, all resident at once, instead of 2 752 one-run workgroups in two and a half rounds of the chip
            static_assert(sizeof(f32x4) * 4 * 64 >= TSM_LDS_WORDS(64) * 4, "table_sum_multi's window lives in the reduction's LDS");
            table_sum_multi(ra.touch_M, ra.touch_row_bits, reinterpret_cast<const tkey_t *>(touch + ra.touch_keys),
                            reinterpret_cast<const int *>(touch + ra.touch_perm), ra.DG, D, ra.tabs, table_store & 1,
                            (long long)(by - ngroups - 1) * gx + bx, &reinterpret_cast<const TouchHeader *>(touch)->pad[0],
                            reinterpret_cast<unsigned *>(part));
            return;
        }
#endif
        table_sum_block(ra.touch_M, ra.touch_row_bits, reinterpret_cast<const tkey_t *>(touch + ra.touch_keys),
                        reinterpret_cast<const int *>(touch + ra.touch_perm), ra.DG, D, ra.tabs, table_store & 1,
                        (long long)(by - ngroups - 1) * gx + bx, &reinterpret_cast<const TouchHeader *>(touch)->pad[0],
                        ra.runs, ra.runs ? ra.runs + ra.touch_M : nullptr);
        return;
    }
    // fused tail: what follows reads what tiles / vector ops of THIS launch wrote (slabs and the post-pass' last vectors and
    // rows of `parts`, all written through) or must come after their last read of the epochs. One lane polls the arrival
    // counter (agent scope, bounded), then the workgroup's barrier. No acquire fence: nothing on this XCD has read those
    // lines before in this launch. Everything that does NOT depend on them -- the group's record, its rank-1 records, the u
    // vectors (pre-pass), the old value -- is requested before the wait.
    auto wait_phase1 = [&]() {
        if (ra.arrive) {
            if (threadIdx.x == 0) {
                // (hundreds of workgroups wait on ONE word: polled every ~1.5 us while more than a few arrivals are missing --
                // at one poll per 0.25 us each they saturated the word's L2 channel and the post-pass next to them took
                // 31 us instead of 15 -- and quickly only for the last few)
                for (int spins = 0;; ++spins) {
                    const unsigned have = uop_poll(ra.arrive);
                    if ((int)(have - ra.phase1) >= 0) break;
                    if (spins >= UOP_SPIN_LIMIT) {
                        flag_error(ra.err, MPQE_FLAG_INTERNAL | 0x1000);
                        break;
                    }
#ifndef MPQE_EMU
                    if (ra.phase1 - have > 3u) __builtin_amdgcn_s_sleep(48);
                    else __builtin_amdgcn_s_sleep(2);
#endif
                }
            }
            __syncthreads();
        }
    };
    if (by == ngroups) {       // one extra workgroup row: the loss reduction rides along
        if (bx == 0) {
            wait_phase1();
            // the step is over: the next step's granules (forward pre-pass, backward post-pass) get new tags, and the
            // count of finished transpose workgroups a new target (step_uniform.h, step_chain.h)
            if (epoch_b && threadIdx.x == 0) {
                *epoch_b = *epoch_b + 1u;
                *(epoch_b - 16) = *(epoch_b - 16) + 1u;       // epoch_f
                if (table_store & 2) *(epoch_b + 32) = *(epoch_b + 32) + 1u;      // merged launch: its own epoch (DoneMeta)
                *(epoch_b + 24) = 0u;       // the grid barrier of the next step's in-launch sort starts from zero (step_touch.h)
            }
            // a touch plan whose build could not finish (its workgroups were not all resident: step_touch.h): the table rows
            // above stored nothing; the caller rebuilds the plan and sums them again (mpqe_step_table_rows)
            if (touch && threadIdx.x == 0 && reinterpret_cast<const TouchHeader *>(touch)->pad[0]) flag_error(ra.err, MPQE_FLAG_TOUCH_RETRY);
            // (every launch that reads the ids is over: the chain / tail launches come before this one in stream order)
            if (ra.notify && threadIdx.x == 0) {
                ra.notify[1] = ra.err ? (unsigned)*ra.err : 0u;
#ifndef MPQE_EMU
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#endif
                ra.notify[0] = ra.notify_value;
            }
            if (ra.early) return;
            if (ra.lm.chain) loss_block_chain(ra.lm, ra.bterms, ra.loss, reinterpret_cast<float *>(part), 4);
            else loss_block(ra.sd, ra.terms, ra.loss, reinterpret_cast<float *>(part), 4);
        }
        return;
    }
    const RGroup g = groups[by];
    const bool wide = g.kind >= 4;          // a column block of a [D, 2 D] matrix: rows 2 D apart
    const long long elems = (g.kind <= 1 || wide) ? (long long)D * D : D;
    if (g.kind >= 2 && !wide && vec && (256 % (D / 4)) == 0) {
        // a vector group (bias / mode row): hundreds of partial rows of D floats (one per chain block), ONE
        // workgroup: D/4 lanes cover a row, the 256 / (D/4) row groups each walk every RG-th row with 8 loads
        // in flight, then the row groups' sums are added in order (fixed order: reproducible)
        // (VEC_SLICES > 1: workgroup bx takes the columns [bx D / VEC_SLICES, ...) of every row -- 128 bytes of a row at
        // D = 128: more rows in flight per workgroup, VEC_SLICES workgroups per group; a fixed order all the same)
        // (a row of the unpacked grid has gx = ceil(D D / 256) workgroups: D = 16 has ONE -- no slices there, or the columns
        // beyond the first slice were never summed)
        const int NS = (D % (4 * VEC_SLICES) == 0 && 256 % (D / 4 / VEC_SLICES) == 0 && (ra.nmat >= 0 || gx >= VEC_SLICES)) ? VEC_SLICES : 1;
        if (bx >= NS) return;
        wait_phase1();
        const int LQ = D / 4 / NS, RG = 256 / LQ;
        const int c4 = bx * LQ + threadIdx.x % LQ, rg = threadIdx.x / LQ;
        const float *pv = partial + (long long)g.start * D + 4 * c4;
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
        for (int i = rg; i < g.count; i += RG * 8) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = i + RG * q;
                v[q] = gload4(pv + (long long)(k < g.count ? k : i) * D);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i + RG * q < g.count) acc4 += v[q];
        }
        f32x4 *flat = &part[0][0];
        flat[threadIdx.x] = acc4;
        __syncthreads();
        // the row groups' sums: groups rg, rg + 4, ... into four, then the four (two short chains instead of one long one)
        const int cl = threadIdx.x % LQ;
        f32x4 t = acc4;
        if (rg < 4)
            for (int q = rg + 4; q < RG; q += 4) t += flat[q * LQ + cl];
        __syncthreads();
        if (rg < 4) flat[threadIdx.x] = t;
        __syncthreads();
        if (rg != 0) return;
        float *dstv = g.kind == 2 ? gp.bias[g.layer] : (gp.mode_emb ? gp.mode_emb + g.row * D : nullptr);
        if (!dstv) return;
        for (int q = 1; q < 4 && q < RG; ++q) t += flat[q * LQ + cl];
#pragma unroll
        for (int k = 0; k < 4; ++k) dstv[4 * c4 + k] = zeroed ? t[k] : dstv[4 * c4 + k] + t[k];
        return;
    }
    const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const long long idx = ((long long)bx * 64 + el) * 4;
    if ((long long)bx * 256 >= elems) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float *p = ((g.kind <= 1 || wide) ? slabs : partial) + (long long)g.start * elems + idx;
    float *dst;
    if (g.kind == 0) dst = gp.basis[g.layer] ? gp.basis[g.layer] + g.row * elems : nullptr;
    else if (g.kind == 1) dst = gp.root[g.layer];
    else if (wide) {
        // (element idx of the block = row idx / D, column idx % D of it: `dst + idx` then IS its address)
        const long long cb = g.row & 255, nbk = g.row >> 8;
        dst = gp.root[g.layer] ? gp.root[g.layer] + (idx / D) * (nbk - 1) * D + cb * D : nullptr;
    }
    else if (g.kind == 2) dst = gp.bias[g.layer];
    else dst = gp.mode_emb ? gp.mode_emb + g.row * D : nullptr;
    // rank-1 terms of a matrix group (sources whose input state is one vector per batch: out[i][j] += u[i] v[j], v = the
    // column sum of the destination's gradient rows; chain form, D % 64 == 0). Wave sg takes terms sg, sg + 4, ... in
    // order into its partial sum: their records are requested together, then their u / v pieces together -- two round
    // trips next to the slab loads whatever the count (a loop of dependent record -> vector loads per term, and then a
    // staged version with two workgroup barriers per eight terms, were the launch's long pole).
    const bool r1 = g.kind <= 1 && g.r1_count > 0 && vec && idx + 3 < elems && !(STEP_DBG & 2);
    const int ri = (int)(idx / D), rj = (int)(idx % D);
    Rank1 rk0[R1_CHUNK];
    float u0[R1_CHUNK];
    if (r1) {       // first chunk of this wave's terms: records, then the u pieces (vectors of the forward pre-pass)
#pragma unroll
        for (int q = 0; q < R1_CHUNK; ++q) rk0[q] = rank1[g.r1_start + (sg + 4 * q < g.r1_count ? sg + 4 * q : (sg < g.r1_count ? sg : 0))];
#pragma unroll
        for (int q = 0; q < R1_CHUNK; ++q) u0[q] = gload1(VT + (long long)rk0[q].u * D + ri);
    }
    f32x4 old4 = {0.f, 0.f, 0.f, 0.f};       // accumulate mode: the old value travels with the other loads, not after them
    if (vec && dst && !zeroed && sg == 0 && idx + 3 < elems) old4 = gload4(dst + idx);
    wait_phase1();
    if (r1) {
        for (int t0 = sg; t0 < g.r1_count; t0 += 4 * R1_CHUNK) {
            Rank1 rk[R1_CHUNK];
            float u[R1_CHUNK];
            f32x4 v[R1_CHUNK];
            if (t0 == sg) {
#pragma unroll
                for (int q = 0; q < R1_CHUNK; ++q) {
                    rk[q] = rk0[q];
                    u[q] = u0[q];
                }
            } else {
#pragma unroll
                for (int q = 0; q < R1_CHUNK; ++q) rk[q] = rank1[g.r1_start + (t0 + 4 * q < g.r1_count ? t0 + 4 * q : t0)];
#pragma unroll
                for (int q = 0; q < R1_CHUNK; ++q) u[q] = gload1(VT + (long long)rk[q].u * D + ri);
            }
#pragma unroll
            for (int q = 0; q < R1_CHUNK; ++q) v[q] = gload4(VT + (long long)rk[q].v * D + rj);
#pragma unroll
            for (int q = 0; q < R1_CHUNK; ++q)
                if (t0 + 4 * q < g.r1_count) s += u[q] * v[q];
        }
    }
    if (vec) {
        if (idx < elems) {
            // four slabs of this wave in flight at a time (slab i, i+4, i+8, i+12; clamped loads, masked adds)
            for (int i = sg; i < g.count; i += 16) {
                f32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = i + 4 * q;
                    v[q] = gload4(p + (long long)(k < g.count ? k : i) * elems);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i + 4 * q < g.count) s += v[q];
            }
        }
    } else {
        for (int i = sg; i < g.count; i += 4)
            for (int k = 0; k < 4; ++k)
                if (idx + k < elems) s[k] += p[(long long)i * elems + k];
    }
    part[sg][el] = s;
    __syncthreads();
    if (sg != 0 || !dst) return;
    const bool have_old = vec && !zeroed && idx + 3 < elems;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (idx + k < elems) {
            const float sum = (part[0][el][k] + part[1][el][k]) + (part[2][el][k] + part[3][el][k]);
            dst[idx + k] = zeroed ? sum : (have_old ? old4[k] : dst[idx + k]) + sum;
        }
}

// (REDUCE_WAVES = 6 / 8: the launch's 92 VGPRs capped at 80 / 64 for six / eight instead of five workgroups per CU -- its
// 2 752 table-row workgroups are two and a half rounds of the chip -- measured: 9.8 / 11.1 us against 10.3, the step 0.3 -
// 1.2 us SLOWER: the spills land in the group workgroups, the launch's critical path)
#ifndef REDUCE_WAVES
#define REDUCE_WAVES 0
#endif
#if REDUCE_WAVES
__global__ __launch_bounds__(256, REDUCE_WAVES) void step_reduce_kernel(ReduceArgs ra) {
#else
__global__ __launch_bounds__(256) void step_reduce_kernel(ReduceArgs ra) {
#endif
    __shared__ f32x4 part[4][64];
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (ra.nmat >= 0) {
        // packed rows: [0, nmat) the matrix groups (gx workgroups each); row nmat: every vector group's column slices side
        // by side, then the loss; beyond: the entity-table rows -- a row of gx workgroups per VECTOR group left all but its
        // first few without work (9 x 60 of them on the AIFB step, dispatched in front of the table workgroups)
        if (by == ra.nmat) {
            const int g = ra.nmat + bx / VEC_SLICES;
            if (g < ra.ngroups) {
                by = g;
                bx = bx % VEC_SLICES;
            } else if (bx == (ra.ngroups - ra.nmat) * VEC_SLICES) {
                by = ra.ngroups;
                bx = 0;
            } else return;
        } else if (by > ra.nmat) by += ra.ngroups - ra.nmat;
    }
    reduce_block(ra, bx, by, (int)gridDim.x, part);
}
struct TailArgs {
    const WSource *wsrc;
    const WBlock *wblock;
    int nwsrc, wblocks;
    const VSource *vsrc;
    const int *vblock;
    int nvsrc, vblocks;
    const int *anchor_off;
    int nb;
    long long *stamps;       // diagnostics (mpqe_debug_tail_stamps): 8 words per workgroup, or NULL
    const ZMat *zmats;       // untouched gradient matrices, zero-filled by workgroups [wblocks, wblocks + zblocks)
    int zblocks, zper;       // zper = workgroups per matrix
    int ublocks;             // the backward post-pass of the uniform node states: the FIRST ublocks workgroups
    int D;                   // = sd->D, by value: a tile's record is then the first and only load in front of its rows
    int tile_n;              // columns per weight-gradient tile
    int ux;                  // > 0: XCDs set aside for the post-pass' vector ops (step_tail_kernel)
    int runs_front, runs_n;  // > 0: the launch's first runs_front workgroups (runs_n of them at work) compact the touch plan's run
    int *runs_out;           // starts (touch_runs_block) for the reduction launch's table workgroups: runs_out[0 .. M) the
                             // positions, runs_out[M] their number
    int extra0;              // >= 0: workgroups [extra0, ...) of the launch are roles that read only what the CHAIN launch wrote --
    int tm_blocks;           // [extra0] the loss (loss_block_chain), then tm_blocks entity-table workgroups (table_sum_multi):
                             // they were 2 800 + 1 workgroups of the reduction launch; here they run beside the tiles
    ClosureArgs ca;          // ca.ncl > 0: the post-pass as closures (step_closure.h) -- the launch's FIRST ncl workgroups, padded
    int clpad;               // to clpad (a multiple of 8: tile b keeps XCD b % 8); ublocks is 0 then

    const long long *node_map;
    long long map_len;
    const long long *anchor_ids;
    float *slabs, *parts;
};

// workgroup zb of the zero fill of the relation matrices nobody writes this step (zper workgroups per matrix)
__device__ __forceinline__ void zmat_block(const ZMat *__restrict__ zmats, int zper, int zb, int D, const GradPtrs &gp) {
    const ZMat zm = zmats[zb / zper];
    float *base = pick_grad(gp.basis, zm.layer);
    if (!base) return;
    const long long elems = (long long)D * D;
    float *p = base + zm.rel * elems;
    const long long lo = (long long)(zb % zper) * ZMAT_FLOATS_PER_BLOCK;
    for (long long i = lo + threadIdx.x * 4; i < lo + ZMAT_FLOATS_PER_BLOCK && i < elems; i += 1024) {
        if (i + 3 < elems && ((uintptr_t)(p + i) & 15) == 0) *reinterpret_cast<f32x4 *>(p + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        else
            for (long long q = i; q < i + 4 && q < elems; ++q) p[q] = 0.f;
    }
}

// post roles of the merged chain launch (declared with PostArgs, in front of step_chain_kernel)
template <int LDS_TILES>
__device__ __forceinline__ void post_block(const StepDev *__restrict__ sd, const LayerPtrs &lp, const PostArgs &po, int pb,
                                           float *smem) {
    if (pb < po.zmblocks) {
        zmat_block(po.zmats, po.zper, pb, po.D, po.gp);
    } else if (pb < po.zmblocks + po.ublocks) {
        uop_block(pb - po.zmblocks, po.D, lp, po.ub, smem, &po.gp, po.zeroed);
    } else if (pb >= po.ppad && pb < po.ppad + po.wblocks) {
        const int tb = pb - po.ppad;
        long long *dbg = po.stamps ? po.stamps + (long long)tb * 8 : nullptr;
#ifndef MPQE_EMU
        long long tick0 = 0;
        if (dbg && threadIdx.x == 0) {
            tick0 = (long long)__builtin_amdgcn_s_memtime();
            dbg[0] = (long long)wall_clock64();
            dbg[3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                     ((long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
        }
#endif
        grad_w_block<LD_T, LDS_TILES>(sd, nullptr, 0, po.wblock, po.H, po.GH, po.level_stride, po.slabs, tb, po.wblocks,
                                      smem, po.gp, po.zeroed != 0, dbg, po.D, &po, po.tile_n);
#ifndef MPQE_EMU
        if (dbg && threadIdx.x == 0) {
            dbg[1] = (long long)wall_clock64();
            dbg[2] = (long long)__builtin_amdgcn_s_memtime() - tick0;
        }
#endif
    }       // (else: padding)
}

// FUSED (chain form, LD_T; diagnostics switch FUSE_TAIL, off by default): the step's reduction rides in this launch --
// `fa.first` workgroups of tiles / vector ops / zero fill as before, then the reduction's workgroups: entity-table rows and
// the loss (they read what the chain launch wrote) and, waiting for the arrival counter of the tiles and vector ops, the
// reduction groups: two launches per step instead of three. Built, parity-tested (tests/test_step.py), and SLOWER on the
// AIFB step -- 32.3 us against 19.8 + 10.2 -- for two measured reasons: (1) every workgroup of a launch has the launch's
// register footprint, the tile's 228 VGPRs = two workgroups per CU whatever their LDS, so ~2 800 table-row and ~400
// group workgroups queue for the ~270 slots the tiles and vector ops leave (and the waiting groups hold some); (2) the
// post-pass outputs the reduction reads must be written through to reach another XCD inside a launch, and those
// agent-scope stores stretch the post-pass' dependence chain from 15.7 to 20.6 us. DESIGN.md 4.2 (round 3).
struct FuseArgs {
    int first;              // workgroups in front of the reduction's (the un-fused launch's grid); 0: not fused
    int gx, trows;          // the reduction's grid: gx workgroups per group, trows rows of gx table-row workgroups
    int tx;                 // the table-row workgroups are dealt to the first tx XCDs only (8: all): not where the post-pass runs
    int tspan;              // workgroups of the launch the table rows take (holes included)
    unsigned *arrive;       // arrival counter (zeroed by the chain launch)
};
template <int MODE, bool FUSED = false>
__global__ __launch_bounds__(256) void step_tail_kernel(const StepDev *__restrict__ sd, TailArgs ta,
                                                        const float *__restrict__ H, const float *__restrict__ GH,
                                                        long long level_stride, GradPtrs gp, int zeroed, LayerPtrs lp,
                                                        UArgs ua, FuseArgs fa, ReduceArgs ra) {
    // weight-gradient tiles only: the DMA ring takes 64 KB of LDS per workgroup, which would throttle the
    // thousands of light partial-sum / anchor workgroups to 2 per CU if they shared this kernel
    // (LD_T, the chain form: the tiles meet in a 17 KB LDS tile at their end; the post-pass' vector ops use 8 KB)
    __shared__ __attribute__((aligned(16))) float smem[MODE == LD_T ? (FUSED ? GWR_SMEM_FLOATS2 : GWR_SMEM_FLOATS) : (MODE == LD_FAST ? GWD_SMEM_FLOATS : GT_SMEM_FLOATS)];
    // The launch's FIRST ta.runs_front workgroups (a multiple of 8: workgroup b of the rest keeps XCD b % 8) compact the run
    // starts of the step's touch plan for the reduction launch's table workgroups (touch_runs_block): they depend on the chain
    // launch alone and are through before the first tile has its rows
    int bid = (int)blockIdx.x;
    if (ta.runs_front > 0) {
        if (bid < ta.runs_front) {
            if (bid < ta.runs_n)
                touch_runs_block(ra.touch_M, reinterpret_cast<const tkey_t *>(ra.touch + ra.touch_keys), ta.runs_out,
                                 ta.runs_out + ra.touch_M, &reinterpret_cast<const TouchHeader *>(ra.touch)->pad[0],
                                 reinterpret_cast<int *>(smem), bid);
            return;
        }
        bid -= ta.runs_front;
    }
    if constexpr (FUSED) {
        if (bid >= fa.first) {
            int p = bid - fa.first;
            const int T = fa.tspan;
            int bx, by;
            if (p < T) {
                if ((p & 7) >= fa.tx) return;            // (a hole: this XCD is the post-pass')
                p = (p >> 3) * fa.tx + (p & 7);
                if (p >= fa.trows * fa.gx) return;
                bx = p % fa.gx, by = ra.ngroups + 1 + p / fa.gx;
            } else if (p == T) bx = 0, by = ra.ngroups;
            else bx = (p - T - 1) % fa.gx, by = (p - T - 1) / fa.gx;
            reduce_block(ra, bx, by, fa.gx, reinterpret_cast<f32x4(*)[64]>(smem));
            return;
        }
    }
#if MPQE_HAS_EXPERIMENTS
    if (ta.extra0 >= 0 && bid >= ta.extra0) {
        // roles that depend on the chain launch alone: the loss of the step, the entity-table rows (step_touch.h)
        const int e = bid - ta.extra0;
#ifndef MPQE_EMU
        if (ta.stamps && threadIdx.x == 0) ta.stamps[(long long)bid * 8 + 0] = (long long)wall_clock64();
#endif
        if (e == 0) {
            loss_block_chain(ra.lm, ra.bterms, ra.loss, smem, 4);
        } else if (e - 1 < ta.tm_blocks) {
            static_assert(sizeof(smem) >= TSM_LDS_WORDS(64) * 4, "table_sum_multi's window lives in the launch's LDS");
            table_sum_multi(ra.touch_M, ra.touch_row_bits, reinterpret_cast<const tkey_t *>(ra.touch + ra.touch_keys),
                            reinterpret_cast<const int *>(ra.touch + ra.touch_perm), ra.DG, ta.D, ra.tabs, ra.table_store & 1,
                            (long long)(e - 1), &reinterpret_cast<const TouchHeader *>(ra.touch)->pad[0],
                            reinterpret_cast<unsigned *>(smem));
        }
#ifndef MPQE_EMU
        if (ta.stamps && threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ta.stamps[(long long)bid * 8 + 5] = (long long)wall_clock64();
            ta.stamps[(long long)bid * 8 + 6] = 1 + 6;            // kind 6: loss / entity-table rows
        }
#endif
        return;
    }
#endif
    // (fused: a tile / vector-op workgroup counts itself in once its stores -- written through -- are acknowledged)
    auto arrived = [&]() {
        if constexpr (FUSED) {
#ifndef MPQE_EMU
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(fa.arrive, 1u);
        }
    };
#ifndef MPQE_EMU
    long long tick0 = 0;
    if (ta.stamps && threadIdx.x == 0) {
        tick0 = (long long)__builtin_amdgcn_s_memtime();        // shader-clock ticks: word 2 = ticks start -> end
        ta.stamps[(long long)bid * 8 + 0] = (long long)wall_clock64();
        ta.stamps[(long long)bid * 8 + 3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                   ((long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
    }
#endif
    // role of the workgroup. ta.ux > 0 (chain form): the post-pass' vector ops take the last ta.ux XCDs of the chip and the
    // tiles the others (workgroup b runs on XCD b % 8) -- the vector ops are a latency chain of small loads and polls, the
    // tiles stream ~40 MB through their XCDs' L2s and fabric ports
    int ub = -1, tb;
#if MPQE_HAS_EXPERIMENTS
    if (ta.ca.ncl > 0) {       // the post-pass as closures: the launch's first workgroups, one per batch (step_closure.h)
        if (bid < ta.clpad) {
            if (bid < ta.ca.ncl) {
#ifndef MPQE_EMU
                __builtin_amdgcn_s_setprio(3);      // a latency chain next to throughput work
#endif
                closure_block(bid, ta.D, ta.ca, lp, ua, smem, gp, zeroed, ta.stamps ? ta.stamps + (long long)bid * 8 : nullptr);
#ifndef MPQE_EMU
                if (ta.stamps && threadIdx.x == 0) {
                    ta.stamps[(long long)bid * 8 + 5] = (long long)wall_clock64();       // (word 1 stays 0: not a tile)
                    ta.stamps[(long long)bid * 8 + 6] = 1 + 5;                           // kind 5: a closure
                }
#endif
            }
            return;
        }
        tb = bid - ta.clpad;
    } else
#endif
    if (ta.ux > 0) {
        const int x = bid & 7, r = bid >> 3, tx = 8 - ta.ux;
        if (x >= tx) {
            ub = r * ta.ux + (x - tx);
            if (ub >= ta.ublocks) return;
        }
        tb = r * tx + x;
    } else {
        if (bid < ta.ublocks) ub = bid;
        tb = bid - ta.ublocks;
    }
    if (ub >= 0) {         // uniform node states, backward: vector ops on column sums
        uop_block(ub, sd->D, lp, ua, smem, &gp, zeroed);
#ifndef MPQE_EMU
        if (ta.stamps && threadIdx.x == 0) {
            ta.stamps[(long long)bid * 8 + 5] = (long long)wall_clock64();       // (word 1 stays 0: not a tile)
            ta.stamps[(long long)bid * 8 + 6] = 1 + (long long)ua.ops[ub / ua.chunks].kind;
        }
#endif
        arrived();
        return;
    }
    if (tb >= ta.wblocks) {        // zero fill of a gradient matrix nobody writes (uniform branch)
        if (tb - ta.wblocks < ta.zblocks) zmat_block(ta.zmats, ta.zper, tb - ta.wblocks, sd->D, gp);
        return;
    }
    grad_w_block<MODE, (FUSED ? 2 : 4)>(sd, ta.wsrc, ta.nwsrc, ta.wblock, H, GH, level_stride, ta.slabs, tb, ta.wblocks,
                       smem, gp, zeroed != 0, ta.stamps ? ta.stamps + (long long)bid * 8 : nullptr, ta.D, nullptr, ta.tile_n,
                       ta.ux > 0 ? 8 - ta.ux : 8, FUSED);      // zeroed: this call zero-filled the gradients, a store suffices
    arrived();
#ifndef MPQE_EMU
    if (ta.stamps && threadIdx.x == 0) {
        ta.stamps[(long long)bid * 8 + 1] = (long long)wall_clock64();
        ta.stamps[(long long)bid * 8 + 2] = (long long)__builtin_amdgcn_s_memtime() - tick0;
    }
#endif
}

// bias / variable-row partial sums and anchor-table gradients: light, latency-bound roles in one launch
__global__ __launch_bounds__(256) void step_tail_small_kernel(const StepDev *__restrict__ sd, TailArgs ta,
                                                              TablePtrs tabs, const float *__restrict__ H,
                                                              const float *__restrict__ GH,
                                                              long long level_stride) {
    __shared__ float smem[4 * 64];
    const int bid = blockIdx.x;
    if (bid < ta.vblocks)
        vec_partial_block(sd, ta.vsrc, ta.nvsrc, ta.vblock, H, GH, level_stride, ta.parts, bid, smem);
    else
        anchor_bwd_block(sd, tabs, ta.node_map, ta.map_len, ta.anchor_ids, GH, ta.anchor_off, ta.nb,
                         bid - ta.vblocks);
}



// Fused training step: forward + backward of the whole query-graph encoder for ALL batches of
// one step (the reference's post-burn-in step draws 11 formula batches: train_helpers.py:81,
// 97-112; each goes through RGCNEncoderDecoder.margin_loss, model.py:464-494) in ~15 kernel
// launches instead of one launch per op per batch.
//
// At the reference's batch size (B = 512, <= 4 nodes per graph, D = 128) every single op is
// far too small to fill 256 CUs and the path is launch-latency bound. So the unit of a launch
// here is a LEVEL of the whole step: level p applies message-passing pass p of every batch
// that still has a pass to run (batches differ in template, relations and number of passes),
// as one grid of 64x64 MFMA tiles described by a small descriptor table in HBM.
//
//   forward   assemble (anchors: gather + L2 normalise; variables: mode rows; +/- targets)
//             level 0 .. Lmax-1 layer tiles
//             readout + cosine(+/-) + hinge terms per graph;  loss reduction (one workgroup)
//   backward  d hinge -> d cosine -> d readout -> rows of gH[L_b]; target/negative table grads
//             level Lmax-1 .. 0 backward-x tiles
//             weight-gradient tiles of ALL levels (split over K chunks, slabs)
//             bias / variable-row partial sums; anchor table grads
//             one reduction pass: slabs and partials -> gradients, fixed order
//
// Arithmetic is identical to the per-op kernels (same tile bodies, rgcn_template_body.h);
// margin_loss's two encoder passes are one here (the query embedding does not depend on the
// target).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "grad_w_dma.h"
#include "rgcn_template_body.h"

#define STEP_MAX_LEVELS MPQE_STEP_MAX_LAYERS
#define CHAIN_MAX_GRAPHS (1 << 20)
#define UPLOAD_BYTES 2048

struct BatchDev {
    TmplArgs tp;
    int A, V, L, B;
    long long var_id[3];
    int anchor_tab[3];
    int target_tab;
    long long row_off, g_off, anchor_off;
    float weight;
    int pad;
    // live[p]: node slots whose state H[p] can reach the readout (bit n). Everything else is neither
    // computed nor read: its gradient is exactly zero (MPQE_STEP_NO_PRUNE: all slots at every level).
    unsigned live[MPQE_STEP_MAX_LAYERS + 1];
    int pad2;
    // chain form. uvL[n] >= 0: node slot n is still batch-uniform at level L; its state is the pre-pass vector with that
    // granule slot (the readout reads it in every row). lpart[n] >= 0: first row in `parts` of the per-block column sums of
    // gH[L][n] (row = lpart[n] + block index inside the batch).
    int uvL[4], lpart[4];
};

struct StepDev {
    int nb, D, num_layers, readout;
    long long rows_total, graphs_total;
    BatchDev b[MPQE_STEP_MAX_BATCHES];
};

struct LayerPtrs {
    const float *basis[MPQE_STEP_MAX_LAYERS], *root[MPQE_STEP_MAX_LAYERS], *bias[MPQE_STEP_MAX_LAYERS];
};
struct TablePtrs {
    const float *table[MPQE_STEP_MAX_MODES];
    float *grad[MPQE_STEP_MAX_MODES];
    long long rows[MPQE_STEP_MAX_MODES];
};

// Tiles of one (batch, node slot) at one level: rt*ct tiles of equal K length. Per-CU MFMA time is
// what bounds a level, and a level's workgroups are (almost always) all resident at once, dealt
// round-robin over the 256 CUs: block i shares its CU with blocks i+256, i+512, i+768 (measured,
// tools/micro/placement.hip; a speed assumption only, results do not depend on it). place_tiles() orders
// the table so that those per-CU sums are balanced (longest-processing-time-first bin packing).
struct TileGroup {
    int batch, node, tile_off, steps;
};
// One entry per workgroup of a level launch: a workgroup finds its work with ONE 8-byte load instead
// of a binary search over the groups (each probe is a dependent ~0.5 us scalar round trip that sits in
// front of the first MFMA; with launches this short that start-up cost is a visible share).
struct TileRef {
    short batch, node;
    int rem;            // tile index inside the (batch, node) group: row tile * ct + column tile
};

// one weight-gradient source: (batch, level, slot) -> nch K-chunks, each a slab of D*D floats
// one weight-gradient workgroup, everything it needs in one 64-byte record (the host resolved source, batch
// and template: three dependent loads in front of the first DMA otherwise)
struct WBlock {
    long long x_off, g_off;     // float offsets into H / gH: level base + the batch's first row
    long long slab_off;         // float offset of the chunk's slab
    long long rel;              // direct: relation id (-1 = root)
    int xs, xo, go;             // rows of graph q: x at (q * xs + xo), g at (q * xs + go)
    int q0, q1;                 // the K-chunk: graphs [q0, q1)
    int i0, j0;                 // the output tile
    int direct;                 // >= 0: layer whose gradient matrix the tile writes itself; -1: slab
    int d0, dn;                 // merged launch: the `done` counters [d0, d0 + dn) cover the chain workgroups of the K-chunk
    int batch, pad;             // pad = 1: the operands change places -- x rows from gH (at g_off), g rows from H (at x_off): the
                                // tile is then the gradient of an nn.Linear weight [out, in] (a learned readout on the chain)
};
// Merged launch (chain form): the weight-gradient tiles and the backward post-pass are workgroups of the CHAIN launch,
// behind the chain workgroups. A chain workgroup counts itself into the `done` counter of its group of DONE_GRAPHS graphs
// once its H / gH rows and column sums are out (release at agent scope); a tile / vector op waits for the counters of
// the graphs it reads. Counters only grow: target = (merged-launch epoch + 1) x (chain workgroups of the group).
#define STEP_XCDS_MAX 8
#define DONE_GRAPHS 128
struct DoneMeta {
    int base[MPQE_STEP_MAX_BATCHES + 1];       // counters of batch b: [base[b], base[b + 1])
};
struct WSource {
    int batch, level, slot, relu;
    int nch, ch, slab_start, block_start;
    // direct >= 0: this source is the ONLY contribution to its gradient matrix and one K-chunk: the tile adds
    // straight into the gradient (layer `direct`, relation `rel`, -1 = root), no slab, no reduction group
    int direct, pad;
    long long rel;
};
// one partial-vector source: kind 0 = bias colsum of (batch, level), kind 1 = variable row (batch, k)
struct VSource {
    int kind, batch, level_or_k, relu;
    int nblk, part_start, block_start, pad;
};
// one reduction group: out[...] += sum of `count` consecutive slabs/partials starting at `start`
// (+ matrices, chain form with uniform node states: the sum of r1_count rank-1 terms u (x) v from `r1_start` on)
struct RGroup {
    int kind;          // 0 basis, 1 root, 2 bias, 3 mode row; 4: a D x D column block of a [D, n D] matrix (`root` of `layer`:
                       // a learned readout's first Linear layer), row = block | n << 8
    int layer;         // layer index (kinds 0-2)
    long long row;     // relation id (kind 0) / mode id (kind 3)
    int start, count;
    int r1_start, r1_count;
};
struct Rank1 {
    int u, v;          // vector ids: out[i][j] += VT[u][i] * VT[v][j]
};
#ifndef R1_CHUNK
#define R1_CHUNK 8
#endif
#ifndef STEP_DBG
#define STEP_DBG 0      // timing experiments only (wrong results): 1 = no table-sum rows in the reduction launch,
#endif                  // 2 = no rank-1 terms, 3 = neither

// ---- batch-uniform node states (chain form; include/mpqe_amd.h: MPQE_STEP_NO_UNIFORM) --------------------------
// x0's variable rows are ONE mode_embeddings row for every graph of a batch (reference model.py:421), so a node state
// that no anchor has reached yet is one vector per batch. Per batch and level p the node slots split into uniform
// (U) and per-graph (NU) ones: U[0] = the variable slots, n in U[p+1] iff n and all sources of its in-edges are in
// U[p]. Consequences, all exact:
//   forward   a U node update is a matrix-VECTOR product chain, done once per batch (UOP_FWD, pre-pass, rides in the
//             prologue launch). An NU node's U sources add a constant vector to its pre-activation: the pre-pass
//             forms bias + that constant, the chain kernel's epilogue adds it where it added the bias.
//   backward  everything downstream of a U node's gradient needs only its COLUMN SUM over the batch (its inputs
//             are uniform, so its weight gradients are rank-1: u (x) colsum; its ReLU mask is uniform, so masking
//             commutes with the sum; bias / variable-row gradients are column sums anyway). The chain kernel leaves
//             per-block column sums of every NU node's gradient rows in `parts`; the post-pass (rides in the
//             weight-gradient launch) sums them over the blocks (UOP_RED) and runs the U nodes' backward as
//             vector-matrix^T products on those sums (UOP_BWD).
// Vectors live in the vector table VT [id][D] of the workspace; vectors that are handed from one workgroup to
// another INSIDE a launch also travel as {tag, value} granules (8 bytes, one agent-scope atomic store each: the data
// is its own flag -- MI355X guide, inter-workgroup visibility, form R2) in the packed step's descriptor buffer.
#define UOP_FWD 0      // out = act(bias[layer] + sum_t in_t . M_t)
#define UOP_BWD 1      // out = mask(VT[mask_vec] > 0) * sum_t in_t . M_t^T
#define UOP_RED 2      // out = sum of `nrows` consecutive rows of `parts` from row0
#define UOP_COPY 3     // out = mode_emb[mode_row]
#define UOP_R1 4       // gradient matrix (layer r1_layer, relation r1_rel | -1 root) = sum_t VT[u_vec[t]] (x) in_t: a matrix
                       // whose only contributions are rank-1 terms is written here, not by the reduction launch
#define UOP_MAX_TERMS 4
struct UOp {
    int kind, out_vec, out_gran;       // out_gran: granule slot of the output (-1: nobody reads it inside the launch)
    int out_part;                      // >= 0: the output is also written to this row of `parts` (a reduction group's input)
    int nterms;
    int in_vec[UOP_MAX_TERMS];         // vector id (in_kind 0: read from its granules; 2: plain, written by an earlier launch)
    int in_kind[UOP_MAX_TERMS];        // 0 granules, 1 row in_vec of mode_emb, 2 plain VT, 3 sum of in_gran rows of `parts` from row in_vec
    int in_gran[UOP_MAX_TERMS];
    int layer[UOP_MAX_TERMS], mat[UOP_MAX_TERMS];      // matrix: relation id or -1 = root, of layer `layer`
    int bias_layer, relu;              // FWD
    int mask_vec;                      // BWD: -1 = no mask
    int row0, nrows;                   // RED
    long long mode_row;                // COPY
    int u_vec[UOP_MAX_TERMS];          // R1
    int r1_layer, r1_rel;
    unsigned wait_mask;                // merged launch: bit b = the op reads rows the chain workgroups of batch b write
    int through;                       // fused tail: the op's output is read by the reduction (a rank-1 term's v, a row of
                                       // `parts`): written through, so that the same launch's reduction workgroups see it
};

struct Blob {
    char bytes[UPLOAD_BYTES];
};
__global__ void step_upload_kernel(Blob blob, char *dst, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = blob.bytes[i];
}

__device__ __forceinline__ int find_le(const int *__restrict__ off, int n, int t) {
    // largest i in [0, n) with off[i] <= t   (off non-decreasing, off[0] = 0)
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// lp.X[li] with a runtime li would spill the by-value pointer table to scratch; select instead
__device__ __forceinline__ const float *pick_layer(const float *const *arr, int li) {
    const float *r = arr[0];
#pragma unroll
    for (int l = 1; l < MPQE_STEP_MAX_LAYERS; ++l)
        if (l == li) r = arr[l];
    return r;
}

__device__ __forceinline__ int find_group_le(const TileGroup *__restrict__ g, int n, int t) {
    int lo = 0, hi = n - 1;     // largest i in [0, n) with g[i].tile_off <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (g[mid].tile_off <= t) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ int layer_index(int p, int L, int num_layers) {
    return p < L - 1 ? p : num_layers - 1;     // reference model.py:435-441
}

// ------------------------------------------------------------------------------------ assemble
__device__ __forceinline__ long long table_row(const long long *__restrict__ node_map, long long map_len,
                                               long long id, long long rows, int32_t *err) {
    if (id < 0 || id >= map_len) {
        flag_error(err, MPQE_FLAG_BAD_NODE_ID);
        return -1;
    }
    const long long r = node_map[id];
    if (r < 0 || r >= rows) {
        flag_error(err, MPQE_FLAG_BAD_NODE_ID);
        return -1;
    }
    return r;
}

struct GradPtrs {
    float *basis[MPQE_STEP_MAX_LAYERS], *root[MPQE_STEP_MAX_LAYERS], *bias[MPQE_STEP_MAX_LAYERS];
    float *mode_emb;
};
__device__ __forceinline__ float *pick_grad(float *const *arr, int li) {
    float *r = arr[0];
#pragma unroll
    for (int l = 1; l < MPQE_STEP_MAX_LAYERS; ++l)
        if (l == li) r = arr[l];
    return r;
}

#include "step_chain.h"
#include "step_uniform.h"
#if MPQE_HAS_EXPERIMENTS
#include "step_closure.h"
#else
struct ClosureArgs {       // (the post-pass as closures: an experiment, not in this build -- csrc/step_closure.h)
    int ncl;
};
struct ClBlock {
    int batch;
};
#endif
#include "step_touch.h"
#include "grad_w_reg.h"
#include "step_readout.h"
#define LD_T 3          // weight-gradient launch of the chain form: register-only K loop (grad_w_reg.h)

// Prologue roles of the chain launch (they were a launch of their own, 9 us in front of the chain kernel): the forward
// pre-pass of the batch-uniform node states (vector ops, step_uniform.h), transposed copies of the matrices the backward
// chains multiply by (64 x 64 pieces through LDS), zero fill of the gradient buffers (MPQE_STEP_ZERO_GRADS).
struct WtSlot {
    int layer, mat;       // mat < 0: root
    // a learned readout's first Linear layer may be WIDE (targetmlp: [D, 2 D]): the copy is of its D x D column block from
    // column col0 on, row length ld; plain: copied as it is (the backward chains multiply by the block itself), not transposed
    int col0, ld, plain;
};
#define PREP_MAX_SEGS 48
struct ZeroSegs {
    float *p[PREP_MAX_SEGS];
    long long n[PREP_MAX_SEGS];       // floats
    long long block0[PREP_MAX_SEGS + 1];   // first zero-fill workgroup of each segment
    int count;
};
#define PREP_ZERO_FLOATS_PER_BLOCK 8192      // 256 threads x 8 x float4
// Chain form: every chain block left the sum of its (<= 16) hinge terms in block_terms, and everything the
// reduction needs to know about the batches comes by value -- no dependent loads in front of the sums.
struct LossMeta {
    int nb, chain;
    int B[MPQE_STEP_MAX_BATCHES], blk_off[MPQE_STEP_MAX_BATCHES + 1];
    float weight[MPQE_STEP_MAX_BATCHES];
};
__device__ __forceinline__ void loss_block_chain(const LossMeta &lm, const float *__restrict__ bterms,
                                                 float *__restrict__ loss, float *mean, int nwaves) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int bi = w; bi < lm.nb; bi += nwaves) {
        float s = 0.f;
        for (int i = lm.blk_off[bi] + lane; i < lm.blk_off[bi + 1]; i += 64) s += bterms[i];
        s = wave_sum(s);
        if (lane == 0) {
            mean[bi] = s / (float)lm.B[bi];
            loss[1 + bi] = mean[bi];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int bi = 0; bi < lm.nb; ++bi) total += lm.weight[bi] * mean[bi];
        loss[0] = total;
    }
}

// Forward-only step in the chain form: the loss reduction, the epoch advance and the caller's notification ride in the chain
// launch itself -- every workgroup of the launch counts itself out as it ends, and the LAST one to end does what
// step_loss_kernel does as a launch of its own (one launch per margin_loss call of the drop-in entry points instead of two:
// ~2.5 us of host time and ~6 us of device time per call). The chain workgroups' block_terms are written through
// (agent-scope stores, acknowledged before the count) and read back the same way: no L2 write-back.
struct FinArgs {
    unsigned *count;          // workgroups of this launch that have ended (the last one leaves 0 behind); NULL: not this form
    float *loss;
    const float *bterms;
    unsigned *epoch_f;
    int bump_b;
    unsigned *notify;
    unsigned notify_value;
    const int32_t *err;
    const float *reg_norms;   // != NULL: loss[0] += reg_coef * *reg_norms (mpqe_step_extra_t.readout_norms: the readout's regulariser)
    float reg_coef;
    LossMeta lm;
};
__device__ __forceinline__ void chain_finish(const FinArgs &fin, float *smem) {
    __syncthreads();          // (every wave is through its role: the state buffers are free)
    unsigned *flag = reinterpret_cast<unsigned *>(smem);
    float *mean = smem + 16;
    if (threadIdx.x == 0) {
#ifndef MPQE_EMU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (this thread's block_terms store has been acknowledged)
#endif
        *flag = atomicAdd(fin.count, 1u) + 1u == gridDim.x ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u) return;
    const LossMeta &lm = fin.lm;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = (int)(blockDim.x >> 6);
    for (int bi = w; bi < lm.nb; bi += nwaves) {              // (step_loss_kernel's sums, term for term)
        float s = 0.f;
        for (int i = lm.blk_off[bi] + lane; i < lm.blk_off[bi + 1]; i += 64)
            s += agent_load(fin.bterms + i);
        s = wave_sum(s);
        if (lane == 0) {
            mean[bi] = s / (float)lm.B[bi];
            fin.loss[1 + bi] = mean[bi];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int bi = 0; bi < lm.nb; ++bi) total += lm.weight[bi] * mean[bi];
        // (step_ro_reg_kernel's `loss[0] = fma(coef, sum of the norms, loss[0])`, with the norms of mpqe_step_readout_norms)
        if (fin.reg_norms) total = __builtin_fmaf(fin.reg_coef, *fin.reg_norms, total);
        fin.loss[0] = total;
        agent_store(fin.count, 0u);
        // (the next step's forward granules get a new tag, step_uniform.h; the transposed copies' count a new target)
        *fin.epoch_f = *fin.epoch_f + 1u;
        if (fin.bump_b) *(fin.epoch_f + 16) = *(fin.epoch_f + 16) + 1u;
        if (fin.notify) {
            fin.notify[1] = fin.err ? (unsigned)agent_load(fin.err) : 0u;
#ifndef MPQE_EMU
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // (system scope: the flags word is out before the number)
#endif
            fin.notify[0] = fin.notify_value;
        }
    }
}

struct PrepArgs {
    UArgs ua;
    TSortArgs ts;             // the touch plan's sort (MPQE_STEP_BUILD_TOUCH): workgroups [0, sblocks) of the launch --
    int sblocks;              // dealt to the XCDs of `sxrank` only (4 bits per XCD: rank + 1, 0 = none; workgroup b of a
    int sna;                  // launch runs on XCD b % 8): the first sblocks = 8 x rows workgroups of the launch are ROWS of
    unsigned sxrank;          // eight -- the sort's XCDs take sort workgroups, the others go on with the prologue's items
    int strail;               // != 0: the launch's LAST strail workgroups instead (diagnostics switch TSORT_TRAIL)
    int late;                 // diagnostics ("HANDOFF_LATE"): the first transpose workgroup counts itself in ~1 s late -- a producer
                              // that lost its CU to another process: its consumers' bounded waits run out (MPQE_FLAG_INTERNAL)
    unsigned *tail_arrive;    // fused tail: the arrival counter of the step's weight-gradient launch, zeroed here; or NULL
    int *runs_count;          // the number of run starts the weight-gradient launch will compact (touch_runs_block), zeroed here; or NULL
    int ublocks, tblocks;     // vector-op workgroups, transpose workgroups
    int lead;                 // prologue workgroups in front of the chain workgroups: sblocks + ublocks + tblocks rounded
                              // up to a multiple of 8 (chain workgroup b keeps XCD b % 8)
    int nchain;               // chain workgroups (holes of the placement grid included)
    int plast;                // != 0: the prologue's items BEHIND the chain workgroups ([sort rows][chain][items]). A prologue
                              // of more than one workgroup per CU holds slots the chain workgroups are dealt into: those wait
                              // until prologue workgroups end, here and there, and pair up on some CUs while others stay
                              // empty (AIFB step with the MLP readout, ~400 prologue workgroups: 203 of 256 CUs used, chain
                              // launch 172 us; behind the chain: 256 CUs, 133 us. The default step's ~230 fit beside the chain
                              // workgroups and are 1.2 us faster in front). Only when every chain workgroup is resident with
                              // slots to spare: a producer must never wait for a slot held by its consumers
    int plna;                 // ... and those items are dealt only to the plna XCDs whose CUs hold at most ONE chain workgroup
    unsigned plxrank;         // (4 bits per XCD: rank + 1, 0 = none): an XCD whose slots are all taken by chain workgroups
                              // would queue a producer behind its waiting consumers
    const WtSlot *slots;
    float *WT;
    unsigned *wt_count;
    // forward-only step with a learned readout: only the copies its forward multiplies by (tsel_n of them, slots tsel[]) are
    // made; the first transpose workgroup counts the tskip workgroups of the others in as well, so that the count advances
    // by the same number in every launch of the packed step (the chain workgroups' target: epoch x all)
    int tsel_n, tsel[8];
    unsigned tskip;
    unsigned *fwd_done;       // vector-op workgroups finished, ever (uop_wait_prepass); NULL: none
    ZeroSegs zs;
};

// one 64 x 64 piece of a matrix, transposed: T[c][r] = W[r][c] (D % 64 == 0 in the chain form)
__device__ __forceinline__ void prep_transpose_block(const LayerPtrs &lp, const PrepArgs &pa, int D, int tb, float *smem) {
    float(*tile)[65] = reinterpret_cast<float(*)[65]>(smem);
    const int tpd = D / 64, per = tpd * tpd;
    const int si = pa.tsel_n ? pa.tsel[tb / per] : tb / per, tr = (tb % per) / tpd, tc = tb % tpd;
    const WtSlot sl = pa.slots[si];
    const float *W = (sl.mat >= 0 ? pick_layer(lp.basis, sl.layer) + (long long)sl.mat * D * D : pick_layer(lp.root, sl.layer)) + sl.col0;
    const int ld = sl.ld;
    float *T = pa.WT + (long long)si * D * D;
    const int tid = threadIdx.x;
    if (tid < 256) {
        const int c4 = tid & 15, r0 = tid >> 4;               // 16 float4 columns x 16 rows per pass
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 16 * q;
            const f32x4 v = gload4(W + (long long)(tr * 64 + r) * ld + tc * 64 + 4 * c4);
            if (sl.plain) *reinterpret_cast<f32x4 *>(T + (long long)(tr * 64 + r) * D + tc * 64 + 4 * c4) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r][4 * c4 + e] = v[e];
        }
    }
    __syncthreads();
    if (tid < 256 && !sl.plain) {
        const int c4 = tid & 15, r0 = tid >> 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 16 * q;                        // row of T = column of W
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tile[4 * c4 + e][r];
            *reinterpret_cast<f32x4 *>(T + (long long)(tc * 64 + r) * D + tr * 64 + 4 * c4) = v;
        }
    }
    // publish: every storing wave drains its stores, the workgroup meets, ONE lane releases at agent scope and
    // counts the workgroup in (the consumers: chain_block, before the backward levels)
#ifndef MPQE_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (tid == 0) {
#ifndef MPQE_EMU
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        atomicAdd(pa.wt_count, tb == 0 ? 1u + pa.tskip : 1u);
    }
}

__device__ __forceinline__ void prep_zero_block(const ZeroSegs &zs, long long zb) {
    if (threadIdx.x >= 256) return;
    int sg = 0;
    for (int i = 1; i < zs.count; ++i)
        if (zs.block0[i] <= zb) sg = i;
    const long long base = (zb - zs.block0[sg]) * PREP_ZERO_FLOATS_PER_BLOCK;
    float *p = zs.p[0];
    long long n = zs.n[0];
#pragma unroll
    for (int i = 1; i < PREP_MAX_SEGS; ++i)       // (a runtime index into the by-value table would spill it)
        if (i == sg) {
            p = zs.p[i];
            n = zs.n[i];
        }
    if (((uintptr_t)p & 15) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long i = base + (long long)(threadIdx.x + 256 * k) * 4;
            if (i + 3 < n) *reinterpret_cast<f32x4 *>(p + i) = f32x4{0.f, 0.f, 0.f, 0.f};
            else
                for (long long q = i; q < n; ++q) p[q] = 0.f;
        }
    } else {
        for (long long i = base + threadIdx.x; i < base + PREP_ZERO_FLOATS_PER_BLOCK && i < n; i += 256) p[i] = 0.f;
    }
}

// Merged launch: what used to be the weight-gradient launch -- zero fill of untouched relation matrices, the backward
// post-pass of the uniform node states, the weight-gradient tiles -- as workgroups of the chain launch, behind its zero
// fill: [prologue][chain][zero fill, padded to a multiple of 8][untouched matrices | post-pass, padded][tiles].
// They wait for the chain workgroups whose rows they read (DoneMeta) and for nothing that comes after them in the
// launch; with in-order dispatch per XCD every wait ends (and is bounded all the same).
struct ZMat {
    int layer, pad;
    long long rel;
};
#define ZMAT_FLOATS_PER_BLOCK 8192
struct WBlock;
struct PostArgs {
    int zpad;                 // zero-fill workgroups, padding included: the post roles start at lead + nchain + zpad; 0 = none
    int zmblocks, ublocks;    // untouched-matrix workgroups, post-pass workgroups
    int ppad;                 // zmblocks + ublocks rounded up to a multiple of na
    int na;                   // XCDs the post roles are dealt to; xrank: 4 bits per XCD, rank + 1 (0 = none)
    unsigned xrank;
    int wblocks, zper, D, zeroed;
    int tile_n;               // columns per weight-gradient tile
    UArgs ub;
    const WBlock *wblock;
    const ZMat *zmats;
    float *slabs;
    const float *H, *GH;
    long long level_stride;
    GradPtrs gp;
    const unsigned *done;
    const int *done_inc;
    const unsigned *epoch_m;  // epoch of the merged launches of this packed step (bumped by the reduction launch)
    int32_t *err;
    long long *stamps;        // diagnostics (mpqe_debug_tail_stamps): 8 words per tile workgroup, or NULL
};
template <int LDS_TILES>
__device__ __forceinline__ void post_block(const StepDev *__restrict__ sd, const LayerPtrs &lp, const PostArgs &po, int pb,
                                           float *smem);
__device__ __forceinline__ void zmat_block(const ZMat *__restrict__ zmats, int zper, int zb, int D, const GradPtrs &gp);

template <int NCB, int KS, int NW = 4, bool RO = false>
__global__ __launch_bounds__(64 * NW) void step_chain_kernel(const StepDev *__restrict__ sd, LayerPtrs lp,
                                                             TablePtrs tabs, ChainArgs ca, PrepArgs pa, PostArgs po) {
    __shared__ __attribute__((aligned(16))) ChainLds<NCB, KS, NW> S;
#define CHAIN_ROLES_RETURN return
#include "step_chain_roles.h"
#undef CHAIN_ROLES_RETURN
}
// ... and the forward-only step's instance, which ends with chain_finish (an instance of its own: the whole step's kernel
// keeps its arguments and its code as they were)
template <int NCB, int KS, int NW = 4, bool RO = false>
__global__ __launch_bounds__(64 * NW) void step_chain_fwd_kernel(const StepDev *__restrict__ sd, LayerPtrs lp,
                                                                 TablePtrs tabs, ChainArgs ca, PrepArgs pa, PostArgs po,
                                                                 FinArgs fin) {
    __shared__ __attribute__((aligned(16))) ChainLds<NCB, KS, NW> S;
#define CHAIN_ROLES_RETURN goto roles_done
#include "step_chain_roles.h"
#undef CHAIN_ROLES_RETURN
roles_done:
    chain_finish(fin, reinterpret_cast<float *>(&S));
}

// level form: zero fill of the gradient buffers (MPQE_STEP_ZERO_GRADS) as a launch of its own
__global__ __launch_bounds__(256) void step_zero_kernel(ZeroSegs zs) { prep_zero_block(zs, (long long)blockIdx.x); }

// rows [0, rows_total) are node rows of H0, then G positive and G negative targets. D/4 adjacent lanes
// own a row (two rows per wave at D = 128): the kernel is a chain of dependent gathers (id -> LUT ->
// table row), so more rows in flight per wave is what shortens it.
__global__ __launch_bounds__(256) void step_assemble_kernel(
    const StepDev *__restrict__ sd, TablePtrs tabs, const long long *__restrict__ node_map, long long map_len,
    const float *__restrict__ mode_emb, long long num_modes, const long long *__restrict__ anchor_ids,
    const long long *__restrict__ targets, const long long *__restrict__ negs, float *__restrict__ H0,
    float *__restrict__ tpos, float *__restrict__ tneg, int32_t *err, int vec, long long row0, long long R,
    long long g0, long long G) {
    // this launch covers node rows [row0, row0 + R) and graphs [g0, g0 + G) (one stream lane of the step)
    const int D = sd->D;
    const int lpr = lanes_per_row(D, vec), rpw = 64 / lpr;
    const int lane = threadIdx.x & 63, sub = lane & (lpr - 1);
    const long long wl = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + lane / lpr;
    // no early return: every lane takes part in the sub-wave shuffles; out-of-range groups idle
    const bool live = wl < R + 2 * G;
    const long long w = row0 + wl;        // global node row when wl < R
    int32_t *e = sub == 0 ? err : nullptr;
    const float *src = nullptr;
    float *dst = nullptr;
    bool normalise = true;
    if (live && wl < R) {
        int bi = 0;
        for (int i = 1; i < sd->nb; ++i)
            if (sd->b[i].row_off <= w) bi = i;
        const BatchDev &b = sd->b[bi];
        const long long lr = w - b.row_off;
        const long long g = lr / b.tp.N;
        const int n = (int)(lr - g * b.tp.N);
        dst = H0 + w * D;
        if (n < b.A) {
            const int tab = b.anchor_tab[n];
            const long long id = anchor_ids[b.anchor_off + (long long)n * b.B + g];
            const long long row = table_row(node_map, map_len, id, tabs.rows[tab], e);
            if (row >= 0) src = tabs.table[tab] + row * D;
        } else {
            const long long m = b.var_id[n - b.A];
            normalise = false;
            if (m < 0 || m >= num_modes) flag_error(e, MPQE_FLAG_BAD_NODE_ID);
            else src = mode_emb + m * D;
        }
    } else if (live) {
        const long long gi = g0 + (wl - R) % G;
        const bool is_neg = (wl - R) >= G;
        int bi = 0;
        for (int i = 1; i < sd->nb; ++i)
            if (sd->b[i].g_off <= gi) bi = i;
        const int tab = sd->b[bi].target_tab;
        const long long id = is_neg ? negs[gi] : targets[gi];
        const long long row = table_row(node_map, map_len, id, tabs.rows[tab], e);
        if (row >= 0) src = tabs.table[tab] + row * D;
        dst = (is_neg ? tneg : tpos) + gi * D;
    }
    if (live && (!src || !normalise)) {
        if (vec) {
            for (int c = sub * 4; c < D; c += 4 * lpr)
                *reinterpret_cast<f32x4 *>(dst + c) = src ? *reinterpret_cast<const f32x4 *>(src + c)
                                                          : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            for (int c = sub; c < D; c += lpr) dst[c] = src ? src[c] : 0.f;
        }
    }
    // the normalising groups run the shuffles; the others feed them a dummy (valid) row and discard
    const bool norm = live && src && normalise;
    const float *vs = norm ? src : mode_emb;
    if (norm) row_norm_store_sub(vs, dst, D, sub, lpr, vec);
    else {
        float dummy = 0.f;
        for (int off = lpr >> 1; off > 0; off >>= 1) dummy += __shfl_xor(dummy, off, 64);   // keep lanes converged
        (void)dummy;
    }
}

// ------------------------------------------------------------------------------------ layer levels
template <int MODE>
__global__ __launch_bounds__(256) void step_layer_fwd_kernel(const StepDev *__restrict__ sd, LayerPtrs lp, int p,
                                                             const TileRef *__restrict__ tiles,
                                                             const float *__restrict__ Hin,
                                                             float *__restrict__ Hout) {
    __shared__ __attribute__((aligned(16))) float smem[GT_SMEM_FLOATS];
    const TileRef tr = tiles[blockIdx.x];
    const BatchDev &b = sd->b[tr.batch];
    const int D = sd->D;
    const int ct = (D + GT_BN - 1) / GT_BN;
    const int n = tr.node, rem = tr.rem;
    const int li = layer_index(p, b.L, sd->num_layers);
    const TmplArgs tp = b.tp;
    tmpl_fwd_tile<MODE>(tp, b.B, Hin + b.row_off * D, pick_layer(lp.basis, li), pick_layer(lp.root, li),
                        pick_layer(lp.bias, li), D, D, p < b.L - 1,
                       Hout + b.row_off * D, n, (long long)(rem / ct) * GT_BM, (rem % ct) * GT_BN, smem);
}

template <int MODE>
__global__ __launch_bounds__(256) void step_layer_bwd_x_kernel(const StepDev *__restrict__ sd, LayerPtrs lp, int p,
                                                               const TileRef *__restrict__ tiles,
                                                               const float *__restrict__ Gout,
                                                               const float *__restrict__ Hin,
                                                               float *__restrict__ Gin, int add_in) {
    __shared__ __attribute__((aligned(16))) float smem[GT_SMEM_FLOATS];
    const TileRef tr = tiles[blockIdx.x];
    const BatchDev &b = sd->b[tr.batch];
    const int D = sd->D;
    const int ct = (D + GT_BN - 1) / GT_BN;
    const int m = tr.node, rem = tr.rem;
    const int li = layer_index(p, b.L, sd->num_layers);
    const TmplArgs tp = b.tp;
    // Gout is already a pre-activation gradient (masked by whoever wrote it); the gradient written
    // here belongs to H[p], which for p >= 1 is the ReLU output of pass p-1 -> mask it on the way out
    tmpl_bwd_x_tile<MODE>(tp, b.B, Gout + b.row_off * D, (const float *)nullptr, pick_layer(lp.basis, li),
                          pick_layer(lp.root, li), D, D, 0,
                         Gin + b.row_off * D, m, (long long)(rem / ct) * GT_BM, (rem % ct) * GT_BN, smem,
                         p >= 1 ? Hin + b.row_off * D : (const float *)nullptr, b.live[p + 1], add_in && p >= 1);
}

// ------------------------------------------------------------------------------------ score / loss
__device__ __forceinline__ float readout_value(int readout, const float *__restrict__ h, int N, int A, int D,
                                               int c, int *arg) {
    if (readout == MPQE_READOUT_TM) return h[(long long)A * D + c];
    if (readout == MPQE_READOUT_SUM) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += h[(long long)n * D + c];
        return s;
    }
    float best = h[c];
    int a = 0;
    for (int n = 1; n < N; ++n) {
        const float v = h[(long long)n * D + c];
        if (v > best) {
            best = v;
            a = n;
        }
    }
    *arg = a;
    return best;
}

#define STEP_MAX_COLS_PER_LANE 8     // D <= 512 on the fused path

// wave per graph, NJ = ceil(D / 64) columns per lane (compile-time: keeps the per-lane arrays in
// registers). BWD = false: scores and hinge terms. BWD = true: gradient rows of gH[L_b] and the
// positive / negative target-table gradients (through the L2 normalisation).
template <bool BWD, int NJ>
__global__ __launch_bounds__(256) void step_score_kernel(
    const StepDev *__restrict__ sd, const float *__restrict__ H, long long level_stride,
    const float *__restrict__ tpos, const float *__restrict__ tneg, float margin, float eps,
    float *__restrict__ s_pos, float *__restrict__ s_neg, float *__restrict__ terms,
    float *__restrict__ GH, TablePtrs tabs, const long long *__restrict__ node_map, long long map_len,
    const long long *__restrict__ targets, const long long *__restrict__ negs, long long g0, long long ng,
    const float *__restrict__ Q, float *__restrict__ GQ, const float *__restrict__ RY, float *__restrict__ RGY,
    int ro_op) {
    // Q != NULL (MPQE_READOUT_CALLER): the query embedding of graph gi is row gi of Q, its gradient goes to row gi of GQ.
    // RY != NULL (the learned readouts, step_readout.h): the embedding is the add / max / mean (ro_op: MPQE_SCATTER_*) over the
    // graph's rows of RY -- its N node rows, or its N - 1 pair rows (TARGETMLP) -- and the rows' gradients go to RGY.
    const long long gl = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (gl >= ng) return;
    const long long gi = g0 + gl;          // graphs [g0, g0 + ng): one stream lane of the step
    int bi = 0;
    for (int i = 1; i < sd->nb; ++i)
        if (sd->b[i].g_off <= gi) bi = i;
    const BatchDev &b = sd->b[bi];
    const int D = sd->D, N = b.tp.N, A = b.A;
    const long long row0 = b.row_off + (gi - b.g_off) * N;
    const float *h = H + (long long)b.L * level_stride + row0 * D;
    const bool ro_pairs = RY && sd->readout == MPQE_READOUT_TARGETMLP;
    const int ro_cnt = ro_pairs ? N - 1 : N;
    const long long ro_r0 = ro_pairs ? (b.row_off - b.g_off) + (gi - b.g_off) * (N - 1) : row0;
    if (RY) h = RY + ro_r0 * D;
    const float *tp_ = tpos + gi * D, *tn_ = tneg + gi * D;
    float q[NJ];
    int arg[NJ];
    float dp = 0.f, dn = 0.f, qq = 0.f, pp = 0.f, nn = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        q[j] = 0.f;
        arg[j] = 0;
        if (c < D) {
            if (RY) {
                q[j] = readout_value(ro_op == MPQE_SCATTER_MAX ? MPQE_READOUT_MAX : MPQE_READOUT_SUM, h, ro_cnt, 0, D, c, &arg[j]);
                if (ro_op == MPQE_SCATTER_MEAN) q[j] /= (float)ro_cnt;
            } else q[j] = Q ? Q[gi * D + c] : readout_value(sd->readout, h, N, A, D, c, &arg[j]);
            const float a = tp_[c], bb = tn_[c];
            dp += q[j] * a;
            dn += q[j] * bb;
            qq += q[j] * q[j];
            pp += a * a;
            nn += bb * bb;
        }
    }
    dp = wave_sum(dp);
    dn = wave_sum(dn);
    qq = wave_sum(qq);
    pp = wave_sum(pp);
    nn = wave_sum(nn);
    const float rq = sqrtf(qq), rp = sqrtf(pp), rn = sqrtf(nn);
    const float nq = fmaxf(rq, eps), np_ = fmaxf(rp, eps), nn_ = fmaxf(rn, eps);
    const float sp = dp / (nq * np_), sn = dn / (nq * nn_);
    const float v = margin - (sp - sn);
    if (lane == 0) {        // the backward instance also emits the scores: no separate forward launch then
        s_pos[gi] = sp;
        s_neg[gi] = sn;
        terms[gi] = v > 0.f ? v : 0.f;
    }
    if (!BWD) return;
    // d loss / d sp = -w/B on active terms, d/d sn = +w/B   (loss = sum_b w_b mean_b hinge)
    const float act = v >= 0.f ? b.weight / (float)b.B : 0.f;
    const float gsp = -act, gsn = act;
    const float inv_p = 1.f / (nq * np_), inv_n = 1.f / (nq * nn_);
    const float kq = rq > eps ? (gsp * sp + gsn * sn) / (nq * nq) : 0.f;
    const float ktp = rp > eps ? sp / (np_ * np_) : 0.f;
    const float ktn = rn > eps ? sn / (nn_ * nn_) : 0.f;
    // target rows: y = v/|v| (unit norm), dv = (g - y (y.g)) / |v|; |v| from the table row
    const int tab = b.target_tab;
    const long long prow = table_row(node_map, map_len, targets[gi], tabs.rows[tab], nullptr);
    const long long nrow = table_row(node_map, map_len, negs[gi], tabs.rows[tab], nullptr);
    float gyp[NJ], gyn[NJ];
    float yg_p = 0.f, yg_n = 0.f, ssp = 0.f, ssn = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        gyp[j] = gyn[j] = 0.f;
        if (c < D) {
            const float a = tp_[c], bb = tn_[c];
            gyp[j] = gsp * (q[j] * inv_p - ktp * a);
            gyn[j] = gsn * (q[j] * inv_n - ktn * bb);
            yg_p += a * gyp[j];
            yg_n += bb * gyn[j];
            if (prow >= 0) {
                const float t = tabs.table[tab][prow * D + c];
                ssp += t * t;
            }
            if (nrow >= 0) {
                const float t = tabs.table[tab][nrow * D + c];
                ssn += t * t;
            }
            const float gq = gsp * tp_[c] * inv_p + gsn * tn_[c] * inv_n - kq * q[j];
            float *gh = GH + (long long)b.L * level_stride + row0 * D + c;
            if (Q) GQ[gi * D + c] = gq;
            if (RY) {
                float *gy = RGY + ro_r0 * D + c;
                for (int n = 0; n < ro_cnt; ++n)
                    gy[(long long)n * D] = ro_op == MPQE_SCATTER_ADD ? gq
                                           : (ro_op == MPQE_SCATTER_MEAN ? gq / (float)ro_cnt : (arg[j] == n ? gq : 0.f));
            }
            for (int n = 0; n < N && !Q && !RY; ++n) {
                float gv;
                if (sd->readout == MPQE_READOUT_SUM) gv = gq;
                else if (sd->readout == MPQE_READOUT_TM) gv = n == A ? gq : 0.f;
                else gv = arg[j] == n ? gq : 0.f;
                gh[(long long)n * D] = gv;
            }
        }
    }
    yg_p = wave_sum(yg_p);
    yg_n = wave_sum(yg_n);
    ssp = wave_sum(ssp);
    ssn = wave_sum(ssn);
    const float ivp = 1.f / sqrtf(ssp), ivn = 1.f / sqrtf(ssn);
    float *gt = tabs.grad[tab];
    if (gt) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                if (prow >= 0) atomicAdd(gt + prow * D + c, (gyp[j] - tp_[c] * yg_p) * ivp);
                if (nrow >= 0) atomicAdd(gt + nrow * D + c, (gyn[j] - tn_[c] * yg_n) * ivn);
            }
        }
    }
}

// loss[0] = sum_b w_b * mean_b(terms), loss[1 + b] = mean_b(terms): one workgroup of 16 waves, wave b
// sums batch b (lane-strided, then a butterfly), thread 0 adds the batches in order -> fixed order
// nwaves waves of the calling workgroup take the batches round-robin (fixed order per batch, and the
// total is added in batch order by thread 0): reproducible.
__device__ __forceinline__ void loss_block(const StepDev *__restrict__ sd, const float *__restrict__ terms,
                                           float *__restrict__ loss, float *mean /*LDS, MAX_BATCHES floats*/,
                                           int nwaves) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int bi = w; bi < sd->nb; bi += nwaves) {
        const BatchDev &b = sd->b[bi];
        float s = 0.f;
        for (int i = lane; i < b.B; i += 64) s += terms[b.g_off + i];
        s = wave_sum(s);
        if (lane == 0) {
            mean[bi] = s / (float)b.B;
            loss[1 + bi] = mean[bi];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int bi = 0; bi < sd->nb; ++bi) total += sd->b[bi].weight * mean[bi];
        loss[0] = total;
    }
}

__global__ __launch_bounds__(1024) void step_loss_kernel(const StepDev *__restrict__ sd,
                                                         const float *__restrict__ terms,
                                                         float *__restrict__ loss, LossMeta lm,
                                                         const float *__restrict__ bterms, unsigned *epoch_f,
                                                         int bump_b = 0, unsigned *notify = nullptr, unsigned notify_value = 0,
                                                         const int32_t *err = nullptr) {
    __shared__ float mean[MPQE_STEP_MAX_BATCHES];
    // (mpqe_step_extra_t.notify: this is the forward-only call's last launch -- the launches that read the ids have run)
    if (notify && threadIdx.x == 0) {
        notify[1] = err ? (unsigned)*err : 0u;
#ifndef MPQE_EMU
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // (system scope: the flags word is out before the number)
#endif
        notify[0] = notify_value;
    }
    // (forward-only step: this is its last launch -- the next step's forward granules get a new tag, step_uniform.h; and
    // when its chain launch made transposed copies -- a learned readout's forward reads them -- their count a new target)
    if (epoch_f && threadIdx.x == 0) {
        *epoch_f = *epoch_f + 1u;
        if (bump_b) *(epoch_f + 16) = *(epoch_f + 16) + 1u;
    }
    if (lm.chain) loss_block_chain(lm, bterms, loss, mean, 16);
    else loss_block(sd, terms, loss, mean, 16);
}

#include "step_tail.h"
#include "step_plan.h"

// diagnostics: phase time stamps of the chain kernel's workgroups (tools/chain_timeline.py)
static long long *g_tail_stamps = nullptr;
static size_t g_tail_stamp_blocks = 0;
extern "C" void mpqe_debug_tail_stamps(void *device_buffer, size_t num_blocks) {
    g_tail_stamps = reinterpret_cast<long long *>(device_buffer);
    g_tail_stamp_blocks = num_blocks;
}
static long long *g_chain_stamps = nullptr;
static size_t g_chain_stamp_blocks = 0;
extern "C" void mpqe_debug_chain_stamps(void *device_buffer, size_t num_blocks) {
    g_chain_stamps = reinterpret_cast<long long *>(device_buffer);
    g_chain_stamp_blocks = num_blocks;
}

// Chain kernels (step_chain.h): D = 64 / 128 / 256 with 16-byte aligned weights, every batch within the number of
// passes the kernel's LDS tables cover. Everything else takes the one-launch-per-level form.
static bool want_chain(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb);
// plan for (P, B, lanes): the recent one if it matches, else a fresh one (which becomes the recent one)
static std::shared_ptr<CachedPlan> plan_for(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb,
                                            const mpqe_step_lanes_t *lanes, int *status) {
    *status = MPQE_OK;
    if (!P || !B || nb < 1 || nb > MPQE_STEP_MAX_BATCHES) {
        *status = MPQE_ERR_INVALID_ARG;
        return nullptr;
    }
    const bool ask_chain = want_chain(P, B, nb);
    PlanKey key;
    make_key(P, B, nb, lanes, &key);
    key.chain = ask_chain ? 1 : 0;
    {
        std::lock_guard<std::mutex> lock(g_plan_mu);
        if (g_recent && memcmp(&g_recent->key, &key, sizeof(key)) == 0) return g_recent;
    }
    std::shared_ptr<CachedPlan> fresh = std::make_shared<CachedPlan>();
    fresh->key = key;
    *status = plan_auto(P, B, nb, lanes, ask_chain, &fresh->hp);
    if (*status) return nullptr;
    std::lock_guard<std::mutex> lock(g_plan_mu);
    g_recent = fresh;
    return fresh;
}

static bool want_chain(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb) {
    if (!P || !B || nb < 1 || nb > MPQE_STEP_MAX_BATCHES) return false;
    bool use_chain = !(P->flags & MPQE_STEP_NO_CHAIN) && (P->dim == 64 || P->dim == 128 || P->dim == 256) &&
                     (P->readout < MPQE_READOUT_CALLER || P->readout >= MPQE_READOUT_MLP);
    // (the caller's readout needs the node states in HBM: level form. The learned readouts: two more levels of the chain,
    // while the ReLU bits of their hidden rows have a level to live in and two layer slots are free for their parameters)
    if (!use_chain) return false;
    if (P->readout >= MPQE_READOUT_MLP) {
        if (P->readout == MPQE_READOUT_CONCAT) {        // (one input block per layer)
            for (int i = 0; i < nb; ++i)
                if (B[i].num_passes != P->num_layers) return false;
        }
        if (P->num_layers + 2 > MPQE_STEP_MAX_LAYERS || !P->readout_w0 || !P->readout_w2) return false;
        if (!ptr_vec_ok(P->readout_w0, P->dim) || !ptr_vec_ok(P->readout_w2, P->dim)) return false;
        if ((P->readout_b0 && (uintptr_t)P->readout_b0 % 16 != 0) || (P->readout_b2 && (uintptr_t)P->readout_b2 % 16 != 0)) return false;
        for (int i = 0; i < nb; ++i)
            if (B[i].num_passes + 1 > CH_MASK_LEVELS) return false;
    }
    long long graphs = 0;
    for (int i = 0; i < nb; ++i) graphs += B[i].batch_size;
    use_chain = graphs <= CHAIN_MAX_GRAPHS && P->num_layers > 0 && P->num_layers <= MPQE_STEP_MAX_LAYERS;
    // (5 passes x (3 edges + 4 nodes) x 2 directions = 70 ops <= CH_MAX_OPS; a batch with more than CH_MAX_CV forward
    // node updates is turned away by the planner: plan_auto)
    for (int i = 0; i < nb; ++i) use_chain = use_chain && B[i].num_passes <= CH_MASK_LEVELS + 1;
    for (int l = 0; use_chain && l < P->num_layers; ++l)
        use_chain = P->basis[l] && P->root[l] && ptr_vec_ok(P->basis[l], P->dim) && ptr_vec_ok(P->root[l], P->dim) &&
                    (!P->bias[l] || (uintptr_t)P->bias[l] % 16 == 0);
    for (int m = 0; use_chain && m < P->num_modes && m < MPQE_STEP_MAX_MODES; ++m)
        use_chain = P->tables[m] && (uintptr_t)P->tables[m] % 16 == 0;
    return use_chain && P->mode_emb && (uintptr_t)P->mode_emb % 16 == 0;
}

extern "C" size_t mpqe_step_workspace_bytes(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb,
                                            const mpqe_step_lanes_t *lanes) {
    int st;
    const std::shared_ptr<CachedPlan> cp = plan_for(P, B, nb, lanes, &st);
    return cp ? cp->hp.total : 0;
}
extern "C" size_t mpqe_step_desc_bytes(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb,
                                       const mpqe_step_lanes_t *lanes) {
    // exact: the level form's lanes have their own per-level tile tables, so the table's size depends on the split
    int st;
    const std::shared_ptr<CachedPlan> cp = plan_for(P, B, nb, lanes, &st);
    return cp ? cp->hp.desc_total : 0;
}

extern "C" int mpqe_step_states_layout(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb,
                                       const mpqe_step_lanes_t *lanes, int64_t *states_offset, int64_t *grads_offset,
                                       int64_t *level_stride, int64_t *row_offset, int64_t *queries_offset,
                                       int64_t *query_grads_offset) {
    int st;
    const std::shared_ptr<CachedPlan> cp = plan_for(P, B, nb, lanes, &st);
    if (!cp) return st ? st : MPQE_ERR_INVALID_ARG;
    const HostPlan &hp = cp->hp;
    if (hp.chain || P->readout != MPQE_READOUT_CALLER) return MPQE_ERR_UNSUPPORTED;        // (the chain form keeps the node states in LDS)
    if (states_offset) *states_offset = (int64_t)hp.o_H;
    if (grads_offset) *grads_offset = (int64_t)hp.o_GH;
    if (level_stride) *level_stride = (int64_t)hp.level_stride;
    if (queries_offset) *queries_offset = (int64_t)hp.o_Q;
    if (query_grads_offset) *query_grads_offset = (int64_t)hp.o_GQ;
    for (int i = 0; row_offset && i <= nb; ++i) row_offset[i] = i < nb ? (int64_t)hp.sd.b[i].row_off : (int64_t)hp.sd.rows_total;
    return MPQE_OK;
}

// ---- touch plan (step_touch.h)
static int touch_dims(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb, long long *M, int *row_bits,
                      int *key_bits) {
    if (!P || !B || nb < 1 || nb > MPQE_STEP_MAX_BATCHES || P->num_modes <= 0 || P->num_modes > MPQE_STEP_MAX_MODES)
        return MPQE_ERR_INVALID_ARG;
    *M = touch_entries(B, nb);
    if (*M <= 0 || *M >= (1ll << 31)) return MPQE_ERR_INVALID_ARG;
    long long rows = 1;
    for (int m = 0; m < P->num_modes; ++m) rows = std::max(rows, (long long)P->table_rows[m]);
    *row_bits = touch_bits(rows);
    *key_bits = *row_bits + 5;          // table index (<= 16) above the row; an invalid entry has every bit set
    return MPQE_OK;
}
extern "C" size_t mpqe_step_touch_bytes(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb) {
    long long M;
    int rb, kb;
    if (touch_dims(P, B, nb, &M, &rb, &kb) != MPQE_OK) return 0;
    return touch_layout(M, kb).total;
}
extern "C" size_t mpqe_step_touch_workspace_bytes(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb) {
    long long M;
    int rb, kb;
    if (touch_dims(P, B, nb, &M, &rb, &kb) != MPQE_OK) return 0;
    return touch_layout(M, kb).w_total;
}
extern "C" int mpqe_step_touch_build(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb,
                                     const int64_t *anchor_ids, const int64_t *targets, const int64_t *negs,
                                     void *touch, size_t touch_bytes, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    long long M;
    int rb, kb;
    int st = touch_dims(P, B, nb, &M, &rb, &kb);
    if (st) return st;
    if (!anchor_ids || !targets || !negs || !touch || !workspace || !P->node_map) return MPQE_ERR_INVALID_ARG;
    const TouchLayout L = touch_layout(M, kb);
    if (touch_bytes < L.total || workspace_bytes < L.w_total) return MPQE_ERR_WORKSPACE;
    if ((uintptr_t)touch % 256 != 0 || (uintptr_t)workspace % 256 != 0) return MPQE_ERR_INVALID_ARG;
    hipStream_t s = as_stream(stream);
    TouchMeta tm;
    memset(&tm, 0, sizeof(tm));
    tm.nb = nb;
    tm.row_bits = rb;
    long long anchors = 0, graphs = 0;
    for (int i = 0; i < nb; ++i) {
        const TemplateDesc &t = kTemplates[B[i].query_type];
        tm.B[i] = B[i].batch_size;
        tm.A[i] = t.A;
        tm.anchor_off[i] = anchors;
        tm.g_off[i] = graphs;
        for (int a = 0; a < 3; ++a) {
            tm.anchor_tab[i][a] = a < t.A ? B[i].anchor_mode[a] : 0;
            if (a < t.A && (B[i].anchor_mode[a] < 0 || B[i].anchor_mode[a] >= P->num_modes)) return MPQE_ERR_INVALID_ARG;
        }
        if (B[i].target_mode < 0 || B[i].target_mode >= P->num_modes) return MPQE_ERR_INVALID_ARG;
        tm.target_tab[i] = B[i].target_mode;
        anchors += (long long)B[i].batch_size * t.A;
        graphs += B[i].batch_size;
    }
    tm.anchor_off[nb] = anchors;
    tm.g_off[nb] = graphs;
    for (int m = 0; m < P->num_modes; ++m) tm.table_rows[m] = P->table_rows[m];
    char *tb = reinterpret_cast<char *>(touch), *wb = reinterpret_cast<char *>(workspace);
    TouchHeader th;
    memset(&th, 0, sizeof(th));
    th.M = M;
    th.row_bits = rb;
    th.key_bits = kb;
    static_assert(sizeof(TouchMeta) <= 2048, "touch_layout reserves 2 KB for the batch table");
    if (M <= TSORT_MAX_ENTRIES && kb <= 31 && !dbg_on("TOUCH_MULTI_LAUNCH") && !(P->flags & MPQE_STEP_TOUCH_LIBRARY_SORT)) {
        // the whole plan in one launch (step_touch.h: tsort_block) behind the clear of its barrier counter and the
        // upload of the batch table
        const int nblk = tsort_blocks(M);
        const size_t Mp = (size_t)nblk * TSORT_THREADS * tsort_rounds(M);
        TSortArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.ka = reinterpret_cast<unsigned *>(wb + L.w_keys);
        sa.kb = sa.ka + Mp;                                   // (the 8-byte key array, halved)
        sa.va = reinterpret_cast<unsigned *>(wb + L.w_vals);
        sa.vb = reinterpret_cast<unsigned *>(wb + L.w_svals);
        sa.hist = reinterpret_cast<unsigned *>(wb + L.w_hist);
        sa.counter = sa.hist + 4 * 256 * 256;
        TouchMeta *tmd = reinterpret_cast<TouchMeta *>(sa.counter + 64);
        sa.tm = tmd;
        (void)hipMemsetAsync(sa.counter, 0, sizeof(unsigned), s);
        upload(s, reinterpret_cast<char *>(tmd), &tm, sizeof(tm));
        sa.anchor_ids = reinterpret_cast<const long long *>(anchor_ids);
        sa.targets = reinterpret_cast<const long long *>(targets);
        sa.negs = reinterpret_cast<const long long *>(negs);
        sa.node_map = reinterpret_cast<const long long *>(P->node_map);
        sa.map_len = (long long)P->node_map_len;
        sa.keys_out = reinterpret_cast<tkey_t *>(tb + L.keys);
        sa.perm = reinterpret_cast<int *>(tb + L.perm);
        sa.erow = reinterpret_cast<int *>(tb + L.erow);
        sa.th_out = reinterpret_cast<TouchHeader *>(tb);
        sa.M = (int)M;
        sa.key_bits = kb;
        sa.row_bits = rb;
        sa.nblk = nblk;
        sa.rounds = tsort_rounds(M);
        hipLaunchKernelGGL(touch_sort_kernel, dim3((unsigned)nblk), dim3(TSORT_THREADS), 0, s, sa);
        return mpqe_launch_status();
    }
    // (the header rides along as an argument of the keys kernel; the sort leaves perm = the entries in sorted order)
    tkey_t *keys = reinterpret_cast<tkey_t *>(wb + L.w_keys);
    int *vals = reinterpret_cast<int *>(wb + L.w_vals);
    hipLaunchKernelGGL(touch_keys_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, tm,
                       reinterpret_cast<const long long *>(anchor_ids), reinterpret_cast<const long long *>(targets),
                       reinterpret_cast<const long long *>(negs), reinterpret_cast<const long long *>(P->node_map),
                       (long long)P->node_map_len, keys, vals, reinterpret_cast<int *>(tb + L.erow), M, th,
                       reinterpret_cast<TouchHeader *>(tb));
    // stable: entries of one destination row keep their entry order, so the per-row sums have ONE order (radix_sort.h: the
    // library's own multi-launch sort -- nothing in it depends on how many workgroups are resident at once)
    if (radix_sort_pairs_own<tkey_t>(wb + L.w_tmp, (const tkey_t *)keys, reinterpret_cast<tkey_t *>(tb + L.keys), (const int *)vals,
                                     reinterpret_cast<int *>(tb + L.perm), (long long)M, kb, s))
        return MPQE_ERR_LAUNCH;
    return mpqe_launch_status();
}

// ---- row-sparse Adam over the touched table rows (include/mpqe_amd.h: mpqe_adam_rows_step)
struct RowAdamPtrs {
    float *p[MPQE_STEP_MAX_MODES], *m[MPQE_STEP_MAX_MODES], *v[MPQE_STEP_MAX_MODES];
    const float *g[MPQE_STEP_MAX_MODES];
};
__global__ __launch_bounds__(256) void adam_rows_kernel(const TouchHeader *__restrict__ th, const tkey_t *__restrict__ keys,
                                                        int D, RowAdamPtrs rp, float omb1, float omb2, float eps,
                                                        float neg_step_size) {
    // one group of D / 4 lanes per sorted position; the first position of a run of equal keys owns the row
    const int lpr = D / 4, per = 256 / lpr;
    const long long k = (long long)blockIdx.x * per + threadIdx.x / lpr;
    const int c = (threadIdx.x % lpr) * 4;
    if (k >= th->M) return;
    const tkey_t key = keys[k];
    if (key == TOUCH_INVALID || (k > 0 && keys[k - 1] == key)) return;
    const int tab = (int)(key >> th->row_bits);
    const long long off = (long long)(key & ((1ull << th->row_bits) - 1ull)) * D + c;
    float *p = rp.p[0], *m = rp.m[0], *v = rp.v[0];
    const float *g = rp.g[0];
#pragma unroll
    for (int t = 1; t < MPQE_STEP_MAX_MODES; ++t)
        if (t == tab) {
            p = rp.p[t];
            m = rp.m[t];
            v = rp.v[t];
            g = rp.g[t];
        }
    if (!p || !g || !m || !v) return;
    const f32x4 gg = gload4(g + off);
    f32x4 pp = *reinterpret_cast<f32x4 *>(p + off), mm = *reinterpret_cast<f32x4 *>(m + off);
    f32x4 vv = *reinterpret_cast<f32x4 *>(v + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // torch.optim.SparseAdam, operation for operation (no contraction into fused multiply-adds: the same roundings)
        const float m_upd = __fmul_rn(__fsub_rn(gg[e], mm[e]), omb1);
        const float v_upd = __fmul_rn(__fsub_rn(__fmul_rn(gg[e], gg[e]), vv[e]), omb2);
        const float numer = __fadd_rn(m_upd, mm[e]);
        const float v_new = __fadd_rn(v_upd, vv[e]);
        const float denom = __fadd_rn(sqrtf(v_new), eps);
        mm[e] = __fadd_rn(mm[e], m_upd);
        vv[e] = v_new;
        pp[e] = __fadd_rn(pp[e], __fmul_rn(neg_step_size, __fdiv_rn(numer, denom)));
    }
    *reinterpret_cast<f32x4 *>(p + off) = pp;
    *reinterpret_cast<f32x4 *>(m + off) = mm;
    *reinterpret_cast<f32x4 *>(v + off) = vv;
}

extern "C" int64_t mpqe_step_touch_entries(const mpqe_step_batch_t *B, int nb) {
    if (!B || nb < 1 || nb > MPQE_STEP_MAX_BATCHES) return -1;
    return touch_entries(B, nb);
}

// ---- a plan from (table, row) keys that are given, not looked up (the data-parallel row exchange: every rank's touched
// rows, all-gathered): keys sorted (stable), perm[k] = index of sorted position k in the input order
extern "C" size_t mpqe_rows_plan_bytes(int64_t n) { return n > 0 ? touch_layout(n, 0).total : 0; }
extern "C" size_t mpqe_rows_plan_workspace_bytes(int64_t n, int key_bits) {
    return n > 0 && key_bits > 0 && key_bits <= 64 ? touch_layout(n, key_bits).w_total : 0;
}
__global__ __launch_bounds__(256) void iota_kernel(int *__restrict__ v, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = (int)i;
}
extern "C" int mpqe_rows_plan_build(const uint64_t *keys, int64_t n, int row_bits, int key_bits, void *plan,
                                    size_t plan_bytes, void *workspace, size_t workspace_bytes, void *stream) {
    if (!keys || !plan || !workspace || n <= 0 || n >= (1ll << 31) || row_bits <= 0 || key_bits <= row_bits || key_bits > 64)
        return MPQE_ERR_INVALID_ARG;
    const TouchLayout L = touch_layout(n, key_bits);
    if (plan_bytes < L.total || workspace_bytes < L.w_total) return MPQE_ERR_WORKSPACE;
    if ((uintptr_t)plan % 256 != 0 || (uintptr_t)workspace % 256 != 0) return MPQE_ERR_INVALID_ARG;
    hipStream_t s = as_stream(stream);
    char *tb = reinterpret_cast<char *>(plan), *wb = reinterpret_cast<char *>(workspace);
    TouchHeader th;
    memset(&th, 0, sizeof(th));
    th.M = n;
    th.row_bits = row_bits;
    th.key_bits = key_bits;
    upload(s, tb, &th, sizeof(th));
    int *vals = reinterpret_cast<int *>(wb + L.w_vals);
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, vals, (long long)n);
    if (radix_sort_pairs_own<tkey_t>(wb + L.w_tmp, reinterpret_cast<const tkey_t *>(keys), reinterpret_cast<tkey_t *>(tb + L.keys),
                                     (const int *)vals, reinterpret_cast<int *>(tb + L.perm), (long long)n, key_bits, s))
        return MPQE_ERR_LAUNCH;
    return mpqe_launch_status();
}
// table_grads[t][row] (= or +=) the sum of rows[perm[k]] over the plan's run of key (t, row), in sorted (= input) order
__global__ __launch_bounds__(256) void rows_sum_kernel(const char *__restrict__ plan, size_t o_keys, size_t o_perm,
                                                       const float *__restrict__ rows, int D, TablePtrs tabs, int store) {
    const TouchHeader *th = reinterpret_cast<const TouchHeader *>(plan);
    table_sum_block(th->M, th->row_bits, reinterpret_cast<const tkey_t *>(plan + o_keys),
                    reinterpret_cast<const int *>(plan + o_perm), rows, D, tabs, store, (long long)blockIdx.x);
}
extern "C" int mpqe_table_rows_sum(const void *plan, int64_t n, const float *rows, int64_t dim, float *const *table_grads,
                                   int num_modes, int store, void *stream) {
    if (!plan || !rows || !table_grads || n <= 0 || num_modes <= 0 || num_modes > MPQE_STEP_MAX_MODES)
        return MPQE_ERR_INVALID_ARG;
    if (dim <= 0 || dim % 4 != 0 || dim > 1024 || 256 % (dim / 4) != 0 || (uintptr_t)rows % 16 != 0) return MPQE_ERR_UNSUPPORTED;
    TablePtrs tabs;
    memset(&tabs, 0, sizeof(tabs));
    for (int m = 0; m < num_modes; ++m) {
        tabs.grad[m] = table_grads[m];
        if ((uintptr_t)table_grads[m] % 16 != 0) return MPQE_ERR_INVALID_ARG;
    }
    const TouchLayout L = touch_layout(n, 0);
    const long long per = 256 / (dim / 4);
    hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)((n + per - 1) / per)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const char *>(plan), L.keys, L.perm, rows, (int)dim, tabs, store);
    return mpqe_launch_status();
}

// The entity-table rows of a step's reduction again (include/mpqe_amd.h): a step whose own touch plan could not be built
// (MPQE_FLAG_TOUCH_RETRY) left every entry's gradient row in its workspace; `touch` is a plan built afterwards.
__global__ __launch_bounds__(256) void step_table_rows_kernel(const char *__restrict__ touch, size_t o_keys, size_t o_perm,
                                                              long long M, int row_bits, const float *__restrict__ DG, int D,
                                                              TablePtrs tabs, int store) {
    table_sum_block(M, row_bits, reinterpret_cast<const tkey_t *>(touch + o_keys), reinterpret_cast<const int *>(touch + o_perm),
                    DG, D, tabs, store, (long long)blockIdx.x, &reinterpret_cast<const TouchHeader *>(touch)->pad[0]);
}
extern "C" int mpqe_step_table_rows(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb, const mpqe_step_grads_t *G,
                                    const void *desc, void *workspace, size_t workspace_bytes, const void *touch, void *stream) {
    if (!P || !B || !G || !desc || !workspace || !touch || nb < 1 || nb > MPQE_STEP_MAX_BATCHES) return MPQE_ERR_INVALID_ARG;
    if ((uintptr_t)touch % 256 != 0 || (uintptr_t)workspace % 256 != 0) return MPQE_ERR_INVALID_ARG;
    std::shared_ptr<const CachedPlan> cached;
    {
        PlanKey key;
        make_key(P, B, nb, nullptr, &key);
        key.chain = want_chain(P, B, nb) ? 1 : 0;
        std::lock_guard<std::mutex> lock(g_plan_mu);
        auto it = g_plans.find(const_cast<void *>(desc));
        if (it != g_plans.end() && memcmp(&it->second->key, &key, sizeof(key)) == 0) cached = it->second;
    }
    if (!cached) return MPQE_ERR_INVALID_ARG;           // (not the descriptor buffer of a step that has run with these descriptors)
    const HostPlan &hp = cached->hp;
    if (!hp.chain || hp.touch_M <= 0) return MPQE_ERR_UNSUPPORTED;
    if (workspace_bytes < hp.total) return MPQE_ERR_WORKSPACE;
    const int D = P->dim;
    if (D % 4 != 0 || 256 % (D / 4) != 0) return MPQE_ERR_UNSUPPORTED;
    long long trows = 1;
    TablePtrs tabs;
    memset(&tabs, 0, sizeof(tabs));
    for (int m = 0; m < P->num_modes; ++m) {
        trows = std::max(trows, (long long)P->table_rows[m]);
        tabs.grad[m] = G->tables[m];
        tabs.rows[m] = P->table_rows[m];
        if (G->tables[m] && (uintptr_t)G->tables[m] % 16 != 0) return MPQE_ERR_INVALID_ARG;
    }
    const TouchLayout TL = touch_layout(hp.touch_M, 0);
    const long long per = 256 / (D / 4);
    const int store = ((P->flags & MPQE_STEP_SPARSE_TABLES) || (P->flags & MPQE_STEP_ZERO_GRADS)) ? 1 : 0;
    hipLaunchKernelGGL(step_table_rows_kernel, dim3((unsigned)((hp.touch_M + per - 1) / per)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const char *>(touch), TL.keys, TL.perm, (long long)hp.touch_M, touch_bits(trows),
                       reinterpret_cast<const float *>(reinterpret_cast<const char *>(workspace) + hp.o_DG), D, tabs, store);
    return mpqe_launch_status();
}

extern "C" int mpqe_adam_rows_step(const void *touch, int64_t num_entries, float *const *params,
                                   const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                                   int num_modes, int64_t dim, double lr, double beta1, double beta2, double eps,
                                   int64_t step, void *stream) {
    if (!touch || num_entries <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || step < 1) return MPQE_ERR_INVALID_ARG;
    if (num_modes <= 0 || num_modes > MPQE_STEP_MAX_MODES || dim <= 0 || dim % 4 != 0 || dim > 1024 || 256 % (dim / 4) != 0)
        return MPQE_ERR_UNSUPPORTED;
    if (!(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1.)) return MPQE_ERR_INVALID_ARG;
    const long long M = num_entries;
    RowAdamPtrs rp;
    memset(&rp, 0, sizeof(rp));
    for (int m = 0; m < num_modes; ++m) {
        rp.p[m] = params[m];
        rp.g[m] = grads[m];
        rp.m[m] = exp_avg[m];
        rp.v[m] = exp_avg_sq[m];
        if (((uintptr_t)rp.p[m] | (uintptr_t)rp.g[m] | (uintptr_t)rp.m[m] | (uintptr_t)rp.v[m]) % 16 != 0)
            return MPQE_ERR_INVALID_ARG;
    }
    // torch.optim.SparseAdam: step_size = lr * sqrt(1 - b2^t) / (1 - b1^t) in double (python floats), 1 - beta too
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float neg_step = (float)(-(lr * sqrt(bc2) / bc1));
    const TouchLayout L = touch_layout(M, 0);
    const char *tb = reinterpret_cast<const char *>(touch);
    const long long per = 256 / (dim / 4);
    hipLaunchKernelGGL(adam_rows_kernel, dim3((unsigned)((M + per - 1) / per)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const TouchHeader *>(tb), reinterpret_cast<const tkey_t *>(tb + L.keys), (int)dim, rp,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, neg_step);
    return mpqe_launch_status();
}

// sum_i ||p_i||_2 of up to four parameter tensors and its backward (the regulariser of margin_loss, reference
// model.py:486-490, for the module path; the fused step calls the same kernel inside its own call)
extern "C" int mpqe_l2_norms(const float *const *params, const int64_t *sizes, int count, const float *grad_out, float *out,
                             float *const *grads, void *stream) {
    if (!params || !sizes || count < 1 || count > 4 || (!out && !grads)) return MPQE_ERR_INVALID_ARG;
    RoRegArgs rr;
    memset(&rr, 0, sizeof(rr));
    for (int i = 0; i < count; ++i) {
        if (!params[i] || sizes[i] <= 0) return MPQE_ERR_INVALID_ARG;
        rr.p[i] = params[i];
        rr.n[i] = sizes[i];
        rr.g[i] = grads ? grads[i] : nullptr;
    }
    rr.coef = 1.f;
    rr.loss = out;                 // (+= : the caller zero-fills it)
    rr.gscale = grad_out;
    hipLaunchKernelGGL(step_ro_reg_kernel, dim3(1), dim3(1024), 0, as_stream(stream), rr);
    return mpqe_launch_status();
}

static int D_ok_for_readout(int D) { return D % 4 == 0; }      // (16-byte rows in step_readout.h)
// input width of the learned readout's first Linear layer (reference model.py:497-553: targetmlp [target | node], concat one
// block per layer)
static int readout_kin(const mpqe_step_params_t *P) {
    return P->readout == MPQE_READOUT_TARGETMLP ? 2 * P->dim : (P->readout == MPQE_READOUT_CONCAT ? P->num_layers * P->dim : P->dim);
}

// batch weights of a call with extras: sd->b[i].weight = host weight x *device scalar (one workgroup, in front of the step's
// launches on its stream; wdev[i] NULL: the host weight alone -- which is also how a later call without extras restores them)
struct WeightPatch {
    const float *wdev[MPQE_STEP_MAX_BATCHES];
    float whost[MPQE_STEP_MAX_BATCHES];
    int nb;
};
__global__ __launch_bounds__(64) void step_weights_kernel(StepDev *sd, WeightPatch wp) {
    const int i = threadIdx.x;
    if (i < wp.nb) sd->b[i].weight = wp.wdev[i] ? wp.whost[i] * *wp.wdev[i] : wp.whost[i];
}

extern "C" int mpqe_step_readout_norms(const mpqe_step_params_t *P, float *out, void *stream) {
    if (!P || !out || P->readout < MPQE_READOUT_MLP) return MPQE_ERR_INVALID_ARG;
    if (!P->readout_w0 || !P->readout_b0 || !P->readout_w2 || !P->readout_b2 || P->dim <= 0) return MPQE_ERR_INVALID_ARG;
    const int D = P->dim, kin = D_ok_for_readout(D) ? readout_kin(P) : 0;
    if (kin <= 0) return MPQE_ERR_UNSUPPORTED;
    RoRegArgs rr;
    memset(&rr, 0, sizeof(rr));
    rr.p[0] = P->readout_w0; rr.p[1] = P->readout_b0; rr.p[2] = P->readout_w2; rr.p[3] = P->readout_b2;
    rr.n[0] = (long long)D * kin; rr.n[1] = D; rr.n[2] = (long long)D * D; rr.n[3] = D;
    rr.coef = 1.f;                       // (out = fma(1, sum of the norms, 0): the sum itself)
    rr.loss = out;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(out, 0, sizeof(float), s) != hipSuccess) return MPQE_ERR_LAUNCH;
    hipLaunchKernelGGL(step_ro_reg_kernel, dim3(1), dim3(1024), 0, s, rr);
    return mpqe_launch_status();
}

extern "C" int mpqe_step_forward_backward(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb,
                                          const int64_t *anchor_ids, const int64_t *targets, const int64_t *negs,
                                          float margin, const mpqe_step_grads_t *G, int backward,
                                          float *loss, float *scores_pos, float *scores_neg, void *desc,
                                          size_t desc_bytes, int upload_desc, void *workspace,
                                          size_t workspace_bytes, int32_t *err, const mpqe_step_lanes_t *lanes,
                                          void *const *events, int num_events, void *touch, void *stream) {
    return mpqe_step_forward_backward_ex(P, B, nb, anchor_ids, targets, negs, margin, G, backward, loss, scores_pos, scores_neg,
                                         desc, desc_bytes, upload_desc, workspace, workspace_bytes, err, lanes, events, num_events,
                                         touch, stream, nullptr);
}

static int step_ex(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb, const int64_t *anchor_ids,
                   const int64_t *targets, const int64_t *negs, float margin, const mpqe_step_grads_t *G, int backward,
                   float *loss, float *scores_pos, float *scores_neg, void *desc, size_t desc_bytes, int upload_desc,
                   void *workspace, size_t workspace_bytes, int32_t *err, const mpqe_step_lanes_t *lanes, void *const *events,
                   int num_events, void *touch, void *stream, const mpqe_step_extra_t *extra);

extern "C" int mpqe_step_forward_backward_ex(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb,
                                             const int64_t *anchor_ids, const int64_t *targets, const int64_t *negs,
                                             float margin, const mpqe_step_grads_t *G, int backward,
                                             float *loss, float *scores_pos, float *scores_neg, void *desc,
                                             size_t desc_bytes, int upload_desc, void *workspace,
                                             size_t workspace_bytes, int32_t *err, const mpqe_step_lanes_t *lanes,
                                             void *const *events, int num_events, void *touch, void *stream,
                                             const mpqe_step_extra_t *extra) {
    const int st = step_ex(P, B, nb, anchor_ids, targets, negs, margin, G, backward, loss, scores_pos, scores_neg, desc, desc_bytes,
                           upload_desc, workspace, workspace_bytes, err, lanes, events, num_events, touch, stream, extra);
    // (mpqe_step_extra_t.join_event / join_stream: the consumer's stream waits for this call's launches)
    if (st == MPQE_OK && extra && extra->join_event && extra->join_stream != stream) {
        if (hipEventRecord(reinterpret_cast<hipEvent_t>(extra->join_event), as_stream(stream)) != hipSuccess ||
            hipStreamWaitEvent(as_stream(extra->join_stream), reinterpret_cast<hipEvent_t>(extra->join_event), 0) != hipSuccess)
            return MPQE_ERR_LAUNCH;
    }
    return st;
}

static int step_ex(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb, const int64_t *anchor_ids,
                   const int64_t *targets, const int64_t *negs, float margin, const mpqe_step_grads_t *G, int backward,
                   float *loss, float *scores_pos, float *scores_neg, void *desc, size_t desc_bytes, int upload_desc,
                   void *workspace, size_t workspace_bytes, int32_t *err, const mpqe_step_lanes_t *lanes, void *const *events,
                   int num_events, void *touch, void *stream, const mpqe_step_extra_t *extra) {
    if (!P || !B || nb < 1 || nb > MPQE_STEP_MAX_BATCHES || !desc) return MPQE_ERR_INVALID_ARG;
    const bool ask_chain = want_chain(P, B, nb);
    // The launch plan is a pure function of the descriptors; it is kept on the host next to the device
    // table it describes (same key: the caller's desc buffer), so a steady-state call costs one lookup.
    std::shared_ptr<const CachedPlan> cached;
    {
        PlanKey key;
        make_key(P, B, nb, lanes, &key);
        key.chain = ask_chain ? 1 : 0;
        std::lock_guard<std::mutex> lock(g_plan_mu);
        auto it = g_plans.find(desc);
        if (it != g_plans.end() && memcmp(&it->second->key, &key, sizeof(key)) == 0) cached = it->second;
        else if (it != g_plans.end() && !upload_desc) {
            // desc holds another step's table -- unless only the diagnostics switches changed since it was planned: then the
            // same step is planned again and its table uploaded again by this call
            PlanKey old = it->second->key;
            old.dbg_gen = key.dbg_gen;
            if (memcmp(&old, &key, sizeof(key)) != 0) return MPQE_ERR_INVALID_ARG;
            upload_desc = 1;
        }
        if (!cached && g_recent && memcmp(&g_recent->key, &key, sizeof(key)) == 0) {
            // (the size queries of this packed step have just built it)
            if (g_plans.size() >= 1024) g_plans.clear();      // plans in use stay alive through their shared_ptr
            g_plans[desc] = g_recent;
            cached = g_recent;
        }
        if (!cached) {
            std::shared_ptr<CachedPlan> fresh = std::make_shared<CachedPlan>();
            fresh->key = key;
            int st = plan_auto(P, B, nb, lanes, ask_chain, &fresh->hp);
            if (st) return st;
            if (g_plans.size() >= 1024) g_plans.clear();
            g_plans[desc] = fresh;
            cached = fresh;
        }
    }
    const HostPlan &hp = cached->hp;
    const bool use_chain = hp.chain;
    // backward = 2 .. 5: the step in three calls around a readout the CALLER computes (include/mpqe_amd.h: MPQE_STEP_PHASE_*,
    // MPQE_READOUT_CALLER; level form, every node state live)
    const int phase = backward;
    if (phase < 0 || phase > MPQE_STEP_PHASE_SCORES_ONLY) return MPQE_ERR_INVALID_ARG;
    const bool phase_fwd = phase == MPQE_STEP_PHASE_STATES, phase_bwd = phase == MPQE_STEP_PHASE_FROM_STATES;
    const bool phase_score = phase == MPQE_STEP_PHASE_SCORES || phase == MPQE_STEP_PHASE_SCORES_ONLY;
    if ((phase >= 2) != (P->readout == MPQE_READOUT_CALLER)) return MPQE_ERR_INVALID_ARG;
    const bool learned = P->readout >= MPQE_READOUT_MLP;       // (step_readout.h: the readout's two Linear layers are the library's too)
    if ((phase >= 2 || learned) && ((use_chain && !hp.ro_chain) || hp.nlanes > 1)) return MPQE_ERR_UNSUPPORTED;
    if (learned) {
        if (!P->readout_w0 || !P->readout_b0 || !P->readout_w2 || !P->readout_b2) return MPQE_ERR_INVALID_ARG;
        if (P->readout_scatter < MPQE_SCATTER_ADD || P->readout_scatter > MPQE_SCATTER_MEAN) return MPQE_ERR_INVALID_ARG;
        if (D_ok_for_readout(P->dim) == 0) return MPQE_ERR_UNSUPPORTED;
        if (backward && G && (!G->readout_w0 || !G->readout_b0 || !G->readout_w2 || !G->readout_b2)) return MPQE_ERR_INVALID_ARG;
        if (P->readout == MPQE_READOUT_CONCAT)
            for (int i = 0; i < nb; ++i)
                if (hp.sd.b[i].L != P->num_layers) return MPQE_ERR_INVALID_ARG;     // (model.py:441-446: one input block per layer)
    }
    // (the caller's readout read every level: its gradients of the intermediate levels are in the workspace already)
    const int add_states = ((phase_bwd && (P->flags & MPQE_STEP_ADD_STATE_GRADS)) || P->readout == MPQE_READOUT_CONCAT) ? 1 : 0;
    if (phase == MPQE_STEP_PHASE_SCORES_ONLY) backward = 0;        // (scores and loss from the caller's embeddings, no gradients)
    // touch plan given: the chain form stores per-entry table-gradient rows and sums them per destination (no atomics)
    bool use_touch = touch != nullptr && use_chain && backward;
    // ... BUILD_TOUCH: `touch` is an OUTPUT -- the step builds the plan of the ids it is called with inside its chain launch
    // (the level form has no use for a plan and leaves the buffer alone, as it ignores a plan built at pack time)
    const bool build_touch = use_touch && (P->flags & MPQE_STEP_BUILD_TOUCH) != 0;
    if (build_touch && (hp.ts_blocks <= 0 || (P->flags & MPQE_STEP_EIGHT_WAVES) || hp.nlanes > 1))
        return MPQE_ERR_UNSUPPORTED;        // (a step beyond TSORT_MAX_ENTRIES ids: build the plan at pack time)
    const bool sparse_tables = (P->flags & MPQE_STEP_SPARSE_TABLES) != 0;
    if (sparse_tables && backward && !use_touch) return MPQE_ERR_INVALID_ARG;      // (needs the touch plan and the chain form)
    int touch_row_bits = 1;         // (= the header of the caller's plan: mpqe_step_touch_build derives it the same way)
    if (use_touch) {
        long long trows = 1;
        for (int m = 0; m < P->num_modes; ++m) trows = std::max(trows, (long long)P->table_rows[m]);
        touch_row_bits = touch_bits(trows);
        if ((uintptr_t)touch % 256 != 0) return MPQE_ERR_INVALID_ARG;
        for (int m = 0; m < P->num_modes; ++m)
            if (G->tables[m] && (uintptr_t)G->tables[m] % 16 != 0) return MPQE_ERR_INVALID_ARG;
    }
    for (int l = 1; l < hp.nlanes; ++l)           // handles are per call, not part of the cached plan
        if (!lanes->fork_event || !lanes->aux_stream[l] || !lanes->join_event[l]) return MPQE_ERR_INVALID_ARG;
    if (!anchor_ids || !targets || !negs || !loss || !workspace) return MPQE_ERR_INVALID_ARG;
    if (desc_bytes < hp.desc_total) return MPQE_ERR_WORKSPACE;
    if ((uintptr_t)desc % 256 != 0) return MPQE_ERR_INVALID_ARG;
    if (backward && !G) return MPQE_ERR_INVALID_ARG;
    if (workspace_bytes < hp.total) return MPQE_ERR_WORKSPACE;
    if ((uintptr_t)workspace % 256 != 0) return MPQE_ERR_INVALID_ARG;
    if (!P->node_map || !P->mode_emb) return MPQE_ERR_INVALID_ARG;
    hipStream_t s = as_stream(stream);
    char *wb = reinterpret_cast<char *>(workspace);
    char *db = reinterpret_cast<char *>(desc);
    const int D = P->dim;
    const int NL = hp.nlanes;
    hipStream_t ls[MPQE_STEP_MAX_LANES];
    ls[0] = s;
    for (int l = 1; l < NL; ++l) ls[l] = as_stream(lanes->aux_stream[l]);
    // optional timing: event pair k brackets one launch, recorded on the stream of that launch
    // (see mpqe_amd.h for the order)
    int ev = 0;
    auto mark = [&](hipStream_t on) {
        if (events && ev < num_events) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(events[ev]), on);
        ++ev;
    };

    LayerPtrs lp;
    GradPtrs gp;
    TablePtrs tabs;
    memset(&lp, 0, sizeof(lp));
    memset(&gp, 0, sizeof(gp));
    memset(&tabs, 0, sizeof(tabs));
    int vec = D % 4 == 0;
    const bool fast_dims = D % GT_BN == 0;     // D is both K (multiple of 32) and the tile width (64)
    for (int l = 0; l < P->num_layers; ++l) {
        if (!P->basis[l] || !P->root[l]) return MPQE_ERR_INVALID_ARG;
        lp.basis[l] = P->basis[l];
        lp.root[l] = P->root[l];
        lp.bias[l] = P->bias[l];
        vec = vec && ptr_vec_ok(P->basis[l], D) && ptr_vec_ok(P->root[l], D);
        if (backward) {
            gp.basis[l] = G->basis[l];
            gp.root[l] = G->root[l];
            gp.bias[l] = G->bias[l];
        }
    }
    if (hp.ro_chain) {       // the readout's Linear layers: virtual layers of the chain form (HostPlan.ro_chain)
        lp.root[hp.ro_layer] = P->readout_w0;
        lp.bias[hp.ro_layer] = P->readout_b0;
        lp.root[hp.ro_layer + 1] = P->readout_w2;
        lp.bias[hp.ro_layer + 1] = P->readout_b2;
        if (backward) {
            gp.root[hp.ro_layer] = G->readout_w0;
            gp.bias[hp.ro_layer] = G->readout_b0;
            gp.root[hp.ro_layer + 1] = G->readout_w2;
            gp.bias[hp.ro_layer + 1] = G->readout_b2;
        }
    }
    int vec_tab = D % 4 == 0;
    for (int m = 0; m < P->num_modes; ++m) {
        if (!P->tables[m]) return MPQE_ERR_INVALID_ARG;
        vec_tab = vec_tab && (uintptr_t)P->tables[m] % 16 == 0;
        tabs.table[m] = P->tables[m];
        tabs.rows[m] = P->table_rows[m];
        tabs.grad[m] = backward ? G->tables[m] : nullptr;
    }
    if (backward) gp.mode_emb = G->mode_emb;

    const bool fast = vec && fast_dims;
    const StepDev *sd = reinterpret_cast<const StepDev *>(db + hp.o_sd);
    if (upload_desc) {
        // the descriptor table: ONE copy of the host image the plan keeps (the plan outlives the call: the cache holds
        // it); then the hand-off state of this packed step: epochs 0, every granule tagged 0 (a live tag is >= 1)
        (void)hipMemcpyAsync(db, hp.image.data(), hp.image.size(), hipMemcpyHostToDevice, s);
        (void)hipMemsetAsync(db + hp.o_epoch, 0, hp.desc_total - hp.o_epoch, s);
    }
    bool dev_weights = false;
    for (int i = 0; extra && i < nb; ++i) dev_weights = dev_weights || extra->batch_weight[i] != nullptr;
    unsigned *notify = extra ? reinterpret_cast<unsigned *>(extra->notify) : nullptr;
    const unsigned notify_value = extra ? extra->notify_value : 0u;
    if (extra && extra->query_out && !use_chain) return MPQE_ERR_UNSUPPORTED;       // (the chain workgroups' score phase writes it)
    if (dev_weights || (cached->weights_patched && !upload_desc)) {
        WeightPatch wp;
        memset(&wp, 0, sizeof(wp));
        wp.nb = nb;
        for (int i = 0; i < nb; ++i) {
            wp.whost[i] = hp.sd.b[i].weight;
            wp.wdev[i] = dev_weights ? extra->batch_weight[i] : nullptr;
        }
        hipLaunchKernelGGL(step_weights_kernel, dim3(1), dim3(64), 0, s, const_cast<StepDev *>(sd), wp);
    }
    cached->weights_patched = dev_weights;
    unsigned *epoch_f = reinterpret_cast<unsigned *>(db + hp.o_epoch), *epoch_b = epoch_f + 16;
    float *VT = reinterpret_cast<float *>(wb + hp.o_VT);
    UArgs ua;
    memset(&ua, 0, sizeof(ua));
    ua.chunks = D / 64;
    ua.VT = VT;
    ua.gran = reinterpret_cast<u64 *>(db + hp.o_gran);
    ua.mode_emb = P->mode_emb;
    ua.num_modes = (long long)P->num_modes;
    ua.parts = reinterpret_cast<float *>(wb + hp.o_parts);
    ua.err = err;
    float *H = reinterpret_cast<float *>(wb + hp.o_H), *GH = reinterpret_cast<float *>(wb + hp.o_GH);
    float *tpos = reinterpret_cast<float *>(wb + hp.o_tpos), *tneg = reinterpret_cast<float *>(wb + hp.o_tneg);
    float *spos = scores_pos ? scores_pos : reinterpret_cast<float *>(wb + hp.o_spos);
    float *sneg = scores_neg ? scores_neg : reinterpret_cast<float *>(wb + hp.o_sneg);
    float *terms = reinterpret_cast<float *>(wb + hp.o_terms);
    const long long *ids = reinterpret_cast<const long long *>(anchor_ids);
    const long long *tg = reinterpret_cast<const long long *>(targets), *ng = reinterpret_cast<const long long *>(negs);
    const long long *nm = reinterpret_cast<const long long *>(P->node_map);

    // Stream lanes: lane l runs the whole dependent chain (assemble -> levels -> score -> levels back)
    // of ITS batches on its own stream, so the ~8 us a short launch costs regardless of its size
    // overlaps with the other lanes' work; the lanes meet again before the weight gradients.
    long long row0[MPQE_STEP_MAX_LANES + 1], gr0[MPQE_STEP_MAX_LANES + 1];
    for (int l = 0; l <= NL; ++l) {
        const int b = hp.lane_begin[l];
        row0[l] = b < nb ? hp.sd.b[b].row_off : hp.sd.rows_total;
        gr0[l] = b < nb ? hp.sd.b[b].g_off : hp.sd.graphs_total;
    }
    float *WT = reinterpret_cast<float *>(wb + hp.o_WT);
    // prologue work: the forward pre-pass of the batch-uniform node states; backward: transposed weight copies for the
    // backward chains, zero fill of the gradients. Chain form: roles of the chain launch itself (PrepArgs); level form:
    // a zero-fill launch.
    PrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    long long zblocks = 0;
    int wt_all = 0;            // transposed-copy workgroups of a full launch of this packed step
    // merged launch: tiles + post-pass ride in the chain launch (include/mpqe_amd.h: MPQE_STEP_MERGE_TAIL)
    // Measured (AIFB mix, D = 128, B per batch 32 / 64 / 128 / 256 / 384 / 512 / 8192): merged 48.9 / 50.3 / 52.6 / 56.8 /
    // 61.3 / 68.4 / 569 us per step against 59.8 / 59.4 / 62.0 / 62.9 / 64.8 / 65.2 / 550 -- it wins while the chain
    // workgroups leave a free slot on (almost) every CU, and loses once the tiles have to share CUs with running chain
    // workgroups and queue behind them. Hence: merged up to 9/8 x CUs chain workgroups unless a flag says otherwise.
    const bool merged = use_chain && backward && NL == 1 && !(P->flags & MPQE_STEP_SPLIT_TAIL) &&
                        ((P->flags & MPQE_STEP_MERGE_TAIL) || hp.blk_off[nb] <= STEP_CUS + STEP_CUS / 8);
    // ... or only the POST-PASS (its vector ops are few, light, and a three-level dependence chain: it then runs while the
    // slower batches' chain workgroups are still at work, and the weight-gradient launch is its tiles alone). Experiment:
    // mpqe_debug_option POST_IN_CHAIN
    const bool post_only = use_chain && backward && NL == 1 && !merged && D % 64 == 0 && !hp.uops_b.empty() &&
                           hp.closures.empty() && exp_on("POST_IN_CHAIN") && !exp_on("FUSE_TAIL");
    const bool pic = merged || post_only;          // the post-pass rides in the chain launch
    {
        ZeroSegs &zs = pa.zs;
        if (backward && !phase_bwd && !phase_score && (P->flags & MPQE_STEP_ZERO_GRADS)) {     // (step in several calls: the first one fills)
            auto seg = [&](float *ptr, long long n) {
                if (!ptr || n <= 0) return;
                // (merged launch: a root matrix that tiles / a rank-1 op of the SAME launch write whole is not zero-filled
                // -- the fill would race with its writers, who store instead of adding)
                for (size_t k = 0; pic && k < hp.whole_roots.size(); ++k)
                    if (gp.root[hp.whole_roots[k]] == ptr) return;
                for (int k = 0; k < zs.count; ++k)
                    if (zs.p[k] == ptr) return;                  // shared layers repeat their buffers
                if (zs.count >= PREP_MAX_SEGS) return;
                zs.p[zs.count] = ptr;
                zs.n[zs.count] = n;
                zs.block0[zs.count] = zblocks;
                zblocks += (n + PREP_ZERO_FLOATS_PER_BLOCK - 1) / PREP_ZERO_FLOATS_PER_BLOCK;
                zs.count++;
            };
            for (int l = 0; l < P->num_layers; ++l) {
                // (relation matrices: the written ones are stored by their writers, the untouched ones are zero-filled
                // by spare workgroups of the weight-gradient launch, off the critical path: ZMat)
                seg(G->root[l], (long long)D * D);
                seg(G->bias[l], D);
            }
            seg(G->mode_emb, (long long)P->num_modes * D);
            if (learned) {
                seg(G->readout_w0, (long long)D * hp.ro_kin);
                seg(G->readout_b0, D);
                seg(G->readout_w2, (long long)D * D);
                seg(G->readout_b2, D);
            }
            // (SPARSE_TABLES: only the touched rows of the table gradients are ever read; they are written, not accumulated)
            if (use_chain && !use_touch && !sparse_tables) {
                // chain form WITHOUT a touch plan: the chain workgroups add into the tables with atomics -- a zero fill inside
                // their own launch would race with them: a launch of its own in front
                ZeroSegs zt;
                memset(&zt, 0, sizeof(zt));
                long long ztb = 0;
                for (int m = 0; m < P->num_modes && zt.count < PREP_MAX_SEGS; ++m) {
                    if (!G->tables[m] || P->table_rows[m] <= 0) continue;
                    bool dup = false;
                    for (int k = 0; k < zt.count; ++k) dup = dup || zt.p[k] == G->tables[m];
                    if (dup) continue;
                    zt.p[zt.count] = G->tables[m];
                    zt.n[zt.count] = (long long)P->table_rows[m] * D;
                    zt.block0[zt.count] = ztb;
                    ztb += (zt.n[zt.count] + PREP_ZERO_FLOATS_PER_BLOCK - 1) / PREP_ZERO_FLOATS_PER_BLOCK;
                    zt.count++;
                }
                zt.block0[zt.count] = ztb;
                if (ztb > 0) hipLaunchKernelGGL(step_zero_kernel, dim3((unsigned)ztb), dim3(256), 0, s, zt);
            } else
                for (int m = 0; m < P->num_modes && !sparse_tables; ++m) seg(G->tables[m], (long long)P->table_rows[m] * D);
            zs.block0[zs.count] = zblocks;
        }
        if (use_chain) {
            const int tpd = D / 64;
            pa.ua = ua;
            pa.ua.ops = reinterpret_cast<const UOp *>(db + hp.o_uopf);
            pa.ua.nops = (int)hp.uops_f.size();
            pa.ua.epoch = epoch_f;
            pa.ublocks = pa.ua.nops * pa.ua.chunks;
            // (forward only: just the copies a learned readout's forward multiplies by -- the plan lists them last... not
            // sorted: all of them are made, the backward levels' are then unused)
            pa.tblocks = (backward || hp.ro_chain) ? (int)hp.wt_slots.size() * tpd * tpd : 0;
            wt_all = pa.tblocks;
            if (!backward && hp.ro_chain && !dbg_on("FWD_ALL_COPIES")) {
                // (the readout's forward multiplies by the TRANSPOSED blocks of its own two layers; the relation matrices'
                // copies and the plain column blocks belong to the backward programmes)
                int n = 0;
                bool fits = true;
                for (size_t k = 0; k < hp.wt_slots.size(); ++k)
                    if (hp.wt_slots[k].mat < 0 && !hp.wt_slots[k].plain && hp.wt_slots[k].layer >= hp.ro_layer) {
                        if (n < 8) pa.tsel[n] = (int)k;
                        else fits = false;
                        ++n;
                    }
                if (fits && n > 0 && n < (int)hp.wt_slots.size()) {
                    pa.tsel_n = n;
                    pa.tblocks = n * tpd * tpd;
                    pa.tskip = (unsigned)(wt_all - pa.tblocks);
                }
            }
            pa.sblocks = 0;
            if (build_touch) {
                const TouchLayout TL = touch_layout(hp.touch_M, 0);
                char *tb = reinterpret_cast<char *>(touch);
                const size_t Mp = (size_t)hp.ts_blocks * TSORT_THREADS * tsort_rounds(hp.touch_M);
                TSortArgs &ts = pa.ts;
                ts.tm = reinterpret_cast<const TouchMeta *>(db + hp.o_tmeta);
                ts.anchor_ids = ids;
                ts.targets = tg;
                ts.negs = ng;
                ts.node_map = nm;
                ts.map_len = (long long)P->node_map_len;
                ts.ka = reinterpret_cast<unsigned *>(wb + hp.o_tsort);
                ts.kb = ts.ka + Mp;
                ts.va = ts.kb + Mp;
                ts.vb = ts.va + Mp;
                ts.hist = ts.vb + Mp;
                ts.counter = epoch_f + 40;
                ts.keys_out = reinterpret_cast<tkey_t *>(tb + TL.keys);
                ts.perm = reinterpret_cast<int *>(tb + TL.perm);
                ts.erow = nullptr;
                ts.th_out = reinterpret_cast<TouchHeader *>(tb);
                ts.M = (int)hp.touch_M;
                ts.key_bits = hp.ts_key_bits;
                ts.row_bits = hp.ts_row_bits;
                ts.nblk = hp.ts_blocks;
                ts.rounds = tsort_rounds(hp.touch_M);
                ts.fail = dbg_on("TSORT_FAIL") ? 1 : 0;
                ts.stamps = nullptr;
                if (dbg_on("TSORT_TRAIL")) pa.strail = hp.ts_blocks;
                else {
                    pa.sna = hp.sort_na;
                    pa.sxrank = 0;
                    for (int x = 0; x < STEP_XCDS; ++x) pa.sxrank |= (unsigned)(hp.sort_rank[x] + 1) << (4 * x);
                    pa.sblocks = (hp.ts_blocks + pa.sna - 1) / pa.sna * 8;
                }
            }
            // The chain workgroups wait for vectors / matrices that the prologue workgroups produce, so the prologue
            // workgroups come first in the launch: a producer is never queued behind a consumer. (Every wait is bounded
            // all the same: a launch that could not make progress reports MPQE_FLAG_INTERNAL instead of hanging.)
            // (Dealing the prologue workgroups only to the XCDs the chain workgroups leave room on was measured and is
            // worse: those are the XCDs of the heaviest batches, whose workgroups then lose their CU to themselves --
            // chain kernel 53.5 us against 41.4 with the prologue spread over all eight.)
            // (sblocks = 8 x rows; a row holds sna sort workgroups and 8 - sna prologue items)
            pa.lead = (pa.sblocks / 8 * pa.sna + pa.ublocks + pa.tblocks + 7) / 8 * 8;
            if (pa.lead < pa.sblocks) pa.lead = pa.sblocks;
            pa.nchain = (int)hp.crefs.size();
            if (NL == 1) {
                // (two workgroups per CU by registers and LDS; D = 256: one)
                const int slots = (D == 256 || (P->flags & MPQE_STEP_EIGHT_WAVES)) ? STEP_CUS : 2 * STEP_CUS;
#ifdef MPQE_EMU
                const bool fits = false && slots;       // (the host emulator runs a launch's workgroups one after the other, in order)
#else
                // (the placement grid's holes leave at once: only the real chain workgroups hold slots)
                const bool fits = pa.sblocks + hp.blk_off[nb] + 32 <= slots;
#endif
                const int force = mpqe_dbg_value("PROLOGUE_LAST", -1);       // (timing experiments)
                pa.plast = fits && hp.pl_na > 0 && (force >= 0 ? force != 0 : pa.lead > slots / 2) ? 1 : 0;
                if (pa.plast) {
                    pa.plna = hp.pl_na;
                    pa.plxrank = 0;
                    for (int x = 0; x < STEP_XCDS; ++x) pa.plxrank |= (unsigned)(hp.pl_rank[x] + 1) << (4 * x);
                    const int held = pa.sblocks / 8 * (8 - pa.sna);          // items the sort rows hold
                    const int rest = pa.ublocks + pa.tblocks > held ? pa.ublocks + pa.tblocks - held : 0;
                    pa.lead = pa.sblocks + (rest + pa.plna - 1) / pa.plna * 8;
                }
            }
            // (mpqe_step_extra_t.xcd_shift: idle workgroups in front of the chain workgroups move every one of them that many
            // XCDs on -- forward-only steps on several streams at once)
            if (!backward && extra && extra->xcd_shift > 0 && extra->xcd_shift < STEP_XCDS && !pa.plast && pa.sblocks == 0)
                pa.lead += extra->xcd_shift;
            if (dbg_on("DUMP_PLAN"))
                fprintf(stderr, "launch: sort rows %d (x8) | pre-pass %d transposes %d | lead %d | chain %d of %d | prologue behind the chain %d (XCDs %d)\n",
                        pa.sblocks / 8, pa.ublocks, pa.tblocks, pa.lead, hp.blk_off[nb], pa.nchain, pa.plast, pa.plna);
            pa.slots = reinterpret_cast<const WtSlot *>(db + hp.o_wtslots);
            pa.WT = WT;
            pa.wt_count = epoch_f + 32;
            pa.tail_arrive = nullptr;      // (set below once the launch form is known)
            pa.late = dbg_on("HANDOFF_LATE") ? 1 : 0;
            pa.fwd_done = pic && pa.ublocks > 0 ? epoch_f + 33 : nullptr;
            pa.ua.vt_through = pic ? 1 : 0;
        } else if (zblocks > 0) {
            hipLaunchKernelGGL(step_zero_kernel, dim3((unsigned)zblocks), dim3(256), 0, s, zs);
        }
    }
    LossMeta lm;
    memset(&lm, 0, sizeof(lm));
    lm.nb = nb;
    lm.chain = use_chain ? 1 : 0;
    for (int i = 0; i < nb; ++i) {
        lm.B[i] = hp.sd.b[i].B;
        lm.weight[i] = hp.sd.b[i].weight;
        lm.blk_off[i] = hp.blk_off[i];
    }
    lm.blk_off[nb] = hp.blk_off[nb];
    const float *bterms = reinterpret_cast<const float *>(wb + hp.o_bterms);
    if (NL > 1) {       // fork: the lanes start after the descriptor uploads and the prologue
        (void)hipEventRecord(reinterpret_cast<hipEvent_t>(lanes->fork_event), s);
        for (int l = 1; l < NL; ++l) (void)hipStreamWaitEvent(ls[l], reinterpret_cast<hipEvent_t>(lanes->fork_event), 0);
    }
    float *slabs = reinterpret_cast<float *>(wb + hp.o_slabs), *parts = reinterpret_cast<float *>(wb + hp.o_parts);
    TailArgs ta;
    ta.wsrc = reinterpret_cast<const WSource *>(db + hp.o_wsrc);
    ta.wblock = reinterpret_cast<const WBlock *>(db + hp.o_wblock);
    ta.nwsrc = (int)hp.wsrc.size();
    ta.wblocks = hp.wblocks_total;
    ta.vsrc = reinterpret_cast<const VSource *>(db + hp.o_vsrc);
    ta.vblock = reinterpret_cast<const int *>(db + hp.o_vblock);
    ta.nvsrc = (int)hp.vsrc.size();
    ta.vblocks = hp.vblocks_total;
    ta.anchor_off = reinterpret_cast<const int *>(db + hp.o_anchor);
    ta.nb = nb;
    ta.D = D;
    ta.tile_n = hp.tile_n;
    ta.ux = 0;
    memset(&ta.ca, 0, sizeof(ta.ca));
    ta.clpad = 0;
    ta.extra0 = -1;
    ta.runs_front = ta.runs_n = 0;
    ta.runs_out = nullptr;
    ta.tm_blocks = 0;
    ta.node_map = nm;
    ta.map_len = (long long)P->node_map_len;
    ta.anchor_ids = ids;
    ta.slabs = slabs;
    ta.parts = parts;
    // weight-gradient launch over the block table entries [first, first + count) on stream `on`
    ta.zmats = reinterpret_cast<const ZMat *>(db + hp.o_zmats);
    ta.zper = (int)(((long long)D * D + ZMAT_FLOATS_PER_BLOCK - 1) / ZMAT_FLOATS_PER_BLOCK);
    ta.zblocks = 0;
    ta.ublocks = 0;
    ta.stamps = nullptr;
    UArgs ub = ua;
    ub.ops = reinterpret_cast<const UOp *>(db + hp.o_uopb);
    ub.nops = (int)hp.uops_b.size();
    ub.epoch = epoch_b;
    // the step's reduction: a launch of its own, or (chain form, split tail, switch FUSE_TAIL) trailing workgroups of the
    // weight-gradient launch
    ReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.nmat = -1;
    const long long r_elems = (long long)D * D;
    const unsigned r_gx = (unsigned)((r_elems + 255) / 256);
    unsigned r_trows = 0;         // entity-table gradient rows: 256 / (D / 4) sorted positions per workgroup
    // (mpqe_debug_option ROWS_MULTI: a range of sorted positions per table workgroup, 344 instead of 2 752 workgroups for the
    // AIFB step -- measured 3 us SLOWER per step: a range is 3 - 4 dependent round trips per lane group where 2 752
    // independent one-run workgroups, two rounds of the chip at 6 waves per SIMD, need two each)
    const bool rows_multi = use_touch && D % 4 == 0 && 256 % (D / 4) == 0 && D >= 64 && exp_on("ROWS_MULTI");
    if (use_touch && !(STEP_DBG & 1)) {
        const long long per = rows_multi ? (256 / (D / 4)) * TSM_OWN : 256 / (D / 4), tblk = (hp.touch_M + per - 1) / per;
        r_trows = (unsigned)((tblk + r_gx - 1) / r_gx);
    }
    ra.groups = reinterpret_cast<const RGroup *>(db + hp.o_groups);
    ra.ngroups = (int)hp.groups.size();
    ra.D = D;
    ra.gp = gp;
    ra.slabs = slabs;
    ra.partial = parts;
    ra.vec = (int)(D % 4 == 0);
    ra.zeroed = (P->flags & MPQE_STEP_ZERO_GRADS) ? 1 : 0;
    ra.sd = sd;
    ra.terms = terms;
    ra.loss = loss;
    ra.lm = lm;
    ra.bterms = bterms;
    ra.rank1 = reinterpret_cast<const Rank1 *>(db + hp.o_rank1);
    ra.VT = VT;
    ra.epoch_b = use_chain ? epoch_b : nullptr;
    ra.touch = use_touch ? reinterpret_cast<const char *>(touch) : nullptr;
    ra.touch_keys = touch_layout(hp.touch_M, 0).keys;
    ra.touch_perm = touch_layout(hp.touch_M, 0).perm;
    ra.DG = reinterpret_cast<const float *>(wb + hp.o_DG);
    ra.tabs = tabs;
    ra.table_store = ((sparse_tables || (P->flags & MPQE_STEP_ZERO_GRADS)) ? 1 : 0) | (pic ? 2 : 0);
    ra.touch_M = (long long)hp.touch_M;
    ra.touch_row_bits = touch_row_bits;
    ra.rows_multi = rows_multi ? 1 : 0;
    ra.err = err;
    ra.notify = notify;
    ra.notify_value = notify_value;
    const bool fuse_tail = use_chain && backward && !pic && NL == 1 && D % 64 == 0 && exp_on("FUSE_TAIL");
    // split tail launch of the chain form: the loss and the entity-table rows depend on the chain launch alone -- they run as
    // trailing workgroups of the weight-gradient launch, beside its tiles (136 of 256 CUs busy on the AIFB step), instead of
    // in the reduction launch behind it (mpqe_debug_option LATE_ROWS = 1: as before)
    const bool can_early = use_chain && backward && !pic && !fuse_tail && NL == 1 && D % 4 == 0 && 256 % (D / 4) == 0;
    // (as trailing workgroups of the weight-gradient launch itself, mpqe_debug_option EARLY_ROWS = 1: measured slower -- that
    // launch's 230 VGPRs allow two workgroups per CU, a table workgroup took 8.6 us and the launch 6 us longer)
    const bool early_roles = can_early && exp_on("EARLY_ROWS");
    // (as a light launch of their own beside the weight-gradient launch -- enqueued behind it with hipExtAnyOrderLaunch, i.e.
    // without the queue's barrier bit -- was tried too: the flag is not honoured on gfx9 boards (hip_ext.h says so): the
    // launch ran in order and the step took 4.4 us longer)
    ra.early = early_roles ? 1 : 0;
    if (early_roles) r_trows = 0;
    // The table workgroups of the reduction launch take the plan's RUN STARTS, compacted by one workgroup of the weight-gradient
    // launch (touch_runs_block), instead of every sorted position: a step's distinct rows are at most the tables' rows -- the
    // launch is sized for that bound (AIFB step: 326 workgroups instead of 2 752). mpqe_debug_option NO_RUNS = 1: as before.
    const bool use_runs = use_touch && !pic && !fuse_tail && !early_roles && !rows_multi && NL == 1 && D % 4 == 0 &&
                          256 % (D / 4) == 0 && !(STEP_DBG & 1) && !dbg_on("NO_RUNS");
    if (use_runs) {
        long long total_rows = 0;
        for (int m = 0; m < P->num_modes; ++m) total_rows += P->table_rows[m];
        const long long rmax = std::min<long long>(hp.touch_M, total_rows), per = 256 / (D / 4);
        r_trows = (unsigned)(((rmax + per - 1) / per + r_gx - 1) / r_gx);
        ra.runs = reinterpret_cast<const int *>(wb + hp.o_runs);
        pa.runs_count = reinterpret_cast<int *>(wb + hp.o_runs) + hp.touch_M;
    }
    pa.tail_arrive = fuse_tail ? epoch_f + 41 : nullptr;
    bool reduced = false;
    bool zmats_done_in_chain = false;
    auto launch_grad_w = [&](hipStream_t on, int first, int count) {
        TailArgs tl = ta;
        tl.wblock = ta.wblock + first;
        tl.wblocks = count;
        if (first == 0 && (P->flags & MPQE_STEP_ZERO_GRADS) && !zmats_done_in_chain) tl.zblocks = (int)hp.zmats.size() * ta.zper;
        if (first == 0 && !post_only) tl.ublocks = ub.nops * ub.chunks;
        int nblocks = tl.ublocks + count + tl.zblocks;
#if MPQE_HAS_EXPERIMENTS
        const bool closures = use_chain && first == 0 && !hp.closures.empty() && !fuse_tail;
        if (closures) {
            tl.ublocks = 0;
            tl.ca.blocks = reinterpret_cast<const int *>(db + hp.o_closures);
            tl.ca.ncl = (int)hp.closures.size();
            tl.clpad = (tl.ca.ncl + 7) / 8 * 8;
            nblocks = tl.clpad + count + tl.zblocks;
        } else
#endif
        {
            // chain form: two of the eight XCDs for the post-pass' vector ops, six for the tiles (AIFB step, same box,
            // three runs each: 64.95 / 65.15 / 65.04 us against 65.70 / 65.60 / 65.47 with both kinds everywhere; one
            // or three XCDs: 65.8 / 66.0). mpqe_debug_option TAIL_UX overrides (0 = everywhere).
            const int uxv = mpqe_dbg_value("TAIL_UX", 2);
            // (only while the tiles are all resident at once on the other XCDs -- two per CU: with more of them the vector
            // ops' XCDs would stand idle for most of the launch. AIFB step with the MLP readout, 988 tiles: 64.3 -> 52.9 us)
            if (use_chain && first == 0 && tl.ublocks >= 4 && uxv > 0 && uxv < 8 &&
                (count <= (8 - uxv) * 2 * (STEP_CUS / STEP_XCDS) || mpqe_dbg_value("TAIL_UX", -1) > 0)) {
                tl.ux = uxv;
                const int ra = (tl.ublocks + tl.ux - 1) / tl.ux, rb = (count + tl.zblocks + (8 - tl.ux) - 1) / (8 - tl.ux);
                nblocks = 8 * (ra > rb ? ra : rb);
            }
        }
        if (early_roles && first == 0 && count == hp.wblocks_total) {
            const int lpr = D / 4, pos = (256 / lpr) * TSM_OWN;
            tl.extra0 = (nblocks + 7) / 8 * 8;
            tl.tm_blocks = use_touch ? (int)((hp.touch_M + pos - 1) / pos) : 0;
            nblocks = tl.extra0 + 1 + tl.tm_blocks;
        }
        if (use_runs && first == 0 && count == hp.wblocks_total) {       // a few workgroups in front: the touch plan's run starts
            tl.runs_n = (int)((hp.touch_M + TRUNS_PER - 1) / TRUNS_PER);
            tl.runs_front = (tl.runs_n + 7) / 8 * 8;
            tl.runs_out = reinterpret_cast<int *>(wb + hp.o_runs);
            nblocks += tl.runs_front;
        }
        tl.stamps = g_tail_stamps && (size_t)nblocks <= g_tail_stamp_blocks ? g_tail_stamps : nullptr;
        if (nblocks <= 0) return;
        dim3 tgrid((unsigned)nblocks);
        const int zeroed = (P->flags & MPQE_STEP_ZERO_GRADS) ? 1 : 0;
        FuseArgs fa;
        memset(&fa, 0, sizeof(fa));
#if MPQE_HAS_EXPERIMENTS
        if (use_chain && fuse_tail && first == 0 && count == hp.wblocks_total) {
            // [tiles / vector ops / zero fill as before][table rows][loss][reduction groups]: the groups wait for the tiles
            // and vector ops, which come before them in the launch
            fa.first = (nblocks + 7) / 8 * 8;
            fa.gx = (int)r_gx;
            fa.trows = (int)r_trows;
            fa.tx = 8;
            fa.tspan = (fa.trows * fa.gx + fa.tx - 1) / fa.tx * 8;
            fa.arrive = epoch_f + 41;
            ReduceArgs rf = ra;
            rf.arrive = fa.arrive;
            rf.phase1 = (unsigned)(tl.ublocks + count);
            UArgs uf = ub;
            uf.vt_through = 2;      // (only the outputs the reduction reads: UOp.through)
            dim3 fgrid((unsigned)(fa.first + fa.tspan + 1 + (int)(hp.groups.size() * r_gx)));
            hipLaunchKernelGGL((step_tail_kernel<LD_T, true>), fgrid, dim3(256), 0, on, sd, tl, (const float *)H,
                               (const float *)GH, hp.level_stride, gp, zeroed, lp, uf, fa, rf);
            reduced = true;
        } else
#endif
        if (use_chain)
            // (ONE tile workgroup per CU -- the launch's LDS padded beyond half a CU's -- was measured on the 988-tile step of
            // the MLP readout: 71 - 75 us against 64; two per CU stay)
            hipLaunchKernelGGL(step_tail_kernel<LD_T>, tgrid, dim3(256), 0, on, sd, tl, (const float *)H,
                               (const float *)GH, hp.level_stride, gp, zeroed, lp, ub, fa, ra);
        else if (fast && hp.whole_ksteps)
            hipLaunchKernelGGL(step_tail_kernel<LD_FAST>, tgrid, dim3(256), 0, on, sd, tl, (const float *)H,
                               (const float *)GH, hp.level_stride, gp, zeroed, lp, ub, fa, ra);
        else if (vec)
            hipLaunchKernelGGL(step_tail_kernel<LD_PRED>, tgrid, dim3(256), 0, on, sd, tl, (const float *)H,
                               (const float *)GH, hp.level_stride, gp, zeroed, lp, ub, fa, ra);
        else
            hipLaunchKernelGGL(step_tail_kernel<LD_SCALAR>, tgrid, dim3(256), 0, on, sd, tl, (const float *)H,
                               (const float *)GH, hp.level_stride, gp, zeroed, lp, ub, fa, ra);
    };
    // the readout's regulariser (model.py:486-490), after the launch that writes loss[0]
    auto ro_regulariser = [&](bool with_grads) {
        float wsum = 0.f;
        for (int i = 0; i < nb; ++i) wsum += hp.sd.b[i].weight;
        if (!(P->readout_weight_decay > 0.f)) return;
        RoRegArgs rr;
        memset(&rr, 0, sizeof(rr));
        rr.p[0] = P->readout_w0; rr.p[1] = P->readout_b0; rr.p[2] = P->readout_w2; rr.p[3] = P->readout_b2;
        rr.n[0] = (long long)D * hp.ro_kin; rr.n[1] = D; rr.n[2] = (long long)D * D; rr.n[3] = D;
        if (with_grads) { rr.g[0] = G->readout_w0; rr.g[1] = G->readout_b0; rr.g[2] = G->readout_w2; rr.g[3] = G->readout_b2; }
        rr.coef = P->readout_weight_decay * wsum;
        rr.loss = loss;
        if (dev_weights && with_grads) {        // (the gradients' coefficient: weight_decay x sum_i host_i x *device_i, formed on the device)
            rr.nw = nb;
            rr.wd = P->readout_weight_decay;
            for (int i = 0; i < nb; ++i) {
                rr.whost[i] = hp.sd.b[i].weight;
                rr.wdev[i] = extra->batch_weight[i];
            }
        }
        hipLaunchKernelGGL(step_ro_reg_kernel, dim3(1), dim3(1024), 0, s, rr);
    };
    bool loss_in_chain = false, reg_in_chain = false;
    if (use_chain) {
        // assemble -> levels -> scores (-> levels back -> anchor-table gradients): one launch per lane
        ChainArgs ca;
        ca.refs = reinterpret_cast<const ChainRef *>(db + hp.o_cref);
        ca.ops = reinterpret_cast<const ChainOp *>(db + hp.o_cops);
        ca.node_map = nm;
        ca.map_len = (long long)P->node_map_len;
        ca.mode_emb = P->mode_emb;
        ca.num_modes = (long long)P->num_modes;
        ca.anchor_ids = ids;
        ca.targets = tg;
        ca.negs = ng;
        ca.H = H;
        ca.GH = GH;
        ca.WT = WT;
        ca.VT = VT;
        ca.epoch_f = epoch_f;
        ca.DG = use_touch ? reinterpret_cast<float *>(wb + hp.o_DG) : nullptr;
        // (a plan built at pack time also holds the id -> table row hop of every entry; a step that builds its own plan
        // resolves the ids itself)
        ca.erow = use_touch && !build_touch ? reinterpret_cast<const int *>(reinterpret_cast<const char *>(touch) +
                                                                            touch_layout(hp.touch_M, 0).erow) : nullptr;
        ca.Manchor = (long long)hp.anchor_off[nb];
        ca.Gtot = hp.sd.graphs_total;
        ca.parts = reinterpret_cast<float *>(wb + hp.o_parts);
        ca.block_terms = reinterpret_cast<float *>(wb + hp.o_bterms);
        ca.level_stride = hp.level_stride;
        ca.margin = margin;
        ca.eps = 1e-8f;
        ca.s_pos = spos;
        ca.s_neg = sneg;
        ca.terms = terms;
        ca.q_out = extra ? extra->query_out : nullptr;
        ca.err = err;
        ca.backward = backward ? 1 : 0;
        ca.stamps = g_chain_stamps && 2 * hp.crefs.size() + (size_t)hp.ts_blocks <= g_chain_stamp_blocks ? g_chain_stamps : nullptr;
        if (ca.stamps && build_touch) pa.ts.stamps = g_chain_stamps + 16 * (long long)hp.crefs.size();     // (behind the chain entries)
        {
            ca.cb = 0;
            ca.nchain = pa.nchain;
            ca.cv_gran = pa.ublocks > 0 ? reinterpret_cast<const unsigned long long *>(db + hp.o_gran) : nullptr;
            ca.epoch_b = epoch_b;
            ca.wt_count = pa.tblocks > 0 ? pa.wt_count : nullptr;
            ca.wt_blocks = wt_all;
            // (counters and their epoch advance on merged steps only: targets are epoch x count)
            ca.done = pic ? reinterpret_cast<unsigned *>(db + hp.o_done) : nullptr;
            ca.arrive = ca.done ? ca.done + hp.done_inc.size() : nullptr;
            ca.done_inc = reinterpret_cast<const int *>(db + hp.o_done_inc);
            ca.ro = hp.ro_chain ? 1 : 0;
            ca.wt_early = hp.ro_chain && P->readout == MPQE_READOUT_CONCAT ? 1 : 0;
            ca.ro_layer = hp.ro_layer;
            ca.ro_scatter = P->readout_scatter;
            PostArgs po;
            memset(&po, 0, sizeof(po));
            long long grid_blocks = pa.lead + pa.nchain + zblocks;
            // (split tail: the untouched relation matrices' zero fill rides behind the chain workgroups; mpqe_debug_option
            // ZMATS_IN_TAIL = 1: by workgroups of the weight-gradient launch, as before)
            const bool zm_here = !pic && backward && (P->flags & MPQE_STEP_ZERO_GRADS) && !hp.zmats.empty() && !dbg_on("ZMATS_IN_TAIL") &&
                                 !phase_bwd && !phase_score;
            if (zm_here) {
                po.zmblocks = (int)hp.zmats.size() * ta.zper;
                po.zmats = ta.zmats;
                po.zper = ta.zper;
                po.D = D;
                po.gp = gp;
                grid_blocks += po.zmblocks;
                zmats_done_in_chain = true;
            }
            if (pic) {
                unsigned *done = reinterpret_cast<unsigned *>(db + hp.o_done);
                po.zpad = (int)((zblocks + 7) / 8 * 8);
                if (po.zpad == 0) po.zpad = 8;              // (> 0 marks the merged launch)
                po.zmblocks = (merged && (P->flags & MPQE_STEP_ZERO_GRADS)) ? (int)hp.zmats.size() * ta.zper : 0;
                po.ublocks = ub.nops * ub.chunks;
                po.na = hp.post_na;
                po.xrank = 0;
                for (int x = 0; x < STEP_XCDS; ++x) po.xrank |= (unsigned)(hp.post_rank[x] + 1) << (4 * x);
                po.ppad = (po.zmblocks + po.ublocks + po.na - 1) / po.na * po.na;
                po.wblocks = merged ? hp.wblocks_total : 0;      // (post-pass only: the tiles stay a launch of their own)
                po.zper = ta.zper;
                po.D = D;
                po.tile_n = hp.tile_n;
                po.zeroed = (P->flags & MPQE_STEP_ZERO_GRADS) ? 1 : 0;
                po.ub = ub;
                po.ub.done = done;
                po.ub.done_inc = reinterpret_cast<const int *>(db + hp.o_done_inc);
                po.ub.dm = hp.dm;
                po.ub.fwd_done = pa.fwd_done;
                po.ub.epoch_m = epoch_f + 48;
                po.ub.fwd_blocks = pa.ublocks;
                po.wblock = ta.wblock;
                po.zmats = ta.zmats;
                po.slabs = slabs;
                po.H = H;
                po.GH = GH;
                po.level_stride = hp.level_stride;
                po.gp = gp;
                po.done = done;
                po.done_inc = po.ub.done_inc;
                po.epoch_m = epoch_f + 48;
                po.err = err;
                po.stamps = g_tail_stamps && (size_t)po.wblocks <= g_tail_stamp_blocks ? g_tail_stamps : nullptr;
                grid_blocks = pa.lead + pa.nchain + po.zpad +
                              (long long)(po.ppad + po.wblocks + po.na - 1) / po.na * 8;       // (8 workgroups per `na` items)
            }
            grid_blocks += pa.strail;
            dim3 cgrid((unsigned)grid_blocks);
            // forward-only: loss, epoch advance and notification by the launch's last workgroup (chain_finish) instead of a
            // launch behind it. mpqe_debug_option LOSS_LAUNCH = 1: step_loss_kernel as before
            FinArgs fin;
            memset(&fin, 0, sizeof(fin));
            if (!backward && NL == 1 && !dbg_on("LOSS_LAUNCH")) {
                fin.count = epoch_f + 42;
                fin.loss = loss;
                fin.bterms = bterms;
                fin.epoch_f = epoch_f;
                fin.bump_b = pa.tblocks > 0 ? 1 : 0;
                fin.notify = notify;
                fin.notify_value = notify_value;
                fin.err = err;
                fin.lm = lm;
                loss_in_chain = true;
                if (learned && extra && extra->readout_norms && P->readout_weight_decay > 0.f) {
                    float wsum = 0.f;
                    for (int i = 0; i < nb; ++i) wsum += hp.sd.b[i].weight;
                    fin.reg_norms = extra->readout_norms;
                    fin.reg_coef = P->readout_weight_decay * wsum;
                    reg_in_chain = true;
                }
            }
            mark(s);
#define LAUNCH_CHAIN(THREADS, ...)                                                                                       \
    do {                                                                                                                 \
        if (fin.count)                                                                                                   \
            hipLaunchKernelGGL((step_chain_fwd_kernel<__VA_ARGS__>), cgrid, dim3(THREADS), 0, s, sd, lp, tabs, ca, pa, po, fin); \
        else                                                                                                             \
            hipLaunchKernelGGL((step_chain_kernel<__VA_ARGS__>), cgrid, dim3(THREADS), 0, s, sd, lp, tabs, ca, pa, po);  \
    } while (0)
            if (hp.ro_chain) {
                // (a learned readout on the chain: its own instances -- the others' code stays as it was)
                if (D == 64) LAUNCH_CHAIN(256, 1, 1, 4, true);
                else if (D == 128 && (P->flags & MPQE_STEP_NO_KSPLIT)) LAUNCH_CHAIN(256, 2, 1, 4, true);
                else if (D == 128) LAUNCH_CHAIN(256, 4, 2, 4, true);
                else LAUNCH_CHAIN(256, 4, 1, 4, true);
            } else if (D == 64) LAUNCH_CHAIN(256, 1, 1);
            else if (D == 128 && (P->flags & MPQE_STEP_NO_KSPLIT)) LAUNCH_CHAIN(256, 2, 1);
            else if (D == 128 && (P->flags & MPQE_STEP_EIGHT_WAVES)) LAUNCH_CHAIN(512, 2, 2, 8);
            else if (D == 128) LAUNCH_CHAIN(256, 4, 2);
            else LAUNCH_CHAIN(256, 4, 1);
#undef LAUNCH_CHAIN
            mark(s);
        }
        if (!backward) {
            for (int l = 1; l < NL; ++l) {
                (void)hipEventRecord(reinterpret_cast<hipEvent_t>(lanes->join_event[l]), ls[l]);
                (void)hipStreamWaitEvent(s, reinterpret_cast<hipEvent_t>(lanes->join_event[l]), 0);
            }
            if (!loss_in_chain)
                hipLaunchKernelGGL(step_loss_kernel, dim3(1), dim3(1024), 0, s, sd, (const float *)terms, loss, lm, bterms,
                                   use_chain ? epoch_f : (unsigned *)nullptr, pa.tblocks > 0 ? 1 : 0, notify, notify_value,
                                   (const int32_t *)err);
            if (learned && !reg_in_chain) ro_regulariser(false);
            return mpqe_launch_status();
        }
        // (a side stream for the post-pass / table rows beside the tiles was measured: the cross-stream fork and join
        // cost more than the overlap gains -- 93.8 us per step against 81.8 with everything on one stream)
        mark(s);
        if (!merged) launch_grad_w(s, 0, hp.wblocks_total);
        mark(s);
    }
    // ---- forward
    const float *Qc = P->readout == MPQE_READOUT_CALLER ? reinterpret_cast<const float *>(wb + hp.o_Q) : nullptr;
    float *GQc = P->readout == MPQE_READOUT_CALLER ? reinterpret_cast<float *>(wb + hp.o_GQ) : nullptr;
    // learned readouts: gather -> Linear - ReLU - Linear -> reduction over each graph's rows, and the way back
    RoArgs roa;
    memset(&roa, 0, sizeof(roa));
    roa.kind = P->readout;
    roa.op = P->readout_scatter;
    roa.mrows = hp.ro_rows;
    roa.kin = hp.ro_kin;
    roa.level_stride = hp.level_stride;
    float *ro_x = nullptr, *ro_gx = nullptr, *ro_h = nullptr, *ro_y = nullptr, *ro_gy = nullptr, *ro_gh = nullptr;
    if (learned) {
        const long long lv = (long long)hp.sd.b[0].L * hp.level_stride;
        ro_x = hp.ro_direct ? H + lv : reinterpret_cast<float *>(wb + hp.o_rx);
        ro_gx = hp.ro_direct ? GH + lv : reinterpret_cast<float *>(wb + hp.o_rgx);
        ro_h = reinterpret_cast<float *>(wb + hp.o_rh);
        ro_y = reinterpret_cast<float *>(wb + hp.o_ry);
        ro_gy = reinterpret_cast<float *>(wb + hp.o_rgy);
        ro_gh = reinterpret_cast<float *>(wb + hp.o_rgh);
    }
    auto ro_blocks = [](long long threads) { return dim3((unsigned)((threads + 255) / 256)); };
    auto ro_forward = [&]() -> int {
        if (!hp.ro_direct)
            hipLaunchKernelGGL(step_ro_gather_kernel, ro_blocks(roa.mrows * (roa.kin / 4)), dim3(256), 0, s, sd, roa,
                               (const float *)H, ro_x);
        int st = mpqe_linear_fwd(ro_x, roa.mrows, P->readout_w0, roa.kin, P->readout_b0, roa.kin, D, 1, 0, ro_h, s);
        if (st) return st;
        st = mpqe_linear_fwd(ro_h, roa.mrows, P->readout_w2, D, P->readout_b2, D, D, 0, 0, ro_y, s);
        if (st) return st;
        return MPQE_OK;         // (the reduction over each graph's rows: inside the score kernel)
    };
    auto ro_backward = [&]() -> int {
        void *lw = wb + hp.o_rlin;         // (the score kernel has written the rows' gradients)
        int st = mpqe_linear_bwd(ro_h, roa.mrows, P->readout_w2, D, ro_y, ro_gy, D, D, 0, 0, ro_gh, G->readout_w2, D,
                                 G->readout_b2, lw, hp.rlin_bytes, s);
        if (st) return st;
        st = mpqe_linear_bwd(ro_x, roa.mrows, P->readout_w0, roa.kin, ro_h, ro_gh, roa.kin, D, 1, 0, ro_gx,
                             G->readout_w0, roa.kin, G->readout_b0, lw, hp.rlin_bytes, s);
        if (st) return st;
        if (!hp.ro_direct)
            hipLaunchKernelGGL(step_ro_spread_kernel, ro_blocks(hp.sd.rows_total * (D / 4)), dim3(256), 0, s, sd, roa,
                               (const float *)ro_gx, GH);
        return MPQE_OK;
    };
    for (int l = 0; !use_chain && !phase_bwd && !phase_score && l < NL; ++l) {
        const long long nr = row0[l + 1] - row0[l], ngr = gr0[l + 1] - gr0[l];
        const long long waves = nr + 2 * ngr;
        const int lpr_h = [&] { if (!vec_tab) return 64; int q = 1; while (q < 64 && q * 4 < D) q <<= 1; return q; }();
        const long long per_block = 4 * (64 / lpr_h);
        hipLaunchKernelGGL(step_assemble_kernel, dim3((unsigned)((waves + per_block - 1) / per_block)), dim3(256), 0,
                           ls[l], sd, tabs, nm, (long long)P->node_map_len, P->mode_emb, (long long)P->num_modes, ids,
                           tg, ng, H, tpos, tneg, err, vec_tab, row0[l], nr, gr0[l], ngr);
    }
    for (int p = 0; !use_chain && !phase_bwd && !phase_score && p < hp.Lmax; ++p)
        for (int l = 0; l < NL; ++l) {
            if (p >= hp.lane_Lmax[l]) continue;
            const float *hin = H + (long long)p * hp.level_stride;
            float *hout = H + (long long)(p + 1) * hp.level_stride;
            const TileRef *gf = reinterpret_cast<const TileRef *>(db + hp.o_tf[l][p]);
            dim3 grid((unsigned)hp.tfwd[l][p].size());
            mark(ls[l]);
            if (fast)
                hipLaunchKernelGGL(step_layer_fwd_kernel<LD_FAST>, grid, dim3(256), 0, ls[l], sd, lp, p, gf, hin, hout);
            else if (vec)
                hipLaunchKernelGGL(step_layer_fwd_kernel<LD_PRED>, grid, dim3(256), 0, ls[l], sd, lp, p, gf, hin, hout);
            else
                hipLaunchKernelGGL(step_layer_fwd_kernel<LD_SCALAR>, grid, dim3(256), 0, ls[l], sd, lp, p, gf, hin,
                                   hout);
            mark(ls[l]);
        }
#define LAUNCH_SCORE(BWD, NJ, GHP, L)                                                                               \
    hipLaunchKernelGGL((step_score_kernel<BWD, NJ>), dim3((unsigned)((gr0[L + 1] - gr0[L] + 3) / 4)), dim3(256), 0, \
                       ls[L], sd, (const float *)H, hp.level_stride, (const float *)tpos, (const float *)tneg,     \
                       margin, 1e-8f, spos, sneg, terms, GHP, tabs, nm, (long long)P->node_map_len, tg, ng, gr0[L], \
                       gr0[L + 1] - gr0[L], (const float *)Qc, GQc, \
                       (const float *)(learned ? ro_y : nullptr), learned ? ro_gy : (float *)nullptr, roa.op)
#define LAUNCH_SCORE_D(BWD, GHP, L)                  \
    if (D <= 64) LAUNCH_SCORE(BWD, 1, GHP, L);       \
    else if (D <= 128) LAUNCH_SCORE(BWD, 2, GHP, L); \
    else if (D <= 256) LAUNCH_SCORE(BWD, 4, GHP, L); \
    else LAUNCH_SCORE(BWD, 8, GHP, L)
    auto join = [&]() {
        for (int l = 1; l < NL; ++l) {
            (void)hipEventRecord(reinterpret_cast<hipEvent_t>(lanes->join_event[l]), ls[l]);
            (void)hipStreamWaitEvent(s, reinterpret_cast<hipEvent_t>(lanes->join_event[l]), 0);
        }
    };
    if (phase_fwd) return mpqe_launch_status();     // the node states of every level are in the workspace (mpqe_step_states_layout)
    if (learned && !use_chain) {
        const int st = ro_forward();
        if (st) return st;
    }
    if (!backward) {      // (not reached with the chain kernel)
        for (int l = 0; l < NL; ++l) { LAUNCH_SCORE_D(false, (float *)nullptr, l); }
        join();
        hipLaunchKernelGGL(step_loss_kernel, dim3(1), dim3(1024), 0, s, sd, (const float *)terms, loss, lm, bterms,
                           use_chain ? epoch_f : (unsigned *)nullptr, 0, notify, notify_value, (const int32_t *)err);
        if (learned) ro_regulariser(false);
        return mpqe_launch_status();
    }

    // ---- backward (the score kernel's backward instance writes scores and hinge terms too; the loss
    // itself is reduced by the last launch of the step)
    // (the caller's readout: its own call for the scores -- embeddings in, their gradients out --, then the caller writes the
    // rows of gH[L_b] and the last call takes it from there)
    for (int l = 0; !use_chain && !phase_bwd && l < NL; ++l) { LAUNCH_SCORE_D(true, GH, l); }
    if (phase_score) return mpqe_launch_status();
    if (learned && !use_chain) {
        const int st = ro_backward();
        if (st) return st;
    }
#undef LAUNCH_SCORE_D
#undef LAUNCH_SCORE
    for (int p = hp.Lmax - 1; !use_chain && p >= 0; --p)
        for (int l = 0; l < NL; ++l) {
            if (p >= hp.lane_Lmax[l]) continue;
            const float *gout = GH + (long long)(p + 1) * hp.level_stride;
            const float *hin = H + (long long)p * hp.level_stride;
            float *gin = GH + (long long)p * hp.level_stride;
            const TileRef *gb = reinterpret_cast<const TileRef *>(db + hp.o_tb[l][p]);
            dim3 grid((unsigned)hp.tbwd[l][p].size());
            mark(ls[l]);
            if (fast)
                hipLaunchKernelGGL(step_layer_bwd_x_kernel<LD_FAST>, grid, dim3(256), 0, ls[l], sd, lp, p, gb, gout,
                                   hin, gin, add_states);
            else if (vec)
                hipLaunchKernelGGL(step_layer_bwd_x_kernel<LD_PRED>, grid, dim3(256), 0, ls[l], sd, lp, p, gb, gout,
                                   hin, gin, add_states);
            else
                hipLaunchKernelGGL(step_layer_bwd_x_kernel<LD_SCALAR>, grid, dim3(256), 0, ls[l], sd, lp, p, gb,
                                   gout, hin, gin, add_states);
            mark(ls[l]);
        }
    join();
    if (!use_chain) {
        mark(s);
        launch_grad_w(s, 0, hp.wblocks_total);
        mark(s);
        // bias / variable-row partials and anchor-table gradients (the chain kernel does them itself)
        const unsigned small_blocks = (unsigned)(ta.vblocks + (hp.anchor_off[nb] + 3) / 4);
        if (small_blocks)
            hipLaunchKernelGGL(step_tail_small_kernel, dim3(small_blocks), dim3(256), 0, s, sd, ta, tabs,
                               (const float *)H, (const float *)GH, hp.level_stride);
    }
    if (!reduced) {
        // (matrix groups first in the table, vector groups behind them: then the vector groups share ONE row of the launch)
        int nmat = 0;
        const int ng = (int)hp.groups.size();
        while (nmat < ng && (hp.groups[nmat].kind <= 1 || hp.groups[nmat].kind >= 4)) ++nmat;
        bool packed = ra.vec && D % 4 == 0 && 256 % (D / 4) == 0 && !dbg_on("REDUCE_ROWS") && (ng - nmat) * VEC_SLICES + 1 <= (int)r_gx;
        for (int k = nmat; k < ng; ++k) packed = packed && (hp.groups[k].kind == 2 || hp.groups[k].kind == 3);
        ra.nmat = packed ? nmat : -1;
        dim3 grid(r_gx, (unsigned)(packed ? nmat + 1 : ng + 1) + r_trows);
        mark(s);
        hipLaunchKernelGGL(step_reduce_kernel, grid, dim3(256), 0, s, ra);
        mark(s);
    }
    if (learned) ro_regulariser(true);
    return mpqe_launch_status();
}

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
#include "step_closure.h"
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
    unsigned *fwd_done;       // vector-op workgroups finished, ever (uop_wait_prepass); NULL: none
    ZeroSegs zs;
};

// one 64 x 64 piece of a matrix, transposed: T[c][r] = W[r][c] (D % 64 == 0 in the chain form)
__device__ __forceinline__ void prep_transpose_block(const LayerPtrs &lp, const PrepArgs &pa, int D, int tb, float *smem) {
    float(*tile)[65] = reinterpret_cast<float(*)[65]>(smem);
    const int tpd = D / 64, per = tpd * tpd;
    const int si = tb / per, tr = (tb % per) / tpd, tc = tb % tpd;
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
        atomicAdd(pa.wt_count, 1u);
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
    // role of this workgroup (uniform): chain workgroup, prologue work in front of / behind them, zero fill, then the
    // post roles: a producer is never queued behind a consumer that waits for it
    int bid = (int)blockIdx.x, role;
    if (bid == 0 && threadIdx.x == 0 && pa.tail_arrive) *pa.tail_arrive = 0u;       // (read by the NEXT launch)
    if (bid == 0 && threadIdx.x == 0 && pa.runs_count) *pa.runs_count = 0;
    if (pa.strail && bid >= (int)gridDim.x - pa.strail) {
        if (NW == 4 || threadIdx.x < TSORT_THREADS) tsort_block(pa.ts, bid - ((int)gridDim.x - pa.strail), reinterpret_cast<unsigned *>(S.xs));
        return;
    }
    if (pa.plast) {
        if (bid < pa.sblocks) role = 1;
        else if (bid < pa.sblocks + pa.nchain) role = 0, bid -= pa.sblocks;
        else if (bid < pa.lead + pa.nchain) {
            const int t = bid - pa.sblocks - pa.nchain;
            const int rk = (int)((pa.plxrank >> (4 * (t & 7))) & 15u) - 1;
            if (rk < 0) return;
            role = 1, bid = pa.sblocks + (t >> 3) * pa.plna + rk;      // (numbered on from the items the sort rows held)
        } else role = 2, bid -= pa.lead + pa.nchain;
    } else if (bid < pa.lead) role = 1;
    else if (bid < pa.lead + pa.nchain) role = 0, bid -= pa.lead;
    else role = 2, bid -= pa.lead + pa.nchain;
    if (role == 0) {
#ifndef MPQE_EMU
        if (po.zpad > 0) __builtin_amdgcn_s_setprio(1);      // (merged launch: over the tiles that may share the CU)
#endif
        ca.cb = bid;
        ca.nchain = pa.nchain;
        chain_block<NCB, KS, NW, RO>(sd, lp, tabs, ca, S);
    } else if (role == 1) {
        constexpr int D = 16 * NCB * NW / KS;
        // The touch plan of THIS step's ids (step_touch.h): the first workgroups of the launch, so all of them are
        // resident before any other workgroup is dispatched (they synchronise among themselves); nothing in the launch
        // waits for them -- the plan is read by the step's last launch.
        if (bid < pa.sblocks) {
            static_assert(sizeof(S.xs) >= TSORT_LDS_WORDS * sizeof(unsigned), "the sort's LDS tables live in the state buffers");
            const int rk = (int)((pa.sxrank >> (4 * (bid & 7))) & 15u) - 1;
            if (rk >= 0) {
                const int sb = (bid >> 3) * pa.sna + rk;
                if (sb < pa.ts.nblk && (NW == 4 || threadIdx.x < TSORT_THREADS))
                    tsort_block(pa.ts, sb, reinterpret_cast<unsigned *>(S.xs));
                return;
            }
            // (not one of the sort's XCDs: the next prologue item -- no hole in front of the chain workgroups of the XCDs
            // whose CUs are all needed; its rank among the other XCDs)
            int orank = 0;
            for (int x = 0; x < (bid & 7); ++x) orank += ((pa.sxrank >> (4 * x)) & 15u) == 0u;
            bid = (bid >> 3) * (8 - pa.sna) + orank;
        } else {
            bid -= pa.sblocks / 8 * pa.sna;               // (items the rows above have taken: sblocks / 8 x (8 - sna))
        }
        if (bid < pa.ublocks) {
            if (NW == 4 || threadIdx.x < 256) uop_block(bid, D, lp, pa.ua, S.xs, nullptr, 0);
            // merged launch: the post-pass reads the pre-pass' vectors from VT. Wave 0 made the stores (write-through):
            // once they are acknowledged the workgroup counts itself in (uop_wait_prepass)
            if (pa.fwd_done && threadIdx.x < 64) {
#ifndef MPQE_EMU
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                if (threadIdx.x == 0) atomicAdd(pa.fwd_done, 1u);
            }
        } else if (bid < pa.ublocks + pa.tblocks) {
#ifndef MPQE_EMU
            if (pa.late && bid == pa.ublocks)
                for (int q = 0; q < (1 << 18); ++q) __builtin_amdgcn_s_sleep(127);      // (~1 s: 2^18 x 8 128 cycles; uniform)
#endif
            prep_transpose_block(lp, pa, D, bid - pa.ublocks, S.xs);
        }       // (else: padding)
    } else if (po.zpad == 0 || bid < po.zpad) {
        if ((long long)bid < pa.zs.block0[pa.zs.count]) prep_zero_block(pa.zs, bid);
        // (split tail: the relation matrices nothing writes this step are zero-filled HERE, behind the chain workgroups -- the
        // launch has idle CUs from the moment its light batches are through -- not by workgroups of the weight-gradient launch)
        else if (po.zpad == 0 && po.zmblocks > 0 && (long long)bid < pa.zs.block0[pa.zs.count] + po.zmblocks) {
            if (NW == 4 || threadIdx.x < 256) zmat_block(po.zmats, po.zper, bid - (int)pa.zs.block0[pa.zs.count], po.D, po.gp);
        }
    } else {
        // (post roles only on the XCDs picked for them: workgroup b runs on XCD b % 8; the others leave at once)
        const int pb = bid - po.zpad;
        const int rk = (int)((po.xrank >> (4 * (pb & 7))) & 15u) - 1;
        if (rk >= 0 && (NW == 4 || threadIdx.x < 256))
            post_block<(sizeof(S) >= 4 * 64 * GWR_LDT * sizeof(float)) ? 4 : 1>(sd, lp, po, (pb >> 3) * po.na + rk,
                                                                                reinterpret_cast<float *>(&S));
    }
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
    } else if (ta.ux > 0) {
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
            if (!will_merge && hp->nlanes == 1 && dbg_on("CLOSURE") && !dbg_on("FUSE_TAIL")) {
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
    hp->o_closures = take(hp->closures.size() * (size_t)CL_BLOCK_WORDS * 4);
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
        for (size_t k = 0; k < hp->closures.size(); ++k)
            put(hp->o_closures + k * (size_t)CL_BLOCK_WORDS * 4, &hp->closures[k], sizeof(ClBlock));
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
        for (size_t k = 0; k < hp->closures.size(); ++k)
            fprintf(stderr, "  closure %zu batch %d pre %d items %d r1 %d\n", k, hp->closures[k].batch, hp->closures[k].npre,
                    hp->closures[k].nitems, hp->closures[k].nr1);
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

extern "C" int mpqe_step_forward_backward_ex(const mpqe_step_params_t *P, const mpqe_step_batch_t *B, int nb,
                                             const int64_t *anchor_ids, const int64_t *targets, const int64_t *negs,
                                             float margin, const mpqe_step_grads_t *G, int backward,
                                             float *loss, float *scores_pos, float *scores_neg, void *desc,
                                             size_t desc_bytes, int upload_desc, void *workspace,
                                             size_t workspace_bytes, int32_t *err, const mpqe_step_lanes_t *lanes,
                                             void *const *events, int num_events, void *touch, void *stream,
                                             const mpqe_step_extra_t *extra) {
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
                           hp.closures.empty() && dbg_on("POST_IN_CHAIN") && !dbg_on("FUSE_TAIL");
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
    const bool rows_multi = use_touch && D % 4 == 0 && 256 % (D / 4) == 0 && D >= 64 && dbg_on("ROWS_MULTI");
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
    const bool fuse_tail = use_chain && backward && !pic && NL == 1 && D % 64 == 0 && dbg_on("FUSE_TAIL");
    // split tail launch of the chain form: the loss and the entity-table rows depend on the chain launch alone -- they run as
    // trailing workgroups of the weight-gradient launch, beside its tiles (136 of 256 CUs busy on the AIFB step), instead of
    // in the reduction launch behind it (mpqe_debug_option LATE_ROWS = 1: as before)
    const bool can_early = use_chain && backward && !pic && !fuse_tail && NL == 1 && D % 4 == 0 && 256 % (D / 4) == 0;
    // (as trailing workgroups of the weight-gradient launch itself, mpqe_debug_option EARLY_ROWS = 1: measured slower -- that
    // launch's 230 VGPRs allow two workgroups per CU, a table workgroup took 8.6 us and the launch 6 us longer)
    const bool early_roles = can_early && dbg_on("EARLY_ROWS");
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
        const bool closures = use_chain && first == 0 && !hp.closures.empty() && !fuse_tail;
        if (closures) {
            tl.ublocks = 0;
            tl.ca.blocks = reinterpret_cast<const int *>(db + hp.o_closures);
            tl.ca.ncl = (int)hp.closures.size();
            tl.clpad = (tl.ca.ncl + 7) / 8 * 8;
            nblocks = tl.clpad + count + tl.zblocks;
        } else
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
        } else if (use_chain)
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
            ca.wt_blocks = pa.tblocks;
            // (counters and their epoch advance on merged steps only: targets are epoch x count)
            ca.done = pic ? reinterpret_cast<unsigned *>(db + hp.o_done) : nullptr;
            ca.arrive = ca.done ? ca.done + hp.done_inc.size() : nullptr;
            ca.done_inc = reinterpret_cast<const int *>(db + hp.o_done_inc);
            ca.ro = hp.ro_chain ? 1 : 0;
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
            mark(s);
            if (hp.ro_chain) {
                // (a learned readout on the chain: its own instances -- the others' code stays as it was)
                if (D == 64) hipLaunchKernelGGL((step_chain_kernel<1, 1, 4, true>), cgrid, dim3(256), 0, s, sd, lp, tabs, ca, pa, po);
                else if (D == 128 && (P->flags & MPQE_STEP_NO_KSPLIT))
                    hipLaunchKernelGGL((step_chain_kernel<2, 1, 4, true>), cgrid, dim3(256), 0, s, sd, lp, tabs, ca, pa, po);
                else if (D == 128) hipLaunchKernelGGL((step_chain_kernel<4, 2, 4, true>), cgrid, dim3(256), 0, s, sd, lp, tabs, ca, pa, po);
                else hipLaunchKernelGGL((step_chain_kernel<4, 1, 4, true>), cgrid, dim3(256), 0, s, sd, lp, tabs, ca, pa, po);
            } else if (D == 64) hipLaunchKernelGGL((step_chain_kernel<1, 1>), cgrid, dim3(256), 0, s, sd, lp, tabs, ca, pa, po);
            else if (D == 128 && (P->flags & MPQE_STEP_NO_KSPLIT))
                hipLaunchKernelGGL((step_chain_kernel<2, 1>), cgrid, dim3(256), 0, s, sd, lp, tabs, ca, pa, po);
            else if (D == 128 && (P->flags & MPQE_STEP_EIGHT_WAVES))
                hipLaunchKernelGGL((step_chain_kernel<2, 2, 8>), cgrid, dim3(512), 0, s, sd, lp, tabs, ca, pa, po);
            else if (D == 128)
                hipLaunchKernelGGL((step_chain_kernel<4, 2>), cgrid, dim3(256), 0, s, sd, lp, tabs, ca, pa, po);
            else hipLaunchKernelGGL((step_chain_kernel<4, 1>), cgrid, dim3(256), 0, s, sd, lp, tabs, ca, pa, po);
            mark(s);
        }
        if (!backward) {
            for (int l = 1; l < NL; ++l) {
                (void)hipEventRecord(reinterpret_cast<hipEvent_t>(lanes->join_event[l]), ls[l]);
                (void)hipStreamWaitEvent(s, reinterpret_cast<hipEvent_t>(lanes->join_event[l]), 0);
            }
            hipLaunchKernelGGL(step_loss_kernel, dim3(1), dim3(1024), 0, s, sd, (const float *)terms, loss, lm, bterms,
                               use_chain ? epoch_f : (unsigned *)nullptr, pa.tblocks > 0 ? 1 : 0, notify, notify_value,
                               (const int32_t *)err);
            if (learned) ro_regulariser(false);
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
        hipLaunchKernelGGL(step_reduce_kernel, grid, dim3(256), 0, s, ra);
    }
    if (learned) ro_regulariser(true);
    return mpqe_launch_status();
}

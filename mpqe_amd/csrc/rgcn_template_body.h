// Device-side tile bodies of the template R-GCN layer, shared by the per-batch kernels
// (rgcn_template.hip) and the fused whole-step kernels (step.hip). One call = one 64x64
// output tile computed by one 256-thread workgroup; `smem` is GT_SMEM_FLOATS floats of LDS.
#pragma once
#include "gemm_core.h"

struct TmplArgs {
    int N, E;
    int src[3], dst[3];
    long long rel[3];
};

// One K-block of a layer tile = (row slot inside the graph, weight matrix): an incoming template
// edge (source slot, basis[rel]) or the self term (own slot, root). At most 3 edges + self.
struct KBlocks {
    int nk;
    int s0, s1, s2, s3;
    const float *W0, *W1, *W2, *W3;
    __device__ __forceinline__ void add(int sv, const float *Wv) {
        if (nk == 0) { s0 = sv; W0 = Wv; }
        else if (nk == 1) { s1 = sv; W1 = Wv; }
        else if (nk == 2) { s2 = sv; W2 = Wv; }
        else { s3 = sv; W3 = Wv; }
        ++nk;
    }
    // The current block is always entry 0; moving to the next block shifts the table down. Plain
    // selects on wave-uniform values (s_cselect) -- indexing the table by a runtime block number
    // compiles to a nest of scalar branches in the middle of the MFMA loop.
    __device__ __forceinline__ void shift_if(bool wrap) {
        s0 = wrap ? s1 : s0;
        s1 = wrap ? s2 : s1;
        s2 = wrap ? s3 : s2;
        W0 = wrap ? W1 : W0;
        W1 = wrap ? W2 : W1;
        W2 = wrap ? W3 : W2;
    }
};

// Loader of the layer forward / backward-x tiles.
//   A (R-type): rows of `a` ([B*N, K] row-major): tile row r -> graph b0 + r, node slot of the current
//               K-block; contiguous along k.
//   B: TRANS = false (forward)    W[k][n]:  K-type image, rows k, 64 columns n0..
//      TRANS = true  (backward-x) W[n][k]:  R-type image, rows n0.. (output columns), contiguous in k
// The K-blocks are walked as ONE pipelined K loop: pointers are bumped per step and rebuilt only
// when the block changes.
template <int MODE, bool TRANS>
struct LayerLoader {
    KBlocks kb;
    const float *abase, *mask, *wsafe;
    int K, ldw, wrows;          // A row length; W leading dim (= Dout); rows of W the B tile may touch
    int spb, ls, left;
    int ca;                     // column of this thread's A slots in the current step: ls*32 + ac
    int bcol0;                  // K-type: n0 + bc (fixed); R-type: bc (column inside the step)
    bool rok0, rok1;            // A rows inside the batch
    long long arow0, arow1;     // (clamped graph) * N * K : offset of the graph's first node row
    int brow0, brow1;           // B image rows of this thread's slots (K-type: k; R-type: n0 + r)
    const float *pa0, *pa1, *pm0, *pm1, *pb0, *pb1;

    __device__ __forceinline__ void set_block() {
        const long long so = (long long)kb.s0 * K;
        pa0 = abase + arow0 + so;
        pa1 = abase + arow1 + so;
        if (mask) {
            pm0 = mask + arow0 + so;
            pm1 = mask + arow1 + so;
        }
        const float *W = kb.W0;
        pb0 = W + (long long)brow0 * ldw;       // forward: row k = brow of step 0; backward: row n = brow
        pb1 = W + (long long)brow1 * ldw;
    }

    __device__ __forceinline__ void init(const KBlocks &blocks, const float *a_, const float *mask_,
                                         const float *wsafe_, long long B, int N, int K_, int ldw_, int wrows_,
                                         long long b0, int n0) {
        kb = blocks;
        abase = a_;
        mask = mask_;
        wsafe = wsafe_;
        K = K_;
        ldw = ldw_;
        wrows = wrows_;
        spb = (K + GT_BK - 1) / GT_BK;
        ls = 0;
        left = kb.nk * spb;
        ca = stage_col(false);
        pm0 = pm1 = nullptr;
        const long long g0 = b0 + stage_row(false, 0), g1 = b0 + stage_row(false, 1);
        rok0 = g0 < B;
        rok1 = g1 < B;
        // out-of-range rows: LD_FAST reads a clamped (valid) row whose result is discarded,
        // the other modes never dereference the pointer
        arow0 = (rok0 ? g0 : B - 1) * (long long)N * K;
        arow1 = (rok1 ? g1 : B - 1) * (long long)N * K;
        if (!TRANS) {
            brow0 = stage_row(true, 0);
            brow1 = stage_row(true, 1);
            bcol0 = n0 + stage_col(true);
        } else {
            brow0 = n0 + stage_row(false, 0);
            brow1 = n0 + stage_row(false, 1);
            bcol0 = stage_col(false);
        }
        set_block();
    }

    __device__ __forceinline__ f32x4 a(int slot, bool &ok) {
        f32x4 v = ld4_pred<MODE>(abase, slot ? pa1 : pa0, ca, K, slot ? rok1 : rok0, ok);
        if (mask) {     // per-op backward only; the fused step masks in the producer's epilogue
            bool ok2;
            f32x4 o = ld4_pred<MODE>(mask, slot ? pm1 : pm0, ca, K, slot ? rok1 : rok0, ok2);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = o[q] > 0.f ? v[q] : 0.f;
        }
        return v;
    }

    __device__ __forceinline__ f32x4 b(int slot, bool &ok) {
        if (!TRANS) {
            const int kk = ls * GT_BK + (slot ? brow1 : brow0);
            return ld4_pred<MODE>(wsafe, slot ? pb1 : pb0, bcol0, ldw, kk < wrows, ok);
        }
        const int nn = slot ? brow1 : brow0;
        return ld4_pred<MODE>(wsafe, slot ? pb1 : pb0, ls * GT_BK + bcol0, ldw, nn < wrows, ok);
    }

    // Written with selects only (the conditions are wave-uniform, so they become s_cselect): a
    // taken scalar branch per condition per step would cost about as much issue time as the MFMAs
    // it sits between.
    __device__ __forceinline__ void next() {
        const bool go = left > 1;            // freeze on the last step (surplus pipeline loads)
        left -= go ? 1 : 0;
        const bool wrap = go && (ls + 1 == spb);
        ls = wrap ? 0 : (go ? ls + 1 : ls);
        kb.shift_if(wrap);
        ca = wrap ? stage_col(false) : (go ? ca + GT_BK : ca);
        const long long so = (long long)kb.s0 * K;
        const float *W = kb.W0;
        const float *nb0 = W + (long long)brow0 * ldw, *nb1 = W + (long long)brow1 * ldw;
        const long long bump = (!TRANS && go) ? (long long)GT_BK * ldw : 0;
        pa0 = wrap ? abase + arow0 + so : pa0;
        pa1 = wrap ? abase + arow1 + so : pa1;
        if (mask) {
            pm0 = wrap ? mask + arow0 + so : pm0;
            pm1 = wrap ? mask + arow1 + so : pm1;
        }
        pb0 = wrap ? nb0 : pb0 + bump;
        pb1 = wrap ? nb1 : pb1 + bump;
    }
};

__device__ __forceinline__ void kblocks_init(KBlocks &kb, const float *root) {
    kb.nk = 0;
    kb.s0 = kb.s1 = kb.s2 = kb.s3 = 0;
    kb.W0 = kb.W1 = kb.W2 = kb.W3 = root;
}

// out[b0.., n, n0..] = [relu]( sum_{e: dst_e = n} x[:, src_e, :] . basis[rel_e] + x[:, n, :] . root + bias )
template <int MODE>
__device__ __forceinline__ void tmpl_fwd_tile(const TmplArgs &tp, long long B, const float *__restrict__ x,
                                              const float *__restrict__ basis, const float *__restrict__ root,
                                              const float *__restrict__ bias, int Din, int Dout, int relu,
                                              float *__restrict__ out, int n, long long b0,
                                              int n0, float *smem) {
    // K-blocks of this node slot: one per incoming template edge, then the self/root block.
    // Constant indices only: a runtime-indexed tp.src[e] would put the template in scratch and make
    // everything derived from it (block count, loop bounds) look divergent to the compiler.
    KBlocks kb;
    kblocks_init(kb, root);
    if (tp.E > 0 && tp.dst[0] == n) kb.add(tp.src[0], basis + tp.rel[0] * (long long)Din * Dout);
    if (tp.E > 1 && tp.dst[1] == n) kb.add(tp.src[1], basis + tp.rel[1] * (long long)Din * Dout);
    if (tp.E > 2 && tp.dst[2] == n) kb.add(tp.src[2], basis + tp.rel[2] * (long long)Din * Dout);
    kb.add(n, root);
    kb.nk = __builtin_amdgcn_readfirstlane(kb.nk);     // wave-uniform by construction: keep the loop scalar
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    LayerLoader<MODE, false> L;
    L.init(kb, x, nullptr, root, B, tp.N, Din, Dout, Din, b0, n0);
    gemm_block<false, true>(acc, L, kb.nk * L.spb, smem);
    const int col = n0 + acc_col();
    if (col < Dout) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long b = b0 + acc_row(r);
            if (b < B) {
                float v = acc[r] + bv;
                if (relu) v = v > 0.f ? v : 0.f;
                out[(b * tp.N + n) * (long long)Dout + col] = v;
            }
        }
    }
}

// grad_x[:, m, n0..] = sum_{e: src_e = m} gpre[:, dst_e, :] . basis[rel_e]^T + gpre[:, m, :] . root^T
// gpre = g * (out > 0) when the layer applied ReLU (out = the layer's post-ReLU output) and relu != 0;
// with relu = 0 the caller passes g already masked. mask_x (fused step): the ReLU output this gradient
// belongs to -- masking on the way OUT means the next consumer reads a ready pre-activation gradient
// and needs no mask loads of its own.
template <int MODE>
__device__ __forceinline__ void tmpl_bwd_x_tile(const TmplArgs &tp, long long B, const float *__restrict__ g,
                                                const float *__restrict__ out, const float *__restrict__ basis,
                                                const float *__restrict__ root, int Din, int Dout, int relu,
                                                float *__restrict__ grad_x, int m,
                                                long long b0, int n0, float *smem,
                                                const float *__restrict__ mask_x = nullptr,
                                                unsigned live_out = 0xFu, bool add_in = false) {
    // add_in (fused step, caller's readout over every level): grad_x already holds a gradient of the same rows that
    // reaches them by another route; the propagated one is added to it (before the mask).
    // live_out (fused step): node slots whose output gradient can be non-zero; the K-blocks of the
    // others would multiply exact zeros (rows the step never writes) and are left out. The caller
    // guarantees that at least one block remains.
    KBlocks kb;
    kblocks_init(kb, root);
    if (tp.E > 0 && tp.src[0] == m && ((live_out >> tp.dst[0]) & 1u))
        kb.add(tp.dst[0], basis + tp.rel[0] * (long long)Din * Dout);
    if (tp.E > 1 && tp.src[1] == m && ((live_out >> tp.dst[1]) & 1u))
        kb.add(tp.dst[1], basis + tp.rel[1] * (long long)Din * Dout);
    if (tp.E > 2 && tp.src[2] == m && ((live_out >> tp.dst[2]) & 1u))
        kb.add(tp.dst[2], basis + tp.rel[2] * (long long)Din * Dout);
    if ((live_out >> m) & 1u) kb.add(m, root);
    kb.nk = __builtin_amdgcn_readfirstlane(kb.nk);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    LayerLoader<MODE, true> L;      // K runs over Dout; B[k][n] = W[n][k], tile rows = output columns over Din
    L.init(kb, g, relu ? out : nullptr, root, B, tp.N, Dout, Dout, Din, b0, n0);
    gemm_block<false, false>(acc, L, kb.nk * L.spb, smem);
    const int col = n0 + acc_col();
    if (col < Din) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long b = b0 + acc_row(r);
            if (b < B) {
                const long long idx = (b * tp.N + m) * (long long)Din + col;
                const float v = add_in ? acc[r] + grad_x[idx] : acc[r];
                grad_x[idx] = (mask_x && !(mask_x[idx] > 0.f)) ? 0.f : v;
            }
        }
    }
}

// Loader of the weight-gradient tile: both operands K-type, the K dimension runs over rows q:
//   A[k][i] = x[(q*xs + xo)][i0 + i],  B[k][j] = gpre[(q*gs + go)][j0 + j],  q = q0 + step*32 + k
template <int MODE>
struct GradWLoader {
    const float *x, *g, *mask;
    int Din, Dout, left;
    long long q, q1;                 // row of slot 0 in the current step; end of the chunk
    int ca, cb;                      // columns i0 + cc, j0 + cc
    long long stepx, stepg;          // pointer bump per K-step (32 rows)
    long long s16x, s16g;            // slot 1 = slot 0 + 16 rows
    const float *px, *pg, *pm;

    __device__ __forceinline__ void init(const float *x_, const float *g_, const float *mask_, int Din_, int Dout_,
                                         long long xs, long long xo, long long gs, long long go, long long q0,
                                         long long q1_, int i0, int j0, int nsteps) {
        x = x_;
        g = g_;
        mask = mask_;
        Din = Din_;
        Dout = Dout_;
        left = nsteps;
        q = q0 + stage_row(true, 0);
        q1 = q1_;
        ca = i0 + stage_col(true);
        cb = j0 + stage_col(true);
        stepx = (long long)GT_BK * xs * Din;
        stepg = (long long)GT_BK * gs * Dout;
        s16x = 16 * xs * (long long)Din;
        s16g = 16 * gs * (long long)Dout;
        px = x + (q * xs + xo) * (long long)Din;
        pg = g + (q * gs + go) * (long long)Dout;
        pm = mask ? mask + (q * gs + go) * (long long)Dout : nullptr;
    }
    __device__ __forceinline__ f32x4 a(int slot, bool &ok) {
        return ld4_pred<MODE>(x, slot ? px + s16x : px, ca, Din, q + 16 * slot < q1, ok);
    }
    __device__ __forceinline__ f32x4 b(int slot, bool &ok) {
        f32x4 v = ld4_pred<MODE>(g, slot ? pg + s16g : pg, cb, Dout, q + 16 * slot < q1, ok);
        if (mask) {
            bool ok2;
            f32x4 o = ld4_pred<MODE>(mask, slot ? pm + s16g : pm, cb, Dout, q + 16 * slot < q1, ok2);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) v[qq] = o[qq] > 0.f ? v[qq] : 0.f;
        }
        return v;
    }
    __device__ __forceinline__ void next() {
        const bool go = left > 1;    // freeze on the last step: LD_FAST never runs past the tensors
        left -= go ? 1 : 0;
        q += go ? GT_BK : 0;
        px += go ? stepx : 0;
        pg += go ? stepg : 0;
        if (mask) pm += go ? stepg : 0;
    }
};

// slab[i0.., j0..] = sum_{q in [q0, q1)} x[q*xs + xo]^T (x) gpre[q*gs + go]
// (edge slot: xs = gs = N, xo = src, go = dst, q over graphs; root: xs = gs = 1, q over all rows).
// LD_FAST requires q1 - q0 to be a multiple of the K-step (no in-range step needs zero filling).
template <int MODE>
__device__ __forceinline__ void tmpl_grad_w_tile(const float *__restrict__ x, const float *__restrict__ g,
                                                 const float *__restrict__ out, int Din, int Dout, int relu,
                                                 long long xs, long long xo, long long gs, long long go,
                                                 long long q0, long long q1, int i0, int j0,
                                                 float *__restrict__ slab, float *smem, bool accumulate = false) {
    const int nsteps = q1 > q0 ? (int)((q1 - q0 + GT_BK - 1) / GT_BK) : 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (nsteps > 0) {
        GradWLoader<MODE> L;
        L.init(x, g, relu ? out : nullptr, Din, Dout, xs, xo, gs, go, q0, q1, i0, j0, nsteps);
        gemm_block<true, true>(acc, L, nsteps, smem);
    }
    const int col = j0 + acc_col();
    if (col < Dout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + acc_row(r);
            if (row < Din) {
                float *o = slab + (long long)row * Dout + col;
                *o = accumulate ? *o + acc[r] : acc[r];
            }
        }
    }
}

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

// out[b0.., n, n0..] = [relu]( sum_{e: dst_e = n} x[:, src_e, :] . basis[rel_e] + x[:, n, :] . root + bias )
template <int MODE>
__device__ __forceinline__ void tmpl_fwd_tile(const TmplArgs &tp, long long B, const float *__restrict__ x,
                                              const float *__restrict__ basis, const float *__restrict__ root,
                                              const float *__restrict__ bias, int Din, int Dout, int relu,
                                              float *__restrict__ out, int n, long long b0,
                                              int n0, float *smem) {
    // K-blocks of this node slot: one per incoming template edge, then the self/root block.
    // They are walked as ONE pipelined K loop (block kb = step / spb) so the prefetch never drains
    // between blocks; the (source slot, weight) of a block is picked with selects, not an array,
    // to stay in registers.
    const int spb = (Din + GT_BK - 1) / GT_BK;     // steps per block
    int nk = 0;
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const float *W0 = root, *W1 = root, *W2 = root, *W3 = root;
    auto add_block = [&](int sv, const float *Wv) {
        if (nk == 0) { s0 = sv; W0 = Wv; }
        else if (nk == 1) { s1 = sv; W1 = Wv; }
        else if (nk == 2) { s2 = sv; W2 = Wv; }
        else { s3 = sv; W3 = Wv; }
        ++nk;
    };
    // constant indices only: a runtime-indexed tp.src[e] would put the template in scratch and make
    // everything derived from it (block count, loop bounds) look divergent to the compiler
    if (tp.E > 0 && tp.dst[0] == n) add_block(tp.src[0], basis + tp.rel[0] * (long long)Din * Dout);
    if (tp.E > 1 && tp.dst[1] == n) add_block(tp.src[1], basis + tp.rel[1] * (long long)Din * Dout);
    if (tp.E > 2 && tp.dst[2] == n) add_block(tp.src[2], basis + tp.rel[2] * (long long)Din * Dout);
    add_block(n, root);
    nk = __builtin_amdgcn_readfirstlane(nk);       // wave-uniform by construction: keep the loop scalar
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // loader state = (K-block, step inside it); only a block change touches the select chain
    int kb = 0, ls = 0, sv = s0;
    const float *Wv = W0;
    auto aload = [&](int r, int c, bool &ok) -> f32x4 {
        const long long b = b0 + r;
        const long long bv = (MODE == LD_FAST && b >= B) ? B - 1 : b;   // clamped row: result discarded
        const float *p = x + (bv * tp.N + sv) * (long long)Din;
        return ld4_pred<MODE>(x, p, ls * GT_BK + c, Din, b < B, ok);
    };
    auto bload = [&](int k, int c, bool &ok) -> f32x4 {
        const int kk = ls * GT_BK + k;
        return ld4_pred<MODE>(root, Wv + (long long)kk * Dout, n0 + c, Dout, kk < Din, ok);
    };
    auto advance = [&]() {
        if (++ls == spb) {
            ls = 0;
            ++kb;
            sv = kb == 1 ? s1 : (kb == 2 ? s2 : s3);
            Wv = kb == 1 ? W1 : (kb == 2 ? W2 : W3);
        }
    };
    gemm_block<false, true>(acc, aload, bload, advance, nk * spb, smem);
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
// with relu = 0 the caller passes g already masked.
template <int MODE>
__device__ __forceinline__ void tmpl_bwd_x_tile(const TmplArgs &tp, long long B, const float *__restrict__ g,
                                                const float *__restrict__ out, const float *__restrict__ basis,
                                                const float *__restrict__ root, int Din, int Dout, int relu,
                                                float *__restrict__ grad_x, int m,
                                                long long b0, int n0, float *smem,
                                                const float *__restrict__ mask_x = nullptr) {
    const int spb = (Dout + GT_BK - 1) / GT_BK;   // K runs over Dout; steps per block
    int nk = 0;
    int d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    const float *W0 = root, *W1 = root, *W2 = root, *W3 = root;
    auto add_block = [&](int dv, const float *Wv) {
        if (nk == 0) { d0 = dv; W0 = Wv; }
        else if (nk == 1) { d1 = dv; W1 = Wv; }
        else if (nk == 2) { d2 = dv; W2 = Wv; }
        else { d3 = dv; W3 = Wv; }
        ++nk;
    };
    if (tp.E > 0 && tp.src[0] == m) add_block(tp.dst[0], basis + tp.rel[0] * (long long)Din * Dout);
    if (tp.E > 1 && tp.src[1] == m) add_block(tp.dst[1], basis + tp.rel[1] * (long long)Din * Dout);
    if (tp.E > 2 && tp.src[2] == m) add_block(tp.dst[2], basis + tp.rel[2] * (long long)Din * Dout);
    add_block(m, root);
    nk = __builtin_amdgcn_readfirstlane(nk);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int kb = 0, ls = 0, dv = d0;
    const float *Wv = W0;
    auto aload = [&](int r, int c, bool &ok) -> f32x4 {
        const long long b = b0 + r;
        const long long bv = (MODE == LD_FAST && b >= B) ? B - 1 : b;
        const long long off = (bv * tp.N + dv) * (long long)Dout;
        f32x4 v = ld4_pred<MODE>(g, g + off, ls * GT_BK + c, Dout, b < B, ok);
        if (relu) {    // per-op path only; the fused step masks in the producer's epilogue instead
            bool ok2;
            f32x4 o = ld4_pred<MODE>(out, out + off, ls * GT_BK + c, Dout, b < B, ok2);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = o[q] > 0.f ? v[q] : 0.f;
        }
        return v;
    };
    // B[k][n] = W[n][k]: R-type image, tile row = output column n (over Din), contiguous in k
    auto bload = [&](int r, int c, bool &ok) -> f32x4 {
        const int nn = n0 + r;
        return ld4_pred<MODE>(root, Wv + (long long)nn * Dout, ls * GT_BK + c, Dout, nn < Din, ok);
    };
    auto advance = [&]() {
        if (++ls == spb) {
            ls = 0;
            ++kb;
            dv = kb == 1 ? d1 : (kb == 2 ? d2 : d3);
            Wv = kb == 1 ? W1 : (kb == 2 ? W2 : W3);
        }
    };
    gemm_block<false, false>(acc, aload, bload, advance, nk * spb, smem);
    const int col = n0 + acc_col();
    if (col < Din) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long b = b0 + acc_row(r);
            if (b < B) {
                const long long idx = (b * tp.N + m) * (long long)Din + col;
                // mask_x: the ReLU output this gradient belongs to (fused step: the NEXT consumer then
                // reads a ready pre-activation gradient and needs no mask loads of its own)
                grad_x[idx] = (mask_x && !(mask_x[idx] > 0.f)) ? 0.f : acc[r];
            }
        }
    }
}

// slab[i0.., j0..] = sum_{q in [q0, q1)} x[q*xs + xo]^T (x) gpre[q*gs + go]
// (edge slot: xs = gs = N, xo = src, go = dst, q over graphs; root: xs = gs = 1, q over all rows)
// qmax = last valid q of the tensors (LD_FAST clamps the pipeline's surplus tail loads to it; in that
// mode q1 - q0 must be a multiple of the K-step so that no in-range step needs zero filling).
template <int MODE>
__device__ __forceinline__ void tmpl_grad_w_tile(const float *__restrict__ x, const float *__restrict__ g,
                                                 const float *__restrict__ out, int Din, int Dout, int relu,
                                                 long long xs, long long xo, long long gs, long long go,
                                                 long long q0, long long q1, long long qmax, int i0, int j0,
                                                 float *__restrict__ slab, float *smem) {
    const int nsteps = q1 > q0 ? (int)((q1 - q0 + GT_BK - 1) / GT_BK) : 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    long long qs = q0;      // first row of the loader's current K-step
    auto aload = [&](int k, int cc, bool &ok) -> f32x4 {
        const long long q = qs + k;
        const long long qv = (MODE == LD_FAST && q > qmax) ? qmax : q;
        return ld4_pred<MODE>(x, x + (qv * xs + xo) * (long long)Din, i0 + cc, Din, q < q1, ok);
    };
    auto bload = [&](int k, int cc, bool &ok) -> f32x4 {
        const long long q = qs + k;
        const long long qv = (MODE == LD_FAST && q > qmax) ? qmax : q;
        const long long off = (qv * gs + go) * (long long)Dout;
        f32x4 v = ld4_pred<MODE>(g, g + off, j0 + cc, Dout, q < q1, ok);
        if (relu) {
            bool ok2;
            f32x4 o = ld4_pred<MODE>(out, out + off, j0 + cc, Dout, q < q1, ok2);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) v[qq] = o[qq] > 0.f ? v[qq] : 0.f;
        }
        return v;
    };
    auto advance = [&]() { qs += GT_BK; };
    gemm_block<true, true>(acc, aload, bload, advance, nsteps, smem);
    const int col = j0 + acc_col();
    if (col < Dout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + acc_row(r);
            if (row < Din) slab[(long long)row * Dout + col] = acc[r];
        }
    }
}

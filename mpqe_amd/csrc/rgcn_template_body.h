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
__device__ __forceinline__ void tmpl_fwd_tile(const TmplArgs &tp, long long B, const float *__restrict__ x,
                                              const float *__restrict__ basis, const float *__restrict__ root,
                                              const float *__restrict__ bias, int Din, int Dout, int relu,
                                              float *__restrict__ out, int vec_x, int vec_w, int n, long long b0,
                                              int n0, float *smem) {
    // K-blocks of this node slot: one per incoming template edge, then the self/root block.
    // They are walked as ONE pipelined K loop (block kb = step / spb) so the prefetch never drains
    // between blocks; the (source slot, weight) of a block is picked with selects, not an array,
    // to stay in registers.
    const int spb = (Din + GT_BK - 1) / GT_BK;     // steps per block
    int nk = 0;
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const float *W0 = root, *W1 = root, *W2 = root, *W3 = root;
    for (int e = 0; e <= tp.E; ++e) {
        int sv;
        const float *Wv;
        if (e < tp.E) {
            if (tp.dst[e] != n) continue;
            sv = tp.src[e];
            Wv = basis + tp.rel[e] * (long long)Din * Dout;
        } else {
            sv = n;
            Wv = root;
        }
        if (nk == 0) { s0 = sv; W0 = Wv; }
        else if (nk == 1) { s1 = sv; W1 = Wv; }
        else if (nk == 2) { s2 = sv; W2 = Wv; }
        else { s3 = sv; W3 = Wv; }
        ++nk;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto aload = [&](int r, int c, int step) -> f32x4 {
        const long long b = b0 + r;
        if (b >= B) return f32x4{0.f, 0.f, 0.f, 0.f};
        const int kb = step / spb, ls = step - kb * spb;
        const int sv = kb == 0 ? s0 : (kb == 1 ? s1 : (kb == 2 ? s2 : s3));
        const float *p = x + (b * tp.N + sv) * (long long)Din;
        return ld4_guard(p, ls * GT_BK + c, Din, vec_x);
    };
    auto bload = [&](int k, int c, int step) -> f32x4 {
        const int kb = step / spb, ls = step - kb * spb;
        const int kk = ls * GT_BK + k;
        if (kk >= Din) return f32x4{0.f, 0.f, 0.f, 0.f};
        const float *Wv = kb == 0 ? W0 : (kb == 1 ? W1 : (kb == 2 ? W2 : W3));
        return ld4_guard(Wv + (long long)kk * Dout, n0 + c, Dout, vec_w);
    };
    gemm_block<false, true>(acc, aload, bload, nk * spb, smem);
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
// gpre = g * (out > 0) when the layer applied ReLU (out = the layer's post-ReLU output).
__device__ __forceinline__ void tmpl_bwd_x_tile(const TmplArgs &tp, long long B, const float *__restrict__ g,
                                                const float *__restrict__ out, const float *__restrict__ basis,
                                                const float *__restrict__ root, int Din, int Dout, int relu,
                                                float *__restrict__ grad_x, int vec_g, int vec_w, int m,
                                                long long b0, int n0, float *smem) {
    const int spb = (Dout + GT_BK - 1) / GT_BK;   // K runs over Dout; steps per block
    int nk = 0;
    int d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    const float *W0 = root, *W1 = root, *W2 = root, *W3 = root;
    for (int e = 0; e <= tp.E; ++e) {
        int dv;
        const float *Wv;
        if (e < tp.E) {
            if (tp.src[e] != m) continue;
            dv = tp.dst[e];
            Wv = basis + tp.rel[e] * (long long)Din * Dout;
        } else {
            dv = m;
            Wv = root;
        }
        if (nk == 0) { d0 = dv; W0 = Wv; }
        else if (nk == 1) { d1 = dv; W1 = Wv; }
        else if (nk == 2) { d2 = dv; W2 = Wv; }
        else { d3 = dv; W3 = Wv; }
        ++nk;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto aload = [&](int r, int c, int step) -> f32x4 {
        const long long b = b0 + r;
        if (b >= B) return f32x4{0.f, 0.f, 0.f, 0.f};
        const int kb = step / spb, ls = step - kb * spb;
        const int dv = kb == 0 ? d0 : (kb == 1 ? d1 : (kb == 2 ? d2 : d3));
        const long long off = (b * tp.N + dv) * (long long)Dout;
        f32x4 v = ld4_guard(g + off, ls * GT_BK + c, Dout, vec_g);
        if (relu) {
            f32x4 o = ld4_guard(out + off, ls * GT_BK + c, Dout, vec_g);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = o[q] > 0.f ? v[q] : 0.f;
        }
        return v;
    };
    // B[k][n] = W[n][k]: R-type image, tile row = output column n (over Din), contiguous in k
    auto bload = [&](int r, int c, int step) -> f32x4 {
        const int nn = n0 + r;
        if (nn >= Din) return f32x4{0.f, 0.f, 0.f, 0.f};
        const int kb = step / spb, ls = step - kb * spb;
        const float *Wv = kb == 0 ? W0 : (kb == 1 ? W1 : (kb == 2 ? W2 : W3));
        return ld4_guard(Wv + (long long)nn * Dout, ls * GT_BK + c, Dout, vec_w);
    };
    gemm_block<false, false>(acc, aload, bload, nk * spb, smem);
    const int col = n0 + acc_col();
    if (col < Din) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long b = b0 + acc_row(r);
            if (b < B) grad_x[(b * tp.N + m) * (long long)Din + col] = acc[r];
        }
    }
}

// slab[i0.., j0..] = sum_{q in [q0, q1)} x[q*xs + xo]^T (x) gpre[q*gs + go]
// (edge slot: xs = gs = N, xo = src, go = dst, q over graphs; root: xs = gs = 1, q over all rows)
__device__ __forceinline__ void tmpl_grad_w_tile(const float *__restrict__ x, const float *__restrict__ g,
                                                 const float *__restrict__ out, int Din, int Dout, int relu,
                                                 long long xs, long long xo, long long gs, long long go,
                                                 long long q0, long long q1, int i0, int j0,
                                                 float *__restrict__ slab, int vec_x, int vec_g, float *smem) {
    const int nsteps = q1 > q0 ? (int)((q1 - q0 + GT_BK - 1) / GT_BK) : 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto aload = [&](int k, int cc, int step) -> f32x4 {
        const long long q = q0 + (long long)step * GT_BK + k;
        if (q >= q1) return f32x4{0.f, 0.f, 0.f, 0.f};
        return ld4_guard(x + (q * xs + xo) * (long long)Din, i0 + cc, Din, vec_x);
    };
    auto bload = [&](int k, int cc, int step) -> f32x4 {
        const long long q = q0 + (long long)step * GT_BK + k;
        if (q >= q1) return f32x4{0.f, 0.f, 0.f, 0.f};
        const long long off = (q * gs + go) * (long long)Dout;
        f32x4 v = ld4_guard(g + off, j0 + cc, Dout, vec_g);
        if (relu) {
            f32x4 o = ld4_guard(out + off, j0 + cc, Dout, vec_g);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) v[qq] = o[qq] > 0.f ? v[qq] : 0.f;
        }
        return v;
    };
    gemm_block<true, true>(acc, aload, bload, nsteps, smem);
    const int col = j0 + acc_col();
    if (col < Dout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + acc_row(r);
            if (row < Din) slab[(long long)row * Dout + col] = acc[r];
        }
    }
}

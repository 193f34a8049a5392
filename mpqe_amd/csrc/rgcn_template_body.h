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
    const int nsteps = (Din + GT_BK - 1) / GT_BK;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kb = 0; kb <= tp.E; ++kb) {
        int s;
        const float *W;
        if (kb < tp.E) {
            if (tp.dst[kb] != n) continue;
            s = tp.src[kb];
            W = basis + tp.rel[kb] * (long long)Din * Dout;
        } else {
            s = n;
            W = root;
        }
        auto aload = [&](int r, int c, int step) -> f32x4 {
            const long long b = b0 + r;
            if (b >= B) return f32x4{0.f, 0.f, 0.f, 0.f};
            const float *p = x + (b * tp.N + s) * (long long)Din;
            return ld4_guard(p, step * GT_BK + c, Din, vec_x);
        };
        auto bload = [&](int k, int c, int step) -> f32x4 {
            const int kk = step * GT_BK + k;
            if (kk >= Din) return f32x4{0.f, 0.f, 0.f, 0.f};
            return ld4_guard(W + (long long)kk * Dout, n0 + c, Dout, vec_w);
        };
        gemm_block<false, true>(acc, aload, bload, nsteps, smem);
    }
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
    const int nsteps = (Dout + GT_BK - 1) / GT_BK;   // K runs over Dout
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kb = 0; kb <= tp.E; ++kb) {
        int d;
        const float *W;
        if (kb < tp.E) {
            if (tp.src[kb] != m) continue;
            d = tp.dst[kb];
            W = basis + tp.rel[kb] * (long long)Din * Dout;
        } else {
            d = m;
            W = root;
        }
        auto aload = [&](int r, int c, int step) -> f32x4 {
            const long long b = b0 + r;
            if (b >= B) return f32x4{0.f, 0.f, 0.f, 0.f};
            const long long off = (b * tp.N + d) * (long long)Dout;
            f32x4 v = ld4_guard(g + off, step * GT_BK + c, Dout, vec_g);
            if (relu) {
                f32x4 o = ld4_guard(out + off, step * GT_BK + c, Dout, vec_g);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = o[q] > 0.f ? v[q] : 0.f;
            }
            return v;
        };
        // B[k][n] = W[n][k]: R-type image, tile row = output column n (over Din), contiguous in k
        auto bload = [&](int r, int c, int step) -> f32x4 {
            const int nn = n0 + r;
            if (nn >= Din) return f32x4{0.f, 0.f, 0.f, 0.f};
            return ld4_guard(W + (long long)nn * Dout, step * GT_BK + c, Dout, vec_w);
        };
        gemm_block<false, false>(acc, aload, bload, nsteps, smem);
    }
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

// Graph-block chain kernels of the fused step.
//
// Query graphs never interact (reference data_utils.py:405: the batch is block-diagonal), so the
// whole message-passing chain of a block of graphs -- every level forward, or every level backward --
// depends on nothing outside the block. One workgroup therefore takes CH_GB = 16 graphs of one batch
// through ALL its levels: node states stay in LDS between levels (ping-pong), the only barrier is the
// workgroup's own, and nothing waits for the slowest tile of a level as the one-launch-per-level form
// does. HBM sees each state once, on the way out (the weight-gradient kernel and the ReLU masks of the
// backward chain read it from there).
//
// GEMM shape per (node slot, level): [16 graphs] x [K = (in-edges + 1) * D] x [D columns]. A wave owns
// 16 * NCB columns (NCB column blocks, column = n0 + NCB * j + c, so a lane's NCB columns are adjacent
// in memory), D = 64 * NCB. v_mfma_f32_16x16x4_f32: lane l feeds A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; the 4 floats a lane reads from LDS with one ds_read_b128 are the A values of
// four MFMAs (u = 0..3), MFMA u multiplying k = 16 t + 4 (l >> 4) + u -- the sum over k is re-ordered,
// in a fixed order. The weights never pass through LDS: every W element is used by exactly one wave,
// which loads its slice straight into registers one half-block (64 k) ahead of the MFMAs that use it.
// Included by step.hip after StepDev / BatchDev / LayerPtrs / pick_layer.
#pragma once
#include "gemm_core.h"

#ifdef MPQE_EMU
#define CHAIN_PIN(x) (void)(x)
#else
#define CHAIN_PIN(x) asm volatile("" : "+v"(x))
#endif
#define CH_GB 16
#define CH_FIRST 1
#define CH_LAST 2

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const f32x2 __attribute__((address_space(1))) * gvec2_ptr;
__device__ __forceinline__ f32x2 gload2(const float *p) { return *(gvec2_ptr)(p); }

// one K-block of a node update: multiply the LDS rows of node slot `src` by matrix `mat`
// (relation id, or -1 = the layer's root) into the accumulator of node slot `node`
struct ChainOp {
    int src, node, mat, flags;
};
struct ChainRef {
    int batch, g0;
};
// ops of batch b: forward level p = [fwd_off[p], fwd_off[p+1]), backward level p = [bwd_off[p], bwd_off[p+1])
struct ChainBatch {
    int fwd_off[MPQE_STEP_MAX_LAYERS + 1], bwd_off[MPQE_STEP_MAX_LAYERS + 1];
};

template <int NCB>
struct WHalf {
    float v[4][4][NCB];      // [t][u][c]: k = 16 t + 4 kq + u of the half-block, column block c
};

// forward: B[k][n] = W[k][n]; wp = W + (64 h + 4 kq) * D + n0 + NCB * j
template <int NCB>
__device__ __forceinline__ void chain_load_w(WHalf<NCB> &f, const float *wp, int D) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *p = wp + (long long)(16 * t + u) * D;
            if constexpr (NCB == 1) f.v[t][u][0] = gload1(p);
            else if constexpr (NCB == 2) {
                const f32x2 q = gload2(p);
                f.v[t][u][0] = q[0];
                f.v[t][u][1] = q[1];
            } else {
                const f32x4 q = gload4(p);
#pragma unroll
                for (int c = 0; c < NCB; ++c) f.v[t][u][c] = q[c & 3];
            }
        }
}
// backward-x: B[k][n] = W[n][k]; wp = W + (n0 + NCB * j) * D + 64 h + 4 kq
template <int NCB>
__device__ __forceinline__ void chain_load_wt(WHalf<NCB> &f, const float *wp, int D) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const f32x4 q = gload4(wp + (long long)c * D + 16 * t);
#pragma unroll
            for (int u = 0; u < 4; ++u) f.v[t][u][c] = q[u];
        }
}

template <int NCB>
__device__ __forceinline__ void chain_mma(f32x4 (&acc)[NCB], const WHalf<NCB> &f, const float *xp) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(xp + 16 * t);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NCB; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], f.v[t][u][c], acc[c], 0, 0, 0);
    }
}

// BWD = false: X = H. Reads H[0] of the block, writes H[1 .. L].
// BWD = true:  X = gH (pre-activation gradients). Reads gH[L] (written by the score kernel), writes
//              gH[L-1 .. 0], each masked by the ReLU output it belongs to (Hmask = H) for levels >= 1.
//
// The K loop is straight-line code on purpose: every load of the pipeline is unconditional (indices are
// clamped to the level's last half-block, surplus loads touch valid memory and are dropped), because a
// load issued inside a branch makes hipcc's s_waitcnt bookkeeping fall back to vmcnt(0) at the join, which
// would serialise every half-block behind the prefetch that was just issued for the next one.
template <int NCB> struct chain_vec;
template <> struct chain_vec<1> { typedef float type; };
template <> struct chain_vec<2> { typedef f32x2 type; };
template <> struct chain_vec<4> { typedef f32x4 type; };
template <int NCB>
__device__ __forceinline__ void chain_gload(float (&d)[NCB], const float *p) {
    if constexpr (NCB == 1) d[0] = gload1(p);
    else if constexpr (NCB == 2) {
        const f32x2 q = gload2(p);
        d[0] = q[0];
        d[1] = q[1];
    } else {
        const f32x4 q = gload4(p);
#pragma unroll
        for (int c = 0; c < NCB; ++c) d[c] = q[c & 3];
    }
}
template <int NCB>
__device__ __forceinline__ void chain_store(float *p, const float (&v)[NCB]) {
    typename chain_vec<NCB>::type q;
    if constexpr (NCB == 1) q = v[0];
    else {
#pragma unroll
        for (int c = 0; c < NCB; ++c) q[c] = v[c];
    }
    *reinterpret_cast<typename chain_vec<NCB>::type *>(p) = q;
}

template <int NCB, bool BWD>
__device__ __forceinline__ void chain_block(const StepDev *__restrict__ sd, const LayerPtrs &lp,
                                            const ChainRef *__restrict__ refs, const ChainBatch *__restrict__ cbs,
                                            const ChainOp *__restrict__ ops, float *__restrict__ X,
                                            const float *__restrict__ Hmask, long long level_stride, float *xs) {
    constexpr int D = 64 * NCB, LDX = D + 4, BUF = 4 * CH_GB * LDX;
    const ChainRef ref = refs[blockIdx.x];
    const BatchDev &b = sd->b[ref.batch];
    const ChainBatch &cb = cbs[ref.batch];
    const int N = b.tp.N, L = b.L, g0 = ref.g0;
    const int ng = b.B - g0 < CH_GB ? b.B - g0 : CH_GB;
    const long long row0 = b.row_off + (long long)g0 * N;
    {   // stage the block's rows of the entry level: (graph, node)-major and contiguous in HBM. Thread t
        // moves float4 number t + 256 k, k < N * NCB; all loads are issued before the first LDS write.
        const float *src = X + (long long)(BWD ? L : 0) * level_stride + row0 * D;
        const int nrows = ng * N, nk = N * NCB;
        f32x4 v[4 * NCB];
#pragma unroll
        for (int k = 0; k < 4 * NCB; ++k) {
            const int f = threadIdx.x + 256 * (k < nk ? k : 0);
            const int r = f / (D / 4), c4 = f - r * (D / 4);
            v[k] = gload4(src + (long long)(r < nrows ? r : nrows - 1) * D + 4 * c4);
        }
#pragma unroll
        for (int k = 0; k < 4 * NCB; ++k) {
            const int f = threadIdx.x + 256 * k;
            const int r = f / (D / 4), c4 = f - r * (D / 4);
            const int i = r / N, n = r - i * N;
            if (k < nk)
                *reinterpret_cast<f32x4 *>(xs + (n * CH_GB + i) * LDX + 4 * c4) =
                    r < nrows ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int n0 = wave * 16 * NCB;
    const int colb = n0 + NCB * j;               // this lane's NCB adjacent columns
    int cur = 0;
    for (int stepi = 0; stepi < L; ++stepi) {
        const int p = BWD ? L - 1 - stepi : stepi;
        const int li = p < L - 1 ? p : sd->num_layers - 1;        // reference model.py:435-441
        const float *basis = pick_layer(lp.basis, li), *root = pick_layer(lp.root, li);
        const float *bias = pick_layer(lp.bias, li);
        const int o0 = BWD ? cb.bwd_off[p] : cb.fwd_off[p];
        const int T = ((BWD ? cb.bwd_off[p + 1] : cb.fwd_off[p + 1]) - o0) * NCB;     // half-blocks of this level
        const float *Xc = xs + cur * BUF;
        float *Xn = xs + (cur ^ 1) * BUF;
        float *Xout = X + (long long)(BWD ? p : p + 1) * level_stride + row0 * D;
        // ReLU outputs the gradients of this level belong to (levels >= 1; level 0 reads valid rows of
        // H[0] and ignores them)
        const float *Mk = BWD ? Hmask + (long long)p * level_stride + row0 * D : nullptr;
        const bool relu = !BWD && p < L - 1;
        const bool masked = BWD && p >= 1;
        float bv[NCB];
#pragma unroll
        for (int c = 0; c < NCB; ++c) bv[c] = 0.f;
        if (!BWD && bias) chain_gload<NCB>(bv, bias + colb);

        auto get_op = [&](int it) -> ChainOp { return ops[o0 + (it < T ? it : T - 1) / NCB]; };
        auto wptr = [&](const ChainOp &op, int it) -> const float * {
            const int h = (it < T ? it : T - 1) % NCB;
            const float *W = op.mat >= 0 ? basis + (long long)op.mat * D * D : root;
            return BWD ? W + (long long)colb * D + 64 * h + 4 * kq : W + (long long)(64 * h + 4 * kq) * D + colb;
        };
        f32x4 acc[NCB];
        // ReLU masks of a node (backward): loaded BEFORE the weight prefetch that precedes the item, so they
        // are the older loads (vmcnt counts in order: waiting for them never drains the prefetch), and
        // unconditionally; only the masks of a node's last item are used.
        auto load_mask = [&](float (&mk)[4][NCB], const ChainOp &op) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r;
                chain_gload<NCB>(mk[r], Mk + ((long long)(row < ng ? row : 0) * N + op.node) * D + colb);
            }
        };
        auto item = [&](const ChainOp &op, int it, const WHalf<NCB> &f, float (&mk)[4][NCB]) {
            const int h = it % NCB;
            if (h == 0 && (op.flags & CH_FIRST)) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            chain_mma<NCB>(acc, f, Xc + (op.src * CH_GB + j) * LDX + 64 * h + 4 * kq);
            if (BWD) {      // an opaque use after the MFMAs: the mask loads cannot sink into the epilogue branch
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NCB; ++c) CHAIN_PIN(mk[r][c]);
            }
            if (h == NCB - 1 && (op.flags & CH_LAST)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * kq + r;
                    float v[NCB];
#pragma unroll
                    for (int c = 0; c < NCB; ++c) {
                        v[c] = acc[c][r];
                        if (!BWD) {
                            v[c] += bv[c];
                            if (relu) v[c] = v[c] > 0.f ? v[c] : 0.f;
                        } else if (masked) {
                            v[c] = mk[r][c] > 0.f ? v[c] : 0.f;
                        }
                    }
                    chain_store<NCB>(Xn + (op.node * CH_GB + row) * LDX + colb, v);
                    if (row < ng) chain_store<NCB>(Xout + ((long long)row * N + op.node) * D + colb, v);
                }
            }
        };
        auto load = [&](WHalf<NCB> &f, const ChainOp &op, int it) {
            if (BWD) chain_load_wt<NCB>(f, wptr(op, it), D);
            else chain_load_w<NCB>(f, wptr(op, it), D);
        };
        WHalf<NCB> fa, fb;
        float mka[4][NCB], mkb[4][NCB];
        ChainOp opa = get_op(0), opb = get_op(1), opc = get_op(2);     // ops of items it, it + 1, it + 2
        load(fa, opa, 0);
        for (int it = 0; it < T; it += 2) {
            const ChainOp opd = get_op(it + 3), ope = get_op(it + 4);   // next iteration's opb, opc
            // sched_barrier: hipcc's scheduler otherwise sinks every prefetch load down to the MFMA that
            // uses it (one exposed L2 round trip per pair of MFMAs)
            if (BWD) load_mask(mka, opa);
            load(fb, opb, it + 1);
            __builtin_amdgcn_sched_barrier(0);
            item(opa, it, fa, mka);
            __builtin_amdgcn_sched_barrier(0);
            if (BWD) load_mask(mkb, opb);
            load(fa, opc, it + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (NCB >= 2 || it + 1 < T) item(opb, it + 1, fb, mkb);     // T = ops * NCB is even for NCB >= 2
            __builtin_amdgcn_sched_barrier(0);
            opa = opc;
            opb = opd;
            opc = ope;
        }
        __syncthreads();
        cur ^= 1;
    }
}

// Vector ops of the batch-uniform node states (step.hip: UOp): the forward pre-pass and the backward
// post-pass of the chain form. One 256-thread workgroup computes 64 elements of one output vector; the
// workgroups of a launch form a dependence chain of at most L levels (a level's inputs are the level
// before's outputs), handed from workgroup to workgroup INSIDE the launch as {tag, value} granules:
//
//   producer   one 8-byte agent-scope atomic store per element (tag in the high word, the float in the low one):
//              the datum is its own flag, no fence, no separate flag word
//   consumer   one thread per element re-reads its granules with agent-scope atomic loads (they bypass this CU's
//              L1) until the tags match, then the vector goes through LDS to the workgroup; bounded -- a hand-off that never arrives sets MPQE_FLAG_INTERNAL instead
//              of hanging the device
//   tag        (epoch word of the packed step) + 1, the epoch being bumped by a LATER launch of the same step, so
//              it is never a kernel argument (frozen under hipGraph replay) and stale granules of the previous
//              step never match. Every vector is written once per launch.
//
// Deadlock freedom: the ops are ordered by dependence level and take the lowest workgroup numbers of their
// launch, so a producer is always dispatched no later than its consumers and never waits for a later workgroup.
// Included by step.hip after step_chain.h.
#pragma once

typedef unsigned long long u64;
#ifndef MPQE_EMU
typedef u64 __attribute__((address_space(1))) * gu64_ptr;
#endif
#define UOP_SPIN_LIMIT (1 << 18)

__device__ __forceinline__ void gran_store(u64 *g, unsigned tag, float v) {
    const u64 x = ((u64)tag << 32) | (u64)__float_as_uint(v);
#ifdef MPQE_EMU
    *g = x;
#else
    __hip_atomic_store((gu64_ptr)g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
struct UArgs {
    const UOp *ops;
    int nops, chunks;           // chunks = D / 64 workgroups per op
    float *VT;                  // vector table [id][D]
    u64 *gran;                  // granule table [slot][D]
    const unsigned *epoch;      // hand-off epoch of this launch kind in the packed step's descriptor buffer
    const float *mode_emb;
    long long num_modes;
    float *parts;
    int32_t *err;
};

// The op's input vectors, all terms, into LDS (xs[t * D + e]) by the whole workgroup: every element is fetched by ONE
// thread -- a granule is polled by one lane of the workgroup, not by every lane that multiplies by it -- and a thread's
// NP elements are requested together and re-requested together until all of their tags match (one round trip per
// attempt, not one per element). Ends with a workgroup barrier.
template <int D>
__device__ __forceinline__ void uop_fetch_inputs(const UOp &op, const UArgs &ua, unsigned tag, float *xs) {
    constexpr int NP = UOP_MAX_TERMS * D / 256;          // elements per thread: element idx = tid + 256 q
    const int tid = threadIdx.x, total = op.nterms * D;
    float x[NP];
    const u64 *gp[NP];
    bool pend[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int idx = tid + 256 * q;
        x[q] = 0.f;
        gp[q] = nullptr;
        pend[q] = false;
        if (idx >= total) continue;
        const int t = idx / D, e = idx - t * D;
        const int kind = op.in_kind[t];
        if (kind == 0) {
            gp[q] = ua.gran + (long long)op.in_gran[t] * D + e;
            pend[q] = true;
        } else if (kind == 1) {
            const long long m = op.in_vec[t];            // (a bad mode id is flagged by the chain kernel: zero row)
            if (m >= 0 && m < ua.num_modes) x[q] = gload1(ua.mode_emb + m * D + e);
        } else {
            x[q] = gload1(ua.VT + (long long)op.in_vec[t] * D + e);
        }
    }
    for (int spins = 0;; ++spins) {
        u64 g[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            g[q] = 0;
            if (pend[q]) {
#ifdef MPQE_EMU
                g[q] = *gp[q];
#else
                g[q] = __hip_atomic_load((gu64_ptr)gp[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            }
        }
        bool left = false;
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (pend[q]) {
                if ((unsigned)(g[q] >> 32) == tag) {
                    x[q] = __uint_as_float((unsigned)g[q]);
                    pend[q] = false;
                } else {
                    left = true;
                }
            }
        if (!left) break;
        if (spins >= UOP_SPIN_LIMIT) {           // (never, unless the launch is broken: report instead of hanging)
            flag_error(ua.err, MPQE_FLAG_INTERNAL);
            break;
        }
#ifndef MPQE_EMU
        __builtin_amdgcn_s_sleep(1);
#endif
    }
#pragma unroll
    for (int q = 0; q < NP; ++q)
        if (tid + 256 * q < total) xs[tid + 256 * q] = x[q];
    __syncthreads();
}

__device__ __forceinline__ void uop_publish(const UOp &op, int e, int D, float v, const UArgs &ua, unsigned tag) {
    ua.VT[(long long)op.out_vec * D + e] = v;
    if (op.out_part >= 0) ua.parts[(long long)op.out_part * D + e] = v;
    if (op.out_gran >= 0) gran_store(ua.gran + (long long)op.out_gran * D + e, tag, v);
}

template <int D>
__device__ __forceinline__ void uop_run(const UOp &op, int chunk, const LayerPtrs &lp, const UArgs &ua, float *smem) {
    const int tid = threadIdx.x;
    const unsigned tag = *ua.epoch + 1u;
    if (op.kind == UOP_COPY) {
        if (tid < 64) {
            const int e = chunk * 64 + tid;
            const long long m = op.mode_row;
            uop_publish(op, e, D, (m >= 0 && m < ua.num_modes) ? gload1(ua.mode_emb + m * D + e) : 0.f, ua, tag);
        }
        return;
    }
    if (op.kind == UOP_RED) {
        // 64 columns x 4 row groups; row group g adds rows g, g + 4, ... in order, then (0 + 1) + (2 + 3): fixed order
        const int cl = tid & 63, rg = tid >> 6, e = chunk * 64 + cl;
        float acc = 0.f;
        for (int r = rg; r < op.nrows; r += 4) acc += gload1(ua.parts + (long long)(op.row0 + r) * D + e);
        smem[rg * 64 + cl] = acc;
        __syncthreads();
        if (tid < 64) uop_publish(op, e, D, (smem[cl] + smem[64 + cl]) + (smem[128 + cl] + smem[192 + cl]), ua, tag);
        return;
    }
    if (op.kind == UOP_FWD) {
        // out[c] = act(bias[c] + sum_t sum_k in_t[k] M_t[k][c]) for the chunk's 64 columns. Thread (c4, kg): columns
        // 4 c4 .. 4 c4 + 3, rows k = kg + 16 i. The matrix pieces of two terms are requested BEFORE the inputs are
        // waited for: they do not depend on them, and their HBM latency is most of the op.
        constexpr int KI = D / 16;
        const int c4 = tid & 15, kg = tid >> 4;
        const int col = chunk * 64 + 4 * c4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float *xs = smem + 1024;
        for (int t0 = 0; t0 < op.nterms; t0 += 2) {
            f32x4 w[2][KI];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = t0 + tt < op.nterms ? t0 + tt : t0;
                const float *M = op.mat[t] >= 0 ? pick_layer(lp.basis, op.layer[t]) + (long long)op.mat[t] * D * D
                                                : pick_layer(lp.root, op.layer[t]);
#pragma unroll
                for (int i = 0; i < KI; ++i) w[tt][i] = gload4(M + (long long)(kg + 16 * i) * D + col);
            }
            if (t0 == 0) uop_fetch_inputs<D>(op, ua, tag, xs);      // (behind the first matrix requests)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                if (t0 + tt >= op.nterms) break;
#pragma unroll
                for (int i = 0; i < KI; ++i) acc += xs[(t0 + tt) * D + kg + 16 * i] * w[tt][i];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) smem[kg * 64 + 4 * c4 + e] = acc[e];
        __syncthreads();
        if (tid < 64) {
            const int e = chunk * 64 + tid;
            float v = 0.f;
            for (int g = 0; g < 16; ++g) v += smem[g * 64 + tid];
            if (op.bias_layer >= 0) {
                const float *bp = pick_layer(lp.bias, op.bias_layer);
                if (bp) v += bp[e];
            }
            if (op.relu) v = v > 0.f ? v : 0.f;
            uop_publish(op, e, D, v, ua, tag);
        }
        return;
    }
    // UOP_BWD: out[i] = mask_i * sum_t sum_j in_t[j] M_t[i][j] for the chunk's 64 rows i. Thread (l, r): rows
    // r + 16 q, columns 4 l + 64 c; the 16 lanes of a row meet in a DPP row sum.
    constexpr int CJ = D / 64;
    const int l = tid & 15, r = tid >> 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float *xs = smem + 1024;
    for (int t0 = 0; t0 < op.nterms; t0 += 2) {
        f32x4 w[2][4][CJ];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = t0 + tt < op.nterms ? t0 + tt : t0;
            const float *M = op.mat[t] >= 0 ? pick_layer(lp.basis, op.layer[t]) + (long long)op.mat[t] * D * D
                                            : pick_layer(lp.root, op.layer[t]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < CJ; ++c)
                    w[tt][q][c] = gload4(M + (long long)(chunk * 64 + r + 16 * q) * D + 4 * l + 64 * c);
        }
        if (t0 == 0) uop_fetch_inputs<D>(op, ua, tag, xs);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            if (t0 + tt >= op.nterms) break;
#pragma unroll
            for (int c = 0; c < CJ; ++c) {
                const f32x4 sv = *reinterpret_cast<const f32x4 *>(xs + (t0 + tt) * D + 4 * l + 64 * c);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q] += w[tt][q][c][e] * sv[e];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float y = chain_sum16(acc[q]);
        const int i = chunk * 64 + r + 16 * q;
        if (l == 0) {
            const bool on = op.mask_vec < 0 || gload1(ua.VT + (long long)op.mask_vec * D + i) > 0.f;
            uop_publish(op, i, D, on ? y : 0.f, ua, tag);
        }
    }
}

// workgroup `ub` of a launch's vector ops (uniform branch: D is the step's dim, 64 / 128 / 256 in the chain form)
__device__ __forceinline__ void uop_block(int ub, int D, const LayerPtrs &lp, const UArgs &ua, float *smem) {
    const UOp &op = ua.ops[ub / ua.chunks];
    const int chunk = ub % ua.chunks;
    if (D == 64) uop_run<64>(op, chunk, lp, ua, smem);
    else if (D == 128) uop_run<128>(op, chunk, lp, ua, smem);
    else uop_run<256>(op, chunk, lp, ua, smem);
}

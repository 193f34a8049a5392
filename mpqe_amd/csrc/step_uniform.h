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
    // merged launch (the backward post-pass as workgroups of the chain launch, step.hip: DoneMeta): NULL = the rows of
    // `parts` and the vectors of the pre-pass come from an EARLIER launch, nothing to wait for
    const unsigned *done;       // counters of finished chain workgroups
    const int *done_inc;        // chain workgroups per counter and step
    DoneMeta dm;
    const unsigned *fwd_done;   // pre-pass workgroups finished, ever (their plain VT stores are out)
    const unsigned *epoch_m;    // epoch of the merged launches (targets of both counters)
    int fwd_blocks;
    int vt_through;             // 1: merged launch, pre-pass: VT (and `parts`) stores written through (uop_publish); 2: fused
                                // tail, post-pass: those of the ops marked UOp.through
};

// Merged launch: wait until (a) the pre-pass workgroups of this launch have all finished -- the post-pass reads their
// vectors from VT with plain loads -- and (b) the chain workgroups of the batches in `mask` have published their rows.
// One thread per counter polls (agent-scope loads, bounded); ends with a workgroup barrier. Deadlock freedom: the
// pre-pass and the chain workgroups come before every post-pass workgroup in the launch and wait for none of them.
__device__ __forceinline__ unsigned uop_poll(const unsigned *p) {
#ifdef MPQE_EMU
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void uop_wait_until(const unsigned *p, unsigned want, int32_t *err) {
    for (int spins = 0; (int)(uop_poll(p) - want) < 0; ++spins) {
        if (spins >= UOP_SPIN_LIMIT) {
            flag_error(err, MPQE_FLAG_INTERNAL | 0x400);
            break;
        }
#ifndef MPQE_EMU
        __builtin_amdgcn_s_sleep(8);
#endif
    }
}
__device__ __forceinline__ void uop_wait_prepass(const UArgs &ua) {
    if (!ua.done || !ua.fwd_done) return;           // (uniform)
    if (threadIdx.x == 0) uop_wait_until(ua.fwd_done, (*ua.epoch_m + 1u) * (unsigned)ua.fwd_blocks, ua.err);
    __syncthreads();
}
// (kept inline: a real call in these kernels costs a stack and the whole register budget -- measured 150 us per step
// against 65 with one __noinline__ helper)
__device__ __forceinline__ void uop_wait_chain(const UArgs &ua, unsigned mask) {
    if (!ua.done || !mask) return;           // (uniform)
    const int tid = threadIdx.x;
    const unsigned ep = *ua.epoch_m + 1u;
    for (int b = 0; b < MPQE_STEP_MAX_BATCHES; ++b) {
        if (!((mask >> b) & 1u)) continue;
        for (int c = ua.dm.base[b] + tid; c < ua.dm.base[b + 1]; c += 256)
            uop_wait_until(ua.done + c, ep * (unsigned)ua.done_inc[c], ua.err);
    }
    __syncthreads();
}

// p[0] + p[D] + ... (nr terms, in that order)
#define UOP_RED_AHEAD 32
__device__ __forceinline__ float uop_row_sum(const float *pr, int nr, int D) {
    float acc = 0.f;
    for (int r0 = 0; r0 < nr; r0 += UOP_RED_AHEAD) {
        float v[UOP_RED_AHEAD];
#pragma unroll
        for (int r = 0; r < UOP_RED_AHEAD; ++r) v[r] = gload1(pr + (long long)(r0 + r < nr ? r0 + r : r0) * D);
#pragma unroll
        for (int r = 0; r < UOP_RED_AHEAD; ++r)
            if (r0 + r < nr) acc += v[r];
    }
    return acc;
}

// The op's input vectors, all terms, into LDS (xs[t * D + e]) by the whole workgroup: every element is fetched by ONE
// thread -- a granule is polled by one lane of the workgroup, not by every lane that multiplies by it -- and a thread's
// NP elements are requested together and re-requested together until all of their tags match (one round trip per
// attempt, not one per element). Ends with a workgroup barrier.
template <int D>
__device__ __forceinline__ void uop_fetch_inputs(const UOp &op, const UArgs &ua, unsigned tag, float *xs) {
    constexpr int NP = UOP_MAX_TERMS * D / 256;          // elements per thread: element idx = tid + 256 q
    const int tid = threadIdx.x, total = op.nterms * D;
    uop_wait_chain(ua, op.wait_mask);                    // merged launch: the chain workgroups' rows of `parts` (in_kind 3)
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
        } else if (kind == 3) {
            // the column sum itself: rows in_vec .. in_vec + in_gran of `parts` (the chain kernel's per-block sums,
            // an earlier launch), added in row order, UOP_RED_AHEAD requests at a time (B = 512: all 32 rows in one
            // round trip) -- the same order and the same sum as the UOP_RED op of this vector, without waiting for it
            x[q] = uop_row_sum(ua.parts + (long long)op.in_vec[t] * D + e, op.in_gran[t], D);
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
            flag_error(ua.err, MPQE_FLAG_INTERNAL | 0x800);
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
    if (op.out_gran >= 0) gran_store(ua.gran + (long long)op.out_gran * D + e, tag, v);      // (first: somebody may wait)
#ifndef MPQE_EMU
    const bool thr = ua.vt_through == 1 || (ua.vt_through == 2 && op.through);
    if (op.out_part >= 0 && thr) {      // (fused tail: the reduction workgroups of the same launch read the row)
        __hip_atomic_store(ua.parts + (long long)op.out_part * D + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else
#endif
    if (op.out_part >= 0) ua.parts[(long long)op.out_part * D + e] = v;
#ifndef MPQE_EMU
    if (thr) {
        // merged launch, pre-pass: the post-pass workgroups of the SAME launch read these vectors with plain loads, from
        // other XCDs -- written through to memory at agent scope (the write-through store on the hand-off path of the
        // post-pass cost 3.4 us per step: there the granule goes first and VT stays a plain store)
        __hip_atomic_store(ua.VT + (long long)op.out_vec * D + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#endif
    ua.VT[(long long)op.out_vec * D + e] = v;
}

template <int D>
__device__ __forceinline__ void uop_run(const UOp &op, int chunk, const LayerPtrs &lp, const UArgs &ua, float *smem,
                                        const GradPtrs *gp, int zeroed) {
    const int tid = threadIdx.x;
    const unsigned tag = *ua.epoch + 1u;
    if (op.kind == UOP_R1 || op.kind == UOP_RED || op.kind == UOP_BWD) uop_wait_prepass(ua);
    if (op.kind == UOP_RED) uop_wait_chain(ua, op.wait_mask);
    if (op.kind == UOP_R1) {
        // rows [64 chunk, 64 chunk + 64) of the matrix: out[i][j] = sum_t u_t[i] v_t[j], t in table order (what the
        // reduction launch computes for a group without slabs). Thread (r, c4): rows r + 8 q, columns 4 c4 + 128 cc.
        if (!gp) return;
        float *out = op.r1_rel >= 0 ? pick_grad(gp->basis, op.r1_layer) : pick_grad(gp->root, op.r1_layer);
        if (out && op.r1_rel >= 0) out += (long long)op.r1_rel * D * D;
        float *xs = smem + 1024;
        float *us = smem;                                   // u_t[64 chunk + i]: [t][64]
        for (int q = tid; q < op.nterms * 64; q += 256)
            us[q] = gload1(ua.VT + (long long)op.u_vec[q / 64] * D + chunk * 64 + (q & 63));
        uop_fetch_inputs<D>(op, ua, tag, xs);               // (ends with a workgroup barrier)
        if (!out) return;
        constexpr int C4 = D / 4;                           // float4 columns per row
        for (int f = tid; f < 64 * C4; f += 256) {
            const int i = f / C4, c4 = f - i * C4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < op.nterms; ++t) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xs + t * D + 4 * c4);
                acc += us[t * 64 + i] * v;
            }
            f32x4 *dst = reinterpret_cast<f32x4 *>(out + (long long)(chunk * 64 + i) * D + 4 * c4);
            if (zeroed) *dst = acc;
            else *dst = *dst + acc;
        }
        return;
    }
    if (op.kind == UOP_COPY) {
        if (tid < 64) {
            const int e = chunk * 64 + tid;
            const long long m = op.mode_row;
            uop_publish(op, e, D, (m >= 0 && m < ua.num_modes) ? gload1(ua.mode_emb + m * D + e) : 0.f, ua, tag);
        }
        return;
    }
    if (op.kind == UOP_RED) {
        // rows added in row order (the order the in_kind 3 inputs of the UOP_BWD ops use: one value, bit for bit)
        if (tid < 64) {
            const int e = chunk * 64 + tid;
            uop_publish(op, e, D, uop_row_sum(ua.parts + (long long)op.row0 * D + e, op.nrows, D), ua, tag);
        }
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
        float bias_e = 0.f;                 // requested now, added after the hand-offs: off the dependence chain
        if (tid < 64 && op.bias_layer >= 0) {
            const float *bp = pick_layer(lp.bias, op.bias_layer);
            if (bp) bias_e = gload1(bp + chunk * 64 + tid);
        }
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
            v += bias_e;
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
    float mval[4] = {1.f, 1.f, 1.f, 1.f};       // the forward state whose sign masks the row: requested before the hand-offs
    if (l == 0 && op.mask_vec >= 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) mval[q] = gload1(ua.VT + (long long)op.mask_vec * D + chunk * 64 + r + 16 * q);
    }
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
        if (l == 0) uop_publish(op, i, D, mval[q] > 0.f ? y : 0.f, ua, tag);
    }
}

// workgroup `ub` of a launch's vector ops (uniform branch: D is the step's dim, 64 / 128 / 256 in the chain form)
__device__ __forceinline__ void uop_block(int ub, int D, const LayerPtrs &lp, const UArgs &ua, float *smem,
                                          const GradPtrs *gp, int zeroed) {
#ifndef MPQE_EMU
    __builtin_amdgcn_s_setprio(3);      // a latency chain next to throughput work (transposes / weight-gradient tiles)
#endif
    const UOp &op = ua.ops[ub / ua.chunks];
    const int chunk = ub % ua.chunks;
    if (D == 64) uop_run<64>(op, chunk, lp, ua, smem, gp, zeroed);
    else if (D == 128) uop_run<128>(op, chunk, lp, ua, smem, gp, zeroed);
    else uop_run<256>(op, chunk, lp, ua, smem, gp, zeroed);
}

// Backward post-pass of the batch-uniform node states as CLOSURES (split weight-gradient launch of the chain form).
//
// The post-pass of a batch (step.hip: UOp) is a short dependence chain on D-vectors: column sums of the chain kernel's
// per-block rows (UOP_RED) -> up to L levels of vector x matrix^T products (UOP_BWD) -> rank-1 gradient matrices
// (UOP_R1). Its first form (step_uniform.h: one workgroup per op and 64 outputs, handed from workgroup to workgroup as
// {tag, value} granules) paid a cross-XCD hand-off of 2 - 3 us per level next to the streaming weight-gradient tiles:
// the last op of the AIFB step ended at 15.5 us and bounded the launch. A hand-off inside ONE workgroup is an LDS
// write and a barrier. So here one workgroup runs the whole closure of a batch:
//
//   pre     every column sum the closure needs (rows of `parts`, all requests of all sums in flight together) and every
//           pre-pass vector its rank-1 terms read (rows of VT) -> LDS slots
//   BWD     the ops in dependence order; an op = items (64 output rows x one term's matrix), the matrix pieces of the
//           next CL_NBUF items always in flight (they depend on nothing), inputs read from LDS slots, outputs to an LDS
//           slot (+ VT / `parts` for the step's reduction launch) and a workgroup barrier per op
//   R1      out[i][j] = sum_t u_t[i] v_t[j] from LDS, stored whole
//
// No granule, no epoch, no poll: nothing in the launch waits for another workgroup. What bounds a closure is one CU's
// load bandwidth for its matrices (3-chain under the TM readout: 9 terms x 64 KB at D = 128).
// Included by step.hip after step_uniform.h.
#pragma once

#define UOP_IN_LDS 4          // UOp.in_kind: the vector is in LDS slot in_gran[t] of the closure's workgroup
#define CL_MAX_SLOTS 32       // vectors a closure holds in LDS (the host falls back to the vector-op form beyond)
#define CL_PRE_AHEAD 4        // pre ops whose rows are requested together

#define CL_MAX_ITEMS 96       // BWD items of one closure (terms x D / 64 chunks; the host falls back beyond)
struct Closure {
    int first, count;         // ops [first, first + count) of the closure op table, in execution order
    int npre, pad;            // the first npre of them are pre ops (UOP_RED: column sums / copies of VT rows)
    int item_first, nitems;   // its BWD items in the item table, in execution order: (op, 64-row chunk, term), terms innermost
};
// one item of the BWD phase as the host lays it out: 64 output rows [64 chunk, 64 chunk + 64) of one op, one term's matrix.
// The closure's records are copied to LDS once; the item stream then reads nothing from memory but matrices.
#define CLI_FIRST 1           // first term of (op, chunk): the accumulators start from zero
#define CLI_LAST 2            // last term: sum over the lanes, mask, store the 64 outputs
#define CLI_OPEND 4           // ... and the op's vector is whole: workgroup barrier
struct ClItemRec {
    int layer, mat;           // the term's matrix (mat: relation id, -1 = root), row-major [D][D]; out[i] = sum_j in[j] M[i][j]
    int chunk, flags;
    int in_slot, out_slot;    // LDS slots of the term's input vector / the op's output vector
    int out_vec, out_part;    // CLI_LAST: VT row / `parts` row the outputs also go to (-1: none)
    int mask_slot, pad[3];    // CLI_LAST: LDS slot of the forward state whose sign masks the outputs (-1: none)
};
struct ClosureArgs {
    const Closure *cl;
    const UOp *ops;
    const ClItemRec *items;
    int ncl;
};

struct ClItem {
    const float *M;           // the 64 x D piece of the term's matrix
    int k;                    // index of the item's record in LDS
    int valid;
};

template <int D>
struct ClBuf {
    f32x4 w[4][D / 64];       // rows r + 16 q of the chunk, columns 4 l + 64 c
};

template <int D>
__device__ __forceinline__ void closure_run(const Closure cl, const UOp *__restrict__ ops, const ClItemRec *__restrict__ items,
                                            const LayerPtrs &lp, const UArgs &ua, float *smem, const GradPtrs &gp, int zeroed) {
    constexpr int CJ = D / 64, LQ = D / 4, RG = 256 / LQ;
    constexpr int NBUF = D >= 256 ? 2 : 4;
    const int tid = threadIdx.x;
    float *vec = smem;                                   // [CL_MAX_SLOTS][D]
    float *red = smem + CL_MAX_SLOTS * D;                // [CL_PRE_AHEAD][RG][D] partial column sums
    int *recs = reinterpret_cast<int *>(red + CL_PRE_AHEAD * RG * D);      // [nitems] ClItemRec
    static_assert(sizeof(ClItemRec) == 48, "ClItemRec is copied as 12 words");
    for (int q = tid; q < cl.nitems * 12; q += 256) recs[q] = reinterpret_cast<const int *>(items + cl.item_first)[q];
    // ---- pre ops: slot <- sum of nrows consecutive rows of `parts` from row0 (in_kind[0] == 3), or <- VT row in_vec[0]
    // (in_kind[0] == 2, nrows = 1). LQ lanes cover a row (16-byte loads), the RG row groups take every RG-th row; the row
    // groups' sums are added in order: one fixed order per vector.
    {
        const int c4 = tid % LQ, rg = tid / LQ;
        for (int o0 = 0; o0 < cl.npre; o0 += CL_PRE_AHEAD) {
            f32x4 acc[CL_PRE_AHEAD];
            const float *src[CL_PRE_AHEAD];
            int nr[CL_PRE_AHEAD];
#pragma unroll
            for (int q = 0; q < CL_PRE_AHEAD; ++q) {
                const UOp &op = ops[cl.first + (o0 + q < cl.npre ? o0 + q : o0)];
                src[q] = (op.in_kind[0] == 3 ? ua.parts + (long long)op.row0 * D : ua.VT + (long long)op.in_vec[0] * D) + 4 * c4;
                nr[q] = o0 + q < cl.npre ? op.nrows : 0;
                acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            int most = 0;
#pragma unroll
            for (int q = 0; q < CL_PRE_AHEAD; ++q) most = nr[q] > most ? nr[q] : most;
            for (int r0 = rg; r0 < most; r0 += 4 * RG) {             // four rows per op and thread in flight
                f32x4 v[CL_PRE_AHEAD][4];
#pragma unroll
                for (int q = 0; q < CL_PRE_AHEAD; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = r0 + k * RG;
                        v[q][k] = gload4(src[q] + (long long)(r < nr[q] ? r : 0) * D);
                    }
#pragma unroll
                for (int q = 0; q < CL_PRE_AHEAD; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (r0 + k * RG < nr[q]) acc[q] += v[q][k];
            }
#pragma unroll
            for (int q = 0; q < CL_PRE_AHEAD; ++q)
                *reinterpret_cast<f32x4 *>(red + ((q * RG + rg) * D) + 4 * c4) = acc[q];
            __syncthreads();
            const int n = cl.npre - o0 < CL_PRE_AHEAD ? cl.npre - o0 : CL_PRE_AHEAD;
            for (int idx = tid; idx < n * D; idx += 256) {
                const int q = idx / D, e = idx - q * D;
                const UOp &op = ops[cl.first + o0 + q];
                float s = red[(q * RG) * D + e];
                for (int g = 1; g < RG; ++g) s += red[(q * RG + g) * D + e];
                vec[op.out_gran * D + e] = s;
                if (op.out_vec >= 0) ua.VT[(long long)op.out_vec * D + e] = s;      // (a rank-1 term of the reduction launch reads it)
            }
            __syncthreads();
        }
    }
    __syncthreads();            // (the item records are in LDS; a closure without pre ops has not met a barrier yet)
    // ---- BWD ops as ONE stream of items: (op, 64-row chunk, term), terms innermost; records from LDS, wave-uniform
    const int l = tid & 15, r = tid >> 4;
    const int op_end = cl.first + cl.count;
    auto rec = [&](int k, int word) -> int { return __builtin_amdgcn_readfirstlane(recs[k * 12 + word]); };
    int next_k = 0;
    const float *last_M = pick_layer(lp.root, 0);
    auto next_item = [&]() -> ClItem {
        ClItem it;
        if (next_k >= cl.nitems) {
            it.M = last_M;
            it.k = 0;
            it.valid = 0;
            return it;
        }
        const int k = next_k++;
        const int layer = rec(k, 0), mat = rec(k, 1), chunk = rec(k, 2);
        it.k = k;
        it.valid = 1;
        it.M = (mat >= 0 ? pick_layer(lp.basis, layer) + (long long)mat * D * D : pick_layer(lp.root, layer)) + (long long)chunk * 64 * D;
        last_M = it.M;
        return it;
    };
    // (the item stream issues matrix loads and nothing else to memory: no store, no conditional load -- a global store in
    // front of an op's barrier made the barrier's release drain every prefetched matrix piece: 4 us per term)
    auto load = [&](ClBuf<D> &b, const ClItem &it) {       // unconditional: an invalid item re-reads the last matrix piece
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < CJ; ++c) b.w[q][c] = gload4(it.M + (long long)(r + 16 * q) * D + 4 * l + 64 * c);
    };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const ClBuf<D> &b, const ClItem &it) {
        const int chunk = rec(it.k, 2), flags = rec(it.k, 3), in_slot = rec(it.k, 4);
        if (flags & CLI_FIRST) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = 0.f;
        }
        const float *x = vec + in_slot * D;
#pragma unroll
        for (int c = 0; c < CJ; ++c) {
            const f32x4 sv = *reinterpret_cast<const f32x4 *>(x + 4 * l + 64 * c);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[q] += b.w[q][c][e] * sv[e];
        }
        if (flags & CLI_LAST) {
            const int out_slot = rec(it.k, 5), mask_slot = rec(it.k, 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float y = chain_sum16(acc[q]);
                if (l == 0) {
                    const int i = chunk * 64 + r + 16 * q;
                    const float m = mask_slot >= 0 ? vec[mask_slot * D + i] : 1.f;
                    vec[out_slot * D + i] = m > 0.f ? y : 0.f;
                }
            }
            if (flags & CLI_OPEND) __syncthreads();             // the op's vector is whole: the next op may read it
        }
    };
    {
        ClBuf<D> b0, b1, b2, b3;
        ClItem i0 = next_item(), i1 = next_item(), i2, i3;
        load(b0, i0);
        load(b1, i1);
        if constexpr (NBUF == 4) {
            i2 = next_item();
            i3 = next_item();
            load(b2, i2);
            load(b3, i3);
        }
        __builtin_amdgcn_sched_barrier(0);
        while (true) {
#define CL_STEP(B, I)                                 \
    if (!I.valid) break;                              \
    compute(B, I);                                    \
    __builtin_amdgcn_sched_barrier(0);                \
    I = next_item();                                  \
    load(B, I);                                       \
    __builtin_amdgcn_sched_barrier(0);
            CL_STEP(b0, i0)
            CL_STEP(b1, i1)
            if constexpr (NBUF == 4) {
                CL_STEP(b2, i2)
                CL_STEP(b3, i3)
            }
#undef CL_STEP
        }
    }
    __syncthreads();
    // the BWD ops' vectors the step's reduction reads: VT rows (rank-1 terms) / rows of `parts` (bias, mode rows)
    for (int k = 0; k < cl.nitems; ++k) {
        if (!(rec(k, 3) & CLI_OPEND)) continue;
        const int out_slot = rec(k, 5), out_vec = rec(k, 6), out_part = rec(k, 7);
        if (tid < D) {
            const float v = vec[out_slot * D + tid];
            if (out_vec >= 0) ua.VT[(long long)out_vec * D + tid] = v;
            if (out_part >= 0) ua.parts[(long long)out_part * D + tid] = v;
        }
    }
    // ---- R1 ops: gradient matrices made of rank-1 terms only, out[i][j] = sum_t u_t[i] v_t[j] (u, v: LDS slots mat[t], in_gran[t])
    for (int k = cl.first + cl.npre; k < op_end; ++k) {
        const UOp &op = ops[k];
        if (op.kind != UOP_R1) continue;
        float *out = op.r1_rel >= 0 ? pick_grad(gp.basis, op.r1_layer) : pick_grad(gp.root, op.r1_layer);
        if (!out) continue;
        if (op.r1_rel >= 0) out += (long long)op.r1_rel * D * D;
        for (int f = tid; f < D * LQ; f += 256) {
            const int i = f / LQ, c4 = f - i * LQ;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < op.nterms; ++t)
                a += vec[op.mat[t] * D + i] * *reinterpret_cast<const f32x4 *>(vec + op.in_gran[t] * D + 4 * c4);
            f32x4 *dst = reinterpret_cast<f32x4 *>(out + (long long)i * D + 4 * c4);
            if (zeroed) *dst = a;
            else *dst = *dst + a;
        }
    }
}

// workgroup `cb` of the closures of a launch (uniform branch: D = 64 / 128 / 256 in the chain form)
__device__ __forceinline__ void closure_block(int cb, int D, const ClosureArgs &ca, const LayerPtrs &lp, const UArgs &ua,
                                              float *smem, const GradPtrs &gp, int zeroed) {
    const Closure cl = ca.cl[cb];
    if (D == 64) closure_run<64>(cl, ca.ops, ca.items, lp, ua, smem, gp, zeroed);
    else if (D == 128) closure_run<128>(cl, ca.ops, ca.items, lp, ua, smem, gp, zeroed);
    else closure_run<256>(cl, ca.ops, ca.items, lp, ua, smem, gp, zeroed);
}

// Backward post-pass of the batch-uniform node states as CLOSURES (split weight-gradient launch of the chain form).
//
// The post-pass of a batch (step.hip: UOp) is a short dependence chain on D-vectors: column sums of the chain kernel's
// per-block rows (UOP_RED) -> up to L levels of vector x matrix^T products (UOP_BWD) -> rank-1 gradient matrices
// (UOP_R1). Its first form (step_uniform.h: one workgroup per op and 64 outputs, handed from workgroup to workgroup as
// {tag, value} granules) paid a cross-XCD hand-off of 2 - 3 us per level next to the streaming weight-gradient tiles:
// the last op of the AIFB step ended at 15.5 us and bounded the launch. A hand-off inside ONE workgroup is an LDS
// write and a barrier. So here one workgroup runs the whole closure of a batch:
//
//   block   the closure's programme -- counts, pre records, item records, rank-1 records -- is ONE fixed-size block of
//           words the host lays out (ClBlock); the workgroup copies it to LDS in one round trip and reads nothing else
//           from the descriptor table (a first form read UOp fields from memory where it needed them: every op cost
//           dependent round trips of 1 - 2 us next to the streaming tiles -- 8 us for a one-term closure, 42 for nine)
//   pre     every column sum the closure needs (rows of `parts`, all requests of all sums in flight together) and every
//           pre-pass vector its ops read (rows of VT: rank-1 u vectors, ReLU masks) -> LDS slots; the first matrix pieces
//           of the BWD phase are requested in front of it (they depend on nothing)
//   BWD     the ops in dependence order as ONE stream of items (64 output rows x one term's matrix), the matrix pieces of
//           the next CL_NBUF items always in flight, inputs read from LDS slots, outputs to an LDS slot and a workgroup
//           barrier per op; no store and no conditional load inside the stream
//   out     the vectors the step's reduction reads go to VT / `parts`
//   R1      out[i][j] = sum_t u_t[i] v_t[j] from LDS, stored whole
//
// No granule, no epoch, no poll: nothing in the launch waits for another workgroup. What bounds a closure is one CU's
// load bandwidth for its matrices (3-chain under the TM readout: 9 terms x 64 KB at D = 128).
// Included by step.hip after step_uniform.h.
#pragma once

#define UOP_IN_LDS 4          // UOp.in_kind (host-side closure ops): the vector is in LDS slot in_gran[t]
#define CL_MAX_SLOTS 32       // vectors a closure holds in LDS (the host falls back to the vector-op form beyond)
#define CL_PRE_AHEAD 4        // pre ops whose rows are requested together
#define CL_MAX_PRE 32
#define CL_MAX_ITEMS 96       // BWD items of one closure (terms x D / 64 chunks)
#define CL_MAX_R1 8
#define CL_MAX_OUT 16

// pre record: slot <- sum of nrows consecutive rows of `parts` from `row` (kind 3), or <- VT row `row` (kind 2, nrows 1)
struct ClPreRec {
    int kind, row, nrows, slot;
    int out_vec, pad[3];      // >= 0: the sum also goes to this VT row (a rank-1 term of the step's reduction reads it)
};
// item record: the 64 x D piece [64 chunk, 64 chunk + 64) of ONE matrix and every use of it at the item's level -- the ops
// of a level are independent of each other, so a matrix several of them multiply by (the root matrix of a level with
// three uniform nodes) is read once. A use = one term of one op: acc[a] (+)= piece . slot in; ops of a level own one of
// CL_ACCS accumulator sets each.
#define CLI_FIRST 1           // the op's first term in this chunk: its accumulators start from zero
#define CLI_LAST 2            // its last term: sum over the lanes, mask, store the 64 outputs to the op's slot
#define CL_ACCS 3
#define CL_USES 3
// use word: in_slot | acc << 5 | flags << 7 | out_slot << 9 | (mask_slot + 1) << 14
struct ClItemRec {
    int layer, mat;           // the matrix (mat: relation id, -1 = root), row-major [D][D]; out[i] = sum_j in[j] M[i][j]
    int meta;                 // chunk | nuses << 8 | level_end << 16 (workgroup barrier behind the item: the level's vectors are whole)
    int use[CL_USES];
    int pad[2];
};
// a vector of the BWD phase the step's reduction reads: its VT row (rank-1 terms) / row of `parts` (bias, mode rows)
struct ClOutRec {
    int slot, out_vec, out_part, pad;
};
// rank-1 record: gradient matrix (layer, rel | -1 root) = sum_t slot u[t] (x) slot v[t]
struct ClR1Rec {
    int layer, rel, nterms, pad;
    int u[UOP_MAX_TERMS], v[UOP_MAX_TERMS];
};
struct ClBlock {
    int npre, nitems, nr1, batch, nout, pad[3];
    ClPreRec pre[CL_MAX_PRE];
    ClItemRec item[CL_MAX_ITEMS];
    ClR1Rec r1[CL_MAX_R1];
    ClOutRec out[CL_MAX_OUT];
};
#define CL_BLOCK_WORDS 2048
static_assert(sizeof(ClBlock) <= CL_BLOCK_WORDS * 4 && sizeof(ClBlock) % 16 == 0, "a closure's programme is one 8 KB block");
static_assert(sizeof(ClPreRec) == 32 && sizeof(ClItemRec) == 32 && sizeof(ClR1Rec) == 48 && sizeof(ClOutRec) == 16, "record strides below");
#define CL_PRE_W0 8
#define CL_ITEM_W0 (CL_PRE_W0 + CL_MAX_PRE * 8)
#define CL_R1_W0 (CL_ITEM_W0 + CL_MAX_ITEMS * 8)
#define CL_OUT_W0 (CL_R1_W0 + CL_MAX_R1 * 12)

struct ClosureArgs {
    const int *blocks;        // [ncl][CL_BLOCK_WORDS]
    int ncl;
};

typedef int cl_i4 __attribute__((ext_vector_type(4)));
struct ClItem {
    const float *M;           // the 64 x D piece of the item's matrix
    int meta, use[CL_USES];   // its record's words, wave-uniform (read one item-distance ahead)
};

template <int D>
struct ClBuf {
    f32x4 w[4][D / 64];       // rows r + 16 q of the chunk, columns 4 l + 64 c
};

template <int D>
__device__ __forceinline__ void closure_run(const int *__restrict__ gblock, const LayerPtrs &lp, const UArgs &ua,
                                            float *smem, const GradPtrs &gp, int zeroed, long long *stamps) {
#ifndef MPQE_EMU
#define CL_STAMP(slot)                                                            \
    if (stamps && threadIdx.x == 0) {                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        \
        stamps[slot] = (long long)wall_clock64();                                 \
    }
#else
#define CL_STAMP(slot)
#endif
    constexpr int CJ = D / 64, LQ = D / 4, RG = 256 / LQ;
    constexpr int NBUF = D >= 256 ? 2 : 4;
    const int tid = threadIdx.x;
    float *vec = smem;                                   // [CL_MAX_SLOTS][D]
    float *red = smem + CL_MAX_SLOTS * D;                // [CL_PRE_AHEAD][RG][D] partial column sums
    int *blk = reinterpret_cast<int *>(red + CL_PRE_AHEAD * RG * D);      // the closure's programme
    for (int q = tid; q < CL_BLOCK_WORDS / 4; q += 256)                  // 512 x 16 bytes = the whole block, one round trip
        reinterpret_cast<f32x4 *>(blk)[q] = gload4(reinterpret_cast<const float *>(gblock) + 4 * q);
    __syncthreads();
    CL_STAMP(2)
    auto word = [&](int w) -> int { return __builtin_amdgcn_readfirstlane(blk[w]); };
    const int npre = word(0), nitems = word(1), nr1 = word(2);
    // ---- the BWD stream's first matrix pieces: requested now, in front of the pre phase
    const int l = tid & 15, r = tid >> 4;
    int next_k = 0;
    auto item_matrix = [&](int layer, int mat, int chunk) -> const float * {
        return (mat >= 0 ? pick_layer(lp.basis, layer) + (long long)mat * D * D : pick_layer(lp.root, layer)) + (long long)chunk * 64 * D;
    };
    auto next_item = [&]() -> ClItem {
        ClItem it;
        const int k = next_k < nitems ? next_k : 0;      // (past the end: the first record again, its matrix piece re-read, unused)
        ++next_k;
        // the record's words in two 16-byte LDS reads (one wait), read NBUF items ahead of their use
        const cl_i4 *rp = reinterpret_cast<const cl_i4 *>(blk + CL_ITEM_W0 + 8 * k);
        const cl_i4 a = rp[0], b = rp[1];
        const int layer = __builtin_amdgcn_readfirstlane(a[0]), mat = __builtin_amdgcn_readfirstlane(a[1]);
        it.meta = __builtin_amdgcn_readfirstlane(a[2]);
        it.use[0] = __builtin_amdgcn_readfirstlane(a[3]);
        it.use[1] = __builtin_amdgcn_readfirstlane(b[0]);
        it.use[2] = __builtin_amdgcn_readfirstlane(b[1]);
        it.M = item_matrix(layer, mat, it.meta & 255);
        return it;
    };
    // (the item stream issues matrix loads and nothing else to memory: no store, no conditional load)
    auto load = [&](ClBuf<D> &b, const ClItem &it) {       // unconditional: past the end the last matrix piece is read again
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < CJ; ++c) b.w[q][c] = gload4(it.M + (long long)(r + 16 * q) * D + 4 * l + 64 * c);
    };
    ClBuf<D> b0, b1, b2, b3;
    // (in buffer order, pinned: the loop's waits are static code and count on it)
    ClItem i0 = next_item(), i1 = next_item(), i2 = i1, i3 = i1;
    load(b0, i0);
    __builtin_amdgcn_sched_barrier(0);
    load(b1, i1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NBUF == 4) {
        i2 = next_item();
        i3 = next_item();
        load(b2, i2);
        __builtin_amdgcn_sched_barrier(0);
        load(b3, i3);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- pre ops. LQ lanes cover a row (16-byte loads), the RG row groups take every RG-th row; the row groups' sums are
    // added in order: one fixed order per vector.
    {
        const int c4 = tid % LQ, rg = tid / LQ;
        for (int o0 = 0; o0 < npre; o0 += CL_PRE_AHEAD) {
            f32x4 acc[CL_PRE_AHEAD];
            const float *src[CL_PRE_AHEAD];
            int nr[CL_PRE_AHEAD];
#pragma unroll
            for (int q = 0; q < CL_PRE_AHEAD; ++q) {
                const int base = CL_PRE_W0 + 8 * (o0 + q < npre ? o0 + q : o0);
                const int kind = word(base), row = word(base + 1);
                src[q] = (kind == 3 ? ua.parts : ua.VT) + (long long)row * D + 4 * c4;
                nr[q] = o0 + q < npre ? word(base + 2) : 0;
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
                        const int rr = r0 + k * RG;
                        v[q][k] = gload4(src[q] + (long long)(rr < nr[q] ? rr : 0) * D);
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
            const int n = npre - o0 < CL_PRE_AHEAD ? npre - o0 : CL_PRE_AHEAD;
            for (int idx = tid; idx < n * D; idx += 256) {
                const int q = idx / D, e = idx - q * D;
                const int base = CL_PRE_W0 + 8 * (o0 + q);
                float s = red[(q * RG) * D + e];
                for (int g = 1; g < RG; ++g) s += red[(q * RG + g) * D + e];
                vec[blk[base + 3] * D + e] = s;
                const int out_vec = blk[base + 4];
                if (out_vec >= 0) ua.VT[(long long)out_vec * D + e] = s;
            }
            __syncthreads();
        }
    }
    CL_STAMP(3)
    // ---- BWD ops as ONE stream of items: (op, 64-row chunk, term), terms innermost; records from LDS, wave-uniform
    float acc[CL_ACCS][4];
#pragma unroll
    for (int a = 0; a < CL_ACCS; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][q] = 0.f;
    auto compute = [&](const ClBuf<D> &b, const ClItem &it) {
        const int chunk = it.meta & 255, nuses = (it.meta >> 8) & 255;
#pragma unroll
        for (int u = 0; u < CL_USES; ++u) {
            if (u >= nuses) break;                               // (uniform)
            const int w = it.use[u];
            const int in_slot = w & 31, a_idx = (w >> 5) & 3, flags = (w >> 7) & 3, out_slot = (w >> 9) & 31, mask_slot = ((w >> 14) & 63) - 1;
            const float *x = vec + in_slot * D;
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CJ; ++c) {
                const f32x4 sv = *reinterpret_cast<const f32x4 *>(x + 4 * l + 64 * c);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) part[q] += b.w[q][c][e] * sv[e];
            }
            // (the accumulator set is a run-time number: a uniform branch per set instead of indexed registers)
#pragma unroll
            for (int a = 0; a < CL_ACCS; ++a) {
                if (a != a_idx) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[a][q] = (flags & CLI_FIRST) ? part[q] : acc[a][q] + part[q];
                if (flags & CLI_LAST) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float y = chain_sum16(acc[a][q]);
                        if (l == 0) {
                            const int i = chunk * 64 + r + 16 * q;
                            const float m = mask_slot >= 0 ? vec[mask_slot * D + i] : 1.f;
                            vec[out_slot * D + i] = m > 0.f ? y : 0.f;
                        }
                    }
                }
            }
        }
        if ((it.meta >> 16) & 1) __syncthreads();               // the level's vectors are whole: the next level may read them
    };
    // (no exit inside a trip: the host pads the item list to whole trips with items that do nothing -- with a `break`
    // behind every step hipcc routed one exit path through the loop header, where the buffer refilled last is the one
    // consumed next, and its static wait became vmcnt(0) on every trip)
    for (int k = 0; k < nitems; k += 4) {
#define CL_STEP(B, I)                                 \
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
        } else {
            CL_STEP(b0, i0)
            CL_STEP(b1, i1)
        }
#undef CL_STEP
    }
    __syncthreads();
    CL_STAMP(4)
    // ---- the BWD ops' vectors the step's reduction reads: VT rows (rank-1 terms) / rows of `parts` (bias, mode rows)
    {
        const int nout = word(4);
        for (int k = 0; k < nout; ++k) {
            const int base = CL_OUT_W0 + 4 * k;
            const int out_slot = word(base), out_vec = word(base + 1), out_part = word(base + 2);
            if (tid < D) {
                const float v = vec[out_slot * D + tid];
                if (out_vec >= 0) ua.VT[(long long)out_vec * D + tid] = v;
                if (out_part >= 0) ua.parts[(long long)out_part * D + tid] = v;
            }
        }
    }
    // ---- R1: gradient matrices made of rank-1 terms only, out[i][j] = sum_t u_t[i] v_t[j] (u, v: LDS slots)
    for (int k = 0; k < nr1; ++k) {
        const int base = CL_R1_W0 + 12 * k;
        const int layer = word(base), rel = word(base + 1), nterms = word(base + 2);
        float *out = rel >= 0 ? pick_grad(gp.basis, layer) : pick_grad(gp.root, layer);
        if (!out) continue;
        if (rel >= 0) out += (long long)rel * D * D;
        // (the slots are wave-uniform words read ONCE: read inside the loop they were two dependent LDS round trips per term
        // and float4 -- 2.9 us per matrix)
        int us[UOP_MAX_TERMS], vs[UOP_MAX_TERMS];
#pragma unroll
        for (int t = 0; t < UOP_MAX_TERMS; ++t) {
            us[t] = word(base + 4 + (t < nterms ? t : 0));
            vs[t] = word(base + 8 + (t < nterms ? t : 0));
        }
        for (int f = tid; f < D * LQ; f += 256) {
            const int i = f / LQ, c4 = f - i * LQ;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < UOP_MAX_TERMS; ++t)
                if (t < nterms) a += vec[us[t] * D + i] * *reinterpret_cast<const f32x4 *>(vec + vs[t] * D + 4 * c4);
            f32x4 *dst = reinterpret_cast<f32x4 *>(out + (long long)i * D + 4 * c4);
            if (zeroed) *dst = a;
            else *dst = *dst + a;
        }
    }
}

// workgroup `cb` of the closures of a launch (uniform branch: D = 64 / 128 / 256 in the chain form)
__device__ __forceinline__ void closure_block(int cb, int D, const ClosureArgs &ca, const LayerPtrs &lp, const UArgs &ua,
                                              float *smem, const GradPtrs &gp, int zeroed, long long *stamps = nullptr) {
    const int *gblock = ca.blocks + (long long)cb * CL_BLOCK_WORDS;
    if (D == 64) closure_run<64>(gblock, lp, ua, smem, gp, zeroed, stamps);
    else if (D == 128) closure_run<128>(gblock, lp, ua, smem, gp, zeroed, stamps);
    else closure_run<256>(gblock, lp, ua, smem, gp, zeroed, stamps);
}

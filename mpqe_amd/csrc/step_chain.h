// Graph-block chain kernel of the fused step.
//
// Query graphs never interact (reference data_utils.py:405: the batch is block-diagonal), so everything
// the step does to a block of graphs before the weight gradients -- feature assembly, every message-passing
// level forward, readout + scores + hinge terms, and every level backward -- depends on nothing outside the
// block. One workgroup takes CH_GB = 16 graphs of one batch through that whole chain: node states stay in
// LDS between levels (ping-pong), the only barrier is the workgroup's own, nothing waits for the slowest
// tile of a level or for a launch, and the latency chains of the gather / score phases overlap with the
// MFMA phases of the other workgroup on the CU. HBM sees each state once, on the way out (H and gH of
// every level: the weight-gradient kernel reads them from there).
//
// GEMM shape per (node slot, level): [16 graphs] x [K = (in-edges + 1) * D] x [D columns]. A wave owns
// 16 * NCB columns (NCB column blocks, column = n0 + NCB * j + c, so a lane's NCB columns are adjacent
// in memory), D = 64 * NCB. v_mfma_f32_16x16x4_f32: lane l feeds A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; the 4 floats a lane reads from LDS with one ds_read_b128 are the A values of
// four MFMAs (u = 0..3), MFMA u multiplying k = 16 t + 4 (l >> 4) + u -- the sum over k is re-ordered,
// in a fixed order. The weights never pass through LDS: every W element is used by exactly one wave,
// which loads its slice straight into registers one half-block (64 k) ahead of the MFMAs that use it.
// Included by step.hip after StepDev / BatchDev / LayerPtrs / TablePtrs / pick_layer / table_row.
#pragma once
#include <type_traits>
#include "gemm_core.h"

#define CH_GB 16
#define CH_FIRST 1        // first K-block of a node update: clear the accumulator
#define CH_LAST 2         // last K-block: epilogue
#define CH_LEVEL_END 4    // last K-block of a level: workgroup barrier, swap the LDS buffers
#define CH_RELU 8         // forward epilogue applies ReLU (and records the mask bits)
#define CH_MASK 16        // backward epilogue masks with the bits recorded for this (level, node)
#define CH_NOSTORE 32     // the result stays in LDS: nothing after the chain kernel reads H[L] or gH[0]
#define CH_ADDG 64        // (learned readouts, concat) the epilogue adds what is stored at the update's own rows of H / gH: a
                          // partial sum of the readout's first layer over the levels (forward) / the readout's share of a
                          // level's state gradient (backward, in front of the ReLU mask)
#define CH_NOBIAS 128      // ... forward: no constant vector is added (a partial sum)
#define CH_TSLOT_ON (1 << 20)   // ChainOp.pad: bits 16-19 = the LDS tile slot the update writes (else: its node slot)
#define CH_MASK_LEVELS 4  // ReLU outputs live at levels 1 .. L-1: chains up to L = 5 passes
#define CH_MAX_OPS 80     // forward + backward K-blocks of one batch (5 passes x 7 x 2 = 70; 3 passes + the concat readout: 74)
#define CH_MAX_CV 12      // forward node updates of one batch (3 passes x 4 node slots; the host checks)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const f32x2 __attribute__((address_space(1))) * gvec2_ptr;
__device__ __forceinline__ f32x2 gload2(const float *p) { return *(gvec2_ptr)(p); }

// one K-block of a node update: multiply the LDS rows of node slot `src` by matrix `mat` (relation id, or
// -1 = root) of layer `layer` into the accumulator of node slot `node`; the result is level `level`'s row
// block of H (forward) / gH (backward)
struct ChainOp {
    unsigned char src, node, layer, level;
    int mat, flags;
    int wt_slot;          // backward ops: slot of the matrix's transposed copy in the step workspace (-1: the parameter itself --
                          // a readout Linear's weight [out, in] IS the transposed form); forward, last K-block of a node
                          // update: the update's number in the forward programme (< CH_MAX_CV; readout ops: 0 / 1)
    int aux;              // forward, last K-block of a node update: id of the node's constant vector (bias + the products
                          // of its batch-uniform sources, formed once per batch by the pre-pass), -1 = the layer's bias;
                          // backward, last K-block: first row in `parts` of the node's per-block column sums, -1 = none
    int pad;              // forward ops of a learned readout: 1 + slot of the transposed copy they multiply by (0: the parameter)
};
// one workgroup: graphs [g0, g0 + 16) of `batch`; its forward / backward programmes in the op table
struct ChainRef {
    int batch, g0, fwd_begin, fwd_count, bwd_begin, bwd_count;
    int tb;               // number of this block among all blocks of the step (batch order): its slot in block_terms
    int done;             // merged launch: the `done` counter of its group of graphs (step.hip: DoneMeta)
    // what the first loads of the workgroup need, so that they depend on this record alone (not on the batch's):
    int e0;               // entry (step_touch.h) of anchor slot 0 of graph g0: anchor_off + g0; slot n adds n * B
    int gi0;              // number of graph g0 among the step's graphs: g_off + g0
    int B;
    unsigned meta;        // N | A << 4 | anchor table 0 / 1 / 2 << 8 / 12 / 16 | target table << 20
    int rof;              // learned readout on the chain (MPQE_READOUT_MLP / _TARGETMLP / _CONCAT): forward ops of its two Linear layers, between the
                          // forward and the backward programme (whose first ops are then the readout's backward)
};

template <int NCB>
struct WHalf {
    float v[4][4][NCB];      // [t][u][c]: k = 16 t + 4 kq + u of the half-block, column block c
};
template <int NCB> struct chain_vec;
template <> struct chain_vec<1> { typedef float type; };
template <> struct chain_vec<2> { typedef f32x2 type; };
template <> struct chain_vec<4> { typedef f32x4 type; };
template <int NCB> struct chain_bits { typedef unsigned char type; };      // 4 * NCB mask bits per lane
template <> struct chain_bits<4> { typedef unsigned short type; };

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

template <int NCB>
__device__ __forceinline__ void chain_lload(float (&d)[NCB], const float *p) {
    const typename chain_vec<NCB>::type q = *reinterpret_cast<const typename chain_vec<NCB>::type *>(p);
    if constexpr (NCB == 1) d[0] = q;
    else {
#pragma unroll
        for (int c = 0; c < NCB; ++c) d[c] = q[c];
    }
}
// 1 / sqrt(x) and 1 / x in one instruction (v_rsq_f32 / v_rcp_f32, 1 ulp) where the result only scales a row or a
// gradient: the IEEE sequences hipcc emits for sqrtf and `/` are 10 - 15 VALU instructions each, none of them hidden
// (the scores themselves keep the exact division).
__device__ __forceinline__ unsigned long long chain_gran_load(const unsigned long long *g) {
#ifdef MPQE_EMU
    return *g;
#else
    typedef unsigned long long __attribute__((address_space(1))) * gp_t;
    return __hip_atomic_load((gp_t)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ unsigned chain_count_load(const unsigned *p) {
#ifdef MPQE_EMU
    return *p;
#else
    typedef unsigned __attribute__((address_space(1))) * gu_t;
    return __hip_atomic_load((gu_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// four consecutive {tag, value} granules (step_uniform.h) once all their tags match; bounded
__device__ __forceinline__ f32x4 chain_gran_read4(const unsigned long long *g, unsigned tag, int32_t *err) {
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    for (int spins = 0;; ++spins) {
        unsigned long long x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = chain_gran_load(g + e);
        bool ok = true;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ok = ok && (unsigned)(x[e] >> 32) == tag;
            q[e] = __uint_as_float((unsigned)x[e]);
        }
        if (ok) break;
        if (spins >= (1 << 18)) {           // (never, unless the launch is broken: report, do not hang)
            flag_error(err, MPQE_FLAG_INTERNAL | 0x100);
            break;
        }
    }
    return q;
}
// one granule (bounded like the above)
__device__ __forceinline__ float chain_gran_read1(const unsigned long long *g, unsigned tag, int32_t *err) {
    for (int spins = 0;; ++spins) {
        const unsigned long long x = chain_gran_load(g);
        if ((unsigned)(x >> 32) == tag) return __uint_as_float((unsigned)x);
        if (spins >= (1 << 18)) {
            flag_error(err, MPQE_FLAG_INTERNAL | 0x100);
            return 0.f;
        }
    }
}
__device__ __forceinline__ float chain_rsq(float x) {
#ifdef MPQE_EMU
    return 1.f / sqrtf(x);
#else
    return __builtin_amdgcn_rsqf(x);
#endif
}
__device__ __forceinline__ float chain_rcp(float x) {
#ifdef MPQE_EMU
    return 1.f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
// v + (v of the lane `off` further, rotating inside the row of 16 lanes), off = 8 / 4 / 2 / 1 in turn: the sum over
// a group of 16 lanes in every lane, by the same pairs as the xor butterfly (so bitwise the same result) but as
// four DPP adds instead of four LDS permutes of ~100 cycles each.
template <int OFF>
__device__ __forceinline__ float chain_row_ror(float v) {
#ifdef MPQE_EMU
    return __shfl_xor(v, OFF, 64);      // (the sums are symmetric: xor and rotation pair the same partial sums)
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + OFF, 0xF, 0xF, false));
#endif
}
__device__ __forceinline__ float chain_sum16(float v) {
    v += chain_row_ror<8>(v);
    v += chain_row_ror<4>(v);
    v += chain_row_ror<2>(v);
    v += chain_row_ror<1>(v);
    return v;
}
// ReLU of one output value + its mask bit, and the masking of one gradient value by that bit, at two VALU
// instructions each: the bits of a lane's 4 x NCB values travel through the carry flag (forward: shifted
// in at the bottom, first value ends up highest; backward: shifted out at the top, same order).
// hipcc's own code for `v > 0 ? v : 0` plus `bits |= (v > 0) << n` is 4 - 5 instructions per value, and a
// wave64 VALU instruction is 4 cycles: the node-update epilogue is instruction bound.
__device__ __forceinline__ void chain_relu_push(float &v, unsigned &bits) {
#ifdef MPQE_EMU
    const bool p = v > 0.f;
    v = p ? v : 0.f;
    bits = (bits << 1) | (p ? 1u : 0u);
#else
    // (the value goes in and out through different operands: tied to one register, hipcc copied each value out of
    // the register pair its packed add had produced)
    float o;
    asm volatile("v_cmp_lt_f32 vcc, 0, %2\n\tv_cndmask_b32 %0, 0, %2, vcc\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                 : "=&v"(o), "+v"(bits)
                 : "v"(v)
                 : "vcc");
    v = o;
#endif
}
__device__ __forceinline__ void chain_mask_pop(float &v, unsigned &bits) {
#ifdef MPQE_EMU
    const bool p = (bits >> 31) != 0u;
    bits <<= 1;
    v = p ? v : 0.f;
#else
    float o;
    asm volatile("v_add_co_u32 %1, vcc, %1, %1\n\tv_cndmask_b32 %0, 0, %2, vcc" : "=&v"(o), "+v"(bits) : "v"(v) : "vcc");
    v = o;
#endif
}

// B[k][n] = M[k][n] with M row-major: forward M = W, backward-x M = W^T (a transposed copy made by
// step_prep_kernel, so both directions read whole 128-byte row pieces: 16 lanes x NCB adjacent floats).
// wp = M + (64 h + 4 kq) * D + n0 + NCB * j; one t-step = 16 k.
#ifndef CHAIN_ZERO_C
#define CHAIN_ZERO_C 1  // 0: clear the accumulators after every node update instead of starting the next one from C = 0
#endif
#ifndef CHAIN_DBG
#define CHAIN_DBG 0     // experiments only: 1 = no weight loads in the K loop, 2 = no MFMAs
#endif
#ifndef CHAIN_LATE_DECODE
#define CHAIN_LATE_DECODE 1   // K split form: an op's LDS words become scalars BEHIND the item in front of which they were requested
#endif
#ifndef CHAIN_NOLOAD
#define CHAIN_NOLOAD 0  // experiments only, with CHAIN_DBG = 6 (cycle trace): 1 = no weight loads in the K loop (wrong results)
#endif
template <int NCB>
__device__ __forceinline__ void chain_load_t(WHalf<NCB> &f, const float *wp, int D, int t) {
    if (CHAIN_DBG == 1) return;
#pragma unroll
    for (int u = 0; u < 4; ++u) chain_gload<NCB>(f.v[t][u], wp + (long long)(16 * t + u) * D);
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

// LDS of one workgroup (D = 128: 77 KB -> two workgroups per CU; D = 256: 151 KB)
// NCB: column blocks (of 16) a wave owns; KS: ways the K range of a K-block is split between the waves.
// 4 waves = (D / (16 NCB)) column groups x KS.  D = 64: <1,1>.  D = 128: <4,2> (a wave owns 64 columns and half
// of K: dwordx4 loads, 64 MFMAs per item; the two K-halves meet in the LDS output tile) -- <2,1> is the form
// without the split (MPQE_STEP_NO_KSPLIT).  D = 256: <4,1>.
template <int NCB, int KS, int NW = 4>
struct ChainLds {
    static constexpr int D = 16 * NCB * NW / KS, LDX = D + 4, BUF = 4 * CH_GB * LDX, MT = 64 * NW / KS;
    float xs[2 * BUF];                                   // node states, ping-pong: [node][graph][LDX]
    static constexpr int NCV = CH_MAX_CV;
    // per forward node update: its constant vector (see ChainOp.aux). Dead once the forward levels are done: the
    // column-sum scratch of the score / backward phases (`red`, 256 floats) lives in its first kilobyte.
    float cv[NCV * D];
    typename chain_bits<NCB>::type mbits[CH_MASK_LEVELS * 4 * MT];    // ReLU bits per (level, node, finishing thread)
    const float *rowp[4 * CH_GB + 2 * CH_GB];            // source row of every node row, then +/- targets
    float *gradp[4 * CH_GB];                             // entity-table gradient row of every anchor row
    float nrm[4 * CH_GB];                                // 1 / |v| of the anchor rows
    const float *wp[CH_MAX_OPS];                         // weight matrix of every op of the block's programme
    int opw[CH_MAX_OPS][2];                              // its (src | node << 8 | layer << 16 | level << 24, flags)
    int opp[CH_MAX_OPS];                                 // backward ops: ChainOp.aux (row of the column sums in `parts`); forward: cv slot
    int cvid[NCV];                                       // cv slot -> vector id (-1: the bias of layer cvl, -2: slot unused)
    int cvl[NCV];
    __device__ __forceinline__ float *red() { return cv; }
#if CHAIN_DBG == 6
    long long *trace;                                    // diagnostic build: per-item cycle stamps of one wave
    int trace_n;
#endif
};

// parts[row][:] = sum over the block's graphs i < ng and the node slots in `mask` of the LDS rows of buffer
// `X` (fixed order). Called by the whole workgroup, between barriers of its own.
template <int NCB, int KS, int NW>
__device__ __forceinline__ void chain_colsum(ChainLds<NCB, KS, NW> &S, const float *X, unsigned mask, int ng,
                                             float *__restrict__ dst) {
    constexpr int D = 16 * NCB * NW / KS, LDX = D + 4, NP = 256 / D;
    const int col = threadIdx.x % D, part = threadIdx.x / D;
    if (threadIdx.x < 256) {        // (the first four waves; the others only keep the barriers)
        float s = 0.f;
        for (int n = 0; n < 4; ++n) {
            if (!((mask >> n) & 1u)) continue;
            for (int i = part; i < ng; i += NP) s += X[(n * CH_GB + i) * LDX + col];
        }
        S.red()[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x < D && dst) {
        float t = S.red()[threadIdx.x];
        for (int q = 1; q < NP; ++q) t += S.red()[threadIdx.x + q * D];
        dst[threadIdx.x] = t;
    }
    __syncthreads();
}
// an op as the K loop sees it: wave-uniform, read from LDS (a vector load from HBM here would sit in vmcnt
// behind the weight prefetch and drain it)
struct ChainStep {
    int src, node, tslot, level, flags, part;      // tslot: LDS tile slot of the output (= node, but for the concat readout's scratch)
};

// The K loop of one direction: the block's programme (T half-blocks) as ONE software pipeline across node
// updates and levels. Straight-line on purpose: every load of the pipeline is unconditional (indices are
// clamped to the last half-block, surplus loads touch valid memory and are dropped), because a load issued
// inside a branch makes hipcc's s_waitcnt bookkeeping fall back to vmcnt(0) at the join, which would
// serialise every half-block behind the prefetch just issued for the next one; sched_barriers keep hipcc's
// scheduler from sinking the prefetch loads down to the MFMAs that use them.
template <int NCB, int KS, bool BWD, int NW, bool RO = false>
__device__ __forceinline__ void chain_run(ChainLds<NCB, KS, NW> &S, const int first_op, int T /* items */, int N, int ng,
                                          float *__restrict__ Xrows, long long level_stride, int &cur,
                                          float *parts = nullptr, int blk = 0,
                                          const unsigned long long *cv_gran = nullptr, unsigned cv_tag = 0,
                                          int32_t *cv_err = nullptr) {
    constexpr int D = 16 * NCB * NW / KS, LDX = D + 4, BUF = 4 * CH_GB * LDX;
    constexpr int CW = NW / KS;                 // column groups of 16 NCB columns
    constexpr int IPO = D / KS / 64;            // items (64 k each) per K-block and wave
    constexpr int MT = 64 * NW / KS;            // threads that finish node updates (the last K part's waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int ks = wave / CW;                            // which part of K this wave multiplies
    const int kbase = ks * (D / KS);
    const int colb = (wave % CW) * 16 * NCB + NCB * j;   // this lane's NCB adjacent columns
    const bool finisher = __builtin_amdgcn_readfirstlane(ks) == KS - 1;      // (a scalar: real branches, not exec masks)
    auto get_op = [&](int it) -> ChainStep {
        const int k = first_op + (it < T ? it : T - 1) / IPO;
        const int w = __builtin_amdgcn_readfirstlane(S.opw[k][0]);
        ChainStep o;
        o.src = w & 0xff;
        o.node = (w >> 8) & 0xff;
        o.tslot = RO ? (w >> 16) & 0xff : o.node;
        o.level = (w >> 24) & 0xff;
        o.flags = __builtin_amdgcn_readfirstlane(S.opw[k][1]);
        o.part = __builtin_amdgcn_readfirstlane(S.opp[k]);      // (forward: the node update's cv slot)
        return o;
    };
    // the same in two halves: the op's words are requested from LDS in front of an item and turned into scalars behind it --
    // a readfirstlane right behind its ds_read waits out the LDS round trip (~130 cycles per item: the ops of one wave per
    // SIMD hide nothing), behind the item's 64 MFMAs the words have long landed
    struct RawOp {
        int w0, w1, w2;
    };
    auto get_raw = [&](int it) -> RawOp {
        const int k = first_op + (it < T ? it : T - 1) / IPO;
        return RawOp{S.opw[k][0], S.opw[k][1], S.opp[k]};
    };
    auto decode = [&](const RawOp &r) -> ChainStep {
        const int w = __builtin_amdgcn_readfirstlane(r.w0);
        ChainStep o;
        o.src = w & 0xff;
        o.node = (w >> 8) & 0xff;
        o.tslot = RO ? (w >> 16) & 0xff : o.node;
        o.level = (w >> 24) & 0xff;
        o.flags = __builtin_amdgcn_readfirstlane(r.w1);
        o.part = __builtin_amdgcn_readfirstlane(r.w2);
        return o;
    };
    // matrix pointers come from LDS (filled in phase A1): no scalar memory round trip, no branch, per item
    auto wptr = [&](int it) -> const float * {
        const int itc = it < T ? it : T - 1;
        const int h = itc % IPO;
        const float *W = S.wp[first_op + itc / IPO];
        return W + (long long)(kbase + 64 * h + 4 * kq) * D + colb;
    };
    f32x4 acc[NCB];
#if CHAIN_DBG == 6
    int trace_i = BWD ? 2048 : 0;
#endif
    // one half-block: 4 t-steps of NCB x 4 MFMAs; the weights of the half-block two items ahead are loaded
    // into `fn` (from `wn`) on the way
    float bs[NCB];          // backward: column sums of a node's gradient rows over the block's graphs
#pragma unroll
    for (int c = 0; c < NCB; ++c) bs[c] = 0.f;
    // K split: the finished rows of a node update go to HBM (H / gH, for the weight-gradient launch) from the waves of
    // the FIRST K part: they would otherwise idle until the finishing waves reach the next hand-off, and the stores
    // (plus their address arithmetic: VALU time a wave cannot hide behind its own MFMAs) leave the finishing waves'
    // path. They copy the node's LDS tile one barrier after it was written: behind the next hand-off / level barrier.
    // In the backward direction the same waves also form the rows' column sums: one row of `parts` per (level, node
    // slot) and block -- summed over blocks they are the bias / variable-row gradients and what the batch-uniform
    // part of the backward pass runs on (step.hip: uniform node states).
    const float *pend_tile = nullptr;
    float *pend_out = nullptr;
    bool pend_on = false, pend_store = false;
    int pend_part = -1;             // >= 0: first row in `parts` of this node's column sums (row = pend_part + blk)
    auto flush_rows = [&]() {
        float l[4][NCB];
#pragma unroll
        for (int r = 0; r < 4; ++r) chain_lload<NCB>(l[r], pend_tile + r * LDX);
        if (BWD) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) bs[c] = 0.f;
        }
        if (ng == CH_GB) {
            if (pend_store) {
#pragma unroll
                for (int r = 0; r < 4; ++r) chain_store<NCB>(pend_out + (long long)r * N * D, l[r]);
            }
            if (BWD) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NCB; ++c) bs[c] += l[r][c];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * kq + r < ng) {
                    if (pend_store) chain_store<NCB>(pend_out + (long long)r * N * D, l[r]);
                    if (BWD) {
#pragma unroll
                        for (int c = 0; c < NCB; ++c) bs[c] += l[r][c];
                    }
                }
        }
        if (BWD && pend_part >= 0) {
            float t[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                t[c] = bs[c] + __shfl_xor(bs[c], 16, 64);
                t[c] += __shfl_xor(t[c], 32, 64);
            }
            if (kq == 0) chain_store<NCB>(parts + (long long)(pend_part + blk) * D + colb, t);
        }
        pend_on = false;
    };
    // (f and fn may be ONE buffer: every group of NCB MFMAs is then followed by the load that refills the registers
    // it has just read with the next item's weights)
    auto item = [&](const ChainStep &op, int it, WHalf<NCB> &f, WHalf<NCB> &fn, const float *wn, auto inplace_tag) {
        constexpr bool INPLACE = decltype(inplace_tag)::value;
        const int h = it % IPO;
#if CHAIN_DBG == 6
#define CHAIN_TRACE(tag)                                                                        \
    if (S.trace && (threadIdx.x & 127) == 0 && S.trace_n < 4000) {                              \
        const long long tt_ = (long long)__builtin_amdgcn_s_memtime();                          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                     \
        S.trace[(threadIdx.x >> 7) * 4096 + S.trace_n * 0 + trace_i] = tt_ * 8 + (tag);         \
        ++trace_i;                                                                              \
    }
        CHAIN_TRACE(0)
#else
#define CHAIN_TRACE(tag)
#endif
        // all four A fragments of the item up front: one exposed LDS round trip per item
        const float *xp = S.xs + cur * BUF + (op.src * CH_GB + j) * LDX + kbase + 64 * h + 4 * kq;
        // (CH_ADDG: the rows to add are requested here, by the waves that will finish the update, and land under the MFMAs.
        // L1-bypassing loads: a row may have been read -- and cached -- at an earlier level, before its last writer)
        unsigned long long ga[RO ? 4 : 1][RO ? (NCB + 1) / 2 : 1];
        if constexpr (RO) {
            if (finisher && h == IPO - 1 && (op.flags & CH_LAST) && (op.flags & CH_ADDG)) {
                const float *Xadd = Xrows + (long long)op.level * level_stride + ((long long)(4 * kq) * N + op.node) * D + colb;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c2 = 0; c2 < (NCB + 1) / 2; ++c2) {
                        const bool in = 4 * kq + r < ng;
                        if constexpr (NCB % 2 == 0)
                            ga[r][c2] = in ? chain_gran_load(reinterpret_cast<const unsigned long long *>(Xadd + (long long)r * N * D) + c2) : 0ull;
                        else        // (D = 64: one column per lane, a 4-byte piece)
                            ga[r][c2] = in ? (unsigned long long)chain_count_load(reinterpret_cast<const unsigned *>(Xadd + (long long)r * N * D)) : 0ull;
                    }
            }
        }
        f32x4 av[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) av[t] = *reinterpret_cast<const f32x4 *>(xp + 16 * t);
        // ONE weight load (and its address arithmetic) between every NCB MFMAs: an MFMA occupies the matrix pipe
        // for 32 cycles but the issue port only for 8, so what follows it issues in its shadow -- a burst of
        // loads between groups of 8 MFMAs instead leaves the pipe idle for the length of the burst
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 a = av[t];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (CHAIN_DBG != 1 && !CHAIN_NOLOAD && !INPLACE) chain_gload<NCB>(fn.v[t][u], wn + (long long)(16 * t + u) * D);
                __builtin_amdgcn_sched_barrier(0);
#if CHAIN_DBG == 2
#pragma unroll
                for (int c = 0; c < NCB; ++c) asm volatile("" ::"v"(f.v[t][u][c]), "v"(a[u]));
#else
                if (CHAIN_ZERO_C && t == 0 && u == 0 && h == 0 && (op.flags & CH_FIRST)) {
                    // a node update's first MFMAs take C = 0 instead of accumulators cleared by 4 NCB v_mov
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], f.v[t][u][c], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                } else {
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], f.v[t][u][c], acc[c], 0, 0, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (CHAIN_DBG != 1 && !CHAIN_NOLOAD && INPLACE) {
                    chain_gload<NCB>(f.v[t][u], wn + (long long)(16 * t + u) * D);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        CHAIN_TRACE(1)
        if (h != IPO - 1) return;
        if (op.flags & CH_LAST) {
            static_assert(KS <= 2, "the hand-off below is written for two K parts");
            // K split: the first K part's waves park their partial sums in the node's LDS output tile (nobody
            // reads it before the level ends); the last part's waves pick them up behind the barrier and finish
            float *Xn = S.xs + (cur ^ 1) * BUF;
            float *tile = Xn + (op.tslot * CH_GB + 4 * kq) * LDX + colb;    // this lane's 4 rows x NCB columns
            if (KS > 1 && !finisher) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv[NCB];
#pragma unroll
                    for (int c = 0; c < NCB; ++c) pv[c] = acc[c][r];
                    chain_store<NCB>(tile + r * LDX, pv);
                }
            }
            float v[4][NCB];
            float bv[NCB];
            if (finisher) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NCB; ++c) v[r][c] = acc[c][r];
                if (!BWD && KS == 1) chain_lload<NCB>(bv, S.cv + op.part * D + colb);
            }
            if (KS > 1) __syncthreads();
            // (K split: the constant is read BEHIND the hand-off barrier -- the other K part's waves may have fetched it
            // into LDS only just before they arrived there, see below)
            if (KS > 1 && finisher && !BWD) chain_lload<NCB>(bv, S.cv + op.part * D + colb);
            if (KS > 1 && !finisher) {
                if (pend_on) flush_rows();
                pend_store = !(op.flags & CH_NOSTORE);
                pend_part = op.part;
                pend_on = pend_store || (BWD && pend_part >= 0);
                pend_tile = tile;
                pend_out = Xrows + (long long)op.level * level_stride + ((long long)(4 * kq) * N + op.node) * D + colb;
                if (!BWD && cv_gran) {
                    // K split, forward: the constant vector of the NEXT node update (a pre-pass vector of this launch:
                    // bias + the products of the node's batch-uniform sources) is fetched HERE, by the waves that do not
                    // finish node updates, one update ahead of its use -- the gather phase waits only for the FIRST
                    // update's constant (the later levels' are published microseconds later: waiting for all of them in
                    // the gather phase held every workgroup until ~8 us into the launch).
                    constexpr int NCVc = ChainLds<NCB, KS, NW>::NCV;
                    const int ns = op.part + 1;
                    if (ns < NCVc && (int)threadIdx.x < D) {
                        const int id = S.cvid[ns];
                        if (id >= 0) S.cv[ns * D + threadIdx.x] = chain_gran_read1(cv_gran + (long long)id * D + threadIdx.x, cv_tag, cv_err);
                    }
                }
            }
            CHAIN_TRACE(2)
            // MODE 0: plain, 1: ReLU + record the mask bits, 2: mask by the recorded bits. One straight-line
            // body per mode (a shared tail makes hipcc copy the 16 values around the inline asm)
            auto finish = [&](auto mode_tag) {
                constexpr int MODE = decltype(mode_tag)::value;
                if (KS > 1) {
                    float l[4][NCB];
#pragma unroll
                    for (int r = 0; r < 4; ++r) chain_lload<NCB>(l[r], tile + r * LDX);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < NCB; ++c) v[r][c] += l[r][c];
                }
                CHAIN_TRACE(4)
                const int mslot = ((op.level - 1) * 4 + op.node) * MT + (threadIdx.x & (MT - 1));   // levels 1 .. L-1
                if (!BWD && !(RO && (op.flags & CH_NOBIAS))) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < NCB; ++c) v[r][c] += bv[c];
                }
                if constexpr (RO) {
                    if (op.flags & CH_ADDG) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int c = 0; c < NCB; ++c)
                                v[r][c] += __uint_as_float((unsigned)(ga[r][c / 2] >> (32 * (c & 1))));
                    }
                }
                if (MODE == 1) {
                    unsigned bits = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < NCB; ++c) chain_relu_push(v[r][c], bits);
                    S.mbits[mslot] = (typename chain_bits<NCB>::type)bits;
                } else if (MODE == 2) {
                    unsigned bits = (unsigned)S.mbits[mslot] << (32 - 4 * NCB);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < NCB; ++c) chain_mask_pop(v[r][c], bits);
                }
                CHAIN_TRACE(5)
#pragma unroll
                for (int r = 0; r < 4; ++r) chain_store<NCB>(tile + r * LDX, v[r]);
                CHAIN_TRACE(6)
                // the node's column sums over the block's graphs (one row of `parts`)
                const bool var_row = KS == 1 && BWD && op.part >= 0;
                if (var_row) {
#pragma unroll
                    for (int c = 0; c < NCB; ++c) bs[c] = 0.f;
                }
                float *Xout = Xrows + (long long)op.level * level_stride + ((long long)(4 * kq) * N + op.node) * D + colb;
                if (ng == CH_GB) {          // (all but a batch's last block)
                    if (KS == 1 && !(op.flags & CH_NOSTORE)) {      // (K split: the other K part's waves copy the rows out)
#pragma unroll
                        for (int r = 0; r < 4; ++r) chain_store<NCB>(Xout + (long long)r * N * D, v[r]);
                    }
                    if (BWD && KS == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int c = 0; c < NCB; ++c) bs[c] += v[r][c];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (4 * kq + r < ng) {
                            if (KS == 1 && !(op.flags & CH_NOSTORE)) chain_store<NCB>(Xout + (long long)r * N * D, v[r]);
                            if (BWD && KS == 1) {
#pragma unroll
                                for (int c = 0; c < NCB; ++c) bs[c] += v[r][c];
                            }
                        }
                    }
                }
                CHAIN_TRACE(7)
                if (var_row) {
                    float t[NCB];
#pragma unroll
                    for (int c = 0; c < NCB; ++c) {
                        t[c] = bs[c] + __shfl_xor(bs[c], 16, 64);
                        t[c] += __shfl_xor(t[c], 32, 64);
                    }
                    if (kq == 0) chain_store<NCB>(parts + (long long)(op.part + blk) * D + colb, t);
                }
            };
            if (finisher) {
                if (!BWD && (op.flags & CH_RELU)) finish(std::integral_constant<int, 1>());
                else if (BWD && (op.flags & CH_MASK)) finish(std::integral_constant<int, 2>());
                else finish(std::integral_constant<int, 0>());
            }
            // the next node update starts from zero (cleared here, inside the uniform branch, rather than by a
            // select at the top of every item: that select made hipcc park the accumulators in registers of
            // a load buffer whose loads were still in flight, and wait for them)
            if (!CHAIN_ZERO_C) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        CHAIN_TRACE(3)
        if (op.flags & CH_LEVEL_END) {       // uniform over the workgroup: every wave runs the same programme
            __syncthreads();
            cur ^= 1;
            if (KS > 1 && !finisher) {
                if (pend_on) flush_rows();     // (the level's last node update)
            }
        }
    };
    if (T <= 0) return;
#pragma unroll
    for (int c = 0; c < NCB; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Register buffers for the weights: prefetch distance PF half-blocks (PF + 1 buffers). With ~36 loads in
    // flight per wave (PF = 2) the loads of a lone workgroup on a CU return slower than the MFMAs consume them
    // (L2 hit rate ~75 %, the rest comes from the Infinity Cache): D <= 128 has the registers for PF = 3.
    ChainStep oc = get_op(0), on = get_op(1);       // ops of items it, it + 1
    int it = 0;
    if constexpr (KS == 2 && CHAIN_DBG != 5) {
        // 64 registers of weights, ONE buffer refilled in place: a register is reloaded with the next item's value
        // right behind the MFMAs that read it, so every load has exactly one item (64 MFMAs, 2048 cycles) to land,
        // no register of the buffer is ever dead (hipcc put temporaries into the dead registers of a second
        // buffer and then waited for the loads refilling it), and the K loop is one item long
        WHalf<NCB> f0;
#pragma unroll
        for (int t = 0; t < 4; ++t) chain_load_t<NCB>(f0, wptr(0), D, t);
        __builtin_amdgcn_sched_barrier(0);
        // (the matrix pointer of the next item's prefetch is read from LDS one item early, like the ops)
        const float *wn1 = wptr(1);
        while (true) {
#if CHAIN_LATE_DECODE
            const RawOp r2_ = get_raw(it + 2);
            const float *w2_ = wptr(it + 2);
            item(oc, it, f0, f0, wn1, std::true_type());
            if (++it >= T) break;
            oc = on;
            on = decode(r2_);
            wn1 = w2_;
#else
            const ChainStep o2_ = get_op(it + 2);
            const float *w2_ = wptr(it + 2);
            item(oc, it, f0, f0, wn1, std::true_type());
            if (++it >= T) break;
            oc = on;
            on = o2_;
            wn1 = w2_;
#endif
        }
    } else if constexpr (NCB <= 2) {
        WHalf<NCB> f0, f1, f2, f3;
#pragma unroll
        for (int t = 0; t < 4; ++t) chain_load_t<NCB>(f0, wptr(0), D, t);
#pragma unroll
        for (int t = 0; t < 4; ++t) chain_load_t<NCB>(f1, wptr(1), D, t);
#pragma unroll
        for (int t = 0; t < 4; ++t) chain_load_t<NCB>(f2, wptr(2), D, t);
        __builtin_amdgcn_sched_barrier(0);
#define CHAIN_STEP(F, FN)                       \
    {                                           \
        const ChainStep o2_ = get_op(it + 2);   \
        item(oc, it, F, FN, wptr(it + 3), std::false_type());      \
        if (++it >= T) break;                   \
        oc = on;                                \
        on = o2_;                               \
    }
        while (true) {
            CHAIN_STEP(f0, f3)
            CHAIN_STEP(f1, f0)
            CHAIN_STEP(f2, f1)
            CHAIN_STEP(f3, f2)
        }
#undef CHAIN_STEP
    } else {
        WHalf<NCB> f0, f1, f2;
#pragma unroll
        for (int t = 0; t < 4; ++t) chain_load_t<NCB>(f0, wptr(0), D, t);
#pragma unroll
        for (int t = 0; t < 4; ++t) chain_load_t<NCB>(f1, wptr(1), D, t);
        __builtin_amdgcn_sched_barrier(0);
#define CHAIN_STEP(F, FN)                       \
    {                                           \
        const ChainStep o2_ = get_op(it + 2);   \
        item(oc, it, F, FN, wptr(it + 2), std::false_type());      \
        if (++it >= T) break;                   \
        oc = on;                                \
        on = o2_;                               \
    }
        while (true) {
            CHAIN_STEP(f0, f2)
            CHAIN_STEP(f1, f0)
            CHAIN_STEP(f2, f1)
        }
#undef CHAIN_STEP
    }
}

struct ChainArgs {
    const ChainRef *refs;
    const ChainOp *ops;
    const long long *node_map;
    long long map_len;
    const float *mode_emb;
    long long num_modes;
    const long long *anchor_ids, *targets, *negs;
    float *H, *GH;
    const float *WT;        // transposed copies of the matrices the backward chains multiply by
    const float *VT;        // vector table [vector id][D]: constants / uniform node states of the pre-pass (step.hip)
    unsigned *epoch_f;      // forward hand-off epoch of this packed step: bumped once per chain launch (step.hip)
    float *DG;              // touch plan (step_touch.h): the table-gradient row of entry e (anchors | + targets | - targets)
                            // is stored to DG[e] instead of added atomically; NULL = fp32 atomics into the tables
    const int *erow;        // pack-time touch plan: row of entry e in its table (-1 bad id, -2 resolve here); NULL: resolve here
    long long Manchor, Gtot;
    float *parts;
    float *block_terms;     // [blocks of the step]: sum of the block's hinge terms (the loss reduction reads these)
    long long level_stride;
    float margin, eps;
    float *s_pos, *s_neg, *terms;
    float *q_out;           // != NULL: the query embeddings [graphs of the step, D] (mpqe_step_extra_t.query_out)
    int32_t *err;
    int backward;
    long long *stamps;      // diagnostics (mpqe_debug_chain_stamps): 8 words per workgroup, or NULL
    // the launch also holds the step's prologue work (step.hip: step_chain_kernel): `cb` = this workgroup's number among
    // the chain workgroups, `nchain` = how many there are
    int cb, nchain;
    const unsigned long long *cv_gran;   // granules of the pre-pass' vectors (this launch produces them): NULL = read VT
    const unsigned *epoch_b;             // backward epoch of the packed step (bumped by the reduction launch)
    const unsigned *wt_count;            // transposed-copy workgroups finished, ever (this launch adds wt_blocks)
    int wt_blocks;
    unsigned *done;                      // merged launch: counters of finished chain workgroups (step.hip: DoneMeta); NULL:
                                         // the weight gradients and the post-pass are a later launch
    unsigned *arrive;                    // ... and of chain workgroups whose stores have reached the L2
    const int *done_inc;                 // chain workgroups per counter and step
    int wt_early;                        // != 0: ops of the FORWARD levels read transposed copies too (concat readout): wait for them first
    int ro;                              // != 0: a learned readout's Linear layers run on the chain (virtual layers ro_layer, + 1)
    int ro_layer, ro_scatter;            // MPQE_SCATTER_* of the reduction over a graph's rows
};

// phase time stamps of a workgroup: the 100 MHz wall clock is one time base for the whole device, so the
// stamps of different workgroups line up into a timeline (tools/chain_timeline.py)
__device__ __forceinline__ void chain_stamp(const ChainArgs &ca, int slot) {
#ifndef MPQE_EMU
    if (ca.stamps && threadIdx.x == 0) {
        ca.stamps[(long long)ca.cb * 8 + slot] = (long long)wall_clock64();
        // shader-clock ticks at the first and the last stamp (beyond the 8 words of every block: second half
        // of the buffer): ticks / wall time = the clock the CU really ran at
        if (slot == 0 || slot == 6) {
            const long long t = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            ca.stamps[((long long)ca.nchain + ca.cb) * 8 + (slot == 0 ? 0 : 1)] = t;
        }
    }
#endif
}
__device__ __forceinline__ void chain_stamp_where(const ChainArgs &ca, int batch, int fwd_ops) {
#ifndef MPQE_EMU
    if (ca.stamps && threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);         // XCC_ID[3:0]
        ca.stamps[(long long)ca.cb * 8 + 7] = (long long)hw | ((long long)xcc << 32) | ((long long)batch << 40) |
                                                   ((long long)fwd_ops << 48);
    }
#endif
}

// NW = 8: the workgroup has eight waves, two per SIMD. Only the K loops use all of them (a wave of each K part on
// every SIMD, so the node-update epilogue of one -- VALU work, which does not overlap with the MFMAs of its own
// wave -- runs under the MFMAs of the other); the row-major phases stay with the first four waves (`four`).
template <int NCB, int KS, int NW, bool RO = false>
__device__ __forceinline__ void chain_block(const StepDev *__restrict__ sd, const LayerPtrs &lp, const TablePtrs &tabs,
                                            const ChainArgs &ca, ChainLds<NCB, KS, NW> &S) {
    constexpr int D = 16 * NCB * NW / KS, LDX = D + 4, BUF = 4 * CH_GB * LDX;
    constexpr int DB = D / 64;                            // float4 passes of 256 threads over one node's 16 rows
    constexpr int IPO = D / KS / 64;                      // K-loop items per K-block and wave
    constexpr int LPR = D / 4;                            // lanes that share one row in the row-major phases
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool four = NW == 4 || tid < 256;
    const ChainRef ref = ca.refs[ca.cb];
    if (ref.batch < 0) return;                            // a hole of the placement grid (uniform)
    const BatchDev &b = sd->b[ref.batch];
    const int N = (int)(ref.meta & 15u), A = (int)((ref.meta >> 4) & 15u), L = b.L, g0 = ref.g0;
    const int ng = ref.B - g0 < CH_GB ? ref.B - g0 : CH_GB;
    const int nrows = ng * N;
    const long long row0 = b.row_off + (long long)g0 * N;
    const long long gi0 = b.g_off + g0;

    chain_stamp(ca, 0);
    chain_stamp_where(ca, ref.batch, ref.fwd_count);
#if CHAIN_DBG == 6
#ifndef CHAIN_TRACE_BATCH
#define CHAIN_TRACE_BATCH 0
#endif
    if (threadIdx.x == 0) {      // trace ONE block (the first of batch CHAIN_TRACE_BATCH): words [2 G * 8 ...) of the stamp buffer
        S.trace = (ca.stamps && ref.batch == CHAIN_TRACE_BATCH && g0 == 0) ? ca.stamps + (long long)ca.nchain * 16 : nullptr;
        S.trace_n = 0;
    }
#endif
#if CHAIN_DBG == 4      // experiment: de-synchronise the blocks of an XCD (they read the same matrices in lockstep)
    for (int q = 0; q < (int)((blockIdx.x / 8) % 4) * 6; ++q) __builtin_amdgcn_s_sleep(127);
#endif
    // ---- phase A1: where every row comes from (threads 0 .. 95: node rows, + targets, - targets). The entry number
    // comes from the workgroup's own record; with a pack-time touch plan the id -> LUT -> row hops were done there (erow:
    // ONE round trip between the record and the row gather), otherwise they are done here (two); the programme's ops and
    // the batch record travel beside them.
    const int rof = RO ? ref.rof : 0, nfwd = ref.fwd_count + rof;
    const int nops = nfwd + ref.bwd_count;
    ChainOp op;
    if (tid < nops) op = ca.ops[ref.fwd_begin + tid];
    if (tid < 4 * CH_GB + 2 * CH_GB) {
        const float *src = nullptr;
        float *gdst = nullptr;
        const int rN = (int)(ref.meta & 15u), rA = (int)((ref.meta >> 4) & 15u);
        if (tid < 4 * CH_GB) {
            const int i = tid / rN, n = tid - i * rN;       // row r = i * N + n, as in HBM
            if (tid < nrows) {
                if (n < rA) {
                    const int tab = (int)((ref.meta >> (8 + 4 * n)) & 15u);
                    const long long e = (long long)ref.e0 + (long long)n * ref.B + i;
                    long long row;
                    int er = -2;
                    if (ca.erow) er = ca.erow[e];
                    if (er == -2) row = table_row(ca.node_map, ca.map_len, ca.anchor_ids[e], tabs.rows[tab], ca.err);
                    else {
                        row = er;
                        if (er < 0) flag_error(ca.err, MPQE_FLAG_BAD_NODE_ID);
                    }
                    if (row >= 0) {
                        src = tabs.table[tab] + row * D;
                        if (tabs.grad[tab]) gdst = tabs.grad[tab] + row * D;
                    }
                } else {
                    const long long m = b.var_id[n - rA];
                    if (m < 0 || m >= ca.num_modes) flag_error(ca.err, MPQE_FLAG_BAD_NODE_ID);
                    else src = ca.mode_emb + m * D;
                }
            }
            S.gradp[tid] = gdst;
        } else {
            const int i = (tid - 4 * CH_GB) & (CH_GB - 1);
            const bool is_neg = tid >= 5 * CH_GB;
            if (i < ng) {
                const int tab = (int)((ref.meta >> 20) & 15u);
                const long long e = ca.Manchor + (is_neg ? ca.Gtot : 0) + ref.gi0 + i;
                long long row;
                int er = -2;
                if (ca.erow) er = ca.erow[e];
                if (er == -2)
                    row = table_row(ca.node_map, ca.map_len, is_neg ? ca.negs[ref.gi0 + i] : ca.targets[ref.gi0 + i],
                                    tabs.rows[tab], ca.err);
                else {
                    row = er;
                    if (er < 0) flag_error(ca.err, MPQE_FLAG_BAD_NODE_ID);
                }
                if (row >= 0) src = tabs.table[tab] + row * D;
            }
        }
        S.rowp[tid] = src;
    }
    constexpr int NCV = ChainLds<NCB, KS, NW>::NCV;
    if (tid < NCV) S.cvid[tid] = -2;
    __syncthreads();
    {   // forward ops then backward ops of this block's programme (the host keeps them adjacent; requested above)
        if (tid < nops) {
            if (tid < nfwd || CHAIN_DBG == 3) {     // (3: timing experiment, wrong results)
                S.wp[tid] = op.mat >= 0 ? pick_layer(lp.basis, op.layer) + (long long)op.mat * D * D
                                        : pick_layer(lp.root, op.layer);
                if (RO && (op.pad & 0xffff) > 0) S.wp[tid] = ca.WT + (long long)((op.pad & 0xffff) - 1) * D * D;
            } else {
                S.wp[tid] = ca.WT + (long long)op.wt_slot * D * D;
                if (RO && op.wt_slot < 0) S.wp[tid] = pick_layer(lp.root, op.layer);
            }
            // (byte 2: the LDS tile slot the update writes)
            S.opw[tid][0] = op.src | (op.node << 8) | (((RO && (op.pad & CH_TSLOT_ON)) ? (op.pad >> 16) & 15 : op.node) << 16) | (op.level << 24);
            S.opw[tid][1] = op.flags;
            S.opp[tid] = tid < nfwd ? op.wt_slot : op.aux;
            if (tid < ref.fwd_count && (op.flags & CH_LAST) && !(op.flags & CH_NOBIAS)) {      // one slot per forward node update
                S.cvid[op.wt_slot] = op.aux;
                S.cvl[op.wt_slot] = op.layer;
            }
        }
    }
    __syncthreads();
    chain_stamp(ca, 1);
    // L2 warm-up of the backward matrices (transposed copies the prologue launch has just written: in no L2 yet).
    // The blocks of a batch that share an XCD walk the same matrices in lockstep, so the first touch of every
    // matrix is a miss for all of them at once, in the middle of a K loop. Each block requests one eighth (by
    // its number in the batch) of the lines of its backward programme's matrices while the latency-bound score
    // phase runs; nothing waits for these loads before the backward K loop's first weights are due.
    // (Measured: backward levels of the slowest blocks 27.7 -> 22.0 us. The same for the forward matrices
    // during the gather bought nothing and cost the gather 1.3 us.)
    float wq[4] = {0.f, 0.f, 0.f, 0.f};
    auto warm = [&](int first, int count) {
        constexpr int SL = D * D / 256;                   // 128-byte lines in one eighth of a matrix
        const int slice = (g0 / CH_GB) & 7;
        const int n = count * SL;
        auto touch = [&](int item) -> float {
            const int o = item / SL, l = item - o * SL;
            return gload1(S.wp[first + o] + (long long)(slice * SL + l) * 32);
        };
#pragma unroll
        for (int k = 0; k < 4; ++k) wq[k] = touch(tid + 256 * k < n ? tid + 256 * k : 0);    // (programmes up to 16 ops)
        for (int item = tid + 1024; item < n; item += 256) wq[0] += touch(item);
    };

    // ---- phase A2: gather the rows, L2-normalise the anchors (reference encoders.py:41-43, no eps), write
    // LDS buffer 0 and H[0]. Thread t moves float4 number t + 256 k, k < N * DB; LPR adjacent lanes share a
    // row. All loads are issued before the first use.
    if (four) {
        // the constants of the forward node updates: bias + (uniform mode) the products of the node's batch-uniform
        // sources, one vector per (level, node) written by the pre-pass -- requested before the row gather, used
        // by the first epilogue at the earliest
        for (int f = tid; f < NCV * (D / 4); f += 256) {
            const int slot = f / (D / 4), c4 = f - slot * (D / 4);
            const int id = S.cvid[slot];
            if (id == -2 || (id >= 0 && ca.cv_gran)) continue;
            const float *src = id >= 0 ? ca.VT + (long long)id * D : pick_layer(lp.bias, S.cvl[slot]);
            *reinterpret_cast<f32x4 *>(S.cv + slot * D + 4 * c4) = src ? gload4(src + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int nk = N * DB;
        f32x4 v[4 * DB];
#pragma unroll
        for (int k = 0; k < 4 * DB; ++k) {
            const int f = tid + 256 * (k < nk ? k : 0);
            const int r = f / (D / 4), c4 = f - r * (D / 4);
            const float *src = S.rowp[r];
            v[k] = src ? gload4(src + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (ca.cv_gran) {
            // constants the pre-pass workgroups of THIS launch produce: read as {tag, value} granules (step_uniform.h),
            // four per thread and attempt, behind the row requests above (whose latency the wait shares)
            const unsigned tag = *ca.epoch_f + 1u;
            // (K split: only the first node update's constant -- the others are fetched one update ahead, chain_run)
            for (int f = tid; f < (KS > 1 ? 1 : NCV) * (D / 4); f += 256) {
                const int slot = f / (D / 4), c4 = f - slot * (D / 4);
                const int id = S.cvid[slot];            // (here: the vector's granule slot)
                if (id < 0) continue;
                *reinterpret_cast<f32x4 *>(S.cv + slot * D + 4 * c4) =
                    chain_gran_read4(ca.cv_gran + (long long)id * D + 4 * c4, tag, ca.err);
            }
        }
        float *H0 = ca.H + row0 * D;
#pragma unroll
        for (int k = 0; k < 4 * DB; ++k) {
            const int f = tid + 256 * k;
            const int r = f / (D / 4), c4 = f - r * (D / 4);
            const int i = r / N, n = r - i * N;
            f32x4 q = v[k];
            float ss = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
#pragma unroll
            for (int off = LPR >> 1; off >= 16; off >>= 1) ss += __shfl_xor(ss, off, 64);
            ss = chain_sum16(ss);
            if (k < nk) {
                const bool anchor = n < A && S.rowp[r] != nullptr;
                if (anchor) {      // x * (1 / |v|): within 2 ulp of the reference's x / |v|
                    const float inv = chain_rsq(ss);
                    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
                    if ((f & (D / 4 - 1)) == 0) S.nrm[r] = inv;
                }
                *reinterpret_cast<f32x4 *>(S.xs + (n * CH_GB + i) * LDX + 4 * c4) = q;
                if (r < nrows) *reinterpret_cast<f32x4 *>(H0 + (long long)r * D + 4 * c4) = q;
            }
        }
    }
    __syncthreads();

    chain_stamp(ca, 2);
    // the + and - target rows of the score phase are requested NOW (16 lanes per graph: lane group g of wave w
    // takes graph 4 w + g, lane s of the group the columns s + 16 cc) and ride through the forward levels in
    // registers: their HBM round trip would otherwise sit between the two K loops
    constexpr int CC = D / 16;
    const int sc_i = four ? 4 * wave + (lane >> 4) : 0, sl = lane & 15;
    const float *pp_ = four ? S.rowp[4 * CH_GB + sc_i] : nullptr, *pn_ = four ? S.rowp[5 * CH_GB + sc_i] : nullptr;
    float tp[CC], tn[CC];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
        tp[cc] = pp_ ? gload1(pp_ + sl + 16 * cc) : 0.f;
        tn[cc] = pn_ ? gload1(pn_ + sl + 16 * cc) : 0.f;
    }
    // (the count of finished transpose workgroups, for the wait in front of the backward levels: requested now)
    unsigned wt_have0 = 0;
    if (tid == 0 && (ca.backward || RO) && ca.wt_count) wt_have0 = chain_count_load(ca.wt_count);
    // The transposed copies the backward levels (and a learned readout's forward) multiply by are written by workgroups of
    // THIS launch, which publish them with an agent-scope release and count themselves in (step.hip: prep_transpose_block).
    // One lane compares the count it requested above (normally already complete) with the target and polls on only if it
    // was not; then the workgroup's barrier. No acquire fence: it would invalidate this CU's L1, and no line of the copies
    // can be in it (or in this XCD's L2) -- nothing reads them before this point in the launch, the warm-up requests come
    // after the barrier, and caches do not survive a launch boundary.
    auto wait_for_copies = [&]() {
        if (tid == 0) {
            const unsigned want = (*ca.epoch_b + 1u) * (unsigned)ca.wt_blocks;
            unsigned have = wt_have0;
            for (int spins = 0; (int)(have - want) < 0; ++spins) {
                if (spins >= (1 << 18)) {
                    flag_error(ca.err, MPQE_FLAG_INTERNAL | 0x200);
                    break;
                }
#ifndef MPQE_EMU
                __builtin_amdgcn_s_sleep(2);
#endif
                have = chain_count_load(ca.wt_count);
            }
        }
        __syncthreads();
    };
    // concat (reference model.py:441-446): the first readout layer's products with the states of levels 1 .. L-1 are ops of
    // the FORWARD levels' programme (step_plan.h: ro_cat) and multiply by transposed column blocks of its weight -- copies
    // of this launch: they must be complete before the first forward level, not only before the readout's own ops. (Until
    // round 5 the wait stood behind the forward levels only: those ops then read whatever the copies' slots held -- the
    // previous step's blocks, i.e. weights one optimiser step old; a first step read unwritten memory.)
    const bool early_copies = RO && ca.wt_early && ca.wt_count && (ref.fwd_count > 0 || rof > 0);
    if (early_copies) wait_for_copies();
    // ---- forward levels
    int cur = 0;
    chain_run<NCB, KS, false, NW, RO>(S, 0, ref.fwd_count * IPO, N, ng, ca.H + row0 * D, ca.level_stride, cur, nullptr, 0,
                                      ca.cv_gran, ca.cv_gran ? *ca.epoch_f + 1u : 0u, ca.err);

    chain_stamp(ca, 3);
    if constexpr (RO) {
        // the readout's biases take the first two slots of the forward constants (dead: every forward level is done)
        if (tid < D) {
            const float *b0 = pick_layer(lp.bias, ca.ro_layer), *b2 = pick_layer(lp.bias, ca.ro_layer + 1);
            S.cv[tid] = b0 ? b0[tid] : 0.f;
            S.cv[D + tid] = b2 ? b2[tid] : 0.f;
        }
    }
    if (!early_copies && ((ca.backward && ref.bwd_count > 0) || (RO && rof > 0)) && ca.wt_count) wait_for_copies();
    else if (RO) __syncthreads();
    if (ca.backward && ref.bwd_count > 0 && four) warm(nfwd, ref.bwd_count);
    // node states that are still batch-uniform at level L (no anchor within L hops: possible when a batch runs fewer
    // passes than its diameter) never went through the K loops: the readout sees the pre-pass' vector in every row
    if (b.uvL[0] >= 0 || b.uvL[1] >= 0 || b.uvL[2] >= 0 || b.uvL[3] >= 0) {      // (uniform over the workgroup)
        float *Xc = S.xs + cur * BUF;
        for (int n = 0; n < N; ++n) {
            const int id = b.uvL[n];
            if (id < 0) continue;
            // (a vector of this launch's pre-pass workgroups: read through its granules, uvL = its granule slot)
            const unsigned tag = *ca.epoch_f + 1u;
            for (int f = tid; f < CH_GB * (D / 4) && four; f += 256) {
                const int i = f / (D / 4), c4 = f - i * (D / 4);
                *reinterpret_cast<f32x4 *>(Xc + (n * CH_GB + i) * LDX + 4 * c4) =
                    chain_gran_read4(ca.cv_gran + (long long)id * D + 4 * c4, tag, ca.err);
            }
        }
        __syncthreads();
    }
    if constexpr (RO) {
        // ---- a learned readout's Linear - ReLU - Linear on every node row (reference model.py:497-515, MLPReadout): two
        // more levels of node updates, one K-block each (the node's own row times W^T: a transposed copy of this launch)
        if (rof > 0)
            chain_run<NCB, KS, false, NW, true>(S, ref.fwd_count, rof * IPO, N, ng, ca.H + row0 * D, ca.level_stride, cur);
    }
    // ---- readout, cosine scores against the + and - target, hinge terms (reference model.py:447-462,
    // 483-485); backward: d hinge -> d cosine -> d readout written over H[L] in LDS (a lane group owns whole
    // graphs) and to gH[L]; target-table gradients through the normalisation. 16 lanes per graph (see above).
    if (four) {
        float *Xc = S.xs + cur * BUF;
        // (a learned readout: the rows the reduction reads are its output level, L + 2)
        float *GL = ca.GH + (long long)(RO ? L + 2 : L) * ca.level_stride + row0 * D;
        const int i = sc_i;
        const bool on = i < ng;
        // (a learned readout ends in torch_scatter's add / mean / max over the graph's rows: the sum / max readouts' code)
        const int readout = RO ? (ca.ro_scatter == MPQE_SCATTER_MAX ? MPQE_READOUT_MAX : MPQE_READOUT_SUM) : sd->readout;
        const float rscale = RO && ca.ro_scatter == MPQE_SCATTER_MEAN ? 1.f / (float)__builtin_popcount(b.live[L + 2]) : 1.f;
        auto gsum = [](float v) { return chain_sum16(v); };
        const float *h = Xc + i * LDX;                    // node n at h + n * CH_GB * LDX
        float q[CC];
        int arg[CC];
        // (a learned readout: the reduction runs over the node slots that HAVE a row -- targetmlp: not the target's)
        const unsigned rom = RO ? b.live[L + 2] : 0u;
        if (RO) {
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {             // branch-free: N <= 4 slots, predicated by the row mask
                const int col = sl + 16 * cc;
                float sum = 0.f, best = 0.f;
                int am = -1;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const bool has = (rom >> n) & 1u;
                    const float hv_ = h[(has ? n : 0) * CH_GB * LDX + col];
                    sum += has ? hv_ : 0.f;
                    const bool gt = has && (am < 0 || hv_ > best);      // (lowest row wins ties: torch_scatter's max)
                    best = gt ? hv_ : best;
                    am = gt ? n : am;
                }
                q[cc] = readout == MPQE_READOUT_SUM ? sum * rscale : best;
                arg[cc] = am < 0 ? 0 : am;
            }
        } else if (readout == MPQE_READOUT_TM) {          // (uniform) the target slot's row is the readout
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                q[cc] = h[A * CH_GB * LDX + sl + 16 * cc];
                arg[cc] = 0;
            }
        } else
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {                 // branch-free: N <= 4 slots, predicated
            const int col = sl + 16 * cc;
            float hn[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) hn[n] = h[(n < N ? n : 0) * CH_GB * LDX + col];
            float sum = hn[0], best = hn[0];
            int am = 0;
#pragma unroll
            for (int n = 1; n < 4; ++n) {
                sum += n < N ? hn[n] : 0.f;
                const bool gt = n < N && hn[n] > best;
                best = gt ? hn[n] : best;
                am = gt ? n : am;
            }
            float tm = hn[0];
#pragma unroll
            for (int n = 1; n < 4; ++n) tm = n == A ? hn[n] : tm;
            q[cc] = readout == MPQE_READOUT_TM ? tm : (readout == MPQE_READOUT_SUM ? sum : best);
            arg[cc] = am;
        }
        if (ca.q_out && on) {         // (uniform per launch: the evaluation form's ragged scoring reads these rows)
            float *qo = ca.q_out + (gi0 + i) * D;
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) qo[sl + 16 * cc] = q[cc];
        }
        float ssp = 0.f, ssn = 0.f;
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            ssp += tp[cc] * tp[cc];
            ssn += tn[cc] * tn[cc];
        }
        ssp = gsum(ssp);
        ssn = gsum(ssn);
        const float ip0 = chain_rsq(ssp), in0 = chain_rsq(ssn);
        float dp = 0.f, dn = 0.f, qq = 0.f, pp = 0.f, nn = 0.f;
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            // the normalised target embeddings (DirectEncoder); an invalid id (flagged) scores as a zero row
            tp[cc] = pp_ ? tp[cc] * ip0 : 0.f;
            tn[cc] = pn_ ? tn[cc] * in0 : 0.f;
            dp += q[cc] * tp[cc];
            dn += q[cc] * tn[cc];
            qq += q[cc] * q[cc];
            pp += tp[cc] * tp[cc];
            nn += tn[cc] * tn[cc];
        }
        dp = gsum(dp);
        dn = gsum(dn);
        qq = gsum(qq);
        pp = gsum(pp);
        nn = gsum(nn);
        const float eps = ca.eps;
        const float rq = sqrtf(qq), rp = sqrtf(pp), rn = sqrtf(nn);
        const float nq = fmaxf(rq, eps), np_ = fmaxf(rp, eps), nn_ = fmaxf(rn, eps);
        const float sp = dp / (nq * np_), sn = dn / (nq * nn_);
        const float hv = ca.margin - (sp - sn);
        const long long gi = gi0 + i;
        const float term = on && hv > 0.f ? hv : 0.f;
        if (on && sl == 0) {
            ca.s_pos[gi] = sp;
            ca.s_neg[gi] = sn;
            ca.terms[gi] = term;
        }
        {   // the wave's four graphs (every lane of a group holds its graph's term), then the four waves, in order
            float t4 = term + __shfl_xor(term, 16, 64);
            t4 += __shfl_xor(t4, 32, 64);
            if (lane == 0) S.red()[wave] = t4;
        }
        if (ca.backward) {
            // d loss / d sp = -w/B on active terms, d/d sn = +w/B   (loss = sum_b w_b mean_b hinge)
            const float act = hv >= 0.f ? b.weight / (float)b.B : 0.f;
            const float gsp = -act, gsn = act;
            const float inv_p = chain_rcp(nq * np_), inv_n = chain_rcp(nq * nn_);
            const float kq_ = rq > eps ? (gsp * sp + gsn * sn) * chain_rcp(nq * nq) : 0.f;
            const float ktp = rp > eps ? sp * chain_rcp(np_ * np_) : 0.f;
            const float ktn = rn > eps ? sn * chain_rcp(nn_ * nn_) : 0.f;
            const unsigned tmask = readout == MPQE_READOUT_SUM ? 0xFu : 1u << A;
            const unsigned liveL = b.live[RO ? L + 2 : L];
            float yg_p = 0.f, yg_n = 0.f;
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                const int col = sl + 16 * cc;
                const float gyp = gsp * (q[cc] * inv_p - ktp * tp[cc]);
                const float gyn = gsn * (q[cc] * inv_n - ktn * tn[cc]);
                yg_p += tp[cc] * gyp;
                yg_n += tn[cc] * gyn;
                float gq = gsp * tp[cc] * inv_p + gsn * tn[cc] * inv_n - kq_ * q[cc];
                if (RO) gq *= rscale;
                const unsigned takes = readout == MPQE_READOUT_MAX ? 1u << arg[cc] : tmask;   // slots that get gq
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const float gv = (takes >> n) & 1u ? gq : 0.f;
                    if (n < N && ((liveL >> n) & 1u)) {     // (rows of pruned slots: nothing reads them)
                        Xc[(n * CH_GB + i) * LDX + col] = gv;
                        if (on) GL[((long long)i * N + n) * D + col] = gv;
                    }
                }
            }
            yg_p = gsum(yg_p);
            yg_n = gsum(yg_n);
            float *gt = tabs.grad[b.target_tab];
            if (ca.DG && on) {    // y = v / |v|:  dv = (g - y (y . g)) / |v|, stored as the entry's row (summed per table row later)
                float *dp_ = ca.DG + (ca.Manchor + ref.gi0 + i) * D, *dn_ = dp_ + ca.Gtot * D;      // entries of the + / - target
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) {
                    const int col = sl + 16 * cc;
                    const float gyp = gsp * (q[cc] * inv_p - ktp * tp[cc]);
                    const float gyn = gsn * (q[cc] * inv_n - ktn * tn[cc]);
                    dp_[col] = pp_ ? (gyp - tp[cc] * yg_p) * ip0 : 0.f;
                    dn_[col] = pn_ ? (gyn - tn[cc] * yg_n) * in0 : 0.f;
                }
            } else if (gt && on) {
                const float *tb = tabs.table[b.target_tab];
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) {
                    const int col = sl + 16 * cc;
                    const float gyp = gsp * (q[cc] * inv_p - ktp * tp[cc]);
                    const float gyn = gsn * (q[cc] * inv_n - ktn * tn[cc]);
                    if (CHAIN_DBG == 8) continue;         // (timing experiment: no table atomics, wrong results)
                    if (pp_) atomicAdd(gt + (pp_ - tb) + col, (gyp - tp[cc] * yg_p) * ip0);
                    if (pn_) atomicAdd(gt + (pn_ - tb) + col, (gyn - tn[cc] * yg_n) * in0);
                }
            }
        }
    }
    __syncthreads();
    // (forward-only step: written through -- the launch's last workgroup sums the terms inside this launch, step.hip:
    // chain_finish; a whole step's reduction launch reads them from the L2 they stay in)
    if (tid == 0) {
        const float term = (S.red()[0] + S.red()[1]) + (S.red()[2] + S.red()[3]);
        if (ca.backward) ca.block_terms[ref.tb] = term;
        else agent_store(ca.block_terms + ref.tb, term);
    }
    if (!ca.backward) {
        chain_stamp(ca, 6);
        return;
    }
    chain_stamp(ca, 4);
    const int blk = g0 / CH_GB;
    // column sums of gH[L] per node slot (the readout's gradient rows): the last pass' bias gradient, and what
    // the uniform part of the backward pass starts from
    for (int n = 0; n < N; ++n) {
        const int pr = b.lpart[n];
        if (pr < 0) continue;           // (uniform over the workgroup)
        chain_colsum<NCB, KS, NW>(S, S.xs + cur * BUF, 1u << n, ng, ca.parts + (long long)(pr + blk) * D);
    }

    // ---- backward levels
    chain_run<NCB, KS, true, NW, RO>(S, nfwd, ref.bwd_count * IPO, N, ng, ca.GH + row0 * D, ca.level_stride, cur,
                                     ca.parts, blk);

    chain_stamp(ca, 5);
    if (ca.done) {
        // Merged launch: the H / gH rows and the column sums of this workgroup are what the weight-gradient tiles and
        // the post-pass of its batch wait for (workgroups of THIS launch, behind the chain workgroups). Every wave drains
        // its stores (they are in the XCD's L2 then), the workgroup meets and one lane counts it in as ARRIVED. The
        // workgroup that completes its group of DONE_GRAPHS graphs -- all on one XCD: a batch is dealt out in chunks of 32
        // workgroups per XCD -- writes the L2 back ONCE for the whole group (release at agent scope; a write-back per
        // workgroup cost 16 us per step: each one walks the whole L2) and publishes the group. The anchor rows below are
        // not part of it.
#ifndef MPQE_EMU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        if (tid == 0) {
            const unsigned inc = (unsigned)ca.done_inc[ref.done];
            const unsigned before = atomicAdd(ca.arrive + ref.done, 1u);
            if ((before + 1u) % inc == 0u) {
#ifndef MPQE_EMU
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                atomicAdd(ca.done + ref.done, inc);
            }
        }
    }
    // ---- anchor rows of gH[0] through the L2 normalisation into the entity-table gradients (fp32 atomics:
    // an entity can occur in several graphs). y comes back from H[0] (this workgroup wrote it; L2). (Requested in front
    // of the backward levels instead -- 32 registers held across their K loop -- the step was 0.5 - 1 us SLOWER, same box.)
    if (four) {
        const float *Xc = S.xs + cur * BUF;
        const float *H0 = ca.H + row0 * D;
        const int nk = N * DB;
        const unsigned live0 = b.live[0];
        f32x4 y[4 * DB];
#pragma unroll
        for (int k = 0; k < 4 * DB; ++k) {
            const int f = tid + 256 * (k < nk ? k : 0);
            const int r = f / (D / 4), c4 = f - r * (D / 4);
            y[k] = gload4(H0 + (long long)(r < nrows ? r : 0) * D + 4 * c4);
        }
#pragma unroll
        for (int k = 0; k < 4 * DB; ++k) {
            const int f = tid + 256 * (k < nk ? k : 0);
            const int r = f / (D / 4), c4 = f - r * (D / 4);
            const int i = r / N, n = r - i * N;
            const bool on = k < nk && r < nrows && n < A && ((live0 >> n) & 1u);
            const f32x4 g = *reinterpret_cast<const f32x4 *>(Xc + (n * CH_GB + i) * LDX + 4 * c4);
            float yg = y[k][0] * g[0] + y[k][1] * g[1] + y[k][2] * g[2] + y[k][3] * g[3];
#pragma unroll
            for (int off = LPR >> 1; off >= 16; off >>= 1) yg += __shfl_xor(yg, off, 64);
            yg = chain_sum16(yg);
            if (ca.DG) {          // the entry's gradient row, one 16-byte store (a pruned or invalid anchor: zeros)
                if (k < nk && r < nrows && n < A) {
                    const bool ok = on && S.rowp[r] != nullptr;
                    const float inv = ok ? S.nrm[r] : 0.f;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ok ? (g[e] - y[k][e] * yg) * inv : 0.f;
                    // (entry of anchor slot n of graph i: the batch's anchor ids are slot-major)
                    *reinterpret_cast<f32x4 *>(ca.DG + ((long long)ref.e0 + (long long)n * ref.B + i) * D + 4 * c4) = o;
                }
                continue;
            }
            float *gd = on ? S.gradp[r] : nullptr;
            if (gd && CHAIN_DBG != 7 && CHAIN_DBG != 8) {
                const float inv = S.nrm[r];
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(gd + 4 * c4 + e, (g[e] - y[k][e] * yg) * inv);
            }
        }
    }
    chain_stamp(ca, 6);
#ifndef MPQE_EMU
    asm volatile("" ::"v"(wq[0]), "v"(wq[1]), "v"(wq[2]), "v"(wq[3]));     // (the warm-up loads have a use: hipcc keeps them)
#else
    (void)wq;
#endif
}

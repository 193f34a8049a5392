// fp32 MFMA tile core for gfx950: one 256-thread workgroup (4 waves, 2x2)
// computes a 64x64 fp32 tile with v_mfma_f32_32x32x2_f32 (exact fp32, one
// rounding per product), K advanced 32 at a time through double-buffered LDS.
//
// Operands come through functors so the same core serves
//   * the layer forward      A = gathered/strided rows of x        B = W[k][n]
//   * the layer backward-x   A = rows of (masked) grad_out         B = W[n][k] (transposed use)
//   * the weight gradient    A = x rows as the K dimension         B = grad rows as K
//
// LDS images (floats), both padded against bank conflicts:
//   R-type  [64 rows][32 k]  row stride 36   (operand contiguous along k in HBM)
//   K-type  [32 k][64 cols]  row stride 68   (operand contiguous along the tile's rows/cols)
//
// MFMA lane map (MI355X guide): lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]. The 32 k of a step are split so that lane-half h owns
// k in [16h, 16h+16): MFMA s multiplies k = s (h=0) and k = 16+s (h=1). The
// sum over k is only re-ordered, which fp32 tolerance covers (and which is a
// fixed order, so results are reproducible run to run).
#pragma once
#include "common.h"

#define GT_BM 64
#define GT_BN 64
#define GT_BK 32
#define GT_LDR 36               // R-type row stride (floats)
#define GT_LDK 68               // K-type row stride (floats)
#define GT_TILE_FLOATS 2304     // max(64*36, 32*68)
#define GT_SMEM_FLOATS (4 * GT_TILE_FLOATS)   // A,B x 2 buffers = 36,864 B

// Operand load modes (compile-time):
//   LD_SCALAR  odd dims / unaligned rows: element-wise guarded loads, zeros outside the operand
//   LD_PRED    16-byte loads; a slot outside the operand reads a safe address and is flagged !ok,
//              the pipeline stores zeros for it (no branch between a load and its use)
//   LD_FAST    16-byte loads, nothing to check: the caller guarantees every address it forms is
//              inside the tensor (dims multiples of the tile, out-of-range rows clamped to a
//              valid row whose result is discarded). The loop body is then straight-line code and
//              hipcc keeps two tiles of loads in flight behind counted s_waitcnt.
enum { LD_SCALAR = 0, LD_PRED = 1, LD_FAST = 2 };

// Every operand lives in HBM. Pointers that reach a kernel inside a by-value struct or through a
// select lose their address space and hipcc would emit FLAT loads for them (slower, and they tie
// up both the vm and the lgkm counters); loading through an explicit global-address-space pointer
// keeps them global_load_dwordx4.
typedef const f32x4 __attribute__((address_space(1))) * gvec4_ptr;
typedef const float __attribute__((address_space(1))) * gfloat_ptr;
__device__ __forceinline__ f32x4 gload4(const float *p) { return *(gvec4_ptr)(p); }
__device__ __forceinline__ float gload1(const float *p) { return *(gfloat_ptr)(p); }

template <int MODE>
__device__ __forceinline__ f32x4 ld4_pred(const float *__restrict__ safe, const float *__restrict__ p, int c,
                                          int limit, bool row_ok, bool &ok) {
    if (MODE == LD_FAST) {
        ok = true;
        return gload4(p + c);
    }
    if (MODE == LD_PRED) {
        ok = row_ok && c < limit;
        const float *q = ok ? p + c : safe;
        return gload4(q);
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    ok = true;
    if (row_ok) {
        if (c + 0 < limit) v[0] = gload1(p + c + 0);
        if (c + 1 < limit) v[1] = gload1(p + c + 1);
        if (c + 2 < limit) v[2] = gload1(p + c + 2);
        if (c + 3 < limit) v[3] = gload1(p + c + 3);
    }
    return v;
}

// Staging coordinates of this thread in an operand image: it fills two slots (0, 1) of each
// operand per K-step, 4 consecutive floats each.
//   R-type [64 rows][32 k]: row = t/8 (+32 for slot 1), col = 4*(t%8)   along k
//   K-type [32 k][64 cols]: row = t/16 (+16 for slot 1), col = 4*(t%16)
__device__ __forceinline__ int stage_row(bool ktype, int slot) {
    const int t = threadIdx.x;
    return ktype ? (t >> 4) + 16 * slot : (t >> 3) + 32 * slot;
}
__device__ __forceinline__ int stage_col(bool ktype) {
    const int t = threadIdx.x;
    return ktype ? (t & 15) * 4 : (t & 7) * 4;
}

// A_K / B_K: operand image is K-type (true) or R-type (false).
// The loader L owns all addressing: L.a(slot, ok) / L.b(slot, ok) return this thread's 4 floats of
// staging slot `slot` for the loader's CURRENT K-step (ok = false: outside the operand, stored as
// zeros), L.next() moves to the next K-step. Steps are fetched strictly in order, so a loader keeps
// ready-made pointers and bumps them (a handful of VALU ops per step instead of rebuilding 64-bit
// addresses for every load -- at fp32 MFMA rates that address arithmetic would cost as much issue
// time as the MFMAs). The pipeline calls next() up to two steps past the end and never uses those
// loads; loaders freeze at the last step so they stay inside the tensors.
//
// Pipeline: prefetch distance 2. Two register sets hold the tiles of steps s+1 and s+2 while
// step s is multiplied out of LDS, so every global load has two MFMA phases (~0.9 us) to land;
// at this path's problem sizes a tile is a short dependent chain of <= 16 steps and the load
// latency, not bandwidth or MFMA rate, sets its duration. Every iteration stores one set and
// refills it unconditionally (surplus work at the tail touches only valid memory and the idle LDS
// buffer), so the body is branch-free; it is unrolled by two so that the register sets are
// addressed statically (a runtime-indexed set would live in scratch).
template <bool A_K, bool B_K, class LD>
__device__ __forceinline__ void gemm_block(f32x16 &acc, LD &L, int nsteps, float *smem) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    float *As = smem;                          // 2 buffers
    float *Bs = smem + 2 * GT_TILE_FLOATS;     // 2 buffers

    const int ar0 = stage_row(A_K, 0), ar1 = stage_row(A_K, 1), ac = stage_col(A_K);
    const int br0 = stage_row(B_K, 0), br1 = stage_row(B_K, 1), bc = stage_col(B_K);
    const int lda = A_K ? GT_LDK : GT_LDR;
    const int ldb = B_K ? GT_LDK : GT_LDR;

    if (nsteps <= 0) return;
    f32x4 pa0, pa1, pb0, pb1;      // register set P
    f32x4 qa0, qa1, qb0, qb1;      // register set Q
    bool pm0 = true, pm1 = true, pm2 = true, pm3 = true, qm0 = true, qm1 = true, qm2 = true, qm3 = true;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define GT_LOAD(A0, A1, B0, B1, M0, M1, M2, M3) \
    A0 = L.a(0, M0);                            \
    A1 = L.a(1, M1);                            \
    B0 = L.b(0, M2);                            \
    B1 = L.b(1, M3);                            \
    L.next();
#define GT_STORE(BUF, A0, A1, B0, B1, M0, M1, M2, M3)                                              \
    *reinterpret_cast<f32x4 *>(As + (BUF)*GT_TILE_FLOATS + ar0 * lda + ac) = M0 ? A0 : zero4;      \
    *reinterpret_cast<f32x4 *>(As + (BUF)*GT_TILE_FLOATS + ar1 * lda + ac) = M1 ? A1 : zero4;      \
    *reinterpret_cast<f32x4 *>(Bs + (BUF)*GT_TILE_FLOATS + br0 * ldb + bc) = M2 ? B0 : zero4;      \
    *reinterpret_cast<f32x4 *>(Bs + (BUF)*GT_TILE_FLOATS + br1 * ldb + bc) = M3 ? B1 : zero4;

    auto compute = [&](int buf) {
        const float *Ac = As + buf * GT_TILE_FLOATS;
        const float *Bc = Bs + buf * GT_TILE_FLOATS;
        float a[16], b[16];
        if (A_K) {
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = Ac[(16 * h + k) * GT_LDK + wr * 32 + i];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(Ac + (wr * 32 + i) * GT_LDR + 16 * h + 4 * q);
                a[4 * q + 0] = v[0]; a[4 * q + 1] = v[1]; a[4 * q + 2] = v[2]; a[4 * q + 3] = v[3];
            }
        }
        if (B_K) {
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = Bc[(16 * h + k) * GT_LDK + wc * 32 + i];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(Bc + (wc * 32 + i) * GT_LDR + 16 * h + 4 * q);
                b[4 * q + 0] = v[0]; b[4 * q + 1] = v[1]; b[4 * q + 2] = v[2]; b[4 * q + 3] = v[3];
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
    };

    // prologue: tile 0 -> LDS[0]; tiles 1, 2 in flight in Q, P
    GT_LOAD(pa0, pa1, pb0, pb1, pm0, pm1, pm2, pm3)
    GT_LOAD(qa0, qa1, qb0, qb1, qm0, qm1, qm2, qm3)
    GT_STORE(0, pa0, pa1, pb0, pb1, pm0, pm1, pm2, pm3)
    GT_LOAD(pa0, pa1, pb0, pb1, pm0, pm1, pm2, pm3)
    __syncthreads();
    int s = 0;
    while (true) {
        // even step: tile s in LDS[0]; Q holds tile s+1, P holds tile s+2
        compute(0);
        GT_STORE(1, qa0, qa1, qb0, qb1, qm0, qm1, qm2, qm3)
        GT_LOAD(qa0, qa1, qb0, qb1, qm0, qm1, qm2, qm3)
        __syncthreads();
        if (++s >= nsteps) break;
        // odd step: tile s in LDS[1]; P holds tile s+1, Q holds tile s+2
        compute(1);
        GT_STORE(0, pa0, pa1, pb0, pb1, pm0, pm1, pm2, pm3)
        GT_LOAD(pa0, pa1, pb0, pb1, pm0, pm1, pm2, pm3)
        __syncthreads();
        if (++s >= nsteps) break;
    }
#undef GT_LOAD
#undef GT_STORE
}

// C/D fragment coordinates of accumulator register `reg` for this lane inside
// the workgroup's 64x64 tile (guide: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)).
__device__ __forceinline__ int acc_row(int reg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave >> 1) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave & 1) * 32 + (lane & 31);
}

static inline bool ptr_vec_ok(const void *p, int64_t ld) {
    return (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (ld % 4 == 0);
}

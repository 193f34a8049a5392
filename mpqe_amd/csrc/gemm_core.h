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

// Guarded 4-float load: elements at index >= limit read as 0. `vec` says the
// row base is 16-byte aligned and limit % 4 == 0 (uniform per launch).
__device__ __forceinline__ f32x4 ld4_guard(const float *p, int c, int limit, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
        if (c < limit) v = *reinterpret_cast<const f32x4 *>(p + c);
    } else {
        if (c + 0 < limit) v[0] = p[c + 0];
        if (c + 1 < limit) v[1] = p[c + 1];
        if (c + 2 < limit) v[2] = p[c + 2];
        if (c + 3 < limit) v[3] = p[c + 3];
    }
    return v;
}

// A_K / B_K: operand image is K-type (true) or R-type (false).
// aload(row_or_k, col, step) / bload(...) return the 4 floats at tile-local
// (row, col..col+3) of K-step `step`, zero beyond the operand's extent:
//   R-type: row in [0,64), col in {0,4,..,28} along k
//   K-type: row in [0,32) along k, col in {0,4,..,60}
//
// Pipeline: prefetch distance 2. Two register sets hold the tiles of steps s+1 and s+2 while
// step s is multiplied out of LDS, so every global load has two MFMA phases (~0.9 us) to land;
// at this path's problem sizes a tile is a short dependent chain of <= 16 steps and the load
// latency, not bandwidth or MFMA rate, sets its duration. The loop is unrolled by two so that
// the register sets are addressed statically (a runtime-indexed set would live in scratch).
template <bool A_K, bool B_K, class AF, class BF>
__device__ __forceinline__ void gemm_block(f32x16 &acc, AF aload, BF bload, int nsteps, float *smem) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    float *As = smem;                          // 2 buffers
    float *Bs = smem + 2 * GT_TILE_FLOATS;     // 2 buffers

    // this thread's two staging slots per operand
    const int ar0 = A_K ? (t >> 4) : (t >> 3);
    const int ac = A_K ? ((t & 15) * 4) : ((t & 7) * 4);
    const int ar1 = ar0 + (A_K ? 16 : 32);
    const int br0 = B_K ? (t >> 4) : (t >> 3);
    const int bc = B_K ? ((t & 15) * 4) : ((t & 7) * 4);
    const int br1 = br0 + (B_K ? 16 : 32);
    const int lda = A_K ? GT_LDK : GT_LDR;
    const int ldb = B_K ? GT_LDK : GT_LDR;

    if (nsteps <= 0) return;
    f32x4 pa0, pa1, pb0, pb1;      // register set P
    f32x4 qa0, qa1, qb0, qb1;      // register set Q
#define GT_LOAD(A0, A1, B0, B1, STEP) \
    A0 = aload(ar0, ac, (STEP));       \
    A1 = aload(ar1, ac, (STEP));       \
    B0 = bload(br0, bc, (STEP));       \
    B1 = bload(br1, bc, (STEP));
#define GT_STORE(BUF, A0, A1, B0, B1)                                                   \
    *reinterpret_cast<f32x4 *>(As + (BUF)*GT_TILE_FLOATS + ar0 * lda + ac) = A0;        \
    *reinterpret_cast<f32x4 *>(As + (BUF)*GT_TILE_FLOATS + ar1 * lda + ac) = A1;        \
    *reinterpret_cast<f32x4 *>(Bs + (BUF)*GT_TILE_FLOATS + br0 * ldb + bc) = B0;        \
    *reinterpret_cast<f32x4 *>(Bs + (BUF)*GT_TILE_FLOATS + br1 * ldb + bc) = B1;

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
    GT_LOAD(pa0, pa1, pb0, pb1, 0)
    if (nsteps > 1) { GT_LOAD(qa0, qa1, qb0, qb1, 1) }
    GT_STORE(0, pa0, pa1, pb0, pb1)
    if (nsteps > 2) { GT_LOAD(pa0, pa1, pb0, pb1, 2) }
    __syncthreads();
    int s = 0;
    while (true) {
        // even step: tile s in LDS[0]; Q holds tile s+1, P holds tile s+2
        compute(0);
        if (s + 1 < nsteps) { GT_STORE(1, qa0, qa1, qb0, qb1) }
        if (s + 3 < nsteps) { GT_LOAD(qa0, qa1, qb0, qb1, s + 3) }
        __syncthreads();
        if (++s >= nsteps) break;
        // odd step: tile s in LDS[1]; P holds tile s+1, Q holds tile s+2
        compute(1);
        if (s + 1 < nsteps) { GT_STORE(0, pa0, pa1, pb0, pb1) }
        if (s + 3 < nsteps) { GT_LOAD(pa0, pa1, pb0, pb1, s + 3) }
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

// Weight-gradient tile of the chain form, register-only K loop: out[i0 .. i0+64)[j0 .. j0+64) (+)= sum over the graphs q of a
// K-chunk of x[q][i] g[q][j], x = rows of H[p] of a source node slot, g = rows of gH[p+1] of a destination slot, both in the
// row layout the chain kernel writes (row of graph q and slot s = q * N + s).
//
// v_mfma_f32_16x16x4_f32 (exact fp32): lane l feeds A[position l & 15][k = l >> 4] and B[k = l >> 4][position l & 15], and
// WHICH row / column a position stands for is ours to choose. Lane (pos, kq) loads 16 bytes of graph q = 4 t' + kq: the four
// adjacent elements x[q][i0 + 4 pos .. + 3], and likewise of g; MFMA (m, n) takes component m of the x load and component n of
// the g load, so its position (pr, pc) is the output element (i0 + 4 pr + m, j0 + 4 pc + n): the 16 MFMAs (m, n) cover the
// 64 x 64 tile with the 4 graphs of the lane group as their K. Two coalesced 16-byte loads (256 contiguous bytes per graph
// row) feed 16 MFMAs -- no LDS, no barrier, no cross-lane step in the K loop. The four waves split the graphs (wave w takes
// graphs 4 w .. 4 w + 3 of every 16), keep a whole 64 x 64 accumulator each (64 VGPRs) and GWR_PF graphs-of-four in flight,
// and meet once at the end in LDS (a tile per wave, one barrier), added in wave order (fixed order: reproducible).
// (The LDS-DMA ring form needed one 4-byte LDS read per MFMA operand: 1 700 cycles per 32 graphs against 1 024 of MFMA,
// 13.8 us per 512-graph tile; this loop is bounded by the MFMA pipe: 512 cycles per 16 graphs and wave.)
// Included by step.hip.
#pragma once

#ifndef GWR_DBG
#define GWR_DBG 0
#endif
#ifndef GWR_PF
#define GWR_PF 8            // loads in flight per wave, in iterations (16 graphs of the workgroup = 4 of the wave)
#endif
#define GWR_LDT 68          // row stride of an LDS tile (floats)
#define GWR_SMEM_FLOATS (4 * 64 * GWR_LDT)          // one tile per wave (LDS_TILES = 4; 1: the waves take turns on one;
#define GWR_SMEM_FLOATS2 (2 * 64 * GWR_LDT)         // 2: waves 2, 3 add theirs onto the tiles of waves 0, 1 -- 34 KB)

// a value every lane of the wave holds, moved to SGPRs (the tile's record arrives through a vector load: hipcc cannot know)
__device__ __forceinline__ long long gwr_uniform(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

template <int NJ> struct gwr_bvec;
template <> struct gwr_bvec<4> { typedef f32x4 type; typedef gvec4_ptr gptr; };
template <> struct gwr_bvec<2> {
    typedef f32x2 type;
    typedef const f32x2 __attribute__((address_space(1))) * gptr;
};
// NJ: columns of g per lane -- 4: a 64 x 64 tile (16-byte g loads, 16 MFMAs per iteration), 2: 64 x 32 (8-byte g loads, 8 MFMAs)
template <int LDS_TILES = 4, int NJ = 4>
__device__ __forceinline__ void grad_w_tile_rows(const float *__restrict__ x, const float *__restrict__ g, int D,
                                                 long long xs, long long xo, long long go, long long q0, long long q1,
                                                 int i0, int j0, float *__restrict__ dst, float *smem, bool accumulate,
                                                 long long *dbg = nullptr, bool through = false) {
    // through: the tile is a slab that workgroups of the SAME launch read (fused tail, step.hip): written through to memory
    // at agent scope instead of left dirty in this XCD's L2
#ifndef MPQE_EMU
#define GWR_STAMP(slot, wait)                                 \
    if (dbg && threadIdx.x == 0) {                            \
        asm volatile(wait ::: "memory");                      \
        dbg[slot] = (long long)wall_clock64();                \
    }
#else
#define GWR_STAMP(slot, wait)
#endif
    GWR_STAMP(4, "s_waitcnt vmcnt(0) lgkmcnt(0)")         // the tile's record has arrived
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pos = lane & 15, kq = lane >> 4;
    const long long stride = xs * D;                      // floats between the rows of two consecutive graphs
    // address = wave-uniform base (SGPRs, advanced by scalar adds) + a 32-bit lane offset that never changes: the K loop
    // has NO address arithmetic on the VALU (every VALU instruction between two MFMAs costs ~5 cycles of the MFMA pipe)
    const char *xw = reinterpret_cast<const char *>(
        gwr_uniform(reinterpret_cast<long long>(x + xo * D + i0 + (q0 + 4 * wave) * stride)));
    const char *gw = reinterpret_cast<const char *>(
        gwr_uniform(reinterpret_cast<long long>(g + go * D + j0 + (q0 + 4 * wave) * stride)));
    typedef typename gwr_bvec<NJ>::type bvec;
    typedef typename gwr_bvec<NJ>::gptr bptr;
    const unsigned loff = (unsigned)((kq * stride + 4 * pos) * 4), goff = (unsigned)((kq * stride + NJ * pos) * 4);
    const long long step = gwr_uniform(16 * stride * 4);  // bytes per iteration (16 graphs of the workgroup)
    const int nfull = __builtin_amdgcn_readfirstlane((int)((q1 - q0) / 16));     // iterations with all 16 graphs inside
    f32x4 acc[4][NJ];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NJ; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 A[GWR_PF];
    bvec B[GWR_PF];
    auto load = [&](int slot, int t) {      // t: uniform; beyond the last whole iteration the last one is loaded again (unused)
#if GWR_DBG == 1                    // (timing experiment: every load hits the same rows)
        const int tc = 0;
#else
        const int tc = t < nfull ? t : nfull - 1;
#endif
        const char *xt = xw + (long long)tc * step, *gt = gw + (long long)tc * step;       // (SALU)
        A[slot] = *(gvec4_ptr)(xt + loff);      // global_load_dwordx4 v, v_loff, s[base]
        B[slot] = *(bptr)(gt + goff);
    };
    auto mma = [&](const f32x4 &a, const bvec &b) {
#if GWR_DBG == 2                    // (timing experiment: loads only)
        asm volatile("" ::"v"(a), "v"(b));
        return;
#endif
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NJ; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
    };
    if (nfull > 0) {
#pragma unroll
        for (int s = 0; s < GWR_PF; ++s) {
            load(s, s);
            __builtin_amdgcn_sched_barrier(0);  // in slot order: hipcc issued them last-slot-first, and the loop's first wait
                                                // (for slot 0) then had to be vmcnt(0) on every trip
        }
        GWR_STAMP(5, "s_waitcnt vmcnt(0)")               // first rows landed
        int t = 0;
        for (; t + GWR_PF <= nfull; t += GWR_PF) {
#pragma unroll
            for (int s = 0; s < GWR_PF; ++s) {
                mma(A[s], B[s]);     // unconditional: a branch here makes hipcc drain ALL loads in flight at the top of every
                                     // trip (s_waitcnt vmcnt(2) instead of (14))
                __builtin_amdgcn_sched_barrier(0);        // (or the scheduler sinks the loads to their uses: none in flight)
                load(s, t + s + GWR_PF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int s = 0; s < GWR_PF; ++s)                  // the last nfull % GWR_PF iterations are in their slots already
            if (t + s < nfull) mma(A[s], B[s]);
    }
    if (q0 + 16ll * nfull < q1) {            // ragged end: graphs of a last, partial 16 (clamped address, zero operand)
        const long long q = q0 + 16ll * nfull + 4 * wave + kq;
        const long long qc = q < q1 ? q : q1 - 1;
        f32x4 a = gload4(x + xo * D + i0 + 4 * pos + qc * stride);
        const bvec b = *(bptr)(g + go * D + j0 + NJ * pos + qc * stride);
        if (q >= q1) a = f32x4{0.f, 0.f, 0.f, 0.f};
        mma(a, b);
    }
    GWR_STAMP(6, "s_nop 0")
    // D[position row 4 kq + r][position col pos] of MFMA (m, n) = out[i0 + 4 (4 kq + r) + m][j0 + 4 pos + n]: a lane's four n
    // are 16 adjacent bytes. Every wave leaves its tile in LDS; the tiles are added in wave order (fixed: reproducible).
    if constexpr (LDS_TILES == 4) {
        float *mine = smem + wave * (64 * GWR_LDT);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                bvec v;
#pragma unroll
                for (int n = 0; n < NJ; ++n) v[n] = acc[m][n][r];
                *reinterpret_cast<bvec *>(mine + (4 * (4 * kq + r) + m) * GWR_LDT + NJ * pos) = v;
            }
        __syncthreads();
    } else if constexpr (LDS_TILES == 2) {      // two tiles: waves 0, 1 store, waves 2, 3 add onto them -- (w0 + w2) + (w1 + w3)
        float *mine = smem + (wave & 1) * (64 * GWR_LDT);
        for (int half = 0; half < 2; ++half) {
            if ((wave >> 1) == half) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        bvec *t = reinterpret_cast<bvec *>(mine + (4 * (4 * kq + r) + m) * GWR_LDT + NJ * pos);
                        bvec v;
#pragma unroll
                        for (int n = 0; n < NJ; ++n) v[n] = acc[m][n][r];
                        if (half) v += *t;
                        *t = v;
                    }
            }
            __syncthreads();
        }
    } else {                            // (a workgroup with less LDS: one tile, the waves add theirs one after the other)
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        bvec *t = reinterpret_cast<bvec *>(smem + (4 * (4 * kq + r) + m) * GWR_LDT + NJ * pos);
                        bvec v;
#pragma unroll
                        for (int n = 0; n < NJ; ++n) v[n] = acc[m][n][r];
                        if (w > 0) v += *t;
                        *t = v;
                    }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < NJ; ++k) {              // the tile's 64 x 4 NJ float4s
        const int f = (int)threadIdx.x + 256 * k;
        const int row = f / (4 * NJ), c4 = f % (4 * NJ);
        const float *t0 = smem + row * GWR_LDT + 4 * c4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(t0);
        if constexpr (LDS_TILES == 4) {
#pragma unroll
            for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4 *>(t0 + w * (64 * GWR_LDT));
        } else if constexpr (LDS_TILES == 2) {
            v += *reinterpret_cast<const f32x4 *>(t0 + 64 * GWR_LDT);
        }
        f32x4 *o = reinterpret_cast<f32x4 *>(dst + (long long)(i0 + row) * D + j0 + 4 * c4);
        if (accumulate) v += *o;
#ifndef MPQE_EMU
        if (through) {
            unsigned long long *o8 = reinterpret_cast<unsigned long long *>(o);
            __hip_atomic_store(o8, (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o8 + 1, (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
#endif
        *o = v;
    }
}

// Weight-gradient tile with a deep LDS-DMA pipeline (gfx950 `global_load_lds_dwordx4`).
//
//   slab[i0.., j0..] = sum_{q in [q0, q0 + 32*nsteps)} x[q*xs + xo][i0..]^T (x) g[q*gs + go][j0..]
//
// Both operands are K-type images ([32 k-rows][64 columns], rows = batch entries q): reads go ALONG a
// row (lanes 0..31 = consecutive floats), so the image needs no padding and a wave's DMA piece --
// 64 lanes x 16 B = 4 whole 256-byte rows -- lands exactly where the fragment reads expect it. The
// rows of H / gH this kernel streams are read once and mostly miss the XCD's L2 (PMC: 44 % hit), i.e.
// they come from the Infinity Cache / HBM at ~1.5-2 us; the register pipeline of gemm_core.h keeps 2
// steps (~1 us) in flight, which is not enough. Here a ring of STAGES LDS buffers is filled by DMA
// STAGES-1 steps ahead at no VGPR cost:
//
//   prologue   issue steps 0 .. STAGES-2
//   step s     s_waitcnt vmcnt(4*(STAGES-2))   this wave's 4 pieces of step s have landed
//              s_barrier                        every wave's pieces have; everyone left stage (s-1)
//              issue step s+STAGES-1 into stage (s-1) % STAGES
//              16 MFMAs from stage s % STAGES
//
// Only the unpredicated case (D % 64 == 0, whole K-steps) -- everything else takes the register
// pipeline. All LDS lives in the caller's one __shared__ array; no other VGPR-destination global load
// sits in the loop (either would make hipcc drain the DMA queue with vmcnt(0)).
#pragma once
#include "gemm_core.h"

#define GWD_STAGES 4
#define GWD_TILE_FLOATS 2048                       // 32 x 64
#define GWD_STAGE_FLOATS (2 * GWD_TILE_FLOATS)     // A + B = 16 KB
#define GWD_SMEM_FLOATS (GWD_STAGES * GWD_STAGE_FLOATS)   // 64 KB

typedef const void __attribute__((address_space(1))) * gwd_gptr;
typedef void __attribute__((address_space(3))) * gwd_lptr;

// s_waitcnt simm16 for "vmcnt(n) only" on gfx9+: vmcnt[3:0] bits 3:0, expcnt (7 = no wait) bits 6:4,
// lgkmcnt (15 = no wait) bits 11:8, vmcnt[5:4] bits 15:14
#define GWD_VMCNT(n) ((((n) & 0xF)) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14))

__device__ __forceinline__ void grad_w_tile_dma(const float *__restrict__ x, const float *__restrict__ g, int Din,
                                                int Dout, long long xs, long long xo, long long gs, long long go,
                                                long long q0, int nsteps, int i0, int j0,
                                                float *__restrict__ slab, float *smem, bool accumulate = false,
                                                long long *dbg = nullptr) {
    const int t = threadIdx.x;
#ifndef MPQE_EMU      // diagnostics (mpqe_debug_tail_stamps): wall clock at the tile's phase boundaries
#define GWD_STAMP(slot) if (dbg && t == 0) dbg[slot] = (long long)wall_clock64();
#else
#define GWD_STAMP(slot)
#endif
    GWD_STAMP(4)
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    // this lane's part of a DMA piece: row (lane / 16) of the piece's 4 rows, 4 floats at column 4*(lane % 16)
    const int prow = lane >> 4, pcol = (lane & 15) * 4;
    const int krow = 8 * wave;                         // this wave fills k-rows [8w, 8w+8) of both images
    const long long stepx = (long long)GT_BK * xs * Din, stepg = (long long)GT_BK * gs * Dout;
    const float *px0 = x + ((q0 + krow + prow) * xs + xo) * (long long)Din + i0 + pcol;
    const float *px1 = px0 + 4 * xs * (long long)Din;
    const float *pg0 = g + ((q0 + krow + prow) * gs + go) * (long long)Dout + j0 + pcol;
    const float *pg1 = pg0 + 4 * gs * (long long)Dout;
    int left = nsteps;                                  // loads freeze on the last step (valid memory)

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    auto issue = [&](int stage) {
        float *sa = smem + stage * GWD_STAGE_FLOATS + krow * 64;        // wave-uniform LDS bases
        float *sb = sa + GWD_TILE_FLOATS;
        __builtin_amdgcn_global_load_lds((gwd_gptr)px0, (gwd_lptr)sa, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gwd_gptr)px1, (gwd_lptr)(sa + 4 * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gwd_gptr)pg0, (gwd_lptr)sb, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gwd_gptr)pg1, (gwd_lptr)(sb + 4 * 64), 16, 0, 0);
        const bool go_on = left > 1;
        left -= go_on ? 1 : 0;
        const long long dx = go_on ? stepx : 0, dg = go_on ? stepg : 0;
        px0 += dx;
        px1 += dx;
        pg0 += dg;
        pg1 += dg;
    };

#pragma unroll
    for (int s = 0; s < GWD_STAGES - 1; ++s) issue(s);
    for (int s = 0; s < nsteps; ++s) {
        __builtin_amdgcn_s_waitcnt(GWD_VMCNT(4 * (GWD_STAGES - 2)));
        if (s == 0) GWD_STAMP(5)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue((s + GWD_STAGES - 1) % GWD_STAGES);
        const float *Ac = smem + (s % GWD_STAGES) * GWD_STAGE_FLOATS;
        const float *Bc = Ac + GWD_TILE_FLOATS;
        float a[16], b[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = Ac[(16 * h + k) * 64 + wr * 32 + i];
#pragma unroll
        for (int k = 0; k < 16; ++k) b[k] = Bc[(16 * h + k) * 64 + wc * 32 + i];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
    }
    // drain the surplus tail loads before the LDS can be reused by anyone
    __builtin_amdgcn_s_waitcnt(GWD_VMCNT(0));
    GWD_STAMP(6)
    const int col = j0 + acc_col();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = i0 + acc_row(r);
        float *o = slab + (long long)row * Dout + col;      // accumulate: the tile IS the gradient (one source)
        *o = accumulate ? *o + acc[r] : acc[r];
    }
}


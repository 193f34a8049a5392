// grad_bias[j] += sum_q gpre[q][j] in a fixed order: per-block partial column sums over `rpb` rows
// (4 row groups x 64 columns per workgroup, combined as (0+1)+(2+3)), then one pass over the <= 1024
// partial rows (16 row groups x 64 columns, summed in order; no atomics -> reproducible).
#pragma once
#include "common.h"

static inline int bias_rows_per_block(long long rows) {
    long long r = (rows + 1023) / 1024;      // at most 1024 partial rows (four workgroups per CU: a pass over 33 MB is
    if (r < 32) r = 32;                      // latency-bound with one -- 20 us against 12)
    return (int)r;
}
static inline int bias_num_blocks(long long rows) {
    const int rpb = bias_rows_per_block(rows);
    return (int)((rows + rpb - 1) / rpb);
}

// bits (may be NULL; Dout % 64 == 0): the ReLU mask of `out` as one 64-bit word per (row, 64 columns) -- bit c % 64 of word
// [r][c / 64] = out[r][c] > 0. This pass reads every element of `out` anyway; the gather-GEMM and the weight-gradient
// kernel of the general path then mask with 8 bytes per (row, 64 columns) instead of gathering the row of `out` again
// (rgcn_general.hip). partial may be NULL (mask only).
static __global__ __launch_bounds__(256) void bias_partial_kernel(long long rows, int rpb, const float *__restrict__ g,
                                                                  const float *__restrict__ out, int Dout, int relu,
                                                                  float *__restrict__ partial,
                                                                  unsigned long long *__restrict__ bits = nullptr,
                                                                  const unsigned long long *__restrict__ bits_in = nullptr) {
    // bits_in (may be NULL; Dout % 64 == 0): the mask words already exist (the forward wrote them): `out` is not read
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + cl;
    const long long r0 = (long long)blockIdx.x * rpb;
    long long r1 = r0 + rpb;
    if (r1 > rows) r1 = rows;
    float s = 0.f;
    if (col < Dout) {
        // four rows of the wave per trip: their loads go out together, then the sums / mask words (a mask word needs its
        // row's compare, so one row per trip would wait for every load in turn)
        for (long long r = r0 + rg; r < r1; r += 16) {
            float v[4], o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long rr = r + 4 * q < r1 ? r + 4 * q : r;
                v[q] = g[rr * Dout + col];
                if (bits_in) o[q] = ((bits_in[rr * (Dout / 64) + blockIdx.y] >> cl) & 1ull) ? 1.f : 0.f;
                else o[q] = relu ? out[rr * Dout + col] : 1.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (r + 4 * q >= r1) break;          // (uniform over the wave)
                const bool on = o[q] > 0.f;
                s += on ? v[q] : 0.f;
                if (bits) {         // (Dout % 64 == 0: the whole wave is here)
                    const long long wi = (r + 4 * q) * (Dout / 64) + blockIdx.y;
#ifdef MPQE_EMU
                    int w = on ? (int)(1u << (cl & 31)) : 0;
                    for (int m = 1; m < 32; m <<= 1) w |= __shfl_xor(w, m, 64);
                    if ((cl & 31) == 0) reinterpret_cast<unsigned *>(bits)[2 * wi + (cl >> 5)] = (unsigned)w;
#else
                    const unsigned long long w = __ballot(on);
                    if (cl == 0) bits[wi] = w;
#endif
                }
            }
        }
    }
    part[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < Dout && partial)
        partial[(long long)blockIdx.x * Dout + col] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// ReLU mask words (bit c % 64 of word [row][c / 64] = out[row][c] > 0): the 16 lanes that hold 64 consecutive columns of
// a row (4 each, 16-lane aligned) are one DPP row -- their nibbles are OR-ed with four row rotations, lane 0 of the row
// stores the word.
static __device__ __forceinline__ void mask_word_store(unsigned long long *__restrict__ bits, long long row, int D, int c,
                                                       unsigned nib) {
    const int l16 = (c >> 2) & 15;
    int lo = l16 < 8 ? (int)(nib << (4 * l16)) : 0, hi = l16 >= 8 ? (int)(nib << (4 * (l16 - 8))) : 0;
#ifdef MPQE_EMU
    for (int m = 1; m < 16; m <<= 1) {
        lo |= __shfl_xor(lo, m, 64);
        hi |= __shfl_xor(hi, m, 64);
    }
#else
    lo |= __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);      // row_ror:8
    hi |= __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
    lo |= __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xf, 0xf, false);      // row_ror:4
    hi |= __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xf, 0xf, false);
    lo |= __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xf, 0xf, false);      // row_ror:2
    hi |= __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xf, 0xf, false);
    lo |= __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xf, 0xf, false);      // row_ror:1
    hi |= __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xf, 0xf, false);
#endif
    if (l16 == 0 && row >= 0) bits[row * (D / 64) + (c >> 6)] = (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32);
}

#define BP4_Q 8      // rows in flight per lane (one block per CU: the loads in flight are what fills the memory pipe)
// The same partial sums with 16-byte loads (Dout % 64 == 0, Dout <= 1024, 16-byte aligned rows): Dout / 4 lanes cover a
// row, 256 / (Dout / 4) rows per trip and BP4_Q trips in flight; the row groups' sums meet in LDS and are added in
// group order (fixed order: reproducible). bits / bits_in as above.
static __global__ __launch_bounds__(256) void bias_partial4_kernel(long long rows, int rpb, const float *__restrict__ g,
                                                                   const float *__restrict__ out, int Dout, int relu,
                                                                   float *__restrict__ partial,
                                                                   unsigned long long *__restrict__ bits,
                                                                   const unsigned long long *__restrict__ bits_in) {
    __shared__ f32x4 part[256];
    const int lq = Dout / 4, RG = 256 / lq;
    const int c = ((int)threadIdx.x % lq) * 4, rg = (int)threadIdx.x / lq;
    const long long r0 = (long long)blockIdx.x * rpb;
    long long r1 = r0 + rpb;
    if (r1 > rows) r1 = rows;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    // (every lane makes every trip -- rows beyond the block's range are clamped for the loads and left out of the sums --,
    // so that the mask words' lane exchanges always see whole waves)
    for (long long rb = r0; rb < r1; rb += BP4_Q * RG) {
        f32x4 v[BP4_Q], o[BP4_Q];
        unsigned long long w[BP4_Q];
#pragma unroll
        for (int q = 0; q < BP4_Q; ++q) {
            const long long r = rb + rg + (long long)q * RG, rr = r < r1 ? r : r0;
            v[q] = *reinterpret_cast<const f32x4 *>(g + rr * Dout + c);
            if (bits_in) w[q] = bits_in[rr * (Dout / 64) + (c >> 6)];
            else if (relu) o[q] = *reinterpret_cast<const f32x4 *>(out + rr * Dout + c);
        }
#pragma unroll
        for (int q = 0; q < BP4_Q; ++q) {
            const long long r = rb + rg + (long long)q * RG;
            const bool valid = r < r1;
            unsigned nib = 15u;
            if (bits_in) nib = (unsigned)(w[q] >> (c & 63)) & 15u;
            else if (relu) nib = (o[q][0] > 0.f ? 1u : 0u) | (o[q][1] > 0.f ? 2u : 0u) | (o[q][2] > 0.f ? 4u : 0u) | (o[q][3] > 0.f ? 8u : 0u);
            if (valid) {
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += ((nib >> k) & 1u) ? v[q][k] : 0.f;
            }
            if (bits) mask_word_store(bits, valid ? r : -1, Dout, c, nib);
        }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < lq && partial) {
        f32x4 t = part[threadIdx.x];
        for (int q = 1; q < RG; ++q) {
            const f32x4 u = part[threadIdx.x + q * lq];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += u[k];
        }
        *reinterpret_cast<f32x4 *>(partial + (long long)blockIdx.x * Dout + c) = t;
    }
}

// One workgroup of 1024 threads per 64 columns: 16 row groups each add every 16th partial row (eight requests in flight),
// then the 16 sums are added in order -- a fixed order. (The first form walked the <= 256 partial rows with 4 row groups
// and one dependent load per step: 118 us per call at the stress shape; this one takes one or two round trips.)
static __global__ __launch_bounds__(1024) void bias_final_kernel(int nblk, int Dout, const float *__restrict__ partial,
                                                                 float *__restrict__ grad_bias, int overwrite) {
    __shared__ float part[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (col < Dout)
        for (int b0 = rg; b0 < nblk; b0 += 16 * 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int b = b0 + 16 * q;
                v[q] = partial[(long long)(b < nblk ? b : b0) * Dout + col];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (b0 + 16 * q < nblk) s += v[q];
        }
    part[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < Dout) {
        float t = part[0][cl];
        for (int q = 1; q < 16; ++q) t += part[q][cl];
        grad_bias[col] = overwrite ? t : grad_bias[col] + t;
    }
}

static inline size_t bias_partial_bytes(long long rows, long long Dout) {
    return align_up((size_t)bias_num_blocks(rows) * (size_t)Dout * 4, 256);
}

// grad_bias may be NULL when only the mask words are wanted (bits != NULL)
static inline void launch_bias_grad(long long rows, const float *g, const float *out, int Dout, int relu,
                                    float *partial, float *grad_bias, hipStream_t s, int overwrite = 0,
                                    unsigned long long *bits = nullptr, const unsigned long long *bits_in = nullptr) {
    const int rpb = bias_rows_per_block(rows), nblk = bias_num_blocks(rows);
    const bool v4 = Dout % 64 == 0 && Dout <= 1024 && 256 % (Dout / 4) == 0 && (uintptr_t)g % 16 == 0 &&
                    (!relu || bits_in || (uintptr_t)out % 16 == 0) && (uintptr_t)partial % 16 == 0;
    if (v4)
        hipLaunchKernelGGL(bias_partial4_kernel, dim3(nblk), dim3(256), 0, s, rows, rpb, g, out, Dout, relu,
                           grad_bias ? partial : (float *)nullptr, bits, bits_in);
    else
        hipLaunchKernelGGL(bias_partial_kernel, dim3(nblk, (unsigned)((Dout + 63) / 64)), dim3(256), 0, s, rows, rpb, g,
                           out, Dout, relu, grad_bias ? partial : (float *)nullptr, bits, bits_in);
    if (grad_bias)
        hipLaunchKernelGGL(bias_final_kernel, dim3((unsigned)((Dout + 63) / 64)), dim3(1024), 0, s, nblk, Dout, partial,
                           grad_bias, overwrite);
}

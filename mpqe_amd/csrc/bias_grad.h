// grad_bias[j] += sum_q gpre[q][j] in a fixed order: per-block partial column sums over `rpb` rows
// (4 row groups x 64 columns per workgroup, combined as (0+1)+(2+3)), then one pass over the <= 256
// partial rows (16 row groups x 64 columns, summed in order; no atomics -> reproducible).
#pragma once
#include "common.h"

static inline int bias_rows_per_block(long long rows) {
    long long r = (rows + 255) / 256;        // at most 256 partial rows
    if (r < 64) r = 64;
    return (int)r;
}
static inline int bias_num_blocks(long long rows) {
    const int rpb = bias_rows_per_block(rows);
    return (int)((rows + rpb - 1) / rpb);
}

static __global__ __launch_bounds__(256) void bias_partial_kernel(long long rows, int rpb, const float *__restrict__ g,
                                                                  const float *__restrict__ out, int Dout, int relu,
                                                                  float *__restrict__ partial) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + cl;
    const long long r0 = (long long)blockIdx.x * rpb;
    long long r1 = r0 + rpb;
    if (r1 > rows) r1 = rows;
    float s = 0.f;
    if (col < Dout)
        for (long long r = r0 + rg; r < r1; r += 4) {
            float v = g[r * Dout + col];
            if (relu && !(out[r * Dout + col] > 0.f)) v = 0.f;
            s += v;
        }
    part[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && col < Dout)
        partial[(long long)blockIdx.x * Dout + col] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
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

static inline void launch_bias_grad(long long rows, const float *g, const float *out, int Dout, int relu,
                                    float *partial, float *grad_bias, hipStream_t s, int overwrite = 0) {
    const int rpb = bias_rows_per_block(rows), nblk = bias_num_blocks(rows);
    hipLaunchKernelGGL(bias_partial_kernel, dim3(nblk, (unsigned)((Dout + 63) / 64)), dim3(256), 0, s, rows, rpb, g,
                       out, Dout, relu, partial);
    hipLaunchKernelGGL(bias_final_kernel, dim3((unsigned)((Dout + 63) / 64)), dim3(1024), 0, s, nblk, Dout, partial,
                       grad_bias, overwrite);
}

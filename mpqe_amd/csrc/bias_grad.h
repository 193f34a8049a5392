// grad_bias[j] += sum_q gpre[q][j] in a fixed order: per-block partial column sums over
// BIAS_ROWS rows, then one pass over the partials (no atomics -> reproducible).
#pragma once
#include "common.h"

#define BIAS_ROWS 64

static __global__ __launch_bounds__(256) void bias_partial_kernel(long long rows, const float *__restrict__ g,
                                                                  const float *__restrict__ out, int Dout, int relu,
                                                                  float *__restrict__ partial) {
    const long long r0 = (long long)blockIdx.x * BIAS_ROWS;
    long long r1 = r0 + BIAS_ROWS;
    if (r1 > rows) r1 = rows;
    for (int col = threadIdx.x; col < Dout; col += blockDim.x) {
        float s = 0.f;
        for (long long r = r0; r < r1; ++r) {
            float v = g[r * Dout + col];
            if (relu && !(out[r * Dout + col] > 0.f)) v = 0.f;
            s += v;
        }
        partial[(long long)blockIdx.x * Dout + col] = s;
    }
}

static __global__ __launch_bounds__(256) void bias_final_kernel(int nblk, int Dout, const float *__restrict__ partial,
                                                                float *__restrict__ grad_bias) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= Dout) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += partial[(long long)b * Dout + col];
    grad_bias[col] += s;
}

static inline size_t bias_partial_bytes(long long rows, long long Dout) {
    return align_up((size_t)((rows + BIAS_ROWS - 1) / BIAS_ROWS) * (size_t)Dout * 4, 256);
}

static inline void launch_bias_grad(long long rows, const float *g, const float *out, int Dout, int relu,
                                    float *partial, float *grad_bias, hipStream_t s) {
    const int nblk = (int)((rows + BIAS_ROWS - 1) / BIAS_ROWS);
    hipLaunchKernelGGL(bias_partial_kernel, dim3(nblk), dim3(256), 0, s, rows, g, out, Dout, relu, partial);
    hipLaunchKernelGGL(bias_final_kernel, dim3((unsigned)((Dout + 255) / 256)), dim3(256), 0, s, nblk, Dout, partial,
                       grad_bias);
}

// Cycles per fp32 MFMA on one wave (s_memtime around N instructions), to price the K loops of the step kernels:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(long long *out, float *sink, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
    f32x16 c0 = {0}, c1 = {0};
    f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    float va[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {                    // 32x32x2, one chain
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c0, 0, 0, 0);
            } else if (MODE == 1) {             // 32x32x2, two chains
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            } else if (MODE == 2) {             // 16x16x4, four chains (4 instructions = one 32x32x2 worth of flops x2)
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, d3, 0, 0, 0);
            } else if (MODE == 3) {             // 32x32x2 two chains + two ds_read_b32 per MFMA
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                float x0 = lds[(threadIdx.x + 64 * u) & 8191], x1 = lds[(threadIdx.x + 64 * u + 4096) & 8191];
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
                float x2 = lds[(threadIdx.x + 64 * u + 1024) & 8191], x3 = lds[(threadIdx.x + 64 * u + 5120) & 8191];
                a += (x0 + x1) * 1e-9f;
                b += (x2 + x3) * 1e-9f;
            } else if (MODE >= 10 && MODE < 20) {        // 32x32x2 two chains + (MODE - 10) independent VALU ops per MFMA
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < MODE - 10; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(va[q & 7]) : "v"(b));
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < MODE - 10; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(va[q & 7]) : "v"(b));
            } else if (MODE >= 20 && MODE < 30) {        // 32x32x2 two chains + (MODE - 20) ds_read2_b32 per MFMA
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < MODE - 20; ++q) {
                    const float *pp = lds + ((threadIdx.x & 63) + 256 * ((u + q) & 7));
                    va[q & 7] += pp[0] + pp[32];
                }
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            } else if (MODE == 4) {             // 16x16x4 one chain
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, d0, 0, 0, 0);
            }
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    for (int r = 0; r < 4; ++r) s += d0[r] + d1[r] + d2[r] + d3[r];
    for (int r = 0; r < 8; ++r) s += va[r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *name, int per_iter, int blocks, int threads) {
    long long *out;
    float *sink;
    hipMalloc(&out, blocks * 4 * sizeof(long long));
    hipMalloc(&sink, blocks * threads * sizeof(float));
    const int iters = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, sink, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s blocks %4d x %3d threads: %.1f memtime ticks per MFMA (wave 0)\n", name, blocks, threads,
           (double)h[0] / (iters * per_iter));
    hipFree(out);
    hipFree(sink);
}

int main() {
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int blocks = cfg ? 256 : 1, threads = cfg ? 256 : 64;
        run<0>("32x32x2 f32, one accumulator chain", 16, blocks, threads);
        run<1>("32x32x2 f32, two chains", 16, blocks, threads);
        run<2>("16x16x4 f32, four chains", 32, blocks, threads);
        run<4>("16x16x4 f32, one chain", 16, blocks, threads);
        run<3>("32x32x2 f32, two chains + 2 ds_read_b32 each", 16, blocks, threads);
        run<11>("32x32x2 f32 + 1 independent VALU op per MFMA", 16, blocks, threads);
        run<12>("32x32x2 f32 + 2 independent VALU ops per MFMA", 16, blocks, threads);
        run<14>("32x32x2 f32 + 4 independent VALU ops per MFMA", 16, blocks, threads);
        run<18>("32x32x2 f32 + 8 independent VALU ops per MFMA", 16, blocks, threads);
        run<21>("32x32x2 f32 + 1 ds_read2_b32 (+2 adds) per 2 MFMAs", 16, blocks, threads);
        run<22>("32x32x2 f32 + 2 ds_read2_b32 (+4 adds) per 2 MFMAs", 16, blocks, threads);
    }
    // s_memtime counts at a fixed 100 MHz: ticks * (shader clock / 100 MHz) = shader cycles
    return 0;
}

// The K-step loop of the weight-gradient tile without its DMA: 4 waves, per step a barrier, 16 MFMAs 32x32x2 and
// the 32 fragment reads of the next step. Cycles per step for several ways of reading the fragments.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tile_loop.hip -o tools/micro/tile_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: ds_read_b32 as in the kernel (row k of 64 floats, lanes 0..31 + half h -> rows 16 h + k)
// MODE 1: same without the barrier
// MODE 2: no LDS reads at all (MFMAs + barrier)
// MODE 3: transposed image [col][k] so a lane reads its 16 k values with 4 ds_read_b128
// MODE 4: MODE 0 reads but all 32 issued before the 16 MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void k(long long *out, float *sink, int steps) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 4096 + 64];
    for (int i = threadIdx.x; i < 4 * 4096; i += blockDim.x) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1, i = lane & 31, h = lane >> 5;
    float a[16], b[16], an[16], bn[16];
    for (int q = 0; q < 16; ++q) a[q] = b[q] = 1e-3f * (q + lane);
    f32x16 c0 = {0};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int s = 0; s < steps; ++s) {
        const float *Ac = lds + (s & 1) * 8192, *Bc = Ac + 2048;
        if (MODE != 1) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        }
        if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                an[q] = Ac[(16 * h + q) * 64 + wr * 32 + i];
                bn[q] = Bc[(16 * h + q) * 64 + wc * 32 + i];
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 va = *reinterpret_cast<const f32x4 *>(Ac + (wr * 32 + i) * 36 + 16 * h + 4 * q);
                const f32x4 vb = *reinterpret_cast<const f32x4 *>(Bc + (wc * 32 + i) * 36 + 16 * h + 4 * q);
                for (int u = 0; u < 4; ++u) {
                    an[4 * q + u] = va[u];
                    bn[4 * q + u] = vb[u];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], c0, 0, 0, 0);
            if (MODE == 0 || MODE == 1) {
                an[q] = Ac[(16 * h + q) * 64 + wr * 32 + i];
                bn[q] = Bc[(16 * h + q) * 64 + wc * 32 + i];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE != 2) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                a[q] = an[q];
                b[q] = bn[q];
            }
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += c0[r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks) {
    long long *out;
    float *sink;
    (void)hipMalloc(&out, blocks * 4 * sizeof(long long));
    (void)hipMalloc(&sink, blocks * 256 * sizeof(float));
    const int steps = 64;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, sink, steps);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, sink, steps);
    (void)hipDeviceSynchronize();
    long long h[4];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-56s %4d blocks: %.0f cycles per step (16 MFMAs = 1024)\n", name, blocks, (double)h[0] / steps);
    (void)hipFree(out);
    (void)hipFree(sink);
}

int main() {
    for (int blocks = 1; blocks <= 256; blocks *= 256) {
        run<0>("b32 reads between the MFMAs, barrier per step", blocks);
        run<1>("b32 reads between the MFMAs, no barrier", blocks);
        run<2>("no LDS reads, barrier per step", blocks);
        run<3>("b128 reads of a [col][k] image before the MFMAs", blocks);
        run<4>("b32 reads, all before the MFMAs", blocks);
    }
    return 0;
}

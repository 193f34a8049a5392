// TEST INFRASTRUCTURE ONLY -- a host-side stand-in for <hip/hip_runtime.h> that lets the
// kernel sources under mpqe_amd/csrc be compiled with clang++ for x86 and executed on the
// CPU, one workgroup at a time, every work-item as a cooperative fiber. It exists so the
// indexing / tiling / reduction logic of each kernel can be checked (and run under
// sanitizers) in the GPU-less authoring container before a gpurun call. It is NOT a
// fallback: nothing under mpqe_amd/ loads the library built from it, and the product
// raises if the real gfx950 library is missing.
//
// Emulated: threadIdx/blockIdx/blockDim/gridDim, __syncthreads, __shfl_xor / __shfl_down /
// __shfl (wave = 64), atomics, v_mfma_f32_32x32x2_f32 (lane maps as documented in the
// MI355X guide: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31 row=(r&3)+8(r>>2)+4(l>>5),
// evaluated as a k-ordered fmaf chain), hipLaunchKernelGGL, hipMemsetAsync/hipMemcpyAsync.
#pragma once
#define MPQE_EMU 1
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 {
    unsigned x, y, z;
};
extern emu_uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#ifndef __restrict__
#define __restrict__
#endif

typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return hipSuccess;
}
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    memcpy(d, s, n);
    return hipSuccess;
}

typedef void *hipEvent_t;
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }

namespace emu {
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void block_barrier();
float wave_exchange(float v, int src_lane_xor, int mode, int arg);
void mfma_32x32x2(float a, float b, float *c16);
void mfma_16x16x4(float a, float b, float *c4);
}  // namespace emu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { emu::block_barrier(); }

// mode 0: xor, 1: down, 2: idx
static inline float __shfl_xor(float v, int mask, int width = 64) { return emu::wave_exchange(v, mask, 0, width); }
static inline float __shfl_down(float v, int d, int width = 64) { return emu::wave_exchange(v, d, 1, width); }
static inline float __shfl(float v, int lane, int width = 64) { return emu::wave_exchange(v, lane, 2, width); }
static inline int __shfl_xor(int v, int mask, int width = 64) {
    float f;
    memcpy(&f, &v, 4);
    f = emu::wave_exchange(f, mask, 0, width);
    memcpy(&v, &f, 4);
    return v;
}
static inline int __shfl_down(int v, int d, int width = 64) {
    float f;
    memcpy(&f, &v, 4);
    f = emu::wave_exchange(f, d, 1, width);
    memcpy(&v, &f, 4);
    return v;
}
static inline int __shfl(int v, int lane, int width = 64) {
    float f;
    memcpy(&f, &v, 4);
    f = emu::wave_exchange(f, lane, 2, width);
    memcpy(&v, &f, 4);
    return v;
}

// work-items of a workgroup run one at a time between yield points, so plain RMW is atomic
template <class T>
static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T>
static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T>
static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T>
static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T>
static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
    float t[16];
    for (int i = 0; i < 16; ++i) t[i] = c[i];
    emu::mfma_32x32x2(a, b, t);
    for (int i = 0; i < 16; ++i) c[i] = t[i];
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15 row=4(l>>4)+r, k-ordered fmaf chain
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
    float t[4];
    for (int i = 0; i < 4; ++i) t[i] = c[i];
    emu::mfma_16x16x4(a, b, t);
    for (int i = 0; i < 4; ++i) c[i] = t[i];
    return c;
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
// LDS-DMA: every lane copies `size` bytes from its own global address to (wave-uniform base + lane*size)
static inline void __builtin_amdgcn_global_load_lds(const void __attribute__((address_space(1))) * g,
                                                    void __attribute__((address_space(3))) * l, unsigned size,
                                                    int offset, unsigned) {
    const char *src = (const char *)(const void *)g + offset;
    char *dst = (char *)(void *)l + (threadIdx.x & 63) * size;
    memcpy(dst, src, size);
}
// a wave runs in lockstep on the hardware: when its s_waitcnt returns, every lane's loads (LDS-DMA pieces included)
// have landed. The fibers of a wave meet here (all call sites are wave-uniform).
static inline void __builtin_amdgcn_s_waitcnt(int) { (void)emu::wave_exchange(0.f, 0, 0, 64); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_s_barrier() { emu::block_barrier(); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }

// Optimiser step over flat fp32 buffers. The reference trains every parameter,
// the dense entity tables included, with torch.optim.Adam or torch.optim.SGD at their defaults
// (reference train.py:83-88: `optim.Adam(params, lr=lr)` / `optim.SGD(params, lr=lr, momentum=0)`).
// The fused step already keeps every gradient in ONE flat buffer (the data-parallel bucket); with the
// parameters and the Adam moments flattened the same way the whole update is one HBM-bound launch:
// 16 bytes read + 12 written per element.
#include "common.h"

// torch.optim.Adam (amsgrad = False, maximize = False), step t >= 1:
//   g += wd * p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float omb1, float b2, float omb2, float eps,
                                         float wd, float step_size, float sqrt_bc2) {
    // the same association as torch's kernels: lerp for m, addcmul for v, a division by sqrt(1 - b2^t)
    g += wd * p;
    m = fmaf(omb1, g - m, m);
    v = b2 * v + omb2 * (g * g);
    const float denom = sqrtf(v) / sqrt_bc2 + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, long long n,
                                                   float omb1, float b2, float omb2, float eps, float wd,
                                                   float step_size, float sqrt_bc2, int vec) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    if (vec) {
        for (long long i = tid * 4; i < n; i += stride * 4) {
            if (i + 3 < n) {
                f32x4 pp = *reinterpret_cast<f32x4 *>(p + i), mm = *reinterpret_cast<f32x4 *>(m + i);
                f32x4 vv = *reinterpret_cast<f32x4 *>(v + i);
                const f32x4 gg = *reinterpret_cast<const f32x4 *>(g + i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {       // (vector elements do not bind to references)
                    float pk = pp[k], mk = mm[k], vk = vv[k];
                    adam_one(pk, gg[k], mk, vk, omb1, b2, omb2, eps, wd, step_size, sqrt_bc2);
                    pp[k] = pk;
                    mm[k] = mk;
                    vv[k] = vk;
                }
                *reinterpret_cast<f32x4 *>(p + i) = pp;
                *reinterpret_cast<f32x4 *>(m + i) = mm;
                *reinterpret_cast<f32x4 *>(v + i) = vv;
            } else {
                for (long long q = i; q < n; ++q) adam_one(p[q], g[q], m[q], v[q], omb1, b2, omb2, eps, wd, step_size, sqrt_bc2);
            }
        }
    } else {
        for (long long i = tid; i < n; i += stride) adam_one(p[i], g[i], m[i], v[i], omb1, b2, omb2, eps, wd, step_size, sqrt_bc2);
    }
}

// torch.optim.SGD (momentum = 0, dampening = 0, nesterov = False):  p -= lr * (g + wd * p)
__global__ __launch_bounds__(256) void sgd_kernel(float *__restrict__ p, const float *__restrict__ g, long long n,
                                                  float lr, float wd) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    for (long long i = tid; i < n; i += stride) p[i] -= lr * (g[i] + wd * p[i]);
}

static unsigned optim_grid(int64_t n, int per_thread) {
    long long blocks = (n + 256ll * per_thread - 1) / (256ll * per_thread);
    if (blocks > 256 * 16) blocks = 256 * 16;        // grid-stride beyond 16 workgroups per CU
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

extern "C" int mpqe_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                              double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                              void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) return MPQE_ERR_INVALID_ARG;
    if (!(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1.)) return MPQE_ERR_INVALID_ARG;
    // bias corrections in double on the host, as torch does with python floats
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), sqrt_bc2 = (float)sqrt(bc2);
    // 1 - beta in double first (python floats in torch), then rounded once
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    const int vec = ((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) &&
                    ((uintptr_t)exp_avg_sq % 16 == 0);
    hipLaunchKernelGGL(adam_kernel, dim3(optim_grid(n, 4)), dim3(256), 0, as_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, (long long)n, omb1, (float)beta2, omb2, (float)eps, (float)weight_decay, step_size, sqrt_bc2, vec);
    return mpqe_launch_status();
}

extern "C" int mpqe_sgd_step(float *param, const float *grad, int64_t n, double lr, double weight_decay, void *stream) {
    if (!param || !grad || n <= 0) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sgd_kernel, dim3(optim_grid(n, 1)), dim3(256), 0, as_stream(stream), param, grad, (long long)n,
                       (float)lr, (float)weight_decay);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ negative sampling
// SURVEY.md 8f-2. reference model.py:466-476 draws ONE negative per query with python's random.choice
// from query.neg_samples / query.hard_neg_samples (ragged, per query) or, for 1-chain queries, from
// graph.full_lists[target_mode] (one list for every query). Here the candidate lists sit in HBM as CSR and
// the draw is a counter-based hash of (seed, position in the batch): no state, reproducible on the CPU
// (oracle/ref_cpu.py: sample_negatives). It is NOT python's Mersenne-Twister stream: same distribution
// (uniform over the candidates), different numbers.
__host__ __device__ __forceinline__ unsigned long long mpqe_mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;            // splitmix64
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void sample_negatives_kernel(const long long *__restrict__ cand,
                                                               const long long *__restrict__ offsets,
                                                               const long long *__restrict__ qidx, long long n_cand,
                                                               long long n_lists, long long nq, unsigned long long seed,
                                                               long long *__restrict__ out, int32_t *err) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    long long lo = 0, hi = n_cand;
    if (offsets) {
        const long long q = qidx ? qidx[i] : i;
        if (q < 0 || q >= n_lists) {
            flag_error(err, MPQE_FLAG_BAD_INDEX);
            out[i] = -1;
            return;
        }
        lo = offsets[q];
        hi = offsets[q + 1];
    }
    if (hi <= lo || lo < 0 || hi > n_cand) {      // random.choice([]) raises IndexError in the reference
        flag_error(err, MPQE_FLAG_BAD_INDEX);
        out[i] = -1;
        return;
    }
    const unsigned long long r = mpqe_mix64(seed ^ mpqe_mix64((unsigned long long)i));
    out[i] = cand[lo + (long long)(r % (unsigned long long)(hi - lo))];
}

extern "C" int mpqe_sample_negatives(const int64_t *cand, int64_t n_cand, const int64_t *offsets, int64_t n_lists,
                                     const int64_t *qidx, int64_t nq, uint64_t seed, int64_t *out, int32_t *err,
                                     void *stream) {
    if (!cand || !out || n_cand < 0 || nq <= 0) return MPQE_ERR_INVALID_ARG;
    if (!offsets && qidx) return MPQE_ERR_INVALID_ARG;
    if (offsets && n_lists <= 0) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sample_negatives_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const long long *>(cand), reinterpret_cast<const long long *>(offsets),
                       reinterpret_cast<const long long *>(qidx), (long long)n_cand, (long long)n_lists, (long long)nq,
                       (unsigned long long)seed, reinterpret_cast<long long *>(out), err);
    return mpqe_launch_status();
}

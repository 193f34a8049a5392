// Learned readouts of the fused step (level form): reference MLPReadout / TargetMLPReadout, model.py:497-553, and the
// `concat` input of model.py:441-446. The two Linear layers run on the dense-layer kernels (dense.hip: mpqe_linear_fwd /
// bwd); this file holds what sits around them, all of it HBM-bound row traffic:
//   gather   X[m] = the readout's input row m        mlp: final state of node row m; concat: the states of levels 1 .. L side
//                                                    by side; targetmlp: [target state | non-target state] of pair m
//   reduce   q[g] = add / mean / max over graph g's   (torch_scatter semantics: max keeps the LOWEST row on ties -- strict > --,
//            rows of Y (its N, or N - 1 pair, rows)   mean divides by the row count): INSIDE the score kernel (step.hip:
//            step_score_kernel, RY / RGY), which reads the rows where it used to read the node states and writes the rows'
//            gradients where it used to write the states' -- no launch, no buffer for the embedding
//   spread   gX -> rows of the state-gradient levels  (targetmlp: the target's row gets the sum of its pairs' first halves)
// A graph's rows are consecutive in every layout (graph-major node rows: row_off + g N + n; pairs: pair_off + g (N - 1) + k
// with pair_off = row_off - g_off), so the scatter is a reduction over <= 4 consecutive rows: no index vector, no atomics.
#pragma once

struct RoArgs {
    int kind;                // MPQE_READOUT_MLP / _TARGETMLP / _CONCAT
    int op;                  // MPQE_SCATTER_ADD / _MAX / _MEAN
    long long mrows;         // rows of X / Y
    int kin;                 // columns of X
    long long level_stride;  // floats between the levels of H / GH
};

__device__ __forceinline__ int ro_batch_of_row(const StepDev *__restrict__ sd, long long m, bool pairs) {
    int bi = 0;
    for (int i = 1; i < sd->nb; ++i) {
        const long long o = pairs ? sd->b[i].row_off - sd->b[i].g_off : sd->b[i].row_off;
        if (o <= m) bi = i;
    }
    return bi;
}

// one thread per (row m of X, 4 columns)
__global__ __launch_bounds__(256) void step_ro_gather_kernel(const StepDev *__restrict__ sd, RoArgs ra,
                                                             const float *__restrict__ H, float *__restrict__ X) {
    const int D = sd->D, q = ra.kin / 4;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= ra.mrows * q) return;
    const long long m = t / q;
    const int c = (int)(t % q) * 4;
    const bool pairs = ra.kind == MPQE_READOUT_TARGETMLP;
    const BatchDev &b = sd->b[ro_batch_of_row(sd, m, pairs)];
    const float *src;
    if (ra.kind == MPQE_READOUT_MLP) src = H + (long long)b.L * ra.level_stride + m * D + c;
    else if (ra.kind == MPQE_READOUT_CONCAT) src = H + (long long)(c / D + 1) * ra.level_stride + m * D + c % D;
    else {
        const int N = b.tp.N, A = b.A;
        const long long lm = m - (b.row_off - b.g_off), g = lm / (N - 1);
        const int k = (int)(lm % (N - 1)), n = c < D ? A : (k < A ? k : k + 1);
        src = H + (long long)b.L * ra.level_stride + (b.row_off + g * N + n) * D + (c < D ? c : c - D);
    }
    *reinterpret_cast<f32x4 *>(X + m * ra.kin + c) = *reinterpret_cast<const f32x4 *>(src);
}

// one thread per (node row, 4 columns of D): the gradient rows of the levels the readout read
__global__ __launch_bounds__(256) void step_ro_spread_kernel(const StepDev *__restrict__ sd, RoArgs ra,
                                                             const float *__restrict__ GX, float *__restrict__ GH) {
    const int D = sd->D, q = D / 4;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= sd->rows_total * q) return;
    const long long row = t / q;
    const int c = (int)(t % q) * 4;
    const BatchDev &b = sd->b[ro_batch_of_row(sd, row, false)];
    if (ra.kind == MPQE_READOUT_MLP) {
        *reinterpret_cast<f32x4 *>(GH + (long long)b.L * ra.level_stride + row * D + c) =
            *reinterpret_cast<const f32x4 *>(GX + row * D + c);
    } else if (ra.kind == MPQE_READOUT_CONCAT) {
        for (int p = 0; p < b.L; ++p)
            *reinterpret_cast<f32x4 *>(GH + (long long)(p + 1) * ra.level_stride + row * D + c) =
                *reinterpret_cast<const f32x4 *>(GX + row * ra.kin + p * D + c);
    } else {
        const int N = b.tp.N, A = b.A;
        const long long lr = row - b.row_off, g = lr / N, p0 = (b.row_off - b.g_off) + g * (N - 1);
        const int n = (int)(lr % N);
        f32x4 v;
        if (n == A) {             // the target: first half of each of its graph's pairs, in pair order
            v = *reinterpret_cast<const f32x4 *>(GX + p0 * ra.kin + c);
            for (int k = 1; k < N - 1; ++k) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(GX + (p0 + k) * ra.kin + c);
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] += w[u];
            }
        } else v = *reinterpret_cast<const f32x4 *>(GX + (p0 + (n < A ? n : n - 1)) * ra.kin + D + c);
        *reinterpret_cast<f32x4 *>(GH + (long long)b.L * ra.level_stride + row * D + c) = v;
    }
}

// model.py:486-490: loss += coef * sum_i ||p_i||_2 and (backward) grad_i += coef * p_i / ||p_i||. ONE workgroup walks the
// four parameters in order (fixed order of every sum: reproducible); they are a few hundred KB.
struct RoRegArgs {
    const float *p[4];
    float *g[4];
    long long n[4];
    float coef;
    float *loss;
    const float *gscale;     // != NULL: the gradients are scaled by *gscale too (an upstream gradient on the device)
    // nw > 0: the GRADIENTS' coefficient is wd * sum_i whost[i] * *wdev[i] (wdev[i] NULL: 1) instead of coef -- per-batch upstream
    // gradients on the device (mpqe_step_extra_t.batch_weight); the loss term keeps coef
    int nw;
    float wd;
    float whost[MPQE_STEP_MAX_BATCHES];
    const float *wdev[MPQE_STEP_MAX_BATCHES];
};
__global__ __launch_bounds__(1024) void step_ro_reg_kernel(RoRegArgs a) {
    __shared__ float part[16];
    __shared__ float total;
    float sum_norms = 0.f;
    float gcoef = a.coef;
    if (a.nw > 0) {                 // (every thread forms the same sum in the same order)
        float ws = 0.f;
        for (int i = 0; i < a.nw; ++i) ws += a.wdev[i] ? a.whost[i] * *a.wdev[i] : a.whost[i];
        gcoef = a.wd * ws;
    }
    // every thread's partial sums of all four parameters first -- 16 independent loads at a time, squared and added in index
    // order (a slot beyond the end adds + 0: the sums are what the plain loop gives, bit for bit) --, then the four
    // reductions: one memory round trip in front of them instead of one per element and parameter
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!a.p[i]) continue;
        const float *__restrict__ p = a.p[i];
        const long long n = a.n[i];
        float s = 0.f;
        for (long long k0 = threadIdx.x; k0 < n; k0 += 16 * 1024) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long k = k0 + 1024LL * j;
                v[j] = k < n ? p[k] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) s += v[j] * v[j];
        }
        ps[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!a.p[i]) continue;          // (uniform: fewer than four parameters)
        float s = wave_sum(ps[i]);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tt = 0.f;
            for (int w = 0; w < 16; ++w) tt += part[w];
            total = sqrtf(tt);
        }
        __syncthreads();
        const float nrm = total;
        sum_norms += nrm;
        if (a.g[i] && nrm > 0.f) {
            const float sc = (a.gscale ? gcoef * *a.gscale : gcoef) / nrm;
            for (long long k = threadIdx.x; k < a.n[i]; k += 1024) a.g[i][k] += sc * a.p[i][k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && a.loss) *a.loss = __builtin_fmaf(a.coef, sum_norms, *a.loss);
}

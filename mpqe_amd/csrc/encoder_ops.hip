// The HBM-bound pieces around the R-GCN layers: collation, embedding gather +
// L2 normalise, variable rows, readouts, torch_scatter-style reductions, cosine
// score and hinge loss -- forward and backward. One wave (64 lanes) owns one
// embedding row; rows are read as 16-byte vectors when dim % 4 == 0.
#include "bias_grad.h"
#include "common.h"

#define ROWS_PER_BLOCK 4   // 256 threads = 4 waves = 4 rows

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_in_block() { return threadIdx.x >> 6; }

// ------------------------------------------------------------------------------------ collation
__global__ void collate_kernel(int N, int E, int s0, int s1, int s2, int d0, int d1, int d2, long long r0,
                               long long r1, long long r2, long long B, long long *__restrict__ edge_index,
                               long long *__restrict__ edge_type, long long *__restrict__ batch) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long BE = B * E, BN = B * N;
    if (i < BE) {
        const long long b = i / E;
        const int e = (int)(i - b * E);
        const int s = e == 0 ? s0 : (e == 1 ? s1 : s2);
        const int d = e == 0 ? d0 : (e == 1 ? d1 : d2);
        const long long r = e == 0 ? r0 : (e == 1 ? r1 : r2);
        edge_index[i] = s + b * N;
        edge_index[BE + i] = d + b * N;
        edge_type[i] = r;
    }
    if (i < BN) batch[i] = i / N;
}

extern "C" int mpqe_collate_template(int query_type, int64_t B, const int64_t *et, int64_t *edge_index,
                                     int64_t *edge_type, int64_t *batch, void *stream) {
    if (query_type < 0 || query_type >= MPQE_Q_COUNT || B < 0 || !et) return MPQE_ERR_INVALID_ARG;
    if (B == 0) return MPQE_OK;
    if (!edge_index || !edge_type || !batch) return MPQE_ERR_INVALID_ARG;
    const TemplateDesc &d = kTemplates[query_type];
    const long long total = B * (d.N > d.E ? d.N : d.E);
    hipLaunchKernelGGL(collate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), d.N,
                       d.E, d.src[0], d.src[1], d.src[2], d.dst[0], d.dst[1], d.dst[2], (long long)et[0],
                       (long long)(d.E > 1 ? et[1] : 0), (long long)(d.E > 2 ? et[2] : 0), (long long)B,
                       (long long *)edge_index, (long long *)edge_type, (long long *)batch);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ embedding
__device__ __forceinline__ long long lookup_row(const long long *node_map, long long map_len, long long id,
                                                long long table_rows, int32_t *err) {
    if (!node_map) {                       // identity map: ids are table rows
        if (id < 0 || id >= table_rows) {
            flag_error(err, MPQE_FLAG_BAD_NODE_ID);
            return -1;
        }
        return id;
    }
    if (id < 0 || id >= map_len) {
        flag_error(err, MPQE_FLAG_BAD_NODE_ID);
        return -1;
    }
    const long long row = node_map[id];
    if (row < 0 || row >= table_rows) {
        flag_error(err, MPQE_FLAG_BAD_NODE_ID);
        return -1;
    }
    return row;
}

__global__ __launch_bounds__(256) void embed_l2norm_fwd_kernel(
    const float *__restrict__ table, long long table_rows, int D, const long long *__restrict__ node_map,
    long long map_len, const long long *__restrict__ ids, long long n, float *__restrict__ out,
    long long out_stride, float *__restrict__ inv_norm, int32_t *err, int vec) {
    const long long i = (long long)blockIdx.x * ROWS_PER_BLOCK + wave_in_block();
    if (i >= n) return;
    const int lane = lane_id();
    const long long row = lookup_row(node_map, map_len, ids[i], table_rows, lane == 0 ? err : nullptr);
    float *o = out + i * out_stride;
    if (row < 0) {
        for (int c = lane; c < D; c += 64) o[c] = 0.f;
        if (inv_norm && lane == 0) inv_norm[i] = 0.f;
        return;
    }
    const float nrm = row_norm_store(table + row * D, o, D, lane, vec);
    if (inv_norm && lane == 0) inv_norm[i] = 1.f / nrm;
}

extern "C" int mpqe_embed_l2norm_fwd(const float *table, int64_t table_rows, int64_t dim, const int64_t *node_map,
                                     int64_t node_map_len, const int64_t *ids, int64_t n, float *out,
                                     int64_t out_row_stride, float *inv_norm, int32_t *err, void *stream) {
    if (n < 0 || dim <= 0 || table_rows < 0 || out_row_stride < dim) return MPQE_ERR_INVALID_ARG;
    if (n == 0) return MPQE_OK;
    if (!table || !ids || !out) return MPQE_ERR_INVALID_ARG;
    const int vec = dim % 4 == 0 && out_row_stride % 4 == 0 && (uintptr_t)table % 16 == 0 && (uintptr_t)out % 16 == 0;
    hipLaunchKernelGGL(embed_l2norm_fwd_kernel, dim3((unsigned)((n + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)),
                       dim3(256), 0, as_stream(stream), table, (long long)table_rows, (int)dim,
                       (const long long *)node_map, (long long)node_map_len, (const long long *)ids, (long long)n,
                       out, (long long)out_row_stride, inv_norm, err, vec);
    return mpqe_launch_status();
}

// y = v/|v|:  dv = (g - y (y.g)) / |v|
__global__ __launch_bounds__(256) void embed_l2norm_bwd_kernel(
    const float *__restrict__ g, long long g_stride, const float *__restrict__ table, long long table_rows, int D,
    const long long *__restrict__ node_map, long long map_len, const long long *__restrict__ ids, long long n,
    float *__restrict__ grad_table, int32_t *err) {
    const long long i = (long long)blockIdx.x * ROWS_PER_BLOCK + wave_in_block();
    if (i >= n) return;
    const int lane = lane_id();
    const long long row = lookup_row(node_map, map_len, ids[i], table_rows, lane == 0 ? err : nullptr);
    if (row < 0) return;
    const float *v = table + row * D;
    const float *gi = g + i * g_stride;
    float ss = 0.f, vg = 0.f;
    for (int c = lane; c < D; c += 64) {
        ss += v[c] * v[c];
        vg += v[c] * gi[c];
    }
    ss = wave_sum(ss);
    vg = wave_sum(vg);
    const float nrm = sqrtf(ss);
    const float inv = 1.f / nrm;
    const float ydotg = vg * inv;   // y . g
    float *gt = grad_table + row * D;
    for (int c = lane; c < D; c += 64) {
        const float y = v[c] / nrm;
        atomicAdd(gt + c, (gi[c] - y * ydotg) * inv);
    }
}

extern "C" int mpqe_embed_l2norm_bwd(const float *grad_out, int64_t grad_row_stride, const float *table,
                                     int64_t table_rows, int64_t dim, const int64_t *node_map, int64_t node_map_len,
                                     const int64_t *ids, int64_t n, float *grad_table, int32_t *err, void *stream) {
    if (n < 0 || dim <= 0 || grad_row_stride < dim) return MPQE_ERR_INVALID_ARG;
    if (n == 0) return MPQE_OK;
    if (!grad_out || !table || !ids || !grad_table) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(embed_l2norm_bwd_kernel, dim3((unsigned)((n + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)),
                       dim3(256), 0, as_stream(stream), grad_out, (long long)grad_row_stride, table,
                       (long long)table_rows, (int)dim, (const long long *)node_map, (long long)node_map_len,
                       (const long long *)ids, (long long)n, grad_table, err);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ variable rows
__global__ void var_rows_fwd_kernel(const float *__restrict__ mode_emb, long long num_modes, int D,
                                    const long long *__restrict__ var_ids, int V, long long B, int N, int A,
                                    float *__restrict__ x, int32_t *err) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = B * V * D;
    if (idx >= total) return;
    const int c = (int)(idx % D);
    const long long bk = idx / D;
    const int k = (int)(bk % V);
    const long long b = bk / V;
    const long long m = var_ids[k];
    float v = 0.f;
    if (m < 0 || m >= num_modes) flag_error(err, MPQE_FLAG_BAD_NODE_ID);
    else v = mode_emb[m * D + c];
    x[(b * N + A + k) * D + c] = v;
}

extern "C" int mpqe_var_rows_fwd(const float *mode_emb, int64_t num_modes, int64_t dim, const int64_t *var_ids,
                                 int64_t V, int64_t B, int64_t N, int64_t A, float *x, int32_t *err, void *stream) {
    if (B < 0 || V < 0 || dim <= 0 || A + V != N) return MPQE_ERR_INVALID_ARG;
    if (B == 0 || V == 0) return MPQE_OK;
    if (!mode_emb || !var_ids || !x) return MPQE_ERR_INVALID_ARG;
    const long long total = B * V * dim;
    hipLaunchKernelGGL(var_rows_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       mode_emb, (long long)num_modes, (int)dim, (const long long *)var_ids, (int)V, (long long)B,
                       (int)N, (int)A, x, err);
    return mpqe_launch_status();
}

// grad_mode_emb[var_ids[k]] += sum_b grad_x[b*N + A + k]; slot k also takes later slots with the
// same mode id so that no two workgroups touch one row (fixed summation order, no atomics).
__global__ __launch_bounds__(256) void var_rows_bwd_kernel(const float *__restrict__ gx, long long num_modes, int D,
                                                           const long long *__restrict__ var_ids, int V,
                                                           long long B, int N, int A,
                                                           float *__restrict__ grad_mode, int32_t *err) {
    __shared__ float part[4][64];
    const int k = blockIdx.y;
    const long long m = var_ids[k];
    if (m < 0 || m >= num_modes) {
        if (threadIdx.x == 0) flag_error(err, MPQE_FLAG_BAD_NODE_ID);
        return;
    }
    for (int kk = 0; kk < k; ++kk)
        if (var_ids[kk] == m) return;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < D) {
        for (int kk = k; kk < V; ++kk) {
            if (var_ids[kk] != m) continue;
            for (long long b = rg; b < B; b += 4) s += gx[(b * N + A + kk) * D + c];
        }
    }
    part[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < D) grad_mode[m * D + c] += (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

extern "C" int mpqe_var_rows_bwd(const float *grad_x, int64_t num_modes, int64_t dim, const int64_t *var_ids,
                                 int64_t V, int64_t B, int64_t N, int64_t A, float *grad_mode_emb, int32_t *err,
                                 void *stream) {
    if (B < 0 || V < 0 || dim <= 0 || A + V != N) return MPQE_ERR_INVALID_ARG;
    if (B == 0 || V == 0) return MPQE_OK;
    if (!grad_x || !var_ids || !grad_mode_emb) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(var_rows_bwd_kernel, dim3((unsigned)((dim + 63) / 64), (unsigned)V), dim3(256), 0,
                       as_stream(stream), grad_x, (long long)num_modes, (int)dim, (const long long *)var_ids, (int)V,
                       (long long)B, (int)N, (int)A, grad_mode_emb, err);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ readouts (regular batches)
__global__ void readout_fwd_kernel(int kind, const float *__restrict__ h, long long B, int N, int A, int D,
                                   float *__restrict__ out, int32_t *__restrict__ argmax) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const long long b = idx / D;
    const int c = (int)(idx - b * D);
    const float *p = h + (b * N) * D + c;
    if (kind == MPQE_READOUT_TM) {
        out[idx] = p[(long long)A * D];
    } else if (kind == MPQE_READOUT_SUM) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += p[(long long)n * D];
        out[idx] = s;
    } else {
        float best = p[0];
        int arg = 0;
        for (int n = 1; n < N; ++n) {
            const float v = p[(long long)n * D];
            if (v > best) {
                best = v;
                arg = n;
            }
        }
        out[idx] = best;
        if (argmax) argmax[idx] = arg;
    }
}

extern "C" int mpqe_readout_fwd(int kind, const float *h, int64_t B, int64_t N, int64_t A, int64_t dim, float *out,
                                int32_t *argmax, void *stream) {
    if (kind < 0 || kind > 2 || B < 0 || N <= 0 || dim <= 0 || A < 0 || A >= N) return MPQE_ERR_INVALID_ARG;
    if (B == 0) return MPQE_OK;
    if (!h || !out) return MPQE_ERR_INVALID_ARG;
    const long long total = B * dim;
    hipLaunchKernelGGL(readout_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       kind, h, (long long)B, (int)N, (int)A, (int)dim, out, argmax);
    return mpqe_launch_status();
}

__global__ void readout_bwd_kernel(int kind, const float *__restrict__ g, const int32_t *__restrict__ argmax,
                                   long long B, int N, int A, int D, float *__restrict__ gh) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N * D) return;
    const int c = (int)(idx % D);
    const long long bn = idx / D;
    const int n = (int)(bn % N);
    const long long b = bn / N;
    const float gv = g[b * D + c];
    float v;
    if (kind == MPQE_READOUT_SUM) v = gv;
    else if (kind == MPQE_READOUT_TM) v = (n == A) ? gv : 0.f;
    else v = (argmax[b * D + c] == n) ? gv : 0.f;
    gh[idx] = v;
}

extern "C" int mpqe_readout_bwd(int kind, const float *grad_out, const int32_t *argmax, int64_t B, int64_t N,
                                int64_t A, int64_t dim, float *grad_h, void *stream) {
    if (kind < 0 || kind > 2 || B < 0 || N <= 0 || dim <= 0 || A < 0 || A >= N) return MPQE_ERR_INVALID_ARG;
    if (B == 0) return MPQE_OK;
    if (!grad_out || !grad_h || (kind == MPQE_READOUT_MAX && !argmax)) return MPQE_ERR_INVALID_ARG;
    const long long total = B * N * dim;
    hipLaunchKernelGGL(readout_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       kind, grad_out, argmax, (long long)B, (int)N, (int)A, (int)dim, grad_h);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ torch_scatter-style reductions
// Any index order. add/mean accumulate with fp32 atomics; max uses the ordered-int trick.
__device__ __forceinline__ void atomic_max_float(float *addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

__global__ void scatter_init_kernel(int op, long long total, long long dim_size, float *out, long long *arg,
                                    float *count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        out[i] = (op == MPQE_SCATTER_MAX) ? -INFINITY : 0.f;
        if (arg) arg[i] = 0x7fffffffffffffffLL;
    }
    if (count && i < dim_size) count[i] = 0.f;
}

__global__ void scatter_accum_kernel(int op, const float *__restrict__ src, const long long *__restrict__ index,
                                     long long n_src, int D, long long dim_size, float *out, float *count,
                                     int32_t *err) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_src * D) return;
    const long long j = i / D;
    const int c = (int)(i - j * D);
    const long long t = index[j];
    if (t < 0 || t >= dim_size) {
        flag_error(err, MPQE_FLAG_BAD_INDEX);
        return;
    }
    if (op == MPQE_SCATTER_MAX) atomic_max_float(out + t * D + c, src[i]);
    else atomicAdd(out + t * D + c, src[i]);
    if (count && c == 0) atomicAdd(count + t, 1.f);
}

__global__ void scatter_arg_kernel(const float *__restrict__ src, const long long *__restrict__ index, long long n_src,
                                   int D, long long dim_size, const float *__restrict__ out, long long *arg) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_src * D) return;
    const long long j = i / D;
    const int c = (int)(i - j * D);
    const long long t = index[j];
    if (t < 0 || t >= dim_size) return;
    if (src[i] == out[t * D + c]) atomicMin(arg + t * D + c, j);
}

__global__ void scatter_finish_kernel(int op, long long total, int D, float *out, long long *arg,
                                      const float *__restrict__ count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (op == MPQE_SCATTER_MEAN) {
        const float cn = count[i / D];
        out[i] = out[i] / (cn < 1.f ? 1.f : cn);
    } else if (op == MPQE_SCATTER_MAX) {
        if (arg) {
            if (arg[i] == 0x7fffffffffffffffLL) {
                arg[i] = -1;
                out[i] = 0.f;
            }
        } else if (out[i] == -INFINITY) {
            out[i] = 0.f;
        }
    }
}

extern "C" size_t mpqe_scatter_workspace_bytes(int64_t n_src, int64_t dim_size) {
    (void)n_src;
    return align_up((size_t)(dim_size > 0 ? dim_size : 1) * 4, 256);
}

extern "C" int mpqe_scatter_fwd(int op, const float *src, const int64_t *index, int64_t n_src, int64_t dim,
                                int64_t dim_size, float *out, int64_t *arg, void *workspace, size_t workspace_bytes,
                                int32_t *err, void *stream) {
    if (op < 0 || op > 2 || n_src < 0 || dim <= 0 || dim_size < 0) return MPQE_ERR_INVALID_ARG;
    if (dim_size == 0) return MPQE_OK;
    if (!out || (n_src > 0 && (!src || !index))) return MPQE_ERR_INVALID_ARG;
    float *count = nullptr;
    if (op == MPQE_SCATTER_MEAN) {
        if (!workspace || workspace_bytes < mpqe_scatter_workspace_bytes(n_src, dim_size)) return MPQE_ERR_WORKSPACE;
        count = reinterpret_cast<float *>(workspace);
    }
    if (op != MPQE_SCATTER_MAX) arg = nullptr;
    hipStream_t s = as_stream(stream);
    const long long total = dim_size * dim;
    const long long tmax = total > dim_size ? total : dim_size;
    hipLaunchKernelGGL(scatter_init_kernel, dim3((unsigned)((tmax + 255) / 256)), dim3(256), 0, s, op, total,
                       (long long)dim_size, out, (long long *)arg, count);
    const long long work = n_src * dim;
    if (work > 0) {
        hipLaunchKernelGGL(scatter_accum_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, op, src,
                           (const long long *)index, (long long)n_src, (int)dim, (long long)dim_size, out, count,
                           err);
        if (arg)
            hipLaunchKernelGGL(scatter_arg_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, src,
                               (const long long *)index, (long long)n_src, (int)dim, (long long)dim_size, out,
                               (long long *)arg);
    }
    if (op != MPQE_SCATTER_ADD)
        hipLaunchKernelGGL(scatter_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, op, total,
                           (int)dim, out, (long long *)arg, count);
    return mpqe_launch_status();
}

__global__ void scatter_count_kernel(const long long *__restrict__ index, long long n_src, long long dim_size,
                                     float *count) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_src) return;
    const long long t = index[j];
    if (t >= 0 && t < dim_size) atomicAdd(count + t, 1.f);
}

__global__ void scatter_bwd_kernel(int op, const float *__restrict__ g, const long long *__restrict__ index,
                                   const long long *__restrict__ arg, const float *__restrict__ count,
                                   long long n_src, int D, long long dim_size, float *__restrict__ gs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_src * D) return;
    const long long j = i / D;
    const int c = (int)(i - j * D);
    const long long t = index[j];
    float v = 0.f;
    if (t >= 0 && t < dim_size) {
        const float gv = g[t * D + c];
        if (op == MPQE_SCATTER_ADD) v = gv;
        else if (op == MPQE_SCATTER_MEAN) {
            const float cn = count[t];
            v = gv / (cn < 1.f ? 1.f : cn);
        } else v = (arg[t * D + c] == j) ? gv : 0.f;
    }
    gs[i] = v;
}

extern "C" int mpqe_scatter_bwd(int op, const float *grad_out, const int64_t *index, const int64_t *arg,
                                int64_t n_src, int64_t dim, int64_t dim_size, float *grad_src, void *workspace,
                                size_t workspace_bytes, void *stream) {
    if (op < 0 || op > 2 || n_src < 0 || dim <= 0 || dim_size < 0) return MPQE_ERR_INVALID_ARG;
    if (n_src == 0) return MPQE_OK;
    if (!grad_out || !index || !grad_src || (op == MPQE_SCATTER_MAX && !arg)) return MPQE_ERR_INVALID_ARG;
    hipStream_t s = as_stream(stream);
    float *count = nullptr;
    if (op == MPQE_SCATTER_MEAN) {
        if (!workspace || workspace_bytes < mpqe_scatter_workspace_bytes(n_src, dim_size)) return MPQE_ERR_WORKSPACE;
        count = reinterpret_cast<float *>(workspace);
        hipLaunchKernelGGL(scatter_init_kernel, dim3((unsigned)((dim_size + 255) / 256)), dim3(256), 0, s, 0,
                           (long long)0, (long long)dim_size, (float *)nullptr, (long long *)nullptr, count);
        hipLaunchKernelGGL(scatter_count_kernel, dim3((unsigned)((n_src + 255) / 256)), dim3(256), 0, s,
                           (const long long *)index, (long long)n_src, (long long)dim_size, count);
    }
    const long long work = n_src * dim;
    hipLaunchKernelGGL(scatter_bwd_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, op, grad_out,
                       (const long long *)index, (const long long *)arg, count, (long long)n_src, (int)dim,
                       (long long)dim_size, grad_src);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ cosine score
__global__ __launch_bounds__(256) void cosine_fwd_kernel(const float *__restrict__ q,
                                                         const long long *__restrict__ q_row,
                                                         const float *__restrict__ t, long long n, int D, float eps,
                                                         float *__restrict__ scores) {
    const long long i = (long long)blockIdx.x * ROWS_PER_BLOCK + wave_in_block();
    if (i >= n) return;
    const int lane = lane_id();
    const float *qi = q + (q_row ? q_row[i] : i) * D;
    const float *ti = t + i * D;
    float dot = 0.f, qq = 0.f, tt = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float a = qi[c], b = ti[c];
        dot += a * b;
        qq += a * a;
        tt += b * b;
    }
    dot = wave_sum(dot);
    qq = wave_sum(qq);
    tt = wave_sum(tt);
    const float nq = fmaxf(sqrtf(qq), eps), nt = fmaxf(sqrtf(tt), eps);
    if (lane == 0) scores[i] = dot / (nq * nt);
}

extern "C" int mpqe_cosine_fwd(const float *q, const int64_t *q_row, const float *t, int64_t n, int64_t dim,
                               float eps, float *scores, void *stream) {
    if (n < 0 || dim <= 0) return MPQE_ERR_INVALID_ARG;
    if (n == 0) return MPQE_OK;
    if (!q || !t || !scores) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(cosine_fwd_kernel, dim3((unsigned)((n + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0,
                       as_stream(stream), q, (const long long *)q_row, t, (long long)n, (int)dim, eps, scores);
    return mpqe_launch_status();
}

// s = q.t / (nq nt), nq = max(|q|, eps):  ds/dq = t/(nq nt) - s q/nq^2 (second term only if |q| > eps)
__global__ __launch_bounds__(256) void cosine_bwd_kernel(const float *__restrict__ gs, const float *__restrict__ q,
                                                         const long long *__restrict__ q_row,
                                                         const float *__restrict__ t, long long n, int D, float eps,
                                                         float *grad_q, float *__restrict__ grad_t) {
    const long long i = (long long)blockIdx.x * ROWS_PER_BLOCK + wave_in_block();
    if (i >= n) return;
    const int lane = lane_id();
    const long long qr = q_row ? q_row[i] : i;
    const float *qi = q + qr * D;
    const float *ti = t + i * D;
    float dot = 0.f, qq = 0.f, tt = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float a = qi[c], b = ti[c];
        dot += a * b;
        qq += a * a;
        tt += b * b;
    }
    dot = wave_sum(dot);
    qq = wave_sum(qq);
    tt = wave_sum(tt);
    const float rq = sqrtf(qq), rt = sqrtf(tt);
    const float nq = fmaxf(rq, eps), nt = fmaxf(rt, eps);
    const float inv = 1.f / (nq * nt);
    const float s = dot * inv;
    const float g = gs[i];
    const float kq = rq > eps ? s / (nq * nq) : 0.f;
    const float kt = rt > eps ? s / (nt * nt) : 0.f;
    for (int c = lane; c < D; c += 64) {
        const float a = qi[c], b = ti[c];
        if (grad_q) {
            const float v = g * (b * inv - kq * a);
            if (q_row) atomicAdd(grad_q + qr * D + c, v);
            else grad_q[qr * D + c] = v;
        }
        if (grad_t) grad_t[i * D + c] = g * (a * inv - kt * b);
    }
}

extern "C" int mpqe_cosine_bwd(const float *grad_scores, const float *q, const int64_t *q_row, const float *t,
                               int64_t n, int64_t dim, float eps, float *grad_q, float *grad_t, void *stream) {
    if (n < 0 || dim <= 0) return MPQE_ERR_INVALID_ARG;
    if (n == 0) return MPQE_OK;
    if (!grad_scores || !q || !t) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(cosine_bwd_kernel, dim3((unsigned)((n + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0,
                       as_stream(stream), grad_scores, q, (const long long *)q_row, t, (long long)n, (int)dim, eps,
                       grad_q, grad_t);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ hinge loss
__global__ __launch_bounds__(256) void hinge_fwd_kernel(const float *__restrict__ pos, const float *__restrict__ neg,
                                                        long long n, float margin, float *__restrict__ loss) {
    __shared__ float red[256];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const float v = margin - (pos[i] - neg[i]);
        s += v > 0.f ? v : 0.f;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)n;
}

extern "C" int mpqe_hinge_fwd(const float *pos, const float *neg, int64_t n, float margin, float *loss,
                              void *stream) {
    if (n <= 0 || !pos || !neg || !loss) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(hinge_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), pos, neg, (long long)n, margin,
                       loss);
    return mpqe_launch_status();
}

__global__ void hinge_bwd_kernel(const float *__restrict__ pos, const float *__restrict__ neg, long long n,
                                 float margin, const float *__restrict__ gl, float *__restrict__ gpos,
                                 float *__restrict__ gneg) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = margin - (pos[i] - neg[i]);
    const float g = v >= 0.f ? gl[0] / (float)n : 0.f;   // torch clamp passes the gradient at equality
    if (gpos) gpos[i] = -g;
    if (gneg) gneg[i] = g;
}

extern "C" int mpqe_hinge_bwd(const float *pos, const float *neg, int64_t n, float margin, const float *grad_loss,
                              float *grad_pos, float *grad_neg, void *stream) {
    if (n <= 0 || !pos || !neg || !grad_loss) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(hinge_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), pos, neg,
                       (long long)n, margin, grad_loss, grad_pos, grad_neg);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ (a9) LayerNorm + ReLU
// reference encoders.py:132-146 (LayerNorm of the GraphSAGE-style Encoder: UNBIASED standard deviation, eps added to the
// standard deviation, not the variance) followed by the Encoder's ReLU (encoders.py:127-128), one wave per row:
//   y = act(gamma * (x - mean) / (std + eps) + beta),  std = sqrt(sum (x - mean)^2 / (D - 1))
// stats[row] = {mean, 1 / (std + eps)} for the backward.
__global__ __launch_bounds__(256) void layernorm_relu_fwd_kernel(const float *__restrict__ x, long long rows, int D,
                                                                 const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float eps, int relu,
                                                                 float *__restrict__ y, float *__restrict__ stats) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *xr = x + r * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float d = xr[c] - mean;
        q += d * d;
    }
    const float sd = sqrtf(wave_sum(q) / (float)(D - 1));
    const float inv = 1.f / (sd + eps);
    for (int c = lane; c < D; c += 64) {
        float v = gamma[c] * ((xr[c] - mean) * inv) + beta[c];
        if (relu) v = v > 0.f ? v : 0.f;
        y[r * D + c] = v;
    }
    if (lane == 0) {
        stats[2 * r] = mean;
        stats[2 * r + 1] = inv;
    }
}

// grad_x, and per row the two summands of the parameter gradients (gg = g * xhat for gamma, gb = g for beta, g = grad_y
// masked by the ReLU): their column sums over the rows are formed afterwards in a fixed order (bias_grad.h).
__global__ __launch_bounds__(256) void layernorm_relu_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ x,
                                                                 const float *__restrict__ y, long long rows, int D,
                                                                 const float *__restrict__ gamma,
                                                                 const float *__restrict__ stats, float eps, int relu,
                                                                 float *__restrict__ gx, float *__restrict__ gg,
                                                                 float *__restrict__ gb) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float mean = stats[2 * r], inv = stats[2 * r + 1];
    const float sd = 1.f / inv - eps;
    float sum_d = 0.f, sum_dx = 0.f;
    for (int c = lane; c < D; c += 64) {
        float g = gy[r * D + c];
        if (relu && !(y[r * D + c] > 0.f)) g = 0.f;
        const float xc = x[r * D + c] - mean;
        const float d = g * gamma[c];           // d loss / d xhat
        gg[r * D + c] = g * (xc * inv);
        gb[r * D + c] = g;
        sum_d += d;
        sum_dx += d * xc;
    }
    sum_d = wave_sum(sum_d);
    sum_dx = wave_sum(sum_dx);
    // xhat_i = (x_i - mean) / s, s = std + eps:  d xhat_i / d x_j = (delta_ij - 1/D) / s - (x_i - mean)(x_j - mean) / ((D-1) std s^2)
    const float k = sd > 0.f ? sum_dx * inv * inv / ((float)(D - 1) * sd) : 0.f;
    for (int c = lane; c < D; c += 64) {
        float g = gy[r * D + c];
        if (relu && !(y[r * D + c] > 0.f)) g = 0.f;
        const float xc = x[r * D + c] - mean;
        gx[r * D + c] = (g * gamma[c] - sum_d / (float)D) * inv - xc * k;
    }
}

extern "C" int mpqe_layernorm_relu_fwd(const float *x, int64_t rows, int64_t dim, const float *gamma, const float *beta,
                                       float eps, int relu, float *y, float *stats, void *stream) {
    if (rows < 0 || dim < 2 || !gamma || !beta) return MPQE_ERR_INVALID_ARG;
    if (rows == 0) return MPQE_OK;
    if (!x || !y || !stats) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(layernorm_relu_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x,
                       (long long)rows, (int)dim, gamma, beta, eps, relu, y, stats);
    return mpqe_launch_status();
}

extern "C" size_t mpqe_layernorm_relu_bwd_workspace_bytes(int64_t rows, int64_t dim) {
    if (rows <= 0 || dim <= 0) return 0;
    return 2 * align_up((size_t)rows * (size_t)dim * 4, 256) + bias_partial_bytes(rows, dim) + 256;
}

extern "C" int mpqe_layernorm_relu_bwd(const float *grad_y, const float *x, const float *y, int64_t rows, int64_t dim,
                                       const float *gamma, const float *stats, float eps, int relu, float *grad_x,
                                       float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    if (rows < 0 || dim < 2 || !gamma) return MPQE_ERR_INVALID_ARG;
    if (rows == 0) return MPQE_OK;
    if (!grad_y || !x || !y || !stats || !grad_x || !grad_gamma || !grad_beta || !workspace) return MPQE_ERR_INVALID_ARG;
    if (workspace_bytes < mpqe_layernorm_relu_bwd_workspace_bytes(rows, dim)) return MPQE_ERR_WORKSPACE;
    char *wb = reinterpret_cast<char *>(workspace);
    const size_t plane = align_up((size_t)rows * (size_t)dim * 4, 256);
    float *gg = reinterpret_cast<float *>(wb), *gb = reinterpret_cast<float *>(wb + plane);
    float *part = reinterpret_cast<float *>(wb + 2 * plane);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(layernorm_relu_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, grad_y, x, y,
                       (long long)rows, (int)dim, gamma, stats, eps, relu, grad_x, gg, gb);
    // grad_gamma / grad_beta += column sums, fixed order (the caller zero-fills them: torch accumulates gradients)
    launch_bias_grad((long long)rows, gg, (const float *)nullptr, (int)dim, 0, part, grad_gamma, s);
    launch_bias_grad((long long)rows, gb, (const float *)nullptr, (int)dim, 0, part, grad_beta, s);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ misc
extern "C" const char *mpqe_status_string(int status) {
    switch (status) {
        case MPQE_OK: return "ok";
        case MPQE_ERR_INVALID_ARG: return "invalid argument";
        case MPQE_ERR_UNSUPPORTED: return "unsupported shape";
        case MPQE_ERR_WORKSPACE: return "workspace too small";
        case MPQE_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}
extern "C" int mpqe_abi_version(void) { return 6; }

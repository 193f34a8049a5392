// Dense layer y = [relu](x . W^T + bias) with W as nn.Linear stores it ([out, in], row stride ldw), forward and backward,
// on the same 64 x 64 fp32 MFMA tile core as the R-GCN layer (gemm_core.h, rgcn_template_body.h). The reference uses
// nn.Linear in its MLP readouts (model.py:497-553: Linear -> ReLU -> Linear per node) and one matrix product per
// concatenated neighbour block in Encoder.forward (encoders.py:120-124: compress[mode].mm(combined)); the row stride lets
// the compress matrix be applied block by block without materialising the concatenation.
//   forward     A = x rows (contiguous along k), B[k][n] = W[n][k]: the backward-x loader of the layer (W read in place)
//   grad_x      gpre . W, gpre = g (x) (y > 0): the forward loader with the ReLU mask on the A operand
//   grad_W      x^T . gpre with the rows as K, split into chunks over workgroups; slabs [in, out] summed in a fixed
//               order and written transposed into grad_W [out, in] (no float atomics: reproducible)
//   grad_bias   column sums of gpre (bias_grad.h)
#include "bias_grad.h"
#include "rgcn_template_body.h"

template <int MODE>
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float *__restrict__ x, long long rows,
                                                        const float *__restrict__ W, int ldw,
                                                        const float *__restrict__ bias, int din, int dout, int relu,
                                                        int accumulate, float *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) float smem[GT_SMEM_FLOATS];
    const long long b0 = (long long)blockIdx.x * GT_BM;
    const int n0 = (int)blockIdx.y * GT_BN;
    KBlocks kb;
    kblocks_init(kb, W);
    kb.add(0, W);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    LayerLoader<MODE, true> L;      // K runs over din; B[k][n] = W[n][k], tile rows = output columns
    L.init(kb, x, nullptr, W, rows, 1, din, ldw, dout, b0, n0);
    gemm_block<false, false>(acc, L, L.spb, smem);
    const int col = n0 + acc_col();
    if (col < dout) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long b = b0 + acc_row(r);
            if (b < rows) {
                float *o = y + b * (long long)dout + col;
                float v = acc[r] + bv + (accumulate ? *o : 0.f);
                if (relu) v = v > 0.f ? v : 0.f;
                *o = v;
            }
        }
    }
}

// grad_x[rows, din] = gpre . W  (W [dout, din]: K-type, rows k = output feature, ldw)
template <int MODE>
__global__ __launch_bounds__(256) void dense_bwd_x_kernel(const float *__restrict__ g, const float *__restrict__ mask,
                                                          long long rows, const float *__restrict__ W, int ldw, int din,
                                                          int dout, float *__restrict__ gx) {
    __shared__ __attribute__((aligned(16))) float smem[GT_SMEM_FLOATS];
    const long long b0 = (long long)blockIdx.x * GT_BM;
    const int n0 = (int)blockIdx.y * GT_BN;
    KBlocks kb;
    kblocks_init(kb, W);
    kb.add(0, W);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    LayerLoader<MODE, false> L;
    L.init(kb, g, mask, W, rows, 1, dout, ldw, dout, b0, n0);
    gemm_block<false, true>(acc, L, L.spb, smem);
    const int col = n0 + acc_col();
    if (col < din) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long b = b0 + acc_row(r);
            if (b < rows) gx[b * (long long)din + col] = acc[r];
        }
    }
}

// slab[c][i][j] = sum over the rows q of chunk c of x[q][i] * gpre[q][j]     ([din, dout] per chunk)
template <int MODE>
__global__ __launch_bounds__(256) void dense_grad_w_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                           const float *__restrict__ mask, long long rows, int ch,
                                                           int din, int dout, float *__restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float smem[GT_SMEM_FLOATS];
    const int tiles_j = (dout + GT_BN - 1) / GT_BN;
    const int i0 = ((int)blockIdx.y / tiles_j) * GT_BM, j0 = ((int)blockIdx.y % tiles_j) * GT_BN;
    const long long q0 = (long long)blockIdx.x * ch;
    long long q1 = q0 + ch;
    if (q1 > rows) q1 = rows;
    tmpl_grad_w_tile<MODE>(x, g, mask, din, dout, mask ? 1 : 0, 1, 0, 1, 0, q0, q1, i0, j0,
                          slabs + (long long)blockIdx.x * din * dout, smem);
}

// grad_W[j][i] (+)= sum_c slab[c][i][j], chunks in order. Consecutive threads walk j: the nch slab reads (the bulk of the
// traffic: nch x din x dout floats) are coalesced, the single transposed store is the strided one; eight loads in flight
// per thread, added in chunk order (fixed order: reproducible).
__global__ __launch_bounds__(256) void dense_reduce_w_kernel(const float *__restrict__ slabs, int nch, int din, int dout,
                                                             float *__restrict__ gW, int ldgw, int overwrite) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;      // over [din][dout]
    if (idx >= (long long)din * dout) return;
    const int i = (int)(idx / dout), j = (int)(idx % dout);
    const long long elems = (long long)din * dout;
    const float *p = slabs + idx;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= nch; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(c + u) * elems];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < nch; ++c) s += p[(long long)c * elems];
    float *o = gW + (long long)j * ldgw + i;
    *o = overwrite ? s : *o + s;
}

static void dense_chunks(int64_t rows, int *nch, int *ch) {
    int64_t n = (rows + 255) / 256;
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    int64_t c = (rows + n - 1) / n;
    c = (c + GT_BK - 1) / GT_BK * GT_BK;
    if (c < GT_BK) c = GT_BK;
    n = (rows + c - 1) / c;
    if (n < 1) n = 1;
    *nch = (int)n;
    *ch = (int)c;
}

extern "C" int mpqe_linear_fwd(const float *x, int64_t rows, const float *W, int64_t ldw, const float *bias, int64_t din,
                               int64_t dout, int relu, int accumulate, float *y, void *stream) {
    if (!x || !W || !y || rows < 0 || din <= 0 || dout <= 0 || ldw < din) return MPQE_ERR_INVALID_ARG;
    if (din > (1 << 20) || dout > (1 << 20)) return MPQE_ERR_UNSUPPORTED;
    if (rows == 0) return MPQE_OK;
    dim3 grid((unsigned)((rows + GT_BM - 1) / GT_BM), (unsigned)((dout + GT_BN - 1) / GT_BN));
    const bool vec = ptr_vec_ok(x, din) && ptr_vec_ok(W, ldw);
    hipStream_t s = as_stream(stream);
    if (vec && din % GT_BK == 0 && dout % GT_BN == 0)
        hipLaunchKernelGGL(dense_fwd_kernel<LD_FAST>, grid, dim3(256), 0, s, x, (long long)rows, W, (int)ldw, bias, (int)din,
                           (int)dout, relu, accumulate, y);
    else if (vec)
        hipLaunchKernelGGL(dense_fwd_kernel<LD_PRED>, grid, dim3(256), 0, s, x, (long long)rows, W, (int)ldw, bias, (int)din,
                           (int)dout, relu, accumulate, y);
    else
        hipLaunchKernelGGL(dense_fwd_kernel<LD_SCALAR>, grid, dim3(256), 0, s, x, (long long)rows, W, (int)ldw, bias,
                           (int)din, (int)dout, relu, accumulate, y);
    return mpqe_launch_status();
}

extern "C" size_t mpqe_linear_bwd_workspace_bytes(int64_t rows, int64_t din, int64_t dout) {
    if (rows < 0 || din <= 0 || dout <= 0) return 0;
    int nch, ch;
    dense_chunks(rows, &nch, &ch);
    return align_up((size_t)nch * (size_t)din * (size_t)dout * 4, 256) + bias_partial_bytes((long long)rows, dout) + 256;
}

extern "C" int mpqe_linear_bwd(const float *x, int64_t rows, const float *W, int64_t ldw, const float *y,
                               const float *grad_y, int64_t din, int64_t dout, int relu, int overwrite, float *grad_x,
                               float *grad_W, int64_t ldgw, float *grad_bias, void *workspace, size_t workspace_bytes,
                               void *stream) {
    if (!x || !W || !grad_y || rows < 0 || din <= 0 || dout <= 0 || ldw < din || (relu && !y)) return MPQE_ERR_INVALID_ARG;
    if (grad_W && ldgw < din) return MPQE_ERR_INVALID_ARG;
    hipStream_t s = as_stream(stream);
    if (rows == 0) {
        if (overwrite) {
            if (grad_W)      // (no rows: the column block becomes zeros -- the reduction over zero chunks)
                hipLaunchKernelGGL(dense_reduce_w_kernel, dim3((unsigned)(((long long)din * dout + 255) / 256)), dim3(256), 0, s,
                                   (const float *)nullptr, 0, (int)din, (int)dout, grad_W, (int)ldgw, 1);
            if (grad_bias) (void)hipMemsetAsync(grad_bias, 0, (size_t)dout * 4, s);
        }
        return MPQE_OK;
    }
    const float *mask = relu ? y : nullptr;
    const bool vec = ptr_vec_ok(x, din) && ptr_vec_ok(W, ldw) && ptr_vec_ok(grad_y, dout) && (!relu || ptr_vec_ok(y, dout));
    if (grad_x) {
        dim3 grid((unsigned)((rows + GT_BM - 1) / GT_BM), (unsigned)((din + GT_BN - 1) / GT_BN));
        if (vec && dout % GT_BK == 0 && din % GT_BN == 0)
            hipLaunchKernelGGL(dense_bwd_x_kernel<LD_FAST>, grid, dim3(256), 0, s, grad_y, mask, (long long)rows, W, (int)ldw,
                               (int)din, (int)dout, grad_x);
        else if (vec)
            hipLaunchKernelGGL(dense_bwd_x_kernel<LD_PRED>, grid, dim3(256), 0, s, grad_y, mask, (long long)rows, W, (int)ldw,
                               (int)din, (int)dout, grad_x);
        else
            hipLaunchKernelGGL(dense_bwd_x_kernel<LD_SCALAR>, grid, dim3(256), 0, s, grad_y, mask, (long long)rows, W,
                               (int)ldw, (int)din, (int)dout, grad_x);
    }
    if (grad_W || grad_bias) {
        if (!workspace || workspace_bytes < mpqe_linear_bwd_workspace_bytes(rows, din, dout)) return MPQE_ERR_WORKSPACE;
        int nch, ch;
        dense_chunks(rows, &nch, &ch);
        float *slabs = reinterpret_cast<float *>(workspace);
        float *bias_part = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) +
                                                     align_up((size_t)nch * (size_t)din * (size_t)dout * 4, 256));
        if (grad_W) {
            const int tiles = (int)(((din + GT_BM - 1) / GT_BM) * ((dout + GT_BN - 1) / GT_BN));
            dim3 grid((unsigned)nch, (unsigned)tiles);
            // (LD_FAST: whole K-steps in every chunk -- the rows are the K dimension here)
            if (vec && din % GT_BM == 0 && dout % GT_BN == 0 && rows % GT_BK == 0)
                hipLaunchKernelGGL(dense_grad_w_kernel<LD_FAST>, grid, dim3(256), 0, s, x, grad_y, mask, (long long)rows, ch,
                                   (int)din, (int)dout, slabs);
            else if (vec)
                hipLaunchKernelGGL(dense_grad_w_kernel<LD_PRED>, grid, dim3(256), 0, s, x, grad_y, mask, (long long)rows, ch,
                                   (int)din, (int)dout, slabs);
            else
                hipLaunchKernelGGL(dense_grad_w_kernel<LD_SCALAR>, grid, dim3(256), 0, s, x, grad_y, mask, (long long)rows,
                                   ch, (int)din, (int)dout, slabs);
            const long long elems = (long long)din * dout;
            hipLaunchKernelGGL(dense_reduce_w_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s,
                               (const float *)slabs, nch, (int)din, (int)dout, grad_W, (int)ldgw, overwrite);
        }
        if (grad_bias) launch_bias_grad((long long)rows, grad_y, mask, (int)dout, relu, bias_part, grad_bias, s, overwrite);
    }
    return mpqe_launch_status();
}

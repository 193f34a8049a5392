// R-GCN layer on ARBITRARY graphs (the inner boundary RGCNConv.forward(x, edge_index,
// edge_type), reference mpqe/model.py:269-305): duplicate edges, self loops, isolated
// nodes, unused relations.
//
//   plan     sort edges once per graph (radix_sort.h: the library's own LSD radix sort, stable):
//              by relation   -> message slots p in [0,E), grouped so that one MFMA tile
//                               multiplies 64 gathered rows by ONE relation matrix;
//                               slots [E, E+Nn) are the self/root term of every node
//              by destination-> CSR of message slots per output row   (forward sum)
//              by source     -> CSR of message slots per input row    (backward sum)
//   forward  msg[p] = x[row(p)] . W[rel(p)]      grouped gather-GEMM, fp32 MFMA
//            out[i] = act(bias + msg[E+i] + sum_{p in CSR_dst(i)} msg[p])
//                                                destination-sorted segmented sum: every
//                                                row is read once, coalesced; fixed order
//                                                (no float atomics -> reproducible)
//   backward gmsg[p] = gpre[drow(p)] . W[rel(p)]^T ; grad_x[i] = gmsg[E+i] + sum CSR_src(i)
//            grad W[r] = sum_{p in rel r} x[row(p)]^T (x) gpre[drow(p)]   split over K chunks,
//            slabs reduced in fixed order.
#include "radix_sort.h"

#include "bias_grad.h"
#include "gemm_core.h"

#define GEN_CHUNK 512   // message slots per weight-gradient K chunk (256 before the register-only tiles: a tile's fixed
                        // cost -- records, row ids, first rows, the combine -- is ~3 us against 7.4 us of MFMA time now)

// ------------------------------------------------------------------------------------ plan layout
struct PlanLayout {
    size_t rows_fwd, rows_bwd, rel_ptr, tile_ptr, chunk_ptr, dst_ptr, dst_list, src_ptr, src_list, total;
};
static PlanLayout plan_layout(int64_t Nn, int64_t E, int64_t R) {
    PlanLayout L;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += align_up(n * 4, 256);
        return o;
    };
    L.rows_fwd = take((size_t)(E + Nn));
    L.rows_bwd = take((size_t)(E + Nn));
    L.rel_ptr = take((size_t)R + 2);
    L.tile_ptr = take((size_t)R + 2);
    L.chunk_ptr = take((size_t)R + 2);
    L.dst_ptr = take((size_t)Nn + 1);
    L.dst_list = take((size_t)E + 1);
    L.src_ptr = take((size_t)Nn + 1);
    L.src_list = take((size_t)E + 1);
    L.total = off;
    return L;
}
struct PlanView {
    const int *rows_fwd, *rows_bwd, *rel_ptr, *tile_ptr, *chunk_ptr, *dst_ptr, *dst_list, *src_ptr, *src_list;
};
static PlanView plan_view(const void *plan, int64_t Nn, int64_t E, int64_t R) {
    PlanLayout L = plan_layout(Nn, E, R);
    const char *b = reinterpret_cast<const char *>(plan);
    PlanView v;
    v.rows_fwd = (const int *)(b + L.rows_fwd);
    v.rows_bwd = (const int *)(b + L.rows_bwd);
    v.rel_ptr = (const int *)(b + L.rel_ptr);
    v.tile_ptr = (const int *)(b + L.tile_ptr);
    v.chunk_ptr = (const int *)(b + L.chunk_ptr);
    v.dst_ptr = (const int *)(b + L.dst_ptr);
    v.dst_list = (const int *)(b + L.dst_list);
    v.src_ptr = (const int *)(b + L.src_ptr);
    v.src_list = (const int *)(b + L.src_list);
    return v;
}
static inline int64_t tile_bound(int64_t Nn, int64_t E, int64_t R) { return (E + Nn) / GT_BM + R + 2; }
static inline int64_t chunk_bound(int64_t Nn, int64_t E, int64_t R) { return (E + Nn) / GEN_CHUNK + R + 2; }

// ------------------------------------------------------------------------------------ plan kernels
__global__ void plan_prep_kernel(const long long *__restrict__ edge_index, const long long *__restrict__ edge_type,
                                 long long Nn, long long E, long long R, int *key_rel, int *key_dst, int *key_src,
                                 int *iota, int32_t *err) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    long long s = edge_index[e], d = edge_index[E + e], t = edge_type[e];
    if (s < 0 || s >= Nn || d < 0 || d >= Nn) {
        flag_error(err, MPQE_FLAG_BAD_EDGE);
        s = s < 0 || s >= Nn ? 0 : s;
        d = d < 0 || d >= Nn ? 0 : d;
    }
    if (t < 0 || t >= R) {
        flag_error(err, MPQE_FLAG_BAD_RELATION);
        t = 0;
    }
    key_rel[e] = (int)t;
    key_dst[e] = (int)d;
    key_src[e] = (int)s;
    iota[e] = (int)e;
}

__global__ void plan_slots_kernel(const int *__restrict__ perm, const int *__restrict__ key_src,
                                  const int *__restrict__ key_dst, long long Nn, long long E, int *rows_fwd,
                                  int *rows_bwd, int *pos_of_edge) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < E) {
        const int e = perm[p];
        rows_fwd[p] = key_src[e];
        rows_bwd[p] = key_dst[e];
        pos_of_edge[e] = (int)p;
    } else if (p < E + Nn) {
        rows_fwd[p] = (int)(p - E);
        rows_bwd[p] = (int)(p - E);
    }
}

// ptr[v] = first position in sorted keys[0..n) with key >= v, for v in [0, nvals]
__global__ void plan_lower_bound_kernel(const int *__restrict__ keys, long long n, long long nvals, int *ptr) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > nvals) return;
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    ptr[v] = (int)lo;
}

// rel_ptr[R+1] = E+Nn (the root pseudo relation R owns slots [E, E+Nn)); exclusive scans of the
// per-relation tile / chunk counts. One thread: R+1 is a few hundred at most.
__global__ void plan_scan_kernel(long long Nn, long long E, long long R, int *rel_ptr, int *tile_ptr,
                                 int *chunk_ptr) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    rel_ptr[R] = (int)E;
    rel_ptr[R + 1] = (int)(E + Nn);
    int t = 0, c = 0;
    for (long long r = 0; r <= R; ++r) {
        tile_ptr[r] = t;
        chunk_ptr[r] = c;
        const int cnt = rel_ptr[r + 1] - rel_ptr[r];
        t += (cnt + GT_BM - 1) / GT_BM;
        c += (cnt + GEN_CHUNK - 1) / GEN_CHUNK;
    }
    tile_ptr[R + 1] = t;
    chunk_ptr[R + 1] = c;
}

static int bits_for(int64_t n) {
    int b = 1;
    while (b < 31 && (1ll << b) < n) ++b;
    return b;
}

struct PlanWs {
    size_t key_rel, key_dst, key_src, iota, key_out, perm, pos, sort_tmp, sort_bytes, total;
};
static size_t sort_tmp_bytes(int64_t E, int bits) {
    (void)bits;
    return radix_sort_tmp_bytes<int>((long long)E);
}
static PlanWs plan_ws_layout(int64_t Nn, int64_t E, int64_t R) {
    PlanWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    const size_t n = (size_t)(E > 0 ? E : 1);
    w.key_rel = take(n * 4);
    w.key_dst = take(n * 4);
    w.key_src = take(n * 4);
    w.iota = take(n * 4);
    w.key_out = take(n * 4);
    w.perm = take(n * 4);
    w.pos = take(n * 4);
    size_t bytes = 0;
    if (E > 0) {
        const size_t b1 = sort_tmp_bytes(E, bits_for(R)), b2 = sort_tmp_bytes(E, bits_for(Nn));
        bytes = b1 > b2 ? b1 : b2;
    }
    w.sort_bytes = bytes + 256;
    w.sort_tmp = take(w.sort_bytes);
    w.total = off;
    return w;
}

extern "C" size_t mpqe_rgcn_plan_bytes(int64_t Nn, int64_t E, int64_t R) {
    if (Nn < 0 || E < 0 || R < 0) return 0;
    return plan_layout(Nn, E, R).total;
}
extern "C" size_t mpqe_rgcn_plan_workspace_bytes(int64_t Nn, int64_t E, int64_t R) {
    if (Nn < 0 || E < 0 || R < 0) return 0;
    return plan_ws_layout(Nn, E, R).total;
}

extern "C" int mpqe_rgcn_plan_build(const int64_t *edge_index, const int64_t *edge_type, int64_t Nn, int64_t E,
                                    int64_t R, void *plan, size_t plan_bytes, void *workspace,
                                    size_t workspace_bytes, int32_t *err, void *stream) {
    if (Nn < 0 || E < 0 || R < 0 || !plan) return MPQE_ERR_INVALID_ARG;
    if (E + Nn >= (1ll << 31) || R >= (1ll << 30)) return MPQE_ERR_UNSUPPORTED;
    if (E > 0 && (!edge_index || !edge_type)) return MPQE_ERR_INVALID_ARG;
    PlanLayout L = plan_layout(Nn, E, R);
    PlanWs W = plan_ws_layout(Nn, E, R);
    if (plan_bytes < L.total) return MPQE_ERR_WORKSPACE;
    if (!workspace || workspace_bytes < W.total) return MPQE_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    char *pb = reinterpret_cast<char *>(plan);
    char *wb = reinterpret_cast<char *>(workspace);
    int *rows_fwd = (int *)(pb + L.rows_fwd), *rows_bwd = (int *)(pb + L.rows_bwd);
    int *rel_ptr = (int *)(pb + L.rel_ptr), *tile_ptr = (int *)(pb + L.tile_ptr);
    int *chunk_ptr = (int *)(pb + L.chunk_ptr);
    int *dst_ptr = (int *)(pb + L.dst_ptr), *dst_list = (int *)(pb + L.dst_list);
    int *src_ptr = (int *)(pb + L.src_ptr), *src_list = (int *)(pb + L.src_list);
    int *key_rel = (int *)(wb + W.key_rel), *key_dst = (int *)(wb + W.key_dst), *key_src = (int *)(wb + W.key_src);
    int *iota = (int *)(wb + W.iota), *key_out = (int *)(wb + W.key_out), *perm = (int *)(wb + W.perm);
    int *pos = (int *)(wb + W.pos);
    void *tmp = wb + W.sort_tmp;
    size_t tmp_bytes = W.sort_bytes;

    if (E > 0) {
        hipLaunchKernelGGL(plan_prep_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s,
                           (const long long *)edge_index, (const long long *)edge_type, (long long)Nn, (long long)E,
                           (long long)R, key_rel, key_dst, key_src, iota, err);
        if (radix_sort_pairs_own<int>(tmp, (const int *)key_rel, key_out, (const int *)iota, perm, (long long)E, bits_for(R), s))
            return MPQE_ERR_LAUNCH;
    }
    const long long slots = E + Nn;
    if (slots > 0)
        hipLaunchKernelGGL(plan_slots_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s,
                           (const int *)perm, (const int *)key_src, (const int *)key_dst, (long long)Nn,
                           (long long)E, rows_fwd, rows_bwd, pos);
    // key_out now holds the relation-sorted keys: rel_ptr[r] = lower_bound(r) for r in [0, R]
    hipLaunchKernelGGL(plan_lower_bound_kernel, dim3((unsigned)((R + 1 + 255) / 256)), dim3(256), 0, s,
                       (const int *)key_out, (long long)E, (long long)R, rel_ptr);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(64), 0, s, (long long)Nn, (long long)E, (long long)R,
                       rel_ptr, tile_ptr, chunk_ptr);
    if (E > 0) {
        tmp_bytes = W.sort_bytes;
        if (radix_sort_pairs_own<int>(tmp, (const int *)key_dst, key_out, (const int *)pos, dst_list, (long long)E, bits_for(Nn), s))
            return MPQE_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(plan_lower_bound_kernel, dim3((unsigned)((Nn + 1 + 255) / 256)), dim3(256), 0, s,
                       (const int *)key_out, (long long)E, (long long)Nn, dst_ptr);
    if (E > 0) {
        tmp_bytes = W.sort_bytes;
        if (radix_sort_pairs_own<int>(tmp, (const int *)key_src, key_out, (const int *)pos, src_list, (long long)E, bits_for(Nn), s))
            return MPQE_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(plan_lower_bound_kernel, dim3((unsigned)((Nn + 1 + 255) / 256)), dim3(256), 0, s,
                       (const int *)key_out, (long long)E, (long long)Nn, src_ptr);
    return mpqe_launch_status();
}

// ------------------------------------------------------------------------------------ grouped gather-GEMM
// first r with ptr[r+1] > t  (ptr non-decreasing, ptr[0] = 0, t < ptr[n])
__device__ __forceinline__ int find_group(const int *__restrict__ ptr, int n, int t) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid + 1] > t) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// Loader of the gather-GEMM tile: A rows are gathered through a row-id list (one id per tile row),
// B is one relation matrix: W[k][n] (K-type) forward, W[n][k] (R-type) for the transposed use.
template <int MODE, bool TRANS>
struct GatherLoader {
    const float *abase, *mask, *wsafe;
    int K, Din, Dout, left, ks;           // ks = first k of the current step
    int ac, brow0, brow1, bcol0;
    bool rok0, rok1;
    const float *pa0, *pa1, *pm0, *pm1, *pb0, *pb1;

    __device__ __forceinline__ void init(const int *__restrict__ rows, int nrows, const float *a_,
                                         const float *mask_, const float *W, const float *wsafe_, int K_, int Din_,
                                         int Dout_, int n0, int nsteps) {
        abase = a_;
        mask = mask_;
        wsafe = wsafe_;
        K = K_;
        Din = Din_;
        Dout = Dout_;
        left = nsteps;
        ks = 0;
        ac = stage_col(false);
        const int r0 = stage_row(false, 0), r1 = stage_row(false, 1);
        rok0 = r0 < nrows;
        rok1 = r1 < nrows;
        const long long o0 = rok0 ? (long long)rows[r0] * K : 0, o1 = rok1 ? (long long)rows[r1] * K : 0;
        pa0 = abase + o0;
        pa1 = abase + o1;
        pm0 = mask ? mask + o0 : nullptr;
        pm1 = mask ? mask + o1 : nullptr;
        if (!TRANS) {
            brow0 = stage_row(true, 0);
            brow1 = stage_row(true, 1);
            bcol0 = n0 + stage_col(true);
        } else {
            brow0 = n0 + stage_row(false, 0);
            brow1 = n0 + stage_row(false, 1);
            bcol0 = stage_col(false);
        }
        pb0 = W + (long long)brow0 * Dout;
        pb1 = W + (long long)brow1 * Dout;
    }
    __device__ __forceinline__ f32x4 a(int slot, bool &ok) {
        f32x4 v = ld4_pred<MODE>(abase, slot ? pa1 : pa0, ks + ac, K, slot ? rok1 : rok0, ok);
        if (mask) {
            bool ok2;
            f32x4 o = ld4_pred<MODE>(mask, slot ? pm1 : pm0, ks + ac, K, slot ? rok1 : rok0, ok2);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = o[q] > 0.f ? v[q] : 0.f;
        }
        return v;
    }
    __device__ __forceinline__ f32x4 b(int slot, bool &ok) {
        if (!TRANS) return ld4_pred<MODE>(wsafe, slot ? pb1 : pb0, bcol0, Dout, ks + (slot ? brow1 : brow0) < Din, ok);
        return ld4_pred<MODE>(wsafe, slot ? pb1 : pb0, ks + bcol0, Dout, (slot ? brow1 : brow0) < Din, ok);
    }
    __device__ __forceinline__ void next() {
        if (left <= 1) return;
        --left;
        ks += GT_BK;
        if (!TRANS) {
            pb0 += (long long)GT_BK * Dout;
            pb1 += (long long)GT_BK * Dout;
        }
    }
};

// Loader of the gathered weight gradient: K runs over message slots q; x / gpre rows come through
// the plan's row-id lists.
template <int MODE>
struct GatherGradWLoader {
    const float *x, *g, *mask;
    const int *rows_fwd, *rows_bwd;
    int Din, Dout, left, q, q0, q1, ca, cb;

    __device__ __forceinline__ void init(const int *rf, const int *rb, const float *x_, const float *g_,
                                         const float *mask_, int Din_, int Dout_, int q0_, int q1_, int i0, int j0,
                                         int nsteps) {
        rows_fwd = rf;
        rows_bwd = rb;
        x = x_;
        g = g_;
        mask = mask_;
        Din = Din_;
        Dout = Dout_;
        left = nsteps;
        q0 = q0_;
        q1 = q1_;
        q = q0 + stage_row(true, 0);
        ca = i0 + stage_col(true);
        cb = j0 + stage_col(true);
    }
    __device__ __forceinline__ f32x4 a(int slot, bool &ok) {
        const int qq = q + 16 * slot;
        const int qc = qq < q1 ? qq : q0;          // clamped slot: the row-id read stays in range
        return ld4_pred<MODE>(x, x + (long long)rows_fwd[qc] * Din, ca, Din, qq < q1, ok);
    }
    __device__ __forceinline__ f32x4 b(int slot, bool &ok) {
        const int qq = q + 16 * slot;
        const int qc = qq < q1 ? qq : q0;
        const long long off = (long long)rows_bwd[qc] * Dout;
        f32x4 v = ld4_pred<MODE>(g, g + off, cb, Dout, qq < q1, ok);
        if (mask) {
            bool ok2;
            f32x4 o = ld4_pred<MODE>(mask, mask + off, cb, Dout, qq < q1, ok2);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = o[k] > 0.f ? v[k] : 0.f;
        }
        return v;
    }
    __device__ __forceinline__ void next() {
        if (left <= 1) return;
        --left;
        q += GT_BK;
    }
};

// TRANS = false: msg[p]  = x[rows[p]]    . W[rel]      (K = Din,  cols = Dout)
// TRANS = true : gmsg[p] = gpre[rows[p]] . W[rel]^T    (K = Dout, cols = Din), gpre masked by out > 0
template <bool TRANS, int MODE>
__global__ __launch_bounds__(256) void rgcn_gen_gemm_kernel(
    const int *__restrict__ rows, const int *__restrict__ rel_ptr, const int *__restrict__ tile_ptr, int R,
    const float *__restrict__ a, const float *__restrict__ mask, const float *__restrict__ basis,
    const float *__restrict__ root, int Din, int Dout, float *__restrict__ msg) {
    __shared__ __attribute__((aligned(16))) float smem[GT_SMEM_FLOATS];
    // 1-D grid, 8 row tiles x `ct` column tiles per group: the column tiles of one row tile read the SAME gathered rows,
    // so they go to ONE XCD (workgroup b runs on XCD b % 8) next to each other -- the rows come from HBM once and from
    // that XCD's L2 for the other column tiles. (Row tiles along x and column tiles along y ran the column tiles of a
    // row tile far apart in time: every gathered row from HBM once per column tile.)
    const int ct = ((TRANS ? Din : Dout) + GT_BN - 1) / GT_BN;
    const int grp = (int)blockIdx.x / (8 * ct), rem = (int)blockIdx.x - grp * (8 * ct);
    const int t = grp * 8 + (rem & 7), by = rem >> 3;
    if (t >= tile_ptr[R + 1]) return;
    const int r = find_group(tile_ptr, R + 1, t);
    const int start = rel_ptr[r] + (t - tile_ptr[r]) * GT_BM;
    int nrows = rel_ptr[r + 1] - start;
    if (nrows > GT_BM) nrows = GT_BM;
    const float *W = r < R ? basis + (long long)r * Din * Dout : root;
    const int K = TRANS ? Dout : Din;       // length of an A row
    const int C = TRANS ? Din : Dout;       // output columns
    const int n0 = by * GT_BN;
    const int nsteps = (K + GT_BK - 1) / GT_BK;

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    GatherLoader<MODE, TRANS> L;
    L.init(rows + start, nrows, a, TRANS ? mask : nullptr, W, root, K, Din, Dout, n0, nsteps);
    gemm_block<false, !TRANS>(acc, L, nsteps, smem);
    const int col = n0 + acc_col();
    if (col < C) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = acc_row(q);
            if (row < nrows) msg[(long long)(start + row) * C + col] = acc[q];
        }
    }
}

// The same product with every operand loaded straight into MFMA registers (dims multiples of 64, 16-byte aligned rows):
// no LDS staging, no barrier. v_mfma_f32_16x16x4_f32 sums over its four k slots in any order and lets the kernel choose
// which column a lane position stands for:
//   A   lane (pos, kq) loads the 16 bytes a[row(16 g + pos)][16 it + 4 kq ..]: component u is the A value of MFMA (u, .)
//       -- k = 16 it + 4 kq + u -- for row group g (a workgroup's 64 gathered rows = 4 groups);
//   B   for the same k: W[k][n0 + 4 pos .. + 3] (one 16-byte load per u; TRANS: W[n0 + 4 pos + n][16 it + 4 kq ..], one per
//       n), so position pos stands for the four columns 4 pos + n of MFMAs (., n);
//   D   lane (pos, kq) ends with out[16 g + 4 kq + r][n0 + 4 pos + n], n = 0..3: 16-byte stores, 256 bytes per row.
// A wave owns 64 columns and ALL 64 rows of the tile: 8 (12 with the ReLU mask) 16-byte loads feed 64 MFMAs, and the four
// waves of a workgroup share the gathered rows (one HBM / L2 fetch, L1 hits for the other three). With 256 columns a
// workgroup covers whole output rows: every gathered row is fetched once per launch.
// (The LDS-staged 32x32x2 core above: 96 / 111 us forward / transposed at the stress shape = 0.50 / 0.43 of the roof.)
#ifndef GGM_PF_FWD
#define GGM_PF_FWD 2        // prefetch distance in iterations (forward form at one workgroup per CU: 2 / 4 = 87.3 / 89.5 us)
#endif
#ifndef GGM_PF_TRANS
#define GGM_PF_TRANS 2
#endif
#ifndef GGM_KW_FWD
#define GGM_KW_FWD 4        // consecutive k per lane and iteration: 4 (one 16-byte load per row piece) or 8 (two: whole lines).
                            // Measured at the stress shape: forward 85.4 (4) / 89.4 (8) us, transposed 107 (4) / 123.8 (8, two
                            // stages in flight) / 107.5 (8, one stage) -- half-line loads are not what bounds these kernels
#endif
#ifndef GGM_KW_TRANS
#define GGM_KW_TRANS 4
#endif
template <bool TRANS, bool RELU>
__global__ __launch_bounds__(256) void rgcn_gen_gemm_rows_kernel(
    const int *__restrict__ rows, const int *__restrict__ rel_ptr, const int *__restrict__ tile_ptr, int R,
    const float *__restrict__ a, const unsigned long long *__restrict__ mask, const float *__restrict__ basis,
    const float *__restrict__ root, int Din, int Dout, float *__restrict__ msg) {
    // RELU: `mask` = the ReLU mask of the rows of `a` as bit words (bias_grad.h: bit c % 64 of word [row][c / 64]) -- 8
    // bytes per row and iteration instead of 16 bytes per lane of the gathered `out` row.
    // Occupancy by LDS footprint: the forward form runs ONE workgroup per CU (144 KB claimed, 8 KB used), the transposed form
    // two. Measured at the stress shape, forward: 3 / 2 / 1 workgroups per CU = 139.8 / 114.5 / 86.6 us (transposed: - /
    // 106.8 / 113.3): with more workgroups in flight the gathered half-lines and the streamed matrices evict each other
    // from L1 / L2 before their second use, and the MFMA pipe is already full with one wave per SIMD.
    __shared__ int sp[TRANS ? 2 * 1026 : 36000];
    constexpr int KW = TRANS ? GGM_KW_TRANS : GGM_KW_FWD, H = KW / 4;
    const int K = TRANS ? Dout : Din, C = TRANS ? Din : Dout;
    // A workgroup's four waves take four (row tile, 64-column block) pairs: with 256 or more columns the four blocks of one
    // row tile (cg workgroups per row tile), with 128 / 64 columns the blocks of 2 / 4 consecutive row tiles -- else half /
    // three quarters of the SIMDs idle at one workgroup per CU (D = 128 forward: 46 us = 0.26 of the roof before).
    const int cb = C / 64;                                 // column blocks of a row tile
    const int rt = cb >= 4 ? 1 : (cb == 3 ? 1 : 4 / cb);   // row tiles per workgroup
    const int cg = cb >= 4 ? (cb + 3) / 4 : 1;             // workgroups per row tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pos = lane & 15, kq = lane >> 4;
    const int sub = rt > 1 ? wave / cb : 0, wcol = rt > 1 ? wave % cb : wave;
    const bool lds_tables = R + 2 <= 1026;
    if (lds_tables) {               // both pointer tables in ONE round trip, every search in LDS
        int *sr = sp + 1026;
        for (int i = threadIdx.x; i < R + 2; i += 256) {
            sp[i] = tile_ptr[i];
            sr[i] = rel_ptr[i];
        }
        __syncthreads();
    }
    const int *tp = lds_tables ? sp : tile_ptr, *rp = lds_tables ? sp + 1026 : rel_ptr;
    // PERSISTENT: the launch has one workgroup per CU slot; a workgroup takes the items b, b + G, b + 2 G, ... (item =
    // row tile x column group, eight row tiles of a group on eight XCDs: the column groups of a row tile on ONE XCD). The
    // pointer tables are read once, the next tile's row ids are requested while the current tile multiplies -- a tile's
    // start-up (tables -> ids -> rows -> first MFMA, ~5 us at one workgroup per CU) is paid once per workgroup, not per
    // tile. No barrier below: the four waves walk the same tiles on their own.
    // (one contiguous eighth of the row tiles per XCD -- a relation's tiles and its matrix on one L2 -- was measured and
    // is slower: 158 / 127 us forward / transposed against 140 / 106 dealt round-robin, at three / two workgroups per CU)
    const int ntile = tp[R + 1];
    auto tile_of = [&](int item, int &r, int &start, int &nrows, int &by) -> bool {
        const int grp = item / (8 * cg), rem = item - grp * (8 * cg);
        const int t = (grp * 8 + (rem & 7)) * rt + sub;
        by = rem >> 3;
        if (t >= ntile) return false;
        r = find_group(tp, R + 1, t);
        start = rp[r] + (t - tp[r]) * GT_BM;
        nrows = rp[r + 1] - start;
        if (nrows > GT_BM) nrows = GT_BM;
        return true;
    };
    auto load_ids = [&](int start, int nrows, int (&id)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int rr = 16 * g + pos;
            id[g] = nrows > 0 ? rows[start + (rr < nrows ? rr : nrows - 1)] : 0;      // (clamped: rows beyond the tile are not stored)
        }
    };
    int item = (int)blockIdx.x, r, start, nrows, by;
    if (!tile_of(item, r, start, nrows, by)) return;       // (items only grow: nothing further either)
    const int n0 = (by * 4 + wcol) * 64;                   // (`by` is the same for every item of a workgroup)
    if (n0 >= C) return;                                   // a wave without a column block just leaves
    int idc[4];
    load_ids(start, nrows, idc);
    const int niter = K / (4 * KW);
    constexpr int GGM_PF = TRANS ? GGM_PF_TRANS : GGM_PF_FWD;
    for (;;) {
        int r2 = 0, start2 = 0, nrows2 = 0, by2 = 0, idn[4] = {0, 0, 0, 0};
        const bool more = tile_of(item + (int)gridDim.x, r2, start2, nrows2, by2);
        if (more) load_ids(start2, nrows2, idn);           // (in flight under this tile's K loop)
        if (nrows > 0) {
            const float *W = r < R ? basis + (long long)r * Din * Dout : root;
            const float *pa[4];
            const unsigned long long *pm[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pa[g] = a + (long long)idc[g] * K + KW * kq;
                pm[g] = RELU ? mask + (long long)idc[g] * (K / 64) : nullptr;
            }
            // A lane takes KW = 4 H consecutive k of its row per iteration (H 16-byte loads: with H = 2 the four kq groups
            // cover a whole 128-byte line of the row per load pair). MFMA (h, u) multiplies the k slots
            // {4 KW it + KW kq + 4 h + u}.
            // B: non-TRANS W[(4 KW it + KW kq + 4 h + u)][n0 + 4 pos ..]; TRANS W[(n0 + 4 pos + n)][4 KW it + KW kq + 4 h ..]
            const float *pw = TRANS ? W + (long long)(n0 + 4 * pos) * Dout + KW * kq
                                    : W + (long long)(KW * kq) * Dout + n0 + 4 * pos;
            const long long wstep = TRANS ? 4 * KW : 4ll * KW * Dout;      // floats per iteration
            f32x4 acc[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 A[GGM_PF][4][H], B[GGM_PF][4 * H];
            unsigned long long M[RELU ? GGM_PF : 1][4];
            int MO[RELU ? GGM_PF : 1];         // bit offset of the lane's first k of the stage inside its mask word
            auto load = [&](int s, int it) {
                const int ic = it < niter ? it : niter - 1;        // (beyond the end: the last piece again, unused)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int h = 0; h < H; ++h) A[s][g][h] = gload4(pa[g] + 4 * KW * ic + 4 * h);
                if (RELU) {         // (the 4 KW k of an iteration lie in ONE 64-bit word: 4 KW divides 64)
                    MO[s] = (4 * KW * ic + KW * kq) & 63;
#pragma unroll
                    for (int g = 0; g < 4; ++g) M[s][g] = pm[g][(4 * KW * ic) >> 6];
                }
#pragma unroll
                for (int q = 0; q < 4 * H; ++q)      // non-TRANS: q = 4 h + u, a row of W each; TRANS: q = H n + h
                    B[s][q] = TRANS ? gload4(pw + (long long)ic * wstep + (long long)(q / H) * Dout + 4 * (q % H))
                                    : gload4(pw + (long long)ic * wstep + (long long)q * Dout);
            };
            auto mma = [&](int s) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        f32x4 av = A[s][g][h];
                        if (RELU) {
                            const unsigned nib = (unsigned)(M[s][g] >> (MO[s] + 4 * h));
#pragma unroll
                            for (int u = 0; u < 4; ++u) av[u] = ((nib >> u) & 1u) ? av[u] : 0.f;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int n = 0; n < 4; ++n)
                                acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    av[u], TRANS ? B[s][H * n + h][u] : B[s][4 * h + u][n], acc[g][n], 0, 0, 0);
                    }
            };
#pragma unroll
            for (int s = 0; s < GGM_PF; ++s) {
                load(s, s);
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int it = 0; it < niter; it += GGM_PF) {
#pragma unroll
                for (int s = 0; s < GGM_PF; ++s) {
                    mma(s);             // iteration it + s (K % 64 == 0: niter is a multiple of 2 at KW = 8, of 4 at KW = 4)
                    __builtin_amdgcn_sched_barrier(0);
                    load(s, it + s + GGM_PF);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int row = 16 * g + 4 * kq + rr;
                    if (row < nrows)
                        *reinterpret_cast<f32x4 *>(msg + (long long)(start + row) * C + n0 + 4 * pos) =
                            f32x4{acc[g][0][rr], acc[g][1][rr], acc[g][2][rr], acc[g][3][rr]};
                }
        }
        if (!more) break;
        item += (int)gridDim.x;
        r = r2;
        start = start2;
        nrows = nrows2;
#pragma unroll
        for (int g = 0; g < 4; ++g) idc[g] = idn[g];
    }
}

// ------------------------------------------------------------------------------------ segmented sum
// out[i] = act(bias + msg[E+i] + sum_{k in [ptr[i], ptr[i+1])} msg[list[k]]): one thread owns 4
// consecutive columns of one row, so a row is read by D/4 adjacent lanes in 16-byte pieces
// and no cross-lane step is needed. Message rows are added in CSR order (stable in edge id).
__global__ __launch_bounds__(256) void segment_sum_kernel(const int *__restrict__ ptr, const int *__restrict__ list,
                                                          long long Nn, long long E, int D,
                                                          const float *__restrict__ msg,
                                                          const float *__restrict__ bias, int relu,
                                                          float *__restrict__ out, int vec,
                                                          unsigned long long *__restrict__ bits = nullptr) {
    // bits (may be NULL; needs vec, relu, D % 64 == 0): the ReLU mask of `out` as bit words -- bit c % 64 of word
    // [row][c / 64] = out[row][c] > 0 (bias_grad.h) -- for the backward, which then never reads `out`. The 16 lanes that hold
    // 64 consecutive columns of a row are one DPP row: their nibbles are OR-ed with four row rotations.
    const int per_row = vec ? D / 4 : D;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = idx / per_row;
    if (i >= Nn) return;
    const int c = (int)(idx - i * per_row) * (vec ? 4 : 1);
    const int k0 = ptr[i], k1 = ptr[i + 1];
    if (vec) {
        f32x4 s = *reinterpret_cast<const f32x4 *>(msg + (E + i) * D + c);
        if (bias) {
            f32x4 b = *reinterpret_cast<const f32x4 *>(bias + c);
            s[0] += b[0]; s[1] += b[1]; s[2] += b[2]; s[3] += b[3];
        }
        int k = k0;
        for (; k + 1 < k1; k += 2) {   // two rows in flight, added in order
            f32x4 m0 = *reinterpret_cast<const f32x4 *>(msg + (long long)list[k] * D + c);
            f32x4 m1 = *reinterpret_cast<const f32x4 *>(msg + (long long)list[k + 1] * D + c);
            s[0] += m0[0]; s[1] += m0[1]; s[2] += m0[2]; s[3] += m0[3];
            s[0] += m1[0]; s[1] += m1[1]; s[2] += m1[2]; s[3] += m1[3];
        }
        if (k < k1) {
            f32x4 m0 = *reinterpret_cast<const f32x4 *>(msg + (long long)list[k] * D + c);
            s[0] += m0[0]; s[1] += m0[1]; s[2] += m0[2]; s[3] += m0[3];
        }
        if (relu) {
            s[0] = s[0] > 0.f ? s[0] : 0.f; s[1] = s[1] > 0.f ? s[1] : 0.f;
            s[2] = s[2] > 0.f ? s[2] : 0.f; s[3] = s[3] > 0.f ? s[3] : 0.f;
        }
        *reinterpret_cast<f32x4 *>(out + i * D + c) = s;
        if (bits)
            mask_word_store(bits, i, D, c, (s[0] > 0.f ? 1u : 0u) | (s[1] > 0.f ? 2u : 0u) | (s[2] > 0.f ? 4u : 0u) | (s[3] > 0.f ? 8u : 0u));
    } else {
        float s = msg[(E + i) * D + c] + (bias ? bias[c] : 0.f);
        for (int k = k0; k < k1; ++k) s += msg[(long long)list[k] * D + c];
        if (relu) s = s > 0.f ? s : 0.f;
        out[i * D + c] = s;
    }
}

// ------------------------------------------------------------------------------------ weight gradient
template <int MODE>
__global__ __launch_bounds__(256) void rgcn_gen_grad_w_kernel(
    const int *__restrict__ rows_fwd, const int *__restrict__ rows_bwd, const int *__restrict__ rel_ptr,
    const int *__restrict__ chunk_ptr, int R, const float *__restrict__ x, const float *__restrict__ g,
    const float *__restrict__ out, int Din, int Dout, int relu, float *__restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float smem[GT_SMEM_FLOATS];
    const int c = blockIdx.x;
    if (c >= chunk_ptr[R + 1]) return;
    const int r = find_group(chunk_ptr, R + 1, c);
    const int q0 = rel_ptr[r] + (c - chunk_ptr[r]) * GEN_CHUNK;
    int q1 = q0 + GEN_CHUNK;
    if (q1 > rel_ptr[r + 1]) q1 = rel_ptr[r + 1];
    const int tiles_j = (Dout + GT_BN - 1) / GT_BN;
    const int i0 = (blockIdx.y / tiles_j) * GT_BM;
    const int j0 = (blockIdx.y % tiles_j) * GT_BN;
    const int nsteps = (q1 - q0 + GT_BK - 1) / GT_BK;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    if (nsteps > 0) {
        GatherGradWLoader<MODE> L;
        L.init(rows_fwd, rows_bwd, x, g, relu ? out : nullptr, Din, Dout, q0, q1, i0, j0, nsteps);
        gemm_block<true, true>(acc, L, nsteps, smem);
    }
    float *dst = slabs + (long long)c * Din * Dout;
    const int col = j0 + acc_col();
    if (col < Dout) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = i0 + acc_row(q);
            if (row < Din) dst[(long long)row * Dout + col] = acc[q];
        }
    }
}

// The same 64 x 64 tile of one K-chunk with a register-only K loop (Din, Dout multiples of 64, 16-byte aligned rows): the
// form of the fused step's weight-gradient tiles (grad_w_reg.h) on GATHERED rows. v_mfma_f32_16x16x4_f32: lane (pos, kq)
// loads the 16 bytes x[row_fwd(q)][i0 + 4 pos ..] and g[row_bwd(q)][j0 + 4 pos ..] of message slot q = q0 + 16 t + 4 wave
// + kq; MFMA (m, n) takes component m of one and n of the other, so its position (pr, pc) is output element (i0 + 4 pr + m,
// j0 + 4 pc + n). Two 16-byte row loads (three with the ReLU mask) and two row-id loads feed 16 MFMAs; the row ids are
// requested 2 x GGR_PF iterations ahead, the rows GGR_PF ahead (a row address needs its id). No LDS, no barrier in the
// K loop; the four waves split the slots and meet once in LDS, added in wave order (fixed: reproducible).
// (The LDS-staged 32x32x2 core above spends its K loop on VALU / LDS instructions: 140.9 us = 0.34 of the fp32 MFMA roof
// at the stress shape.)
#ifndef GGR_PF
#define GGR_PF 3        // (2 / 3 / 4 / 6 / 8 measured: 99.4 / 99.2 / 101.4 / 113 / 109 us at the stress shape)
#endif
#define GGR_LDT 68
#ifndef GGR_LDS_TILES
#define GGR_LDS_TILES 1     // 4: a combine tile per wave, one barrier (2 workgroups per CU); 1: the waves take turns on one tile
                            // (102.1 against 101.4 us at prefetch 4: the kernel is bound by its gathered reads, not by either)
#endif
template <bool RELU>
__global__ __launch_bounds__(256) void rgcn_gen_grad_w_rows_kernel(
    const int *__restrict__ rows_fwd, const int *__restrict__ rows_bwd, const int *__restrict__ rel_ptr,
    const int *__restrict__ chunk_ptr, int R, const float *__restrict__ x, const float *__restrict__ g,
    const unsigned long long *__restrict__ out /* RELU: mask bit words of the rows of g (bias_grad.h) */, int Din, int Dout,
    float *__restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float smem[GGR_LDS_TILES * 64 * GGR_LDT];
    // The tiles of one K-chunk read the SAME rows (each 256-byte piece of an x row by Dout / 64 tiles, of a g row by
    // Din / 64): they are given to ONE XCD (workgroup b runs on XCD b % 8) and dispatched together, so a piece comes
    // from HBM once and from that XCD's L2 for the other tiles. 1-D grid; 8 chunks x `tiles` workgroups per group.
    // (With chunks along x and tiles along y the tiles of a chunk ran far apart in time: every piece from HBM for
    // every tile, ~700 MB per launch at the stress shape -- the kernel was bandwidth-bound, not MFMA-bound.)
    const int tiles = (Din / 64) * (Dout / 64);
    const int grp = (int)blockIdx.x / (8 * tiles), rem = (int)blockIdx.x - grp * (8 * tiles);
    const int c = grp * 8 + (rem & 7), tile = rem >> 3;
    // which relation the chunk belongs to: both pointer tables go to LDS in ONE round trip and the search runs there (a
    // binary search over global memory is ~7 dependent loads in front of a tile of 7 us)
    int r, q0, q1;
    if (2 * (R + 2) <= GGR_LDS_TILES * 64 * GGR_LDT) {
        int *sp = reinterpret_cast<int *>(smem), *sr = sp + (R + 2);
        for (int i = threadIdx.x; i < R + 2; i += 256) {
            sp[i] = chunk_ptr[i];
            sr[i] = rel_ptr[i];
        }
        __syncthreads();
        if (c >= sp[R + 1]) return;         // (uniform)
        r = find_group(sp, R + 1, c);
        q0 = sr[r] + (c - sp[r]) * GEN_CHUNK;
        q1 = q0 + GEN_CHUNK;
        if (q1 > sr[r + 1]) q1 = sr[r + 1];
        __syncthreads();                    // (the tables' LDS is the waves' combine space below)
    } else {
        if (c >= chunk_ptr[R + 1]) return;
        r = find_group(chunk_ptr, R + 1, c);
        q0 = rel_ptr[r] + (c - chunk_ptr[r]) * GEN_CHUNK;
        q1 = q0 + GEN_CHUNK;
        if (q1 > rel_ptr[r + 1]) q1 = rel_ptr[r + 1];
    }
    const int tiles_j = Dout / 64;
    const int i0 = (tile / tiles_j) * 64, j0 = (tile % tiles_j) * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pos = lane & 15, kq = lane >> 4;
    const int niter = (q1 - q0 + 15) / 16;
    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q1 > q0) {
        const float *xa = x + i0 + 4 * pos, *gb = g + j0 + 4 * pos;
        const unsigned long long *ob = RELU ? out + (j0 >> 6) : nullptr;      // (a tile's 64 columns: one word per row)
        int IA[GGR_PF], IB[GGR_PF];
        float live[GGR_PF], live_use[GGR_PF];     // 1 / 0 per slot: ids stage, rows stage (the ids stage runs PF ahead)
        f32x4 A[GGR_PF], B[GGR_PF];
        unsigned long long M[RELU ? GGR_PF : 1];
        auto load_ids = [&](int s, int t) {
            const int q = q0 + 16 * t + 4 * wave + kq;
            const int qc = q < q1 ? q : q1 - 1;        // (clamped: a slot beyond the chunk contributes zero)
            IA[s] = rows_fwd[qc];
            IB[s] = rows_bwd[qc];
            live[s] = q < q1 ? 1.f : 0.f;
        };
        auto load_rows = [&](int s) {
            live_use[s] = live[s];
            A[s] = gload4(xa + (long long)IA[s] * Din);       // (the zero of a slot beyond the chunk is applied at the USE:
                                                              // a multiply right behind the load would wait for it)
            B[s] = gload4(gb + (long long)IB[s] * Dout);
            if (RELU) M[s] = ob[(long long)IB[s] * (Dout / 64)];
        };
        auto mma = [&](int s) {
            f32x4 b = B[s];
            const f32x4 a = A[s] * live_use[s];
            if (RELU) {
                const unsigned nib = (unsigned)(M[s] >> (4 * pos));
#pragma unroll
                for (int k = 0; k < 4; ++k) b[k] = ((nib >> k) & 1u) ? b[k] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
        };
        // prologue: ids of iterations [0, PF), their rows, then the ids of [PF, 2 PF)
#pragma unroll
        for (int s = 0; s < GGR_PF; ++s) {
            load_ids(s, s);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < GGR_PF; ++s) {
            load_rows(s);
            __builtin_amdgcn_sched_barrier(0);
            load_ids(s, s + GGR_PF);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int t = 0; t < niter; t += GGR_PF) {
#pragma unroll
            for (int s = 0; s < GGR_PF; ++s) {
                mma(s);                 // iteration t + s (beyond the chunk: zero operand)
                __builtin_amdgcn_sched_barrier(0);
                load_rows(s);           // rows of iteration t + s + PF (ids requested PF iterations ago)
                __builtin_amdgcn_sched_barrier(0);
                load_ids(s, t + s + 2 * GGR_PF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // D[position row 4 kq + r][position col pos] of MFMA (m, n) = element (i0 + 4 (4 kq + r) + m, j0 + 4 pos + n)
#if GGR_LDS_TILES == 4
    float *mine = smem + wave * (64 * GGR_LDT);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            *reinterpret_cast<f32x4 *>(mine + (4 * (4 * kq + rr) + m) * GGR_LDT + 4 * pos) =
                f32x4{acc[m][0][rr], acc[m][1][rr], acc[m][2][rr], acc[m][3][rr]};
    __syncthreads();
#else
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    f32x4 *t = reinterpret_cast<f32x4 *>(smem + (4 * (4 * kq + rr) + m) * GGR_LDT + 4 * pos);
                    f32x4 v = {acc[m][0][rr], acc[m][1][rr], acc[m][2][rr], acc[m][3][rr]};
                    if (w > 0) v += *t;
                    *t = v;
                }
        }
        __syncthreads();
    }
#endif
    float *dst = slabs + (long long)c * Din * Dout;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = (int)threadIdx.x + 256 * k;
        const int row = f >> 4, c4 = f & 15;
        const float *t0 = smem + row * GGR_LDT + 4 * c4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(t0);
#if GGR_LDS_TILES == 4
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4 *>(t0 + w * (64 * GGR_LDT));
#endif
        *reinterpret_cast<f32x4 *>(dst + (long long)(i0 + row) * Dout + j0 + 4 * c4) = v;
    }
}

// grad[r] += sum of the K-chunk slabs of relation r (r == R: the root matrix), fixed order. A workgroup owns 256
// consecutive elements (4 per lane, 16-byte loads); its 4 waves each add every 4th slab with four requests in flight,
// the four sums are combined as (0+1)+(2+3). (One thread per element walking all slabs with dependent loads took
// 72.7 us at the stress shape: the root matrix has one slab per 256 nodes.)
__global__ __launch_bounds__(256) void rgcn_gen_reduce_w_kernel(const int *__restrict__ chunk_ptr, int R, int Din,
                                                                int Dout, const float *__restrict__ slabs,
                                                                float *__restrict__ grad_basis,
                                                                float *__restrict__ grad_root, int overwrite) {
    // overwrite: every gradient matrix is WRITTEN, once -- the sum of its slabs, or zeros for a relation without an edge:
    // no zero fill in front of the call and no read-modify-write (accumulate mode reads and rewrites every touched matrix
    // behind a 33 MB fill at the stress shape)
    __shared__ f32x4 part[4][64];
    const int r = blockIdx.y;
    const long long elems = (long long)Din * Dout;
    const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const long long idx = ((long long)blockIdx.x * 64 + el) * 4;
    const int c0 = chunk_ptr[r], c1 = chunk_ptr[r + 1];
    float *dst = r < R ? grad_basis : grad_root;
    if (!dst) return;
    if (r < R) dst += (long long)r * elems;
    if (c0 == c1) {                     // (uniform over the workgroup)
        if (overwrite && sg == 0)
            for (int k = 0; k < 4; ++k)
                if (idx + k < elems) dst[idx + k] = 0.f;
        return;
    }
    const bool vec = (elems % 4 == 0) && (((uintptr_t)slabs | (uintptr_t)dst) % 16 == 0);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int count = c1 - c0;
    if (idx < elems) {
        const float *p = slabs + (long long)c0 * elems + idx;
        if (vec) {
            for (int i = sg; i < count; i += 16) {
                f32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gload4(p + (long long)(i + 4 * q < count ? i + 4 * q : i) * elems);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i + 4 * q < count) s += v[q];
            }
        } else {
            for (int i = sg; i < count; i += 4)
                for (int k = 0; k < 4; ++k)
                    if (idx + k < elems) s[k] += p[(long long)i * elems + k];
        }
    }
    part[sg][el] = s;
    __syncthreads();
    if (sg != 0 || idx >= elems) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (idx + k < elems) {
            const float t = (part[0][el][k] + part[1][el][k]) + (part[2][el][k] + part[3][el][k]);
            dst[idx + k] = overwrite ? t : dst[idx + k] + t;
        }
}

// ------------------------------------------------------------------------------------ host side
extern "C" size_t mpqe_rgcn_general_workspace_bytes(int64_t Nn, int64_t E, int64_t R, int64_t Din, int64_t Dout,
                                                    int backward) {
    if (Nn < 0 || E < 0 || R < 0 || Din <= 0 || Dout <= 0) return 0;
    const size_t slots = (size_t)(E + Nn);
    if (!backward) return align_up(slots * (size_t)Dout * 4, 256) + 256;
    return align_up(slots * (size_t)Din * 4, 256) +
           align_up((size_t)chunk_bound(Nn, E, R) * (size_t)Din * (size_t)Dout * 4, 256) +
           bias_partial_bytes(Nn, Dout) + align_up((size_t)Nn * (size_t)((Dout + 63) / 64) * 8, 256) + 256;
}

extern "C" int mpqe_rgcn_general_fwd(const void *plan, int64_t Nn, int64_t E, int64_t R, const float *x,
                                     const float *basis, const float *root, const float *bias, int64_t Din,
                                     int64_t Dout, int relu, float *out, uint64_t *relu_mask, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    if (!plan || Nn < 0 || E < 0 || R < 0 || Din <= 0 || Dout <= 0) return MPQE_ERR_INVALID_ARG;
    if (Nn == 0) return MPQE_OK;
    if (!x || !root || !out || (R > 0 && !basis)) return MPQE_ERR_INVALID_ARG;
    if (relu_mask && (!relu || Dout % 64 != 0 || (uintptr_t)relu_mask % 8 != 0)) return MPQE_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < mpqe_rgcn_general_workspace_bytes(Nn, E, R, Din, Dout, 0))
        return MPQE_ERR_WORKSPACE;
    PlanView P = plan_view(plan, Nn, E, R);
    hipStream_t s = as_stream(stream);
    float *msg = reinterpret_cast<float *>(workspace);
    const bool gvec = (!basis || ptr_vec_ok(basis, Dout)) && ptr_vec_ok(root, Dout) && (Din * Dout) % 4 == 0 &&
                      ptr_vec_ok(x, Din);
    dim3 grid((unsigned)(((tile_bound(Nn, E, R) + 7) / 8 * 8) * ((Dout + GT_BN - 1) / GT_BN)));      // (1-D: see the kernel)
    const bool rows64 = gvec && Din % 64 == 0 && Dout % 64 == 0 && !dbg_on("GEN_LDS_GEMM");
    if (rows64) {
        // persistent: one workgroup per CU (the kernel claims the LDS for that), whole groups of 8 x column groups
        const long long cb1 = Dout / 64, rt1 = (cb1 >= 3 || cb1 == 0) ? 1 : 4 / cb1, cg1 = cb1 >= 4 ? (cb1 + 3) / 4 : 1;
        const long long items1 = (((tile_bound(Nn, E, R) + rt1 - 1) / rt1 + 7) / 8 * 8) * cg1;
        // (GEN_SLOTS, tests: a small grid, so that workgroups walk several tiles)
        const long long per1 = 8 * cg1, slots1 = mpqe_dbg_value("GEN_SLOTS", 256) / per1 * per1;
        dim3 g1((unsigned)(items1 < slots1 || slots1 == 0 || dbg_on("GEN_NOT_PERSISTENT") ? items1 : slots1));
        hipLaunchKernelGGL((rgcn_gen_gemm_rows_kernel<false, false>), g1, dim3(256), 0, s, P.rows_fwd, P.rel_ptr, P.tile_ptr,
                           (int)R, x, (const unsigned long long *)nullptr, basis, root, (int)Din, (int)Dout, msg);
    } else
    if (gvec)
        hipLaunchKernelGGL((rgcn_gen_gemm_kernel<false, LD_PRED>), grid, dim3(256), 0, s, P.rows_fwd, P.rel_ptr,
                           P.tile_ptr, (int)R, x, (const float *)nullptr, basis, root, (int)Din, (int)Dout, msg);
    else
        hipLaunchKernelGGL((rgcn_gen_gemm_kernel<false, LD_SCALAR>), grid, dim3(256), 0, s, P.rows_fwd, P.rel_ptr,
                           P.tile_ptr, (int)R, x, (const float *)nullptr, basis, root, (int)Din, (int)Dout, msg);
    const int vec = Dout % 4 == 0 && ptr_vec_ok(out, Dout) && (!bias || (uintptr_t)bias % 16 == 0) &&
                    (uintptr_t)workspace % 16 == 0;
    const long long threads = Nn * (vec ? Dout / 4 : Dout);
    if (relu_mask && !vec) return MPQE_ERR_INVALID_ARG;       // (the mask words come from the 16-byte form)
    hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, P.dst_ptr,
                       P.dst_list, (long long)Nn, (long long)E, (int)Dout, (const float *)msg, bias, relu, out, vec,
                       reinterpret_cast<unsigned long long *>(relu_mask));
    return mpqe_launch_status();
}

extern "C" size_t mpqe_rgcn_general_mask_bytes(int64_t Nn, int64_t Dout) {
    if (Nn < 0 || Dout <= 0 || Dout % 64 != 0) return 0;
    return (size_t)Nn * (size_t)(Dout / 64) * 8;
}

// The scatter-aggregate step of the forward alone: out[i] = act(bias + msg[E + i] + sum of msg[e] over the edges INTO i
// in edge order), msg = [E + Nn, dim] message rows laid out as mpqe_rgcn_general_fwd's gather-GEMM leaves them. For the
// roofline measurement of bench.py (4 dim (E + 2 Nn) algorithmic bytes per launch) and for callers that form their
// messages elsewhere.
extern "C" int mpqe_rgcn_general_aggregate(const void *plan, int64_t Nn, int64_t E, int64_t R, const float *msg,
                                           const float *bias, int64_t dim, int relu, float *out, void *stream) {
    if (!plan || Nn < 0 || E < 0 || R < 0 || dim <= 0) return MPQE_ERR_INVALID_ARG;
    if (Nn == 0) return MPQE_OK;
    if (!msg || !out) return MPQE_ERR_INVALID_ARG;
    PlanView P = plan_view(plan, Nn, E, R);
    const int vec = dim % 4 == 0 && ptr_vec_ok(out, dim) && ptr_vec_ok(msg, dim) && (!bias || (uintptr_t)bias % 16 == 0);
    const long long threads = Nn * (vec ? dim / 4 : dim);
    hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), P.dst_ptr,
                       P.dst_list, (long long)Nn, (long long)E, (int)dim, msg, bias, relu, out, vec);
    return mpqe_launch_status();
}

extern "C" int mpqe_rgcn_general_bwd(const void *plan, int64_t Nn, int64_t E, int64_t R, const float *x,
                                     const float *out, const uint64_t *relu_mask, const float *grad_out,
                                     const float *basis, const float *root,
                                     int64_t Din, int64_t Dout, int relu, int overwrite, float *grad_x, float *grad_basis,
                                     float *grad_root, float *grad_bias, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    if (!plan || Nn < 0 || E < 0 || R < 0 || Din <= 0 || Dout <= 0) return MPQE_ERR_INVALID_ARG;
    if (Nn == 0) {
        if (overwrite) {
            if (grad_basis && R > 0) (void)hipMemsetAsync(grad_basis, 0, (size_t)R * Din * Dout * 4, as_stream(stream));
            if (grad_root) (void)hipMemsetAsync(grad_root, 0, (size_t)Din * Dout * 4, as_stream(stream));
            if (grad_bias) (void)hipMemsetAsync(grad_bias, 0, (size_t)Dout * 4, as_stream(stream));
        }
        return MPQE_OK;
    }
    if (!x || !grad_out || !root || (R > 0 && !basis) || (relu && !out && !relu_mask)) return MPQE_ERR_INVALID_ARG;
    if (relu_mask && (!relu || Dout % 64 != 0 || (uintptr_t)relu_mask % 8 != 0)) return MPQE_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < mpqe_rgcn_general_workspace_bytes(Nn, E, R, Din, Dout, 1))
        return MPQE_ERR_WORKSPACE;
    PlanView P = plan_view(plan, Nn, E, R);
    hipStream_t s = as_stream(stream);
    char *wb = reinterpret_cast<char *>(workspace);
    float *gmsg = reinterpret_cast<float *>(wb);
    float *slabs = reinterpret_cast<float *>(wb + align_up((size_t)(E + Nn) * (size_t)Din * 4, 256));
    float *bias_part = reinterpret_cast<float *>(
        reinterpret_cast<char *>(slabs) + align_up((size_t)chunk_bound(Nn, E, R) * (size_t)Din * (size_t)Dout * 4, 256));
    const bool gvec = (!basis || ptr_vec_ok(basis, Dout)) && ptr_vec_ok(root, Dout) && (Din * Dout) % 4 == 0 &&
                      ptr_vec_ok(grad_out, Dout) && (!relu || !out || ptr_vec_ok(out, Dout)) && ptr_vec_ok(x, Din);
    const float *mask = relu ? out : nullptr;
    // the ReLU mask as bit words for the register-operand kernels (one pass over `out`, shared with the bias sums)
    unsigned long long *mbits = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(bias_part) +
                                                                       bias_partial_bytes(Nn, Dout));
    const bool rows64_any = gvec && Din % 64 == 0 && Dout % 64 == 0 && !(dbg_on("GEN_LDS_GEMM") && dbg_on("GEN_LDS_GRADW"));
    // (the caller's words, written by the forward's segmented sum -- the backward then never reads `out` --, or made here
    // by the bias sums' pass over `out`)
    const bool use_bits = relu && rows64_any && (grad_x || grad_basis || grad_root);
    if (relu_mask) {
        if (!rows64_any) {
            if (!out) return MPQE_ERR_INVALID_ARG;         // (the LDS-staged kernels mask with `out` itself)
        } else mbits = const_cast<unsigned long long *>(reinterpret_cast<const unsigned long long *>(relu_mask));
    }
    if (relu && !out && !(relu_mask && rows64_any)) return MPQE_ERR_INVALID_ARG;
    if (grad_bias || (use_bits && !relu_mask))
        launch_bias_grad((long long)Nn, grad_out, mask, (int)Dout, relu, bias_part, grad_bias, s, overwrite,
                         use_bits && !relu_mask ? mbits : nullptr, relu_mask && rows64_any ? mbits : nullptr);
    if (grad_x) {
        dim3 grid((unsigned)(((tile_bound(Nn, E, R) + 7) / 8 * 8) * ((Din + GT_BN - 1) / GT_BN)));
        const bool rows64g = gvec && Din % 64 == 0 && Dout % 64 == 0 && !dbg_on("GEN_LDS_GEMM");
        const long long cb1 = Din / 64, rt1 = (cb1 >= 3 || cb1 == 0) ? 1 : 4 / cb1, cg1 = cb1 >= 4 ? (cb1 + 3) / 4 : 1;
        const long long items1 = (((tile_bound(Nn, E, R) + rt1 - 1) / rt1 + 7) / 8 * 8) * cg1;
        const long long per1 = 8 * cg1, slots1 = mpqe_dbg_value("GEN_SLOTS", 512) / per1 * per1;       // (two workgroups per CU)
        dim3 g1((unsigned)(items1 < slots1 || slots1 == 0 || dbg_on("GEN_NOT_PERSISTENT") ? items1 : slots1));
        if (rows64g && relu)
            hipLaunchKernelGGL((rgcn_gen_gemm_rows_kernel<true, true>), g1, dim3(256), 0, s, P.rows_bwd, P.rel_ptr, P.tile_ptr,
                               (int)R, grad_out, (const unsigned long long *)mbits, basis, root, (int)Din, (int)Dout, gmsg);
        else if (rows64g)
            hipLaunchKernelGGL((rgcn_gen_gemm_rows_kernel<true, false>), g1, dim3(256), 0, s, P.rows_bwd, P.rel_ptr, P.tile_ptr,
                               (int)R, grad_out, (const unsigned long long *)nullptr, basis, root, (int)Din, (int)Dout, gmsg);
        else if (gvec)
            hipLaunchKernelGGL((rgcn_gen_gemm_kernel<true, LD_PRED>), grid, dim3(256), 0, s, P.rows_bwd, P.rel_ptr,
                               P.tile_ptr, (int)R, grad_out, mask, basis, root, (int)Din, (int)Dout, gmsg);
        else
            hipLaunchKernelGGL((rgcn_gen_gemm_kernel<true, LD_SCALAR>), grid, dim3(256), 0, s, P.rows_bwd, P.rel_ptr,
                               P.tile_ptr, (int)R, grad_out, mask, basis, root, (int)Din, (int)Dout, gmsg);
        const int vec = Din % 4 == 0 && ptr_vec_ok(grad_x, Din) && (uintptr_t)workspace % 16 == 0;
        const long long threads = Nn * (vec ? Din / 4 : Din);
        hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, P.src_ptr,
                           P.src_list, (long long)Nn, (long long)E, (int)Din, (const float *)gmsg,
                           (const float *)nullptr, 0, grad_x, vec);
    }
    if (grad_basis || grad_root) {
        const int tiles = (int)(((Din + GT_BM - 1) / GT_BM) * ((Dout + GT_BN - 1) / GT_BN));
        dim3 grid((unsigned)chunk_bound(Nn, E, R), tiles);
        const bool rows64 = gvec && Din % 64 == 0 && Dout % 64 == 0 && !dbg_on("GEN_LDS_GRADW");
        const dim3 grid1((unsigned)(((chunk_bound(Nn, E, R) + 7) / 8 * 8) * tiles));      // (rows kernel: 1-D, see there)
        if (rows64 && relu && !dbg_on("GEN_NOMASK"))
            hipLaunchKernelGGL(rgcn_gen_grad_w_rows_kernel<true>, grid1, dim3(256), 0, s, P.rows_fwd, P.rows_bwd, P.rel_ptr,
                               P.chunk_ptr, (int)R, x, grad_out, (const unsigned long long *)mbits, (int)Din, (int)Dout, slabs);
        else if (rows64)
            hipLaunchKernelGGL(rgcn_gen_grad_w_rows_kernel<false>, grid1, dim3(256), 0, s, P.rows_fwd, P.rows_bwd, P.rel_ptr,
                               P.chunk_ptr, (int)R, x, grad_out, (const unsigned long long *)nullptr, (int)Din, (int)Dout, slabs);
        else if (gvec)
            hipLaunchKernelGGL(rgcn_gen_grad_w_kernel<LD_PRED>, grid, dim3(256), 0, s, P.rows_fwd, P.rows_bwd, P.rel_ptr,
                               P.chunk_ptr, (int)R, x, grad_out, mask, (int)Din, (int)Dout, relu, slabs);
        else
            hipLaunchKernelGGL(rgcn_gen_grad_w_kernel<LD_SCALAR>, grid, dim3(256), 0, s, P.rows_fwd, P.rows_bwd,
                               P.rel_ptr, P.chunk_ptr, (int)R, x, grad_out, mask, (int)Din, (int)Dout, relu, slabs);
        const long long elems = (long long)Din * Dout;
        dim3 rgrid((unsigned)((elems + 255) / 256), (unsigned)(R + 1));
        hipLaunchKernelGGL(rgcn_gen_reduce_w_kernel, rgrid, dim3(256), 0, s, P.chunk_ptr, (int)R, (int)Din,
                           (int)Dout, (const float *)slabs, grad_basis, grad_root, overwrite);
    }
    return mpqe_launch_status();
}

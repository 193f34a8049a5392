// Entity-table gradients of the fused step WITHOUT float atomics ("touch plan").
//
// The reference's tables are dense nn.Embedding parameters (data_utils.py:31) and autograd scatters the
// gradient rows of the looked-up entities into them with index_add (deterministic on the CPU). The first form
// of the chain kernel did the same with fp32 atomics: 10.7 MB of added bytes per AIFB step against a chip-wide
// atomic rate of ~1.3 TB/s, in wave-instructions of the wrong shape -- measured 11 us of a 42 us kernel, and
// results that changed in the last bits from run to run.
//
// Instead: every looked-up id is an ENTRY e (anchors in the order of the step's anchor_ids array, then the
// positive targets, then the negative ones: the order of the id arrays the caller hands over). The chain kernel
// stores the gradient row of entry e (through the L2 normalisation) with plain 16-byte stores; which entries share a
// destination row is a function of the ids alone, so it is found ONCE per packed step, at pack time
// (mpqe_step_touch_build: LUT gather -> (table, row) keys -> one stable radix sort -> pos[e] = rank of entry e), not
// in the step. The row of entry e goes to DG[pos[e]], so the rows of one destination are ADJACENT, in entry order, and
// table_sum_block adds each run in that fixed order with no indirection: deterministic, every gradient row read once.
// Included by step.hip.
#pragma once
#include <rocprim/device/device_radix_sort.hpp>

typedef unsigned long long tkey_t;
#define TOUCH_INVALID (~0ull)

struct TouchHeader {
    long long M;            // entries
    int row_bits, key_bits; // key = table << row_bits | row
    int pad[12];
};
struct TouchLayout {
    size_t keys, perm, erow, total;     // device buffer: header, sorted keys [M], pos [M] (entry -> rank in sorted order),
                                        // erow [M] (entry -> row of its table, -1 = bad id: the LUT hop, done at pack time)
    size_t w_keys, w_vals, w_svals, w_tmp, w_tmp_bytes, w_total;      // build workspace
};

static inline long long touch_entries(const mpqe_step_batch_t *B, int nb) {
    long long m = 0;
    for (int i = 0; i < nb; ++i) {
        if (B[i].query_type < 0 || B[i].query_type >= MPQE_Q_COUNT || B[i].batch_size <= 0) return -1;
        m += (long long)B[i].batch_size * (kTemplates[B[i].query_type].A + 2);
    }
    return m;
}
static inline TouchLayout touch_layout(long long M, int key_bits /* 0: the device buffer's offsets only */) {
    TouchLayout L;
    memset(&L, 0, sizeof(L));
    size_t off = 256;
    L.keys = off;
    off += align_up((size_t)M * sizeof(tkey_t), 256);
    L.perm = off;
    off += align_up((size_t)M * sizeof(int), 256);
    L.erow = off;
    off += align_up((size_t)M * sizeof(int), 256);
    L.total = off;
    if (key_bits <= 0) return L;
    off = 0;
    L.w_keys = off;
    off += align_up((size_t)M * sizeof(tkey_t), 256);
    L.w_vals = off;
    off += align_up((size_t)M * sizeof(int), 256);
    L.w_svals = off;
    off += align_up((size_t)M * sizeof(int), 256);
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const tkey_t *)nullptr, (tkey_t *)nullptr, (const int *)nullptr,
                                    (int *)nullptr, (size_t)(M > 0 ? M : 1), 0u, (unsigned)key_bits, (hipStream_t) nullptr);
    L.w_tmp = off;
    L.w_tmp_bytes = bytes;
    off += align_up(bytes, 256);
    L.w_total = off;
    return L;
}
static inline int touch_bits(long long v) {
    int b = 1;
    while ((1ll << b) <= v && b < 40) ++b;
    return b;
}

// per batch: where its entries start and which tables they index (by value: <= 16 batches)
struct TouchMeta {
    int nb;
    int B[MPQE_STEP_MAX_BATCHES], A[MPQE_STEP_MAX_BATCHES];
    long long anchor_off[MPQE_STEP_MAX_BATCHES + 1], g_off[MPQE_STEP_MAX_BATCHES + 1];
    int anchor_tab[MPQE_STEP_MAX_BATCHES][3], target_tab[MPQE_STEP_MAX_BATCHES];
    long long table_rows[MPQE_STEP_MAX_MODES];
    int row_bits;
};

// entry e -> key (table << row_bits | row of the entity in its table; TOUCH_INVALID for a bad id) and its table row
__device__ __forceinline__ tkey_t touch_key_of(const TouchMeta &tm, long long e, const long long *__restrict__ anchor_ids,
                                               const long long *__restrict__ targets, const long long *__restrict__ negs,
                                               const long long *__restrict__ node_map, long long map_len, int *er_out) {
    const long long Manchor = tm.anchor_off[tm.nb], G = tm.g_off[tm.nb];
    long long id;
    int tab;
    if (e < Manchor) {
        int bi = 0;
        for (int i = 1; i < tm.nb; ++i)
            if (tm.anchor_off[i] <= e) bi = i;
        const long long lr = e - tm.anchor_off[bi];      // slot-major inside the batch: [A][B]
        tab = tm.anchor_tab[bi][(int)(lr / tm.B[bi])];
        id = anchor_ids[e];
    } else {
        const long long gi = (e - Manchor) % G;
        int bi = 0;
        for (int i = 1; i < tm.nb; ++i)
            if (tm.g_off[i] <= gi) bi = i;
        tab = tm.target_tab[bi];
        id = (e - Manchor) >= G ? negs[gi] : targets[gi];
    }
    tkey_t key = TOUCH_INVALID;
    int er = -1;
    if (id >= 0 && id < map_len) {
        const long long r = node_map[id];
        if (r >= 0 && r < tm.table_rows[tab]) {
            key = ((tkey_t)tab << tm.row_bits) | (tkey_t)r;
            er = r <= 0x7fffffffll ? (int)r : -2;      // (-2: a table beyond 2^31 rows -- the step resolves the id itself)
        }
    }
    *er_out = er;
    return key;
}
__global__ __launch_bounds__(256) void touch_keys_kernel(TouchMeta tm, const long long *__restrict__ anchor_ids,
                                                        const long long *__restrict__ targets,
                                                        const long long *__restrict__ negs,
                                                        const long long *__restrict__ node_map, long long map_len,
                                                        tkey_t *__restrict__ keys, int *__restrict__ vals,
                                                        int *__restrict__ erow, long long M, TouchHeader th,
                                                        TouchHeader *__restrict__ th_out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e == 0 && th_out) *th_out = th;
    if (e >= M) return;
    int er;
    keys[e] = touch_key_of(tm, e, anchor_ids, targets, negs, node_map, map_len, &er);   // (invalid ids sort to the end; the
    vals[e] = (int)e;                                                                   // step itself flags them)
    if (erow) erow[e] = er;
}

// (A single-workgroup sort of the whole plan -- one launch instead of rocPRIM's chain of ~6 -- was built and measured: a
// stable 4-bit LSD radix sort by 1024 threads, 28 entries each. Host time of the build 25 -> 13 us, but 254 us on the
// device against 45: at 128 VGPRs per thread the entries spill to scratch, and one CU's memory latency is all there is
// to hide. Not kept. Replaying copy + keys + library sort + inversion as ONE hipGraph costs 10 - 21 us of host time and
// reproduces the plan (tools/graph_pack_probe.py): the next step for the host side of pack.)

// pos[vals_sorted[k]] = k
__global__ __launch_bounds__(256) void touch_invert_kernel(const int *__restrict__ sorted_vals, int *__restrict__ pos,
                                                          long long M) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k < M) pos[sorted_vals[k]] = (int)k;
}

// One group of LPR = D / 4 lanes per sorted position k: if k starts a run of equal keys, add the run's gradient rows
// DG[k], DG[k + 1], ... in order and write the destination row. The keys and rows of the next TS_AHEAD positions are
// requested together (the rows do not depend on the keys: a row beyond the run's end is loaded and dropped), so a run
// costs one round trip per TS_AHEAD rows, not two per row. store: the call zero-filled the gradients, the row is
// written; otherwise added to what is there.
#define TS_AHEAD 8
// perm != NULL: the row of sorted position k is DG[perm[k]] (rows that arrive in another order: the data-parallel row
// exchange, mpqe_table_rows_sum); NULL: DG[k] (the chain kernel stores its rows in sorted position).
template <class TabsT>
__device__ __forceinline__ void table_sum_block(const TouchHeader *__restrict__ th, const tkey_t *__restrict__ keys,
                                                const int *__restrict__ perm, const float *__restrict__ DG, int D,
                                                const TabsT &tabs, int store, long long block) {
    const int lpr = D / 4, per = 256 / lpr;          // positions per workgroup (D = 64 / 128 / 256: 16 / 8 / 4)
    const long long M = th->M;
    const long long k = block * per + threadIdx.x / lpr;
    const int c = (threadIdx.x % lpr) * 4;
    if (k >= M) return;
    const tkey_t key = keys[k];
    if (key == TOUCH_INVALID || (k > 0 && keys[k - 1] == key)) return;
    f32x4 acc = gload4(DG + (perm ? (long long)perm[k] : k) * D + c);
    for (long long j0 = k + 1; j0 < M; j0 += TS_AHEAD) {
        tkey_t kk[TS_AHEAD];
        f32x4 v[TS_AHEAD];
        if (perm) {
            long long pj[TS_AHEAD];
#pragma unroll
            for (int q = 0; q < TS_AHEAD; ++q) {
                const long long j = j0 + q < M ? j0 + q : M - 1;
                kk[q] = keys[j];
                pj[q] = perm[j];
            }
#pragma unroll
            for (int q = 0; q < TS_AHEAD; ++q) v[q] = gload4(DG + pj[q] * D + c);
        } else {
#pragma unroll
            for (int q = 0; q < TS_AHEAD; ++q) {
                const long long j = j0 + q < M ? j0 + q : M - 1;
                kk[q] = keys[j];
                v[q] = gload4(DG + j * D + c);
            }
        }
        bool more = true;
#pragma unroll
        for (int q = 0; q < TS_AHEAD; ++q) {
            more = more && j0 + q < M && kk[q] == key;
            if (more) acc += v[q];
        }
        if (!more) break;
    }
    const int tab = (int)(key >> th->row_bits);
    const long long row = (long long)(key & ((1ull << th->row_bits) - 1ull));
    float *g = tabs.grad[0];      // (a runtime index into the by-value table would spill it to scratch)
#pragma unroll
    for (int m = 1; m < MPQE_STEP_MAX_MODES; ++m)
        if (m == tab) g = tabs.grad[m];
    if (!g) return;
    f32x4 *dst = reinterpret_cast<f32x4 *>(g + row * D + c);
    if (store) *dst = acc;
    else *dst = *dst + acc;
}

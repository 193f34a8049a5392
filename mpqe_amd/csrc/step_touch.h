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
// stores the gradient row of entry e (through the L2 normalisation) to DG[e] with plain 16-byte stores and needs
// nothing else. Which entries share a destination row is a function of the ids alone: LUT gather -> (table, row)
// keys -> one stable radix sort -> sorted keys + perm[k] = entry of sorted position k; table_sum_block (reduction
// launch) adds, per run of equal keys, the rows DG[perm[k]] in sorted = entry order: deterministic, every gradient
// row read once. The sort is needed only by the step's LAST launch, so it has two homes:
//   * INSIDE the step (MPQE_STEP_BUILD_TOUCH): tsort_block workgroups lead the chain launch's grid and run beside the
//     chain workgroups -- a step with fresh ids costs what a replayed one costs, and nothing id-dependent happens
//     outside mpqe_step_forward_backward (round 2 built the plan at pack time: 21 us of sort + the id -> row hop per
//     step that the timed loop never saw);
//   * at pack time (mpqe_step_touch_build), for callers that want the plan before the step runs (the data-parallel
//     row exchange plans with its keys) or whose steps exceed TSORT_MAX_ENTRIES.
// Included by step.hip.
#pragma once
#include "radix_sort.h"

typedef unsigned long long tkey_t;
#define TOUCH_INVALID (~0ull)

// the one-launch sort (tsort_block, below): workgroups of 256 threads, four entries per thread -- eight for plans beyond
// 256 x 1024 entries (AIFB step: 22 workgroups, sort done at ~29 us with four, 11 workgroups and ~37 us with eight)
#define TSORT_THREADS 256
#define TSORT_ROUNDS 8
#define TSORT_PER_BLOCK (TSORT_THREADS * TSORT_ROUNDS)
#define TSORT_MAX_BLOCKS 256
#define TSORT_MAX_ENTRIES ((long long)TSORT_MAX_BLOCKS * TSORT_PER_BLOCK)
static inline int tsort_rounds(long long M) { return M <= (long long)TSORT_MAX_BLOCKS * TSORT_THREADS * 4 ? 4 : TSORT_ROUNDS; }
static inline int tsort_blocks(long long M) {
    const long long per = (long long)TSORT_THREADS * tsort_rounds(M);
    return (int)((M + per - 1) / per);
}

struct TouchHeader {
    long long M;            // entries
    int row_bits, key_bits; // key = table << row_bits | row
    int pad[12];
};
struct TouchLayout {
    size_t keys, perm, erow, total;     // device buffer: header, sorted keys [M], perm [M] (sorted position -> entry),
                                        // erow [M] (entry -> row of its table, -1 = bad id: the LUT hop; pack-time build only)
    size_t w_keys, w_vals, w_svals, w_hist, w_tmp, w_tmp_bytes, w_total;      // build workspace
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
    const size_t Mp = ((size_t)M + TSORT_PER_BLOCK - 1) / TSORT_PER_BLOCK * TSORT_PER_BLOCK;      // (whole workgroups of the one-launch sort)
    L.w_keys = off;
    off += align_up(Mp * sizeof(tkey_t), 256);
    L.w_vals = off;
    off += align_up(Mp * sizeof(int), 256);
    L.w_svals = off;
    off += align_up(Mp * sizeof(int), 256);
    L.w_hist = off;                                           // [4 passes][256 workgroups][256] digit counts, the barrier
    off += 4 * 256 * 256 * sizeof(unsigned) + 256 + 2048;     // counter (256 bytes), the batch table (TouchMeta, < 2 KB)
    const size_t bytes = radix_sort_tmp_bytes<tkey_t>(M > 0 ? M : 1);
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

// The whole touch plan by workgroups that synchronise among themselves (the library sort is a chain of ~6 launches of
// 3 - 8 us each, latency-bound at these sizes: ~54 us of device time). NB = ceil(M / 2048) workgroups of 256 threads,
// EIGHT entries per thread (entry = 2048 b + 256 r + t in round r: a round is a "virtual" group of four waves), all
// workgroups resident at once; a stable LSD radix sort, 8 bits per pass:
//   1  digit d of my key; my rank among the entries of MY WAVE AND ROUND with the same digit (eight ballots build the
//      mask of equal-digit lanes), the count per (round, wave, digit) to LDS; thread d turns the 16 counts of digit d into
//      an exclusive prefix -> the workgroup's count of digit d, stored to hist[pass][workgroup][d]
//   -- grid barrier --
//   2  thread d sums hist[.][d] over ALL workgroups (total of the digit) and over the workgroups before mine; a
//      256-wide scan of the totals; destination = (entries with a smaller digit) + (same digit, earlier workgroups) +
//      (same digit, earlier rounds / waves of mine) + (same digit, lower lanes of my wave): stable
//   3  scatter (key, entry) to the other buffer     -- grid barrier --     next pass reads its entries from there;
//      the LAST pass writes the plan itself: sorted key (64-bit, invalid = all ones) and perm[rank] = entry.
// Everything that crosses workgroups (keys, entries, histograms) moves through agent-scope atomic stores / loads -- written
// through to memory, read past the L1 -- so the grid barrier is a counter and nothing else: no L2 write-back, no
// invalidate. EVERY wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup's barrier in front of the counter
// add: a wave's relaxed stores may otherwise still be in flight when another workgroup passes the barrier. The counter
// is zero when the launch starts (pack-time build: a 4-byte memset in front of it; inside the step: the reduction launch
// of the previous step leaves it zero); spins are bounded (a launch that cannot make progress leaves the header's
// `pad[0]` = 1 instead of hanging -- e.g. when not all NB workgroups fit on the device together).
#define TSORT_META_WORDS 512      // sizeof(TouchMeta) / 4, rounded up
#define TSORT_LDS_WORDS (TSORT_ROUNDS * 4 * 256 / 2 + 256 + 256 + 4 + TSORT_META_WORDS)      // (the digit counts are 16-bit)
struct TSortArgs {
    const TouchMeta *tm;        // device copy (the packed step's descriptor table / the build's workspace)
    const long long *anchor_ids, *targets, *negs, *node_map;
    long long map_len;
    unsigned *ka, *va, *kb, *vb;        // ping-pong (key, entry) buffers, [nblk * TSORT_PER_BLOCK] each
    unsigned *hist;                     // [passes][nblk][256]
    unsigned *counter;                  // the grid barrier
    tkey_t *keys_out;
    int *perm, *erow;                   // erow: NULL = not wanted
    TouchHeader *th_out;
    int M, key_bits, row_bits, nblk;
    int rounds;                         // entries per thread (tsort_rounds)
    int fail;                           // diagnostics ("TSORT_FAIL"): give up at once, as a sort whose workgroups are not co-resident does
    long long *stamps;                  // diagnostics (mpqe_debug_chain_stamps): 8 words per sort workgroup, or NULL
};
#ifndef MPQE_EMU
__device__ __forceinline__ unsigned tsort_ld(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tsort_st(unsigned *p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool tsort_grid_barrier(unsigned *counter, unsigned target, unsigned *ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: my agent-scope stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned good = 1;
        for (int spins = 0; (int)(tsort_ld(counter) - target) < 0; ++spins) {
            if (spins >= (1 << 21)) {
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        *ok = good;
    }
    __syncthreads();
    return *ok != 0;
}
// one workgroup of the sort; `smem`: TSORT_LDS_WORDS words. Threads >= TSORT_THREADS of a larger workgroup must not call.
__device__ __forceinline__ void tsort_block(const TSortArgs &sa, int b, unsigned *smem) {
    // [round * 4 + wave][digit]: counts of at most 64, their prefix over the workgroup's 2048 entries at most 2048: 16 bits
    unsigned short(*whist)[256] = reinterpret_cast<unsigned short(*)[256]>(smem);
    unsigned *gbase = smem + TSORT_ROUNDS * 4 * 256 / 2, *scan = gbase + 256, *ok = scan + 256;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nblk = sa.nblk, M = sa.M;
    if (sa.fail) {         // (uniform; the plan's arrays stay as they are: garbage that nobody may use)
        if (b == 0 && t == 0) {
            TouchHeader th;
            memset(&th, 0, sizeof(th));
            th.M = M;
            th.row_bits = sa.row_bits;
            th.key_bits = sa.key_bits;
            th.pad[0] = 1;
            *sa.th_out = th;
        }
        return;
    }
    // the batch table through LDS: touch_key_of walks it per entry (from memory that was five dependent round trips in
    // front of the first id load -- 9 to 16 us next to running chain workgroups)
    static_assert(sizeof(TouchMeta) <= TSORT_META_WORDS * 4 && sizeof(TouchMeta) % 4 == 0, "TSORT_META_WORDS");
    unsigned *mw = ok + 4;
    for (int q = t; q < (int)(sizeof(TouchMeta) / 4); q += TSORT_THREADS) mw[q] = reinterpret_cast<const unsigned *>(sa.tm)[q];
    __syncthreads();
    const TouchMeta &tm = *reinterpret_cast<const TouchMeta *>(mw);
    int stamp_i = 0;
    auto stamp = [&]() {         // wall clock (100 MHz) at: start, keys, then after every grid barrier, end (slot 7)
        if (sa.stamps && t == 0 && stamp_i < 7) sa.stamps[(long long)b * 8 + stamp_i++] = (long long)wall_clock64();
    };
    stamp();
    const int R = sa.rounds, per_block = R * TSORT_THREADS;      // (uniform: 4 or 8 entries per thread)
    unsigned k[TSORT_ROUNDS], v[TSORT_ROUNDS];
#pragma unroll
    for (int r = 0; r < TSORT_ROUNDS; ++r) {
        const int i = b * per_block + r * TSORT_THREADS + t;
        k[r] = 0xffffffffu;                                 // (slots beyond M: a key behind every real one, never stored)
        v[r] = (unsigned)i;
        if (r < R && i < M) {
            int er;
            const tkey_t key = touch_key_of(tm, i, sa.anchor_ids, sa.targets, sa.negs, sa.node_map, sa.map_len, &er);
            k[r] = key == TOUCH_INVALID ? 0xffffffffu : (unsigned)key;
            if (sa.erow) sa.erow[i] = er;
        }
    }
    const int passes = (sa.key_bits + 7) / 8;
    unsigned *dk = sa.ka, *dv = sa.va;
    unsigned bar = 0;
    bool good = true;
    stamp();
    for (int p = 0; p < passes; ++p) {
        for (int q = t; q < TSORT_ROUNDS * 4 * 256 / 2; q += TSORT_THREADS) smem[q] = 0;
        __syncthreads();
        unsigned below[TSORT_ROUNDS];
#pragma unroll
        for (int r = 0; r < TSORT_ROUNDS; ++r) {
            if (r >= R) continue;                          // (uniform)
            const unsigned d = (k[r] >> (8 * p)) & 255u;
            unsigned long long peers = ~0ull;              // lanes of my wave whose entry of this round has my digit
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const unsigned long long m = __ballot((d >> bit) & 1u);
                peers &= ((d >> bit) & 1u) ? m : ~m;
            }
            below[r] = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
            if (below[r] == 0) whist[r * 4 + wave][d] = (unsigned short)__popcll(peers);
        }
        __syncthreads();
        {   // exclusive prefix over my workgroup's (round, wave) groups; its count of digit t
            unsigned run = 0;
#pragma unroll
            for (int w = 0; w < TSORT_ROUNDS * 4; ++w) {
                const unsigned c = whist[w][t];
                whist[w][t] = (unsigned short)run;
                run += c;
            }
            tsort_st(sa.hist + ((size_t)p * nblk + b) * 256 + t, run);
        }
        bar += (unsigned)nblk;
        good = tsort_grid_barrier(sa.counter, bar, ok) && good;
        stamp();
        {
            unsigned tot = 0, before = 0;
            // (agent-scope loads -- the rows are re-used by the next step's sort, a line of them may sit in this XCD's L2 --
            // 16 in flight together: one load per workgroup in a dependent row was 28 round trips per pass)
            const unsigned *hp = sa.hist + (size_t)p * nblk * 256 + t;
            for (int b0 = 0; b0 < nblk; b0 += 16) {
                unsigned h[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) h[q] = b0 + q < nblk ? tsort_ld(hp + (size_t)(b0 + q) * 256) : 0u;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    tot += h[q];
                    if (b0 + q < b) before += h[q];
                }
            }
            scan[t] = tot;
            gbase[t] = before - tot;                       // (+ the inclusive scan below = entries with a smaller digit + before)
        }
        {   // inclusive scan of the 256 digit totals: inside each wave by shuffles, the waves' sums through LDS
            unsigned x = scan[t];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned y = __shfl_up(x, off, 64);
                if (lane >= off) x += y;
            }
            if (lane == 63 && wave < 3) ok[1 + wave] = x;  // (ok[1..3]: the sums of waves 0..2; 4 words are reserved)
            __syncthreads();
            unsigned pre = 0;
            for (int w = 0; w < wave; ++w) pre += ok[1 + w];
            scan[t] = x + pre;
        }
        __syncthreads();
        unsigned at[TSORT_ROUNDS];
#pragma unroll
        for (int r = 0; r < TSORT_ROUNDS; ++r) {
            const unsigned d = (k[r] >> (8 * p)) & 255u;
            at[r] = gbase[d] + scan[d] + whist[r * 4 + wave][d] + below[r];
        }
        if (p + 1 < passes) {
#pragma unroll
            for (int r = 0; r < TSORT_ROUNDS; ++r)
                if (r < R) {
                    tsort_st(dk + at[r], k[r]);            // (the padding keys travel too: they stay behind every real key)
                    tsort_st(dv + at[r], v[r]);
                }
            bar += (unsigned)nblk;
            good = tsort_grid_barrier(sa.counter, bar, ok) && good;
            stamp();
#pragma unroll
            for (int r = 0; r < TSORT_ROUNDS; ++r)
                if (r < R) {
                    const int i = b * per_block + r * TSORT_THREADS + t;
                    k[r] = tsort_ld(dk + i);
                    v[r] = tsort_ld(dv + i);
                }
            dk = dk == sa.ka ? sa.kb : sa.ka;
            dv = dv == sa.va ? sa.vb : sa.va;
        } else {
#pragma unroll
            for (int r = 0; r < TSORT_ROUNDS; ++r)
                if (r < R && v[r] < (unsigned)M) {         // the plan: sorted key, entry of the rank
                    sa.keys_out[at[r]] = k[r] == 0xffffffffu ? TOUCH_INVALID : (tkey_t)k[r];
                    sa.perm[at[r]] = (int)v[r];
                }
        }
    }
    if (sa.stamps && t == 0) sa.stamps[(long long)b * 8 + 7] = (long long)wall_clock64();
    if (t == 0 && (b == 0 || !good)) {                     // (a workgroup that gave up says so, whichever it is)
        TouchHeader th;
        memset(&th, 0, sizeof(th));
        th.M = M;
        th.row_bits = sa.row_bits;
        th.key_bits = sa.key_bits;
        if (b == 0 && good) *sa.th_out = th;
        else if (!good) {
            if (b == 0) *sa.th_out = th;
            __hip_atomic_store(&sa.th_out->pad[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
#else
// emulator (workgroups run one after another: no grid barrier): workgroup 0's first thread sorts the whole plan
__device__ __forceinline__ void tsort_block(const TSortArgs &sa, int b, unsigned *) {
    if (b != 0 || threadIdx.x != 0) return;
    const int M = sa.M;
    if (sa.fail) {
        TouchHeader th;
        memset(&th, 0, sizeof(th));
        th.M = M;
        th.row_bits = sa.row_bits;
        th.key_bits = sa.key_bits;
        th.pad[0] = 1;
        *sa.th_out = th;
        return;
    }
    std::vector<std::pair<tkey_t, int>> e((size_t)M);
    for (int i = 0; i < M; ++i) {
        int er;
        e[i].first = touch_key_of(*sa.tm, i, sa.anchor_ids, sa.targets, sa.negs, sa.node_map, sa.map_len, &er);
        e[i].second = i;
        if (sa.erow) sa.erow[i] = er;
    }
    std::stable_sort(e.begin(), e.end(), [](const std::pair<tkey_t, int> &x, const std::pair<tkey_t, int> &y) { return x.first < y.first; });
    for (int i = 0; i < M; ++i) {
        sa.keys_out[i] = e[i].first;
        sa.perm[i] = e[i].second;
    }
    TouchHeader th;
    memset(&th, 0, sizeof(th));
    th.M = M;
    th.row_bits = sa.row_bits;
    th.key_bits = sa.key_bits;
    *sa.th_out = th;
}
#endif
// pack-time build: the sort as a launch of its own
__global__ __launch_bounds__(TSORT_THREADS) void touch_sort_kernel(TSortArgs sa) {
    __shared__ unsigned smem[TSORT_LDS_WORDS];
    tsort_block(sa, (int)blockIdx.x, smem);
}

// One group of LPR = D / 4 lanes per sorted position k: if k starts a run of equal keys, add the run's gradient rows
// DG[k], DG[k + 1], ... in order and write the destination row. The keys and rows of the next TS_AHEAD positions are
// requested together (the rows do not depend on the keys: a row beyond the run's end is loaded and dropped), so a run
// costs one round trip per TS_AHEAD rows, not two per row. store: the call zero-filled the gradients, the row is
// written; otherwise added to what is there.
#define TS_AHEAD 8
// perm != NULL: the row of sorted position k is DG[perm[k]] (the chain kernel stores its rows in entry order; the
// data-parallel row exchange, mpqe_table_rows_sum, in gathered order); NULL: DG[k].
// (M, row_bits: the plan header's fields, by value where the caller knows them -- the fused step does: one round trip less
// in front of the keys)
// failed != NULL: a word that is non-zero when the plan could not be built (the header's pad[0] of a plan the step built
// itself: step.hip) -- nothing is stored then; the word travels with the first keys. The LOADS are made before it is
// known: a plan buffer must therefore hold valid entries at all times -- the caller zero-fills it once, before its first
// use (entry 0 / key 0 are valid), and a build that gives up leaves what an earlier build or the fill left there. (Clamping
// every permutation entry instead cost the reduction launch ~1 us.)
// runs != NULL: the lane groups take the run STARTS listed there (touch_runs_block: runs[0 .. *nruns), any order) instead of
// every sorted position -- a step's distinct rows are at most the tables' rows, four to five times fewer than its entries on
// the AIFB step, and the launch is sized for that bound.
template <class TabsT>
__device__ __forceinline__ void table_sum_block(long long M, int row_bits, const tkey_t *__restrict__ keys,
                                                const int *__restrict__ perm, const float *__restrict__ DG, int D,
                                                const TabsT &tabs, int store, long long block,
                                                const int *__restrict__ failed = nullptr,
                                                const int *__restrict__ runs = nullptr, const int *__restrict__ nruns = nullptr) {
    const int lpr = D / 4, per = 256 / lpr;          // positions per workgroup (D = 64 / 128 / 256: 16 / 8 / 4)
    long long k = block * per + threadIdx.x / lpr;
    const int c = (threadIdx.x % lpr) * 4;
    if (runs) {
        if (k >= (long long)*nruns) return;
        k = runs[k];
    }
    if (k >= M) return;
    const int bad = failed ? *failed : 0;
    // ONE round trip for everything that depends on k alone: my key, my predecessor's, and the keys / permutation entries of
    // the next TS_AHEAD positions (a run is ~8 rows long on the AIFB step); a second one for the rows. (Requested one after
    // the other -- key, then permutation entry, then row, then the next keys ... -- a run cost five dependent round trips.)
    const tkey_t key = keys[k];
    const tkey_t prev = k > 0 ? keys[k - 1] : TOUCH_INVALID;
    const long long p0 = perm ? (long long)perm[k] : k;
    tkey_t kk[TS_AHEAD];
    long long pj[TS_AHEAD];
#pragma unroll
    for (int q = 0; q < TS_AHEAD; ++q) {
        const long long j = k + 1 + q < M ? k + 1 + q : M - 1;
        kk[q] = keys[j];
        pj[q] = perm ? (long long)perm[j] : j;
    }
    if (bad || key == TOUCH_INVALID || (k > 0 && prev == key)) return;
    f32x4 acc = gload4(DG + p0 * D + c);
    {
        f32x4 v[TS_AHEAD];
        bool on[TS_AHEAD];
        bool more = true;
#pragma unroll
        for (int q = 0; q < TS_AHEAD; ++q) {          // (rows beyond the run's end are not requested)
            more = more && k + 1 + q < M && kk[q] == key;
            on[q] = more;
            v[q] = more ? gload4(DG + pj[q] * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < TS_AHEAD; ++q)
            if (on[q]) acc += v[q];
        if (more) {                                   // a longer run: the rest in chunks, as before
            for (long long j0 = k + 1 + TS_AHEAD; j0 < M; j0 += TS_AHEAD) {
                tkey_t k2[TS_AHEAD];
                long long p2[TS_AHEAD];
#pragma unroll
                for (int q = 0; q < TS_AHEAD; ++q) {
                    const long long j = j0 + q < M ? j0 + q : M - 1;
                    k2[q] = keys[j];
                    p2[q] = perm ? (long long)perm[j] : j;
                }
                f32x4 w[TS_AHEAD];
#pragma unroll
                for (int q = 0; q < TS_AHEAD; ++q) w[q] = gload4(DG + p2[q] * D + c);
                bool go = true;
#pragma unroll
                for (int q = 0; q < TS_AHEAD; ++q) {
                    go = go && j0 + q < M && k2[q] == key;
                    if (go) acc += w[q];
                }
                if (!go) break;
            }
        }
    }
    const int tab = (int)(key >> row_bits);
    const long long row = (long long)(key & ((1ull << row_bits) - 1ull));
    float *g = tabs.grad[0];      // (a runtime index into the by-value table would spill it to scratch)
#pragma unroll
    for (int m = 1; m < MPQE_STEP_MAX_MODES; ++m)
        if (m == tab) g = tabs.grad[m];
    if (!g) return;
    f32x4 *dst = reinterpret_cast<f32x4 *>(g + row * D + c);
    if (store) *dst = acc;
    else *dst = *dst + acc;
}

// The run starts of a plan's sorted keys, compacted (a few workgroups of 256 threads, TRUNS_PER positions each; roles of the
// weight-gradient launch, which has the time: step.hip): runs[0 .. *count) = the sorted positions k whose key is valid and
// differs from its predecessor's, in NO particular order (every run is summed on its own: the order of the list changes no
// result) -- a thread's positions are requested together, a workgroup's starts are numbered through LDS and take their
// place in the list with ONE atomic add on *count (zeroed by the chain launch). A plan that could not be built adds nothing.
// lds: 2 ints.
#define TRUNS_EACH 8
#define TRUNS_PER (256 * TRUNS_EACH)
__device__ __forceinline__ void touch_runs_block(long long M, const tkey_t *__restrict__ keys, int *__restrict__ runs,
                                                 int *__restrict__ count, const int *__restrict__ failed, int *lds, int block) {
    const int tid = threadIdx.x;
    const int bad = failed ? *failed : 0;
    const long long base = (long long)block * TRUNS_PER;
    tkey_t key[TRUNS_EACH], prev[TRUNS_EACH];
#pragma unroll
    for (int q = 0; q < TRUNS_EACH; ++q) {              // (coalesced: position base + 256 q + tid)
        const long long k = base + 256 * q + tid;
        key[q] = k < M ? keys[k] : TOUCH_INVALID;
        prev[q] = (k > 0 && k < M) ? keys[k - 1] : TOUCH_INVALID;
    }
    if (tid == 0) lds[0] = 0;
    __syncthreads();
    int c = 0;
#pragma unroll
    for (int q = 0; q < TRUNS_EACH; ++q) {
        const long long k = base + 256 * q + tid;
        c += (!bad && key[q] != TOUCH_INVALID && (k == 0 || prev[q] != key[q])) ? 1 : 0;
    }
    const int mine = c > 0 ? (int)atomicAdd(reinterpret_cast<unsigned *>(&lds[0]), (unsigned)c) : 0;
    __syncthreads();
    if (tid == 0) lds[1] = lds[0] > 0 ? (int)atomicAdd(reinterpret_cast<unsigned *>(count), (unsigned)lds[0]) : 0;
    __syncthreads();
    int at = lds[1] + mine;
#pragma unroll
    for (int q = 0; q < TRUNS_EACH; ++q) {
        const long long k = base + 256 * q + tid;
        if (!bad && key[q] != TOUCH_INVALID && (k == 0 || prev[q] != key[q])) runs[at++] = (int)k;
    }
}

// The same sums by workgroups that take a RANGE of sorted positions each (roles of a launch whose register footprint allows
// few workgroups per CU: thousands of one-run workgroups would queue -- step.hip: the weight-gradient launch). A workgroup
// owns TSM_OWN positions per lane group (D / 4 lanes = one row): it stages the keys and permutation entries of its range
// (+ look-ahead) in LDS with one round trip, then every lane group walks from the first run that STARTS in its own
// positions to the end of the last such run, TS_AHEAD rows per round trip whatever the run lengths (a table of 10^6 rows
// has runs of one, the AIFB step runs of eight), adding in sorted order and storing at every run's end: the same additions
// in the same order as table_sum_block.
// (Smaller ranges -- 2 / 3 / 4 positions per lane group, 1 376 / 918 / 688 workgroups -- measured in the reduction launch of
// the AIFB step: 13.8 / 13.0 / 11.5 us against 10.3 with the one-run workgroups of table_sum_block: the staging round trip and
// its barrier cost more than the second round of the chip they save.)
#ifndef TSM_OWN
#define TSM_OWN 8
#endif
#ifndef TSM_LOOK
#define TSM_LOOK 64
#endif
#define TSM_LDS_WORDS(D) (3 * ((256 / ((D) / 4)) * TSM_OWN + 1 + TSM_LOOK))
template <class TabsT>
__device__ __forceinline__ void table_sum_multi(long long M, int row_bits, const tkey_t *__restrict__ keys,
                                                const int *__restrict__ perm, const float *__restrict__ DG, int D,
                                                const TabsT &tabs, int store, long long wg, const int *__restrict__ failed,
                                                unsigned *lds) {
    const int lpr = D / 4, ngrp = 256 / lpr, POS = ngrp * TSM_OWN, WIN = POS + 1 + TSM_LOOK;
    const long long base = wg * POS;
    if (base >= M) return;
    tkey_t *lk = reinterpret_cast<tkey_t *>(lds);            // keys of positions base - 1 + q
    int *lp = reinterpret_cast<int *>(lds + 2 * WIN);        // entries (rows of DG) of the same positions
    for (int q = threadIdx.x; q < WIN; q += 256) {
        const long long pos = base - 1 + q;
        const bool in = pos >= 0 && pos < M;
        lk[q] = in ? keys[pos] : TOUCH_INVALID;
        lp[q] = (int)(in ? (perm ? (long long)perm[pos] : pos) : 0);
    }
    const int bad = failed ? *failed : 0;
    __syncthreads();
    if (bad) return;
    auto K = [&](long long pos) -> tkey_t {                  // (beyond the window: a run longer than the look-ahead)
        if (pos < 0 || pos >= M) return TOUCH_INVALID;
        const long long q = pos - (base - 1);
        return q < WIN ? lk[q] : keys[pos];
    };
    auto E = [&](long long pos) -> long long {
        const long long q = pos - (base - 1);
        return q < WIN ? (long long)lp[q] : (perm ? (long long)perm[pos] : pos);
    };
    const int g = threadIdx.x / lpr, c = (threadIdx.x % lpr) * 4;
    const long long own0 = base + (long long)g * TSM_OWN;
    const long long own1 = own0 + TSM_OWN < M ? own0 + TSM_OWN : M;
    long long p = own0;
    while (p < own1 && (K(p) == TOUCH_INVALID || K(p) == K(p - 1))) ++p;       // first run that starts in my positions
    if (p >= own1) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    tkey_t cur = K(p);
    bool have = false, done = false;
    auto flush = [&]() {
        const int tab = (int)(cur >> row_bits);
        const long long row = (long long)(cur & ((1ull << row_bits) - 1ull));
        float *gr = tabs.grad[0];      // (a runtime index into the by-value table would spill it to scratch)
#pragma unroll
        for (int m = 1; m < MPQE_STEP_MAX_MODES; ++m)
            if (m == tab) gr = tabs.grad[m];
        if (!gr) return;
        f32x4 *dst = reinterpret_cast<f32x4 *>(gr + row * D + c);
        if (store) *dst = acc;
        else *dst = *dst + acc;
    };
    for (long long j0 = p; !done; j0 += TS_AHEAD) {
        f32x4 v[TS_AHEAD];
#pragma unroll
        for (int q = 0; q < TS_AHEAD; ++q) v[q] = gload4(DG + E(j0 + q < M ? j0 + q : M - 1) * D + c);
#pragma unroll
        for (int q = 0; q < TS_AHEAD; ++q) {
            if (done) continue;
            const long long j = j0 + q;
            const tkey_t kj = K(j);
            if (have && kj != cur) {        // the run has ended: its row is complete
                flush();
                have = false;
            }
            if (!have) {
                if (j >= own1 || kj == TOUCH_INVALID) {       // (position j starts a run of the next lane group / the invalid tail)
                    done = true;
                    continue;
                }
                cur = kj;
                acc = v[q];
                have = true;
            } else {
                acc += v[q];
            }
        }
    }
}

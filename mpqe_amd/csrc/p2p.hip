// Gradient-bucket exchange of the data-parallel step over peer-mapped buffers: ONE-HOP reduce-scatter + all-gather.
//
// The reference has no distributed code; the data-parallel design shards query graphs by rank and sums the
// gradient bucket once per step. On MI355X every GPU has a direct xGMI link to each of its
// seven peers (~153 GB/s each); a ring all-reduce moves 2 (w - 1) / w of the bucket through ONE link per step of the
// ring -- 14 serial hops at w = 8 -- where the links can all carry a shard at once. So, per rank r and step:
//
//   push     my contribution to shard p of the bucket -> rank p's staging slot [r]   (w - 1 peer writes, all links at once)
//   reduce   when all w slots of MY shard have landed: sum them in RANK ORDER (one fixed order, computed once: every
//            replica receives the same bits) and write the sum into shard r of every rank's bucket
//   wait     until all w shards of my bucket have landed
//
// = 2 (n / w) floats over each link, one hop each way. Buffers are fine-grained device allocations (coherent between
// agents while kernels run; hipExtMallocWithFlags) that every rank exports with hipIpcGetMemHandle and maps from its
// peers; hand-offs are epoch-stamped flag words written after a system-scope release and polled with system-scope loads,
// every poll BOUNDED (~25 s): a peer that never arrives sets MPQE_FLAG_INTERNAL | 0x4000 in the caller's error word
// instead of hanging -- the exchange of THAT step is then incomplete (this rank skipped its reduce, its peers will time out
// on its shard too) and the gradients must not be used: the host side (mpqe_amd/parallel.py: StepExchange.check, a
// collective every rank calls before its optimiser step) reads the word, agrees with the other ranks, switches to the RCCL
// all-reduce for good and raises. EXPERIMENTAL: functionally tested with two processes on ONE GPU (IPC handles work
// between processes of one device); what it is FOR -- eight ranks, seven xGMI links, remote writes against the local L2 --
// has not run on hardware (no multi-GPU box in this build's reach).
#include <string.h>

#include <algorithm>

#include "common.h"

#define P2P_MAX_WORLD 16
#define P2P_FLAG_STRIDE 16                 // words between two flags (64 bytes: a line of their own)
#define P2P_SPIN_LIMIT (1 << 22)

struct P2PBufs {
    float *bucket[P2P_MAX_WORLD];          // rank p's bucket [n]
    float *stage[P2P_MAX_WORLD];           // rank p's staging slots [world][shard]
    unsigned *flags[P2P_MAX_WORLD];        // rank p's flags: [0, world): slot filled; [world, 2 world): shard landed; then a counter
};

__device__ __forceinline__ void p2p_flag_store(unsigned *p, unsigned v) {
#ifdef MPQE_EMU
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
__device__ __forceinline__ unsigned p2p_flag_load(const unsigned *p) {
#ifdef MPQE_EMU
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
// every workgroup: drain my stores, release at system scope, count myself in; the LAST one of the launch returns true
__device__ __forceinline__ bool p2p_last_block(unsigned *counter, unsigned nblocks, unsigned *lds_word) {
#ifndef MPQE_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
#ifndef MPQE_EMU
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned before = atomicAdd(counter, 1u);
        *lds_word = before + 1u == nblocks ? 1u : 0u;
        if (before + 1u == nblocks) *counter = 0u;           // (the launch's last arrival: re-armed for the next launch)
    }
    __syncthreads();
    return *lds_word != 0u;
}
__device__ __forceinline__ bool p2p_wait(const unsigned *flag, unsigned epoch, int32_t *err) {
    for (int spins = 0; p2p_flag_load(flag) != epoch; ++spins) {
        if (spins >= P2P_SPIN_LIMIT) {
            flag_error(err, MPQE_FLAG_INTERNAL | 0x4000);
            return false;
        }
#ifndef MPQE_EMU
        // (the flag is a word of this rank's OWN buffer; quick polls while the peer is about to arrive, then ~4 us naps:
        // the bound is ~25 s of real time -- ranks that drift apart by a module load, a checkpoint or an evaluation pass
        // on rank 0 still meet; a peer that never comes is reported, not waited for forever)
        if (spins < 4096) __builtin_amdgcn_s_sleep(16);
        else __builtin_amdgcn_s_sleep(127);
#endif
    }
    return true;
}

// push: bucket[shard p] of rank `rank` -> rank p's stage[rank]; the launch's last workgroup raises flag [rank] at every peer
__global__ __launch_bounds__(256) void p2p_push_kernel(P2PBufs B, int rank, int world, long long n, long long shard,
                                                        unsigned epoch) {
    __shared__ unsigned last;
    const long long per = (long long)gridDim.x * 256 * 4;
    for (int p = 0; p < world; ++p) {
        const long long lo = (long long)p * shard, hi = lo + shard < n ? lo + shard : n;
        const float *src = B.bucket[rank];
        float *dst = B.stage[p] + (long long)rank * shard;
        for (long long i = lo + ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < hi; i += per) {
            if (i + 3 < hi) *reinterpret_cast<f32x4 *>(dst + (i - lo)) = *reinterpret_cast<const f32x4 *>(src + i);
            else
                for (long long q = i; q < hi; ++q) dst[q - lo] = src[q];
        }
    }
    unsigned *counter = B.flags[rank] + 2 * world * P2P_FLAG_STRIDE;
    if (p2p_last_block(counter, gridDim.x, &last) && threadIdx.x < world)
        p2p_flag_store(B.flags[threadIdx.x] + (long long)rank * P2P_FLAG_STRIDE, epoch);
}

// reduce: my shard = sum over ranks (in rank order) of my staging slots, written into shard `rank` of EVERY rank's bucket;
// the launch's last workgroup raises flag [world + rank] at every peer
__global__ __launch_bounds__(256) void p2p_reduce_kernel(P2PBufs B, int rank, int world, long long n, long long shard,
                                                          unsigned epoch, int32_t *err) {
    __shared__ unsigned last, ok;
    if (threadIdx.x == 0) ok = 1u;
    __syncthreads();
    // (one look, no waiting: the one-workgroup launch in front of this one -- p2p_wait_kernel on the slot flags -- has waited
    // for the peers and reported the one that never came; up to 256 workgroups napping on a flag would hold CU slots for as
    // long as a peer is late, and on a GPU shared with that peer's process they are the slots its step needs)
    if (threadIdx.x < world && p2p_flag_load(B.flags[rank] + (long long)threadIdx.x * P2P_FLAG_STRIDE) != epoch) ok = 0u;
    __syncthreads();
#ifndef MPQE_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
#endif
    const long long lo = (long long)rank * shard, hi = lo + shard < n ? lo + shard : n;
    const long long per = (long long)gridDim.x * 256 * 4;
    const float *st = B.stage[rank];
    if (ok)
        for (long long i = lo + ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < hi; i += per) {
            if (i + 3 < hi) {
                f32x4 s = *reinterpret_cast<const f32x4 *>(st + (i - lo));
                for (int p = 1; p < world; ++p) s += *reinterpret_cast<const f32x4 *>(st + (long long)p * shard + (i - lo));
                for (int p = 0; p < world; ++p) *reinterpret_cast<f32x4 *>(B.bucket[p] + i) = s;
            } else {
                for (long long q = i; q < hi; ++q) {
                    float s = st[q - lo];
                    for (int p = 1; p < world; ++p) s += st[(long long)p * shard + (q - lo)];
                    for (int p = 0; p < world; ++p) B.bucket[p][q] = s;
                }
            }
        }
    unsigned *counter = B.flags[rank] + (2 * world + 1) * P2P_FLAG_STRIDE;
    if (p2p_last_block(counter, gridDim.x, &last) && threadIdx.x < world && ok)
        p2p_flag_store(B.flags[threadIdx.x] + (long long)(world + rank) * P2P_FLAG_STRIDE, epoch);
}

// wait (one workgroup): flags [base, base + world) of my buffer carry this exchange's number -- base 0: every peer's slot of
// my shard is filled (in front of the reduce launch); base world: every shard of my bucket has landed
__global__ __launch_bounds__(64) void p2p_wait_kernel(P2PBufs B, int rank, int world, unsigned epoch, int32_t *err, int base) {
    if ((int)threadIdx.x < world) p2p_wait(B.flags[rank] + (long long)(base + threadIdx.x) * P2P_FLAG_STRIDE, epoch, err);
#ifndef MPQE_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
#endif
}

#ifdef MPQE_EMU
extern "C" size_t mpqe_p2p_handle_bytes(void) { return 64; }
#else
extern "C" size_t mpqe_p2p_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }
#endif

// bytes of one rank's communication buffer for a bucket of n floats: [bucket | staging | flags]
static long long p2p_shard(long long n, int world) { return ((n + world - 1) / world + 3) / 4 * 4; }
extern "C" size_t mpqe_p2p_buffer_bytes(int64_t n, int world, int64_t *stage_offset, int64_t *flags_offset) {
    if (n <= 0 || world < 1 || world > P2P_MAX_WORLD) return 0;
    const size_t b = align_up((size_t)n * 4, 256), s = align_up((size_t)world * p2p_shard(n, world) * 4, 256);
    if (stage_offset) *stage_offset = (int64_t)b;
    if (flags_offset) *flags_offset = (int64_t)(b + s);
    return b + s + align_up((size_t)(2 * world + 2) * P2P_FLAG_STRIDE * 4, 256);
}

extern "C" int mpqe_p2p_alloc(size_t bytes, void **ptr, void *handle_out) {
    if (!ptr || !handle_out || bytes == 0) return MPQE_ERR_INVALID_ARG;
#ifdef MPQE_EMU
    return MPQE_ERR_UNSUPPORTED;
#else
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) return MPQE_ERR_LAUNCH;
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return MPQE_ERR_LAUNCH;
    }
    if (hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(handle_out), p) != hipSuccess) {
        (void)hipFree(p);
        return MPQE_ERR_LAUNCH;
    }
    *ptr = p;
    return MPQE_OK;
#endif
}
extern "C" int mpqe_p2p_free(void *ptr) {
#ifdef MPQE_EMU
    return MPQE_ERR_UNSUPPORTED;
#else
    return ptr && hipFree(ptr) == hipSuccess ? MPQE_OK : MPQE_ERR_INVALID_ARG;
#endif
}
extern "C" int mpqe_p2p_open(const void *handle, void **mapped) {
    if (!handle || !mapped) return MPQE_ERR_INVALID_ARG;
#ifdef MPQE_EMU
    return MPQE_ERR_UNSUPPORTED;
#else
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    return hipIpcOpenMemHandle(mapped, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? MPQE_OK : MPQE_ERR_LAUNCH;
#endif
}
extern "C" int mpqe_p2p_close(void *mapped) {
#ifdef MPQE_EMU
    return MPQE_ERR_UNSUPPORTED;
#else
    return mapped && hipIpcCloseMemHandle(mapped) == hipSuccess ? MPQE_OK : MPQE_ERR_INVALID_ARG;
#endif
}

extern "C" int mpqe_p2p_allreduce(void *const *buffers, int rank, int world, int64_t capacity, int64_t n, uint32_t epoch,
                                  int phases, int32_t *err, void *stream) {
    if (!buffers || world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world || n <= 0 || n > capacity || epoch == 0)
        return MPQE_ERR_INVALID_ARG;
    int64_t so = 0, fo = 0;
    if (!mpqe_p2p_buffer_bytes(capacity, world, &so, &fo)) return MPQE_ERR_INVALID_ARG;
    P2PBufs B;
    memset(&B, 0, sizeof(B));
    for (int p = 0; p < world; ++p) {
        if (!buffers[p] || (uintptr_t)buffers[p] % 256 != 0) return MPQE_ERR_INVALID_ARG;
        char *b = reinterpret_cast<char *>(buffers[p]);
        B.bucket[p] = reinterpret_cast<float *>(b);
        B.stage[p] = reinterpret_cast<float *>(b + so);
        B.flags[p] = reinterpret_cast<unsigned *>(b + fo);
    }
    const long long shard = p2p_shard(n, world);
    hipStream_t s = as_stream(stream);
    const unsigned g1 = (unsigned)std::min<long long>(256, (n / 4 + 255) / 256 + 1);
    const unsigned g2 = (unsigned)std::min<long long>(256, (shard / 4 + 255) / 256 + 1);
    if (phases & 1) hipLaunchKernelGGL(p2p_push_kernel, dim3(g1), dim3(256), 0, s, B, rank, world, (long long)n, shard, epoch);
    if (phases & 2) {
        hipLaunchKernelGGL(p2p_wait_kernel, dim3(1), dim3(64), 0, s, B, rank, world, epoch, err, 0);
        hipLaunchKernelGGL(p2p_reduce_kernel, dim3(g2), dim3(256), 0, s, B, rank, world, (long long)n, shard, epoch, err);
    }
    if (phases & 4) hipLaunchKernelGGL(p2p_wait_kernel, dim3(1), dim3(64), 0, s, B, rank, world, epoch, err, world);
    return mpqe_launch_status();
}

// ---- glue of the per-step gradient exchange (mpqe_amd/parallel.py: StepExchange.reduce) as the library's own launches: what
// was torch._foreach_copy_ / index_select / where / unique there.
// spans: dst[do .. do + n) <- src[so .. so + n) for every span of a device table {dst offset, src offset, floats} (the
// touched relation matrices + root / bias / mode rows into the contiguous bucket, and back). One workgroup walks 4 096
// floats of a span; table[nspans] holds the first workgroup of every span (+ the total).
struct SpanRec {
    long long dst, src, n, first_block;
};
__global__ __launch_bounds__(256) void spans_copy_kernel(float *__restrict__ dst, const float *__restrict__ src,
                                                         const SpanRec *__restrict__ spans, int nspans) {
    int lo = 0, hi = nspans - 1;             // the span of this workgroup: the last one whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (spans[mid].first_block <= (long long)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const SpanRec sp = spans[lo];
    const long long base = ((long long)blockIdx.x - sp.first_block) * 4096;
    float *d = dst + sp.dst;
    const float *s = src + sp.src;
    const bool vec = (((uintptr_t)d | (uintptr_t)s) & 15) == 0;
    for (int k = 0; k < 4; ++k) {
        const long long i = base + (long long)(threadIdx.x + 256 * k) * 4;
        if (i >= sp.n) continue;
        if (vec && i + 3 < sp.n) *reinterpret_cast<f32x4 *>(d + i) = *reinterpret_cast<const f32x4 *>(s + i);
        else
            for (long long q = i; q < i + 4 && q < sp.n; ++q) d[q] = s[q];
    }
}
extern "C" int mpqe_spans_copy(float *dst, const float *src, const void *spans_device, int nspans, int64_t total_blocks,
                               void *stream) {
    if (!dst || !src || !spans_device || nspans <= 0 || total_blocks <= 0 || total_blocks >= (1ll << 31)) return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(spans_copy_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), dst, src,
                       reinterpret_cast<const SpanRec *>(spans_device), nspans);
    return mpqe_launch_status();
}

// The row exchange of a step that built its own touch plan: from the plan's SORTED keys [M] (key = table << row_bits | row,
// ~0 = invalid) the first key of every run of equal keys goes out, every other slot invalid (fixed size: no count to
// exchange), with the row of the flat [rows, D] table-gradient view it names (row_base[table] + row; 0 for invalid slots).
struct RowsPrepArgs {
    long long table_rows[MPQE_STEP_MAX_MODES], row_base[MPQE_STEP_MAX_MODES];
    int num_tables, row_bits;
};
__global__ __launch_bounds__(256) void rows_prepare_kernel(const unsigned long long *__restrict__ keys, long long M, long long cap,
                                                           RowsPrepArgs a, unsigned long long *__restrict__ send_keys,
                                                           long long *__restrict__ gidx) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    unsigned long long out = ~0ull;
    long long g = 0;
    if (i < M) {
        const unsigned long long k = keys[i];
        const bool first = i == 0 || keys[i - 1] != k;
        const long long tab = (long long)(k >> a.row_bits), row = (long long)(k & ((1ull << a.row_bits) - 1ull));
        if (first && k != ~0ull && tab < a.num_tables) {
            long long rows = a.table_rows[0], rb = a.row_base[0];
#pragma unroll
            for (int m = 1; m < MPQE_STEP_MAX_MODES; ++m)
                if (m == tab) {
                    rows = a.table_rows[m];
                    rb = a.row_base[m];
                }
            if (row < rows) {
                out = k;
                g = rb + row;
            }
        }
    }
    send_keys[i] = out;
    gidx[i] = g;
}
extern "C" int mpqe_rows_prepare(const uint64_t *sorted_keys, int64_t M, int64_t cap, int row_bits, const int64_t *table_rows,
                                 const int64_t *row_base, int num_tables, uint64_t *send_keys, int64_t *gidx, void *stream) {
    if (!sorted_keys || !send_keys || !gidx || !table_rows || !row_base || M < 0 || cap < M || cap <= 0 || row_bits <= 0 ||
        row_bits > 40 || num_tables <= 0 || num_tables > MPQE_STEP_MAX_MODES)
        return MPQE_ERR_INVALID_ARG;
    RowsPrepArgs a;
    memset(&a, 0, sizeof(a));
    a.num_tables = num_tables;
    a.row_bits = row_bits;
    for (int m = 0; m < num_tables; ++m) {
        a.table_rows[m] = table_rows[m];
        a.row_base[m] = row_base[m];
    }
    hipLaunchKernelGGL(rows_prepare_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const unsigned long long *>(sorted_keys), (long long)M, (long long)cap, a,
                       reinterpret_cast<unsigned long long *>(send_keys), reinterpret_cast<long long *>(gidx));
    return mpqe_launch_status();
}
// out[i][:] = rows[gidx[i]][:]   (D % 4 == 0, 16-byte aligned rows): the gradient rows a rank sends, in its key order
__global__ __launch_bounds__(256) void rows_gather_kernel(const float *__restrict__ rows, const long long *__restrict__ gidx,
                                                          long long n, int D, float *__restrict__ out) {
    const int q = D / 4;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * q) return;
    const long long i = t / q;
    const int c = (int)(t % q) * 4;
    *reinterpret_cast<f32x4 *>(out + i * D + c) = *reinterpret_cast<const f32x4 *>(rows + gidx[i] * D + c);
}
extern "C" int mpqe_rows_gather(const float *rows, const int64_t *gidx, int64_t n, int64_t dim, float *out, void *stream) {
    if (!rows || !gidx || !out || n <= 0 || dim <= 0 || dim % 4 != 0 || (((uintptr_t)rows | (uintptr_t)out) & 15) != 0)
        return MPQE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)((n * (dim / 4) + 255) / 256)), dim3(256), 0, as_stream(stream), rows,
                       reinterpret_cast<const long long *>(gidx), (long long)n, (int)dim, out);
    return mpqe_launch_status();
}

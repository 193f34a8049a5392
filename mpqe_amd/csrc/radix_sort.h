// Stable LSD radix sort of (key, value) pairs as a chain of plain launches: the library's own sort wherever a sort may not
// depend on how many workgroups the device holds at once (step_touch.h's one-launch sort does) -- the pack-time / recovery
// touch plan, the row-exchange plan, the plan of the general-graph path. Integer preprocessing, cached
// per graph or per packed step; it replaces rocPRIM's radix_sort_pairs, which was the library's only third-party device
// code. 8 bits per pass, three launches per pass:
//   hist     per workgroup of RS_TILE items: its count of every digit                    -> hist[digit][workgroup]
//   scan     exclusive prefix over hist in (digit, workgroup) order (one workgroup)       -> first destination of every
//                                                                                           (digit, workgroup) pair
//   scatter  every item again: destination = that prefix + its rank among the workgroup's items with the same digit, in
//            item order (rounds of 256 items; inside a wave by eight ballots, across waves / rounds through LDS): stable
// Passes alternate between the caller's output arrays and a scratch pair so that the last pass lands in the output.
#pragma once
#include "common.h"

#define RS_THREADS 256
#define RS_ROUNDS 8
#define RS_TILE (RS_THREADS * RS_ROUNDS)

static inline long long rs_blocks(long long n) { return (n + RS_TILE - 1) / RS_TILE; }
template <class K>
static inline size_t radix_sort_tmp_bytes(long long n) {
    const size_t m = (size_t)(n > 0 ? n : 1);
    return align_up(m * sizeof(K), 256) + align_up(m * sizeof(int), 256) + align_up((size_t)rs_blocks((long long)m) * 256 * sizeof(unsigned), 256);
}

template <class K>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const K *__restrict__ keys, long long n, int shift,
                                                             unsigned *__restrict__ hist, int nblk) {
    __shared__ unsigned cnt[256];
    cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const long long i = (long long)blockIdx.x * RS_TILE + r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(unsigned)((unsigned long long)keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(long long)threadIdx.x * nblk + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive prefix over v[0 .. total) in place, one workgroup of 1024 threads: a contiguous piece per thread
static __global__ __launch_bounds__(1024) void rs_scan_kernel(unsigned *__restrict__ v, long long total) {
    __shared__ unsigned part[1024];
    const long long per = (total + 1023) / 1024, lo = (long long)threadIdx.x * per, hi = lo + per < total ? lo + per : total;
    unsigned s = 0u;
    for (long long i = lo; i < hi; ++i) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0u;
        for (int t = 0; t < 1024; ++t) {
            const unsigned c = part[t];
            part[t] = run;
            run += c;
        }
    }
    __syncthreads();
    unsigned run = part[threadIdx.x];
    for (long long i = lo; i < hi; ++i) {
        const unsigned c = v[i];
        v[i] = run;
        run += c;
    }
}

template <class K>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const K *__restrict__ kin, const int *__restrict__ vin,
                                                                K *__restrict__ kout, int *__restrict__ vout, long long n,
                                                                int shift, const unsigned *__restrict__ first, int nblk) {
    // whist[round * 4 + wave][digit]: items of that (round, wave) with that digit, then their exclusive prefix in (round, wave) order
    __shared__ unsigned short whist[RS_ROUNDS * 4][256];
#ifdef MPQE_EMU
    __shared__ unsigned char dig[RS_THREADS];
#endif
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int q = t; q < RS_ROUNDS * 4 * 256 / 2; q += RS_THREADS) reinterpret_cast<unsigned *>(whist)[q] = 0u;
    __syncthreads();
    K k[RS_ROUNDS];
    unsigned below[RS_ROUNDS];
    bool on[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const long long i = (long long)blockIdx.x * RS_TILE + r * RS_THREADS + t;
        on[r] = i < n;
        k[r] = on[r] ? kin[i] : (K)0;
        const unsigned d = on[r] ? (unsigned)((unsigned long long)k[r] >> shift) & 255u : 256u;      // (256: matches nobody)
#ifndef MPQE_EMU
        unsigned long long peers = __ballot(on[r]);          // lanes of my wave whose item of this round has my digit
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long m = __ballot((d >> bit) & 1u);
            peers &= ((d >> bit) & 1u) ? m : ~m;
        }
        below[r] = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        if (on[r] && below[r] == 0) whist[r * 4 + wave][d] = (unsigned short)__popcll(peers);
#else
        dig[t] = (unsigned char)(d & 255u);
        __syncthreads();
        unsigned b = 0, c = 0;
        for (int j = wave * 64; j < wave * 64 + 64; ++j) {
            const long long ij = (long long)blockIdx.x * RS_TILE + r * RS_THREADS + j;
            if (ij < n && dig[j] == (unsigned char)(d & 255u)) {
                c += 1;
                b += j < t ? 1 : 0;
            }
        }
        below[r] = b;
        if (on[r] && b == 0) whist[r * 4 + wave][d] = (unsigned short)c;
        __syncthreads();
#endif
    }
    __syncthreads();
    {   // exclusive prefix over the (round, wave) groups, per digit t
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < RS_ROUNDS * 4; ++w) {
            const unsigned c = whist[w][t];
            whist[w][t] = (unsigned short)run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        if (!on[r]) continue;
        const long long i = (long long)blockIdx.x * RS_TILE + r * RS_THREADS + t;
        const unsigned d = (unsigned)((unsigned long long)k[r] >> shift) & 255u;
        const long long at = (long long)first[(long long)d * nblk + blockIdx.x] + whist[r * 4 + wave][d] + below[r];
        kout[at] = k[r];
        vout[at] = vin ? vin[i] : (int)i;
    }
}

// keys_out / vals_out <- (keys_in, vals_in) stably sorted by bits [0, bits) of the key. vals_in == NULL: the item numbers
// 0 .. n - 1. tmp: radix_sort_tmp_bytes<K>(n), 256-byte aligned. keys_in may be modified (it serves no purpose afterwards);
// in / out / tmp must not overlap.
template <class K>
static inline int radix_sort_pairs_own(void *tmp, const K *keys_in, K *keys_out, const int *vals_in, int *vals_out, long long n,
                                       int bits, hipStream_t s) {
    if (n <= 0) return MPQE_OK;
    if (n >= (1ll << 31)) return MPQE_ERR_UNSUPPORTED;
    const int passes = bits <= 0 ? 1 : (bits + 7) / 8;
    const int nblk = (int)rs_blocks(n);
    char *tb = reinterpret_cast<char *>(tmp);
    K *tk = reinterpret_cast<K *>(tb);
    int *tv = reinterpret_cast<int *>(tb + align_up((size_t)n * sizeof(K), 256));
    unsigned *hist = reinterpret_cast<unsigned *>(tb + align_up((size_t)n * sizeof(K), 256) + align_up((size_t)n * sizeof(int), 256));
    const K *ki = keys_in;
    const int *vi = vals_in;
    for (int p = 0; p < passes; ++p) {
        // the last pass writes the output; the ones before alternate so that it can
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        K *ko = to_out ? keys_out : tk;
        int *vo = to_out ? vals_out : tv;
        hipLaunchKernelGGL(rs_hist_kernel<K>, dim3((unsigned)nblk), dim3(RS_THREADS), 0, s, ki, n, 8 * p, hist, nblk);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, s, hist, (long long)256 * nblk);
        hipLaunchKernelGGL(rs_scatter_kernel<K>, dim3((unsigned)nblk), dim3(RS_THREADS), 0, s, ki, vi, ko, vo, n, 8 * p,
                           (const unsigned *)hist, nblk);
        ki = ko;
        vi = vo;
    }
    return mpqe_launch_status();
}

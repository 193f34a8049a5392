// TEST INFRASTRUCTURE ONLY -- host stand-in for the one rocPRIM entry point the plan
// builder uses (stable LSD radix sort of key/value pairs), for the CPU emulation build.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {
template <class K, class V, class Size>
hipError_t radix_sort_pairs(void *tmp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, Size n,
                            unsigned begin_bit = 0, unsigned end_bit = 8 * sizeof(K), hipStream_t = nullptr,
                            bool = false) {
    if (!tmp) {
        bytes = 64;
        return hipSuccess;
    }
    const unsigned long long mask =
        (end_bit - begin_bit >= 64) ? ~0ull : (((1ull << (end_bit - begin_bit)) - 1ull) << begin_bit);
    std::vector<size_t> order((size_t)n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        return ((unsigned long long)kin[a] & mask) < ((unsigned long long)kin[b] & mask);
    });
    std::vector<K> ks((size_t)n);
    std::vector<V> vs((size_t)n);
    for (size_t i = 0; i < (size_t)n; ++i) {
        ks[i] = kin[order[i]];
        vs[i] = vin[order[i]];
    }
    std::copy(ks.begin(), ks.end(), kout);
    std::copy(vs.begin(), vs.end(), vout);
    return hipSuccess;
}
}  // namespace rocprim

// Diagnostics switches of the library (timing experiments, tests that force a rarely taken code path). The data path
// never reads the environment: a switch exists only after mpqe_debug_option(name, value) has set it, and the common
// case -- none set -- costs one relaxed atomic load per query. Process-global and not part of the data path: set them
// before the calls they should affect, from one thread (include/mpqe_amd.h: threading contract).
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>

#include "common.h"

namespace {
std::atomic<int> g_any(0);
std::atomic<int> g_gen(0);
std::mutex g_mu;
std::unordered_map<std::string, int> g_opts;
}  // namespace

int mpqe_dbg_value(const char *name, int unset) {
    if (g_any.load(std::memory_order_relaxed) == 0) return unset;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_opts.find(name);
    return it == g_opts.end() ? unset : it->second;
}

// changes with every mpqe_debug_option call: cached launch plans that a switch may have shaped are keyed by it
int mpqe_dbg_generation() { return g_gen.load(std::memory_order_relaxed); }

extern "C" void mpqe_debug_option(const char *name, int value, int set) {
    if (!name) return;
    std::lock_guard<std::mutex> lock(g_mu);
    g_gen.fetch_add(1, std::memory_order_relaxed);
    if (set) g_opts[name] = value;
    else g_opts.erase(name);
    g_any.store(g_opts.empty() ? 0 : 1, std::memory_order_relaxed);
}

// 1: this build holds the launch forms that were measured slower and taken out of the shipped library (common.h:
// MPQE_EXPERIMENTS) -- their mpqe_debug_option switches work; 0: the switches are accepted and do nothing
extern "C" int mpqe_debug_has_experiments(void) { return MPQE_HAS_EXPERIMENTS; }

// host (pinned) -> device copy on `stream`: hipMemcpyAsync behind the C ABI, so that a host mirror without a HIP binding
// of its own can send the ids of the next step with one asynchronous copy (mpqe_amd/fused.py: pack)
extern "C" int mpqe_copy_to_device(void *dst, const void *src_host, size_t bytes, void *stream) {
    if (!dst || !src_host) return MPQE_ERR_INVALID_ARG;
    return hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)) == hipSuccess ? MPQE_OK : MPQE_ERR_LAUNCH;
}

// random.choice's draws replayed over raw Mersenne-Twister outputs (include/mpqe_amd.h: mpqe_host_random_choice; reference
// model.py:470-476). CPython: choice(seq) = seq[_randbelow(len(seq))]; _randbelow(n): k = n.bit_length();
// r = getrandbits(k) until r < n; getrandbits(k <= 32) = one 32-bit output >> (32 - k). Host only.
extern "C" int mpqe_host_random_choice(const uint32_t *words, int64_t nwords, const int64_t *lens, int64_t len_all,
                                       const int64_t *base, const int64_t *cand, int64_t nq, int64_t *cursor, int64_t *out) {
    if (!words || nwords < 0 || nq < 0 || !cursor || !out || cursor[0] < 0) return MPQE_ERR_INVALID_ARG;
    int64_t q = cursor[0], w = 0;
    int64_t n = 0;
    int shift = 0;
    bool fresh = true;
    while (q < nq && w < nwords) {
        if (fresh) {
            n = lens ? lens[q] : len_all;
            if (n <= 0 || n > 0xffffffffLL) {
                cursor[1] = w;
                return MPQE_ERR_INVALID_ARG;
            }
            shift = 32 - (64 - __builtin_clzll((unsigned long long)n));     // 32 - n.bit_length()
            fresh = false;
        }
        const int64_t r = (int64_t)(words[w++] >> shift);
        if (r < n) {
            out[q] = cand ? cand[(base ? base[q] : 0) + r] : r;
            ++q;
            fresh = true;
        }
    }
    cursor[0] = q;
    cursor[1] = w;
    return MPQE_OK;
}

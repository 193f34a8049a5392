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

// host (pinned) -> device copy on `stream`: hipMemcpyAsync behind the C ABI, so that a host mirror without a HIP binding
// of its own can send the ids of the next step with one asynchronous copy (mpqe_amd/fused.py: pack)
extern "C" int mpqe_copy_to_device(void *dst, const void *src_host, size_t bytes, void *stream) {
    if (!dst || !src_host) return MPQE_ERR_INVALID_ARG;
    return hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)) == hipSuccess ? MPQE_OK : MPQE_ERR_LAUNCH;
}

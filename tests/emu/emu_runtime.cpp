// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <vector>

// Context switch. glibc's swapcontext saves and restores the signal mask with a system call per switch, and a kernel with
// MFMAs switches millions of times (a third of the CPU suite's time was spent in rt_sigprocmask): on x86-64 the fibers
// switch with a dozen instructions instead -- callee-saved registers and the stack pointer, nothing else is live across a
// call. -DEMU_UCONTEXT keeps the portable form (other hosts, sanitizer builds).
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#endif
#endif
#if defined(__x86_64__) && !defined(EMU_UCONTEXT)
#define EMU_FAST_SWITCH 1
extern "C" void emu_switch(void **save_sp, void *const *load_sp);
asm(".text\n"
    ".globl emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "    pushq %rbp\n"
    "    pushq %rbx\n"
    "    pushq %r12\n"
    "    pushq %r13\n"
    "    pushq %r14\n"
    "    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq (%rsi), %rsp\n"
    "    popq %r15\n"
    "    popq %r14\n"
    "    popq %r13\n"
    "    popq %r12\n"
    "    popq %rbx\n"
    "    popq %rbp\n"
    "    ret\n"
    ".size emu_switch, .-emu_switch\n");
#endif

emu_uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace emu {
namespace {
struct Fiber {
#ifdef EMU_FAST_SWITCH
    void *sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char *stack = nullptr;
    unsigned tid = 0;
    bool done = false;
};
constexpr size_t kStack = 256 * 1024;
#ifdef EMU_FAST_SWITCH
void *g_main_sp = nullptr;
#else
ucontext_t g_main;
#endif
std::vector<Fiber> g_fibers;
Fiber *g_cur = nullptr;
const std::function<void()> *g_body = nullptr;
int g_alive = 0, g_block_gen = 0, g_block_arrived = 0;
long g_events = 0;
constexpr int kMaxWaves = 16;
int g_wave_alive[kMaxWaves], g_wave_gen[kMaxWaves], g_wave_arrived[kMaxWaves];
float g_xa[kMaxWaves][64], g_xb[kMaxWaves][64];
int g_wave_order = -1;                  // MPQE_EMU_WAVE_ORDER (read once)
unsigned long long g_wave_rng = 0;

void set_tid(unsigned tid) {
    threadIdx.x = tid % blockDim.x;
    threadIdx.y = (tid / blockDim.x) % blockDim.y;
    threadIdx.z = tid / (blockDim.x * blockDim.y);
}
#ifdef EMU_FAST_SWITCH
#ifdef EMU_ASAN
// AddressSanitizer is told about every stack switch (it tracks the bounds of the running stack): start_switch in front
// of it, finish_switch as the first thing on the other side. A fiber that is done passes NULL (its fake stack is freed).
const void *g_main_bottom = nullptr;
size_t g_main_size = 0;
void *g_main_fake = nullptr;
void yield() {
    Fiber *f = g_cur;
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(f->done ? nullptr : &fake, g_main_bottom, g_main_size);
    emu_switch(&f->sp, &g_main_sp);
    __sanitizer_finish_switch_fiber(fake, &g_main_bottom, &g_main_size);
}
#else
void yield() { emu_switch(&g_cur->sp, &g_main_sp); }
#endif
#else
void yield() { swapcontext(&g_cur->ctx, &g_main); }
#endif
void trampoline() {
#if defined(EMU_FAST_SWITCH) && defined(EMU_ASAN)
    __sanitizer_finish_switch_fiber(nullptr, &g_main_bottom, &g_main_size);      // (first entry of this fiber)
#endif
    (*g_body)();
    g_cur->done = true;
    yield();
    abort();          // (a finished fiber is never resumed)
}
#ifdef EMU_FAST_SWITCH
// a fresh fiber: emu_switch pops six registers, then returns into trampoline with the stack as a call would leave it
// (return-address slot 16-byte aligned, a null return address above it)
void fiber_init(Fiber &f) {
#ifdef EMU_ASAN
    // (the stack's previous run ended inside trampoline -> yield, never unwound: its frames' redzones are still poisoned)
    __asan_unpoison_memory_region(f.stack, kStack);
#endif
    char *top = f.stack + kStack;
    top -= (uintptr_t)top % 16;
    void **sp = reinterpret_cast<void **>(top);
    *--sp = nullptr;                                   // trampoline's (never used) return address
    *--sp = reinterpret_cast<void *>(&trampoline);     // emu_switch's `ret` target
    for (int r = 0; r < 6; ++r) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
    f.sp = sp;
}
#endif
void wave_barrier() {
    const int w = g_cur->tid / 64;
    const int gen = g_wave_gen[w];
    ++g_events;
    if (++g_wave_arrived[w] >= g_wave_alive[w]) {
        g_wave_arrived[w] = 0;
        g_wave_gen[w]++;
    } else {
        while (g_wave_gen[w] == gen) yield();
    }
}
}  // namespace

void block_barrier() {
    const int gen = g_block_gen;
    ++g_events;
    if (++g_block_arrived >= g_alive) {
        g_block_arrived = 0;
        g_block_gen++;
    } else {
        while (g_block_gen == gen) yield();
    }
}

float wave_exchange(float v, int arg, int mode, int width) {
    const int w = g_cur->tid / 64, lane = g_cur->tid % 64;
    g_xa[w][lane] = v;
    wave_barrier();
    int src;
    if (mode == 0) src = lane ^ arg;
    else if (mode == 1) src = lane + arg;
    else src = (lane / width) * width + (arg % width);
    if (mode != 2 && (src / width != lane / width || src >= 64 || src < 0)) src = lane;
    float r = g_xa[w][src];
    wave_barrier();
    return r;
}

void mfma_32x32x2(float a, float b, float *c) {
    const int w = g_cur->tid / 64, lane = g_cur->tid % 64;
    g_xa[w][lane] = a;
    g_xb[w][lane] = b;
    wave_barrier();
    const int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float d = c[r];
        d = fmaf(g_xa[w][row], g_xb[w][col], d);            // k = 0
        d = fmaf(g_xa[w][row + 32], g_xb[w][col + 32], d);  // k = 1
        c[r] = d;
    }
    wave_barrier();
}

void mfma_16x16x4(float a, float b, float *c) {
    const int w = g_cur->tid / 64, lane = g_cur->tid % 64;
    g_xa[w][lane] = a;
    g_xb[w][lane] = b;
    wave_barrier();
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float d = c[r];
        for (int k = 0; k < 4; ++k) d = fmaf(g_xa[w][row + 16 * k], g_xb[w][col + 16 * k], d);
        c[r] = d;
    }
    wave_barrier();
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    const unsigned nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > 1024) {
        fprintf(stderr, "emu: bad block size %u\n", nthreads);
        abort();
    }
    if (g_fibers.size() < nthreads) {
        size_t old = g_fibers.size();
        g_fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; ++i) g_fibers[i].stack = (char *)malloc(kStack);
    }
    if (g_wave_order < 0) {
        const char *e = getenv("MPQE_EMU_WAVE_ORDER");
        g_wave_order = e ? atoi(e) : 0;
        g_wave_rng = (unsigned long long)g_wave_order * 0x9E3779B97F4A7C15ull + 1;
    }
    blockDim = block;
    gridDim = grid;
    g_body = &body;
    // MPQE_EMU_BLOCK_ORDER=1: the workgroups of a launch run in DESCENDING order (for kernels whose workgroups do not depend
    // on each other: a result that depends on which workgroup came last shows up; the fused step's launches hand data from
    // earlier to later workgroups on purpose and need the ascending order)
    static const int block_order = getenv("MPQE_EMU_BLOCK_ORDER") ? atoi(getenv("MPQE_EMU_BLOCK_ORDER")) : 0;
    for (unsigned bz_ = 0; bz_ < grid.z; ++bz_)
        for (unsigned by_ = 0; by_ < grid.y; ++by_)
            for (unsigned bx_ = 0; bx_ < grid.x; ++bx_) {
                const unsigned bx = block_order ? grid.x - 1 - bx_ : bx_, by = block_order ? grid.y - 1 - by_ : by_,
                               bz = block_order ? grid.z - 1 - bz_ : bz_;
                blockIdx.x = bx;
                blockIdx.y = by;
                blockIdx.z = bz;
                g_alive = (int)nthreads;
                g_block_gen = g_block_arrived = 0;
                for (int w = 0; w < kMaxWaves; ++w) {
                    int lo = w * 64, hi = lo + 64;
                    g_wave_alive[w] = (int)nthreads > lo ? ((int)nthreads < hi ? (int)nthreads - lo : 64) : 0;
                    g_wave_gen[w] = g_wave_arrived[w] = 0;
                }
                for (unsigned t = 0; t < nthreads; ++t) {
                    Fiber &f = g_fibers[t];
                    f.tid = t;
                    f.done = false;
#ifdef EMU_FAST_SWITCH
                    fiber_init(f);
#else
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &g_main;
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
#endif
                }
                int remaining = (int)nthreads;
                while (remaining > 0) {
                    int progressed = 0;
                    const long ev0 = g_events;
                    // The order in which the WAVES of the workgroup get their turn: ascending by default; MPQE_EMU_WAVE_ORDER=1
                    // descending, >= 2 a fresh pseudo-random order every round (the value seeds it). Waves are independent
                    // between barriers on the GPU, so a kernel's results must not depend on this -- a hand-off between waves
                    // without a barrier shows up as a difference (lanes keep their order inside a wave: they run in lockstep
                    // there, which the emulator models only at the shuffle / MFMA exchanges).
                    const unsigned nwaves = (nthreads + 63) / 64;
                    unsigned worder[kMaxWaves];
                    for (unsigned w = 0; w < nwaves; ++w) worder[w] = g_wave_order == 1 ? nwaves - 1 - w : w;
                    if (g_wave_order >= 2)
                        for (unsigned w = nwaves; w > 1; --w) {
                            g_wave_rng = g_wave_rng * 6364136223846793005ull + 1442695040888963407ull;
                            const unsigned j = (unsigned)((g_wave_rng >> 33) % w);
                            const unsigned tmp = worder[w - 1];
                            worder[w - 1] = worder[j];
                            worder[j] = tmp;
                        }
                    for (unsigned k = 0; k < nwaves * 64; ++k) {
                        const unsigned t = worder[k / 64] * 64 + k % 64;
                        if (t >= nthreads) continue;
                        Fiber &f = g_fibers[t];
                        if (f.done) continue;
                        g_cur = &f;
                        set_tid(t);
#ifdef EMU_FAST_SWITCH
#ifdef EMU_ASAN
                        __sanitizer_start_switch_fiber(&g_main_fake, f.stack, kStack);
                        emu_switch(&g_main_sp, &f.sp);
                        __sanitizer_finish_switch_fiber(g_main_fake, nullptr, nullptr);
#else
                        emu_switch(&g_main_sp, &f.sp);
#endif
#else
                        swapcontext(&g_main, &f.ctx);
#endif
                        if (f.done) {
                            --remaining;
                            --g_alive;
                            --g_wave_alive[t / 64];
                            ++progressed;
                            // a finished work-item may have been the last one others wait for
                            if (g_alive > 0 && g_block_arrived >= g_alive) {
                                g_block_arrived = 0;
                                g_block_gen++;
                            }
                            int w = t / 64;
                            if (g_wave_alive[w] > 0 && g_wave_arrived[w] >= g_wave_alive[w]) {
                                g_wave_arrived[w] = 0;
                                g_wave_gen[w]++;
                            }
                        }
                    }
                    if (!progressed && g_events == ev0) {
                        fprintf(stderr, "emu: deadlock (divergent barrier?) in block (%u,%u,%u)\n", bx, by, bz);
                        abort();
                    }
                }
            }
    g_body = nullptr;
}
}  // namespace emu

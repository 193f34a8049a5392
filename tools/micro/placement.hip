// Where does the dispatcher put workgroups? (speed experiment only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void probe(unsigned *out, int spin) {
    extern __shared__ float lds[];
    unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    long long t0 = wall_clock64();
    float acc = 0.f;
    for (int i = 0; i < spin; ++i) { lds[threadIdx.x] = acc; __syncthreads(); acc += lds[(threadIdx.x + 1) & 255]; }
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = hw; out[3 * blockIdx.x + 1] = xcc; out[3 * blockIdx.x + 2] = (unsigned)t0 + (acc == 1.5f); }
}
int main() {
    for (int grid : {256, 512, 624, 1024}) {
        unsigned *d; hipMalloc(&d, grid * 12);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 36864, 0, d, 2000);
            hipDeviceSynchronize();
        }
        std::vector<unsigned> h(grid * 3); hipMemcpy(h.data(), d, grid * 12, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> cu;
        for (int b = 0; b < grid; ++b) {
            unsigned hw = h[3 * b], xcc = h[3 * b + 1] & 15;
            unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15);   // xcc, se, cu
            cu[key].push_back(b);
        }
        printf("grid %d: distinct CUs %zu\n", grid, cu.size());
        int hist[16] = {0}, same = 0, tot = 0;
        for (auto &kv : cu) { hist[kv.second.size() < 15 ? kv.second.size() : 15]++; }
        for (int b = 0; b + 256 < grid; ++b) { tot++; for (auto &kv : cu) { bool a = false, c = false; for (int x : kv.second) { a |= x == b; c |= x == b + 256; } if (a && c) same++; } }
        printf("  WGs/CU histogram:"); for (int i = 0; i < 16; ++i) if (hist[i]) printf(" %d:%d", i, hist[i]); printf("\n  b and b+256 on same CU: %d of %d\n", same, tot);
        int k = 0; for (auto &kv : cu) { if (k++ < 6) { printf("  cu %05x:", kv.first); for (int x : kv.second) printf(" %d", x); printf("\n"); } }
        hipFree(d);
    }
    return 0;
}

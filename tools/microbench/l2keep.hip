// Microbenchmark: does an XCD's L2 keep its lines across a kernel boundary (same stream, back-to-back launches)?
// 8 workgroups of one wave (workgroup w lands on XCD w % 8); each chases a random cycle through its own 1 MiB slice
// (8192 lines of 128 B).  Pass A and B in ONE launch: A is cold, B hits L2.  Then a second launch chases the SAME slice
// once (L2 hit only if the lines survived the boundary), a third launch the slice of the neighbouring XCD (never in this L2,
// at best in the memory-side Infinity Cache), a fourth one a slice nobody touched (HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>
constexpr int LINES = 8192, STRIDE = 32;   // ints per 128-byte line
__global__ void chase(const int* __restrict__ buf, int slice_shift, int passes, long long* clk, int* sink) {
    const int* p = buf + (size_t)((blockIdx.x + slice_shift) % 24) * LINES * STRIDE;
    int at = 0;
    for (int q = 0; q < passes; ++q) {
        const long long t0 = wall_clock64();
        for (int i = 0; i < LINES; ++i) at = __builtin_nontemporal_load(p + (size_t)at * STRIDE) ;
        const long long t1 = wall_clock64();
        if (threadIdx.x == 0) clk[blockIdx.x * 4 + q] = t1 - t0;
    }
    if (at == -7) sink[0] = at;
}
__global__ void chase_plain(const int* __restrict__ buf, int slice_shift, int passes, long long* clk, int* sink) {
    const int* p = buf + (size_t)((blockIdx.x + slice_shift) % 24) * LINES * STRIDE;
    int at = 0;
    for (int q = 0; q < passes; ++q) {
        const long long t0 = wall_clock64();
        for (int i = 0; i < LINES; ++i) at = p[(size_t)at * STRIDE];
        const long long t1 = wall_clock64();
        if (threadIdx.x == 0) clk[blockIdx.x * 4 + q] = t1 - t0;
    }
    if (at == -7) sink[0] = at;
}
int main() {
    std::vector<int> h((size_t)24 * LINES * STRIDE, 0);
    std::mt19937 rng(1);
    for (int s = 0; s < 24; ++s) {
        std::vector<int> perm(LINES); std::iota(perm.begin(), perm.end(), 0); std::shuffle(perm.begin() + 1, perm.end(), rng);
        for (int i = 0; i < LINES; ++i) h[((size_t)s * LINES + perm[i]) * STRIDE] = perm[(i + 1) % LINES];
    }
    int *buf, *sink; long long* clk;
    hipMalloc(&buf, h.size() * 4); hipMalloc(&sink, 4); hipMalloc(&clk, 8 * 4 * 8);
    hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    long long c[32];
    auto report = [&](const char* what, int q) {
        hipDeviceSynchronize(); hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost);
        double lo = 1e30, hi = 0; for (int w = 0; w < 8; ++w) { const double v = (double)c[w * 4 + q] / LINES * 10.0; lo = std::min(lo, v); hi = std::max(hi, v); }   // wall clock = 100 MHz
        printf("%-64s %6.0f .. %6.0f ns per dependent load\n", what, lo, hi);
    };
    for (int variant = 0; variant < 2; ++variant) {
        auto k = variant ? chase_plain : chase;
        printf("-- %s loads\n", variant ? "plain" : "nontemporal");
        hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, buf, 0, 2, clk, sink);
        report("launch 1, pass A (cold)", 0); report("launch 1, pass B (same launch: L2)", 1);
        hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, buf, 0, 1, clk, sink);
        report("launch 2, same slice (L2 only if it survives the boundary)", 0);
        hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, buf, 1, 1, clk, sink);
        report("launch 3, the neighbour XCD's slice (Infinity Cache)", 0);
        hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, buf, 8 + variant * 8, 1, clk, sink);
        report("launch 4, an untouched slice (HBM)", 0);
        // back-to-back without a host sync in between (as inside a hipGraph / stream)
        hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, buf, 0, 1, clk, sink);
        hipLaunchKernelGGL(k, dim3(8), dim3(64), 0, 0, buf, 0, 1, clk, sink);
        report("launch 6 right behind launch 5, same slice", 0);
    }
    return 0;
}

// Microbenchmark: what a round of scattered 1 KiB operand loads costs as a function of the SIZE of the region they fall in
// and of how the 64-scenario groups are laid out in it (the factor storage of a 512-scenario batch is 2.2 GB, of a 64-scenario
// batch 0.28 GB; a wave reads one 2x2 block of one entry for its 64 scenarios = 1 KiB).
//   mode 0 (batch-minor, as the engine stores it): entry e of group g at (e * 8 + g) KiB
//   mode 1 (group-major):                          entry e of group g at g * region/8 + e KiB
// Workgroups are 8 waves; workgroup w serves group w % 8 (= the XCD it lands on); every round a wave issues 12 loads to random
// entries and waits for them.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(512) void k(const double2* __restrict__ src, double* out, unsigned entries, int mode, int rounds, int shared) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned g = blockIdx.x & 7;
    double acc = 0;
    for (int r = 0; r < rounds; ++r) {
        double2 v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            // shared != 0: the workgroups of one group draw from the same small pool of entries in a round (L2 hits)
            const unsigned seed = shared ? (unsigned)((r * 131 + (i + wave * 12) % shared)) : (unsigned)(((blockIdx.x * 8 + wave) * 4099 + r) * 12 + i);
            const unsigned e = hash(seed) % entries;
            const size_t kib = mode == 0 ? (size_t)e * 8 + g : (size_t)g * entries + e;
            v[i] = src[kib * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) acc += v[i].x + v[i].y;
        if (acc == 1.2345) break;
    }
    if (acc == 12345.678) out[0] = acc;
}
int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double* out; hipMalloc(&out, 8);
    for (size_t mb : {256, 1024, 2304, 4608}) {
        const size_t bytes = mb << 20;
        double2* src; if (hipMalloc(&src, bytes) != hipSuccess) { printf("alloc %zu MB failed\n", mb); continue; }
        hipMemset(src, 0, bytes);
        const unsigned entries = (unsigned)(bytes / 8192);
        for (int mode : {0, 1}) for (int grid : {8, 80, 640, 2048}) for (int shared : {0, 48}) {
            const int rounds = 200;
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, src, out, entries, mode, 4, shared);
            hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, src, out, entries, mode, rounds, shared); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double gb = (double)grid * 8 * 12 * 1024 * rounds / 1e9;
            printf("region %5zu MB  %-12s grid %4d  %s: %7.3f us per round of 12 loads, %7.1f GB/s requested\n", mb, mode ? "group-major" : "batch-minor", grid,
                   shared ? "pool of 48 per group " : "all different        ", ms * 1e3 / rounds, gb / (ms * 1e-3));
        }
        hipFree(src);
    }
    return 0;
}

// f64 FMA cost per wave-instruction on gfx950: dependent chain vs independent chains, by waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o f64rate f64rate.hip && ./f64rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ __launch_bounds__(1024) void k(int iters, double* out, long long* clk) {
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = 1.0 + threadIdx.x * 1e-3 + c;
    const double y = 0.5;
    const long long t0 = clock64();
    for (int q = 0; q < iters; ++q) {
#pragma unroll
        for (int i = 0; i < 64 / CH; ++i)
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = fma(x[c], 0.999999, y);
    }
    const long long t1 = clock64();
    double s = 0; for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[threadIdx.x >> 6] = t1 - t0;          // every wave reports: the oldest wave wins the issue arbitration
}
int main() {
    double* out; long long* clk;
    (void)hipMalloc(&out, 1024 * 1024 * 8); (void)hipMalloc(&clk, 1024 * 8);
    const int iters = 200;
    for (int threads : {64, 256, 320, 512, 1024}) {
        auto run = [&](auto kern, int ch) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, iters, out, clk);
            hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, iters, out, clk);
            (void)hipDeviceSynchronize();
            long long h[16]; (void)hipMemcpy(h, clk, 16 * 8, hipMemcpyDeviceToHost);
            long long mx = 0, mn = 1LL << 60; for (int w = 0; w < threads / 64; ++w) { mx = h[w] > mx ? h[w] : mx; mn = h[w] < mn ? h[w] : mn; }
            printf("%4d threads (%d waves / SIMD), %d independent chains: %.2f .. %.2f clocks per f64 FMA wave-instruction (fastest .. slowest wave)\n", threads, (threads + 255) / 256, ch, (double)mn / (iters * 64.0), (double)mx / (iters * 64.0));
        };
        run(k<1>, 1); run(k<2>, 2); run(k<4>, 4); run(k<8>, 8);
    }
    return 0;
}

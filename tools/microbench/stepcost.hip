// What does one pivot step of k_fact_top cost?  Clocks per iteration of a loop made of the step's primitives:
//   0 barrier only | 1 barrier + LDS write/read round trip | 2 + 40 dependent f64 FMAs | 3 + two v_rcp_f64 Newton chains
//   4 = 2 but the FMA chain only on wave 4 (the others wait at the barrier) | 5 = dependent FMAs alone, no barrier
// hipcc --offload-arch=gfx950 -O3 -o stepcost stepcost.hip && ./stepcost
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(320) void k(int mode, int iters, double* out, long long* clk) {
    __shared__ double buf[2][64 * 4];
    const int tid = threadIdx.x;
    double x = 1.0 + tid * 1e-3, y = 0.5;
    buf[0][tid & 255] = x; buf[1][tid & 255] = y;
    __syncthreads();
    const long long t0 = clock64();
    for (int q = 0; q < iters; ++q) {
        if (mode >= 1 && mode != 5) {
            buf[(q + 1) & 1][tid & 255] = x;
        }
        if (mode == 2 || mode == 3 || (mode == 4 && tid >= 256) || mode == 5) {
#pragma unroll
            for (int i = 0; i < 40; ++i) x = fma(x, 0.999999, y);
        }
        if (mode == 3) {
            double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); r = fma(fma(-x, r, 1.0), r, r);
            double s = __builtin_amdgcn_rcp(r + 1.0); s = fma(fma(-(r + 1.0), s, 1.0), s, s); s = fma(fma(-(r + 1.0), s, 1.0), s, s);
            x += s;
        }
        if (mode != 5) __syncthreads();
        if (mode >= 1 && mode != 5) y = buf[(q + 1) & 1][(tid + 1) & 255];
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + tid] = x + y;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* clk;
    hipMalloc(&out, 1024 * 320 * 8); hipMalloc(&clk, 1024 * 8);
    const int iters = 1000;
    for (int grid : {1, 64, 512}) for (int mode = 0; mode <= 5; ++mode) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(320), 0, 0, mode, iters, out, clk);
        hipLaunchKernelGGL(k, dim3(grid), dim3(320), 0, 0, mode, iters, out, clk);
        hipDeviceSynchronize();
        long long h[1024]; hipMemcpy(h, clk, grid * 8, hipMemcpyDeviceToHost);
        long long mx = 0; for (int i = 0; i < grid; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("grid %4d mode %d: %.1f clocks per iteration (workgroup 0: %.1f)\n", grid, mode, (double)mx / iters, (double)h[0] / iters);
    }
    return 0;
}

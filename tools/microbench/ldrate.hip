// Microbenchmark: time for ONE workgroup of 16 waves per CU to issue NL dependent-free loads per wave and consume them.
// Variants: 8-byte loads (dwordx2) vs 16-byte loads (dwordx4) per lane; lanes spread over 512 B / 1 KB vs all lanes one address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int W, bool UNIFORM>
__global__ __launch_bounds__(1024) void k(const double* __restrict__ src, double* out, int nl, size_t stride, int reps) {
    const int lane = threadIdx.x, wave = threadIdx.y;
    const size_t b = UNIFORM ? 0 : (size_t)lane * W;
    double acc = 0;
    for (int r = 0; r < reps; ++r) {
        const double* p = src + ((size_t)(blockIdx.x * 16 + wave) * 64 + (size_t)r * 7919 % 1024) * stride + b;
#pragma unroll 4
        for (int i = 0; i < nl; i += 4) {
            if (W == 1) { acc += p[(size_t)i * stride] + p[(size_t)(i + 1) * stride] + p[(size_t)(i + 2) * stride] + p[(size_t)(i + 3) * stride]; }
            else {
                const double2 a0 = *(const double2*)(p + (size_t)i * stride), a1 = *(const double2*)(p + (size_t)(i + 1) * stride);
                const double2 a2 = *(const double2*)(p + (size_t)(i + 2) * stride), a3 = *(const double2*)(p + (size_t)(i + 3) * stride);
                acc += a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y;
            }
        }
    }
    if (acc == 12345.678) out[0] = acc;
}
int main() {
    const size_t stride = 128;  // doubles between consecutive loads of a wave (1 KB apart)
    const size_t n = (size_t)256 * 16 * 64 * stride * 2 + 1024 * stride * 64;
    double *src, *out; hipMalloc(&src, n * 8); hipMalloc(&out, 8); hipMemset(src, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, int grid, int nl) {
        const int reps = 200;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64, 16), 0, 0, src, out, nl, stride, 2);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(64, 16), 0, 0, src, out, nl, stride, reps); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s grid %3d loads/wave %2d: %.3f us per round (16 waves/CU), %.1f ns per wave-load-instr per CU\n", name, grid, nl, ms * 1e3 / reps, ms * 1e6 / reps / (16.0 * nl));
    };
    for (int grid : {1, 256}) for (int nl : {16, 48}) {
        run(k<1, false>, "8B/lane, 512B segment", grid, nl);
        run(k<1, true>, "8B/lane, one address", grid, nl);
        run(k<2, false>, "16B/lane, 1KB segment", grid, nl);
        run(k<2, true>, "16B/lane, one address", grid, nl);
    }
    return 0;
}

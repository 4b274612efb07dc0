// Dependent-launch floor on MI355X: N tiny kernels in one stream, (a) plain launches, (b) one hipGraph.
// Each kernel does one dependent load + store so the chain is real.  hipcc --offload-arch=gfx950 -O3 launchgap.hip -o launchgap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_step(const double* in, double* out) { out[threadIdx.x + blockIdx.x * blockDim.x] = in[threadIdx.x + blockIdx.x * blockDim.x] + 1.0; }
__global__ void k_empty() {}
int main() {
    const int N = 200, G = 64;
    double *a, *b;
    hipMalloc(&a, G * 1024 * 8); hipMalloc(&b, G * 1024 * 8);
    hipMemset(a, 0, G * 1024 * 8);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 3; ++variant) {
        const int grid = variant == 2 ? G : 1;
        auto enqueue = [&]() { for (int i = 0; i < N; ++i) { if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); else hipLaunchKernelGGL(k_step, dim3(grid), dim3(1024), 0, st, (i & 1) ? b : a, (i & 1) ? a : b); } };
        enqueue(); hipStreamSynchronize(st);
        hipEventRecord(e0, st); enqueue(); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal); enqueue(); hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float mg; hipEventElapsedTime(&mg, e0, e1);
        printf("%s: stream %.2f us/kernel, graph %.2f us/kernel\n", variant == 0 ? "empty 1 wave" : (variant == 1 ? "load+store 1 WG of 16 waves" : "load+store 64 WGs"), 1e3 * ms / N, 1e3 * mg / N);
    }
    return 0;
}

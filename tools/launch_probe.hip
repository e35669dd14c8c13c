// Diagnostic: cost of a dependent chain of small kernels on this box, eager vs hipGraph replay.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(float* x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0001f + 1.f; }
int main() {
    float* x; hipMalloc(&x, 1 << 20); hipMemset(x, 0, 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    const int N = 400;
    for (int grid : {1, 32, 256}) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, x, 65536);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, x, 65536);
        auto t1 = std::chrono::high_resolution_clock::now();
        hipStreamSynchronize(s);
        auto t2 = std::chrono::high_resolution_clock::now();
        double host = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
        double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / N;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, x, 65536);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        auto t3 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        auto t4 = std::chrono::high_resolution_clock::now();
        double gr = std::chrono::duration<double, std::micro>(t4 - t3).count() / (5.0 * N);
        printf("grid %3d: eager host-submit %.2f us/launch, eager end-to-end %.2f us/launch, graph replay %.2f us/kernel\n", grid, host, tot, gr);
    }
    return 0;
}

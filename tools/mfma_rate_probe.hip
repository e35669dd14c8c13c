// fp32 MFMA issue rate on MI355X as a function of (independent accumulator chains per wave, waves per SIMD)
// (developer probe, GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o tools/bin/mfma_rate_probe)
// Every wave issues ITERS * 8 v_mfma_f32_32x32x2_f32 (or 16x16x4) from registers; CH chains: consecutive MFMAs rotate over
// CH accumulators.  Prints cycles per MFMA per SIMD (s_memtime) and the clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int CH, int GAP>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* clk, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % CH], 0, 0, 0);
        if (GAP == 1) __builtin_amdgcn_s_barrier();
        if (GAP == 2) { __builtin_amdgcn_s_barrier(); a += 1e-6f; b -= 1e-6f; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int CH>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* clk, int iters) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % CH], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&clk, 4096 * 16));
    unsigned long long h[4096 * 2];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2048;
    auto run = [&](const char* name, auto kern, int wgs, double flop_per_mfma) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, out, clk, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, clk, wgs * 16, hipMemcpyDeviceToHost));
        double cyc = 0, ns = 0; for (int i = 0; i < wgs; ++i) { cyc += h[2 * i]; ns += h[2 * i + 1] * 10.0; }
        cyc /= wgs; ns /= wgs;
        const double wps = wgs / 256.0;      // waves per SIMD
        printf("%-34s wgs %4d: %7.1f us  %6.1f TFLOP/s  | per wave: %6.1f shader cycles per MFMA (%.1f per SIMD slot), clock %.2f GHz\n", name, wgs,
               ms * 1e3, wgs * 4.0 * iters * 8 * flop_per_mfma / ms / 1e9, cyc / (iters * 8), cyc / (iters * 8) / wps, cyc / ns);
    };
    for (int wgs : {256, 512, 1024}) {
        run("32x32x2 1 chain", k32<1, 0>, wgs, 4096);
        run("32x32x2 2 chains", k32<2, 0>, wgs, 4096);
        run("32x32x2 4 chains", k32<4, 0>, wgs, 4096);
        run("32x32x2 1 chain + barrier/8", k32<1, 1>, wgs, 4096);
        run("32x32x2 1 chain + barrier+valu/8", k32<1, 2>, wgs, 4096);
        run("32x32x2 4 chains + barrier+valu/8", k32<4, 2>, wgs, 4096);
        run("16x16x4 1 chain", k16<1>, wgs, 2048);
        run("16x16x4 4 chains", k16<4>, wgs, 2048);
    }
    return 0;
}

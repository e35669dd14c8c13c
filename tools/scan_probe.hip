// Diagnostic micro-benchmark (not part of the product): where do the ~18 us per GRU step of the
// persistent scan go?  One workgroup repeats a step-shaped loop (two wave-GEMMs with A in LDS and
// B streamed from L2 in packed order) in several ablation modes.
//   hipcc --offload-arch=gfx950 -O3 tools/scan_probe.hip -o gpurun_out/scan_probe && ./gpurun_out/scan_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCG, int PDK, bool LOADS, bool MFMA>
__device__ __forceinline__ void wave_gemm(const float* __restrict__ ldsA, int lda, const float4* __restrict__ p,
                                          int KB, int lane, f32x4* acc, float& sink) {
    float4 ring[PDK][NCG];
#pragma unroll
    for (int i = 0; i < PDK; ++i)
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) ring[i][cg] = LOADS ? p[(i * NCG + cg) * 64 + lane] : make_float4(1.f, 2.f, 3.f, 4.f);
    const float* ap = ldsA + (lane & 15) * lda + (lane >> 4);
    int kb0 = 0;
#pragma unroll 1
    for (; kb0 + PDK < KB; kb0 += PDK) {
#pragma unroll
        for (int i = 0; i < PDK; ++i) {
            const int kb = kb0 + i;
            const float a = ap[4 * kb];
            float4 b[NCG];
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                b[cg] = ring[i][cg];
                if (LOADS) ring[i][cg] = p[((kb + PDK) * NCG + cg) * 64 + lane];
            }
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                if (MFMA) {
                    acc[cg * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[cg].x, acc[cg * 4 + 0], 0, 0, 0);
                    acc[cg * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[cg].y, acc[cg * 4 + 1], 0, 0, 0);
                    acc[cg * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[cg].z, acc[cg * 4 + 2], 0, 0, 0);
                    acc[cg * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[cg].w, acc[cg * 4 + 3], 0, 0, 0);
                } else {
                    sink += a * (b[cg].x + b[cg].y + b[cg].z + b[cg].w);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PDK; ++i) {
        const float a = ap[4 * (kb0 + i)];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            if (MFMA) {
                acc[cg * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][cg].x, acc[cg * 4 + 0], 0, 0, 0);
                acc[cg * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][cg].y, acc[cg * 4 + 1], 0, 0, 0);
                acc[cg * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][cg].z, acc[cg * 4 + 2], 0, 0, 0);
                acc[cg * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][cg].w, acc[cg * 4 + 3], 0, 0, 0);
            } else {
                sink += a * (ring[i][cg].x + ring[i][cg].y + ring[i][cg].z + ring[i][cg].w);
            }
        }
    }
}

// H = 256: phase 1 NCB = 8 (NCG 2), phase 2 NCB = 4 (NCG 1), KB = 64
template <bool LOADS, bool MFMA, bool RAWBAR, int PD1, int PD2>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ pk, float* __restrict__ out, int steps) {
    constexpr int H = 256, LDA = H + 2, KB = 64;
    __shared__ float hb[16 * LDA];
    __shared__ float rhb[16 * LDA];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 16 * LDA; i += 256) { hb[i] = 0.01f * (i % 7); rhb[i] = 0.02f * (i % 5); }
    __syncthreads();
    const float4* p1 = reinterpret_cast<const float4*>(pk + (size_t)w * H * 128);
    const float4* p2 = reinterpret_cast<const float4*>(pk + (size_t)2 * H * H + (size_t)w * H * 64);
    float sink = 0.f;
    for (int t = 0; t < steps; ++t) {
        f32x4 acc1[8], acc2[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc1[i] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[i] = f32x4{0, 0, 0, 0};
        wave_gemm<2, PD1, LOADS, MFMA>(hb, LDA, p1, KB, lane, acc1, sink);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r, col = 64 * w + 16 * j + (lane & 15);
                rhb[row * LDA + col] = fminf(fmaxf(0.2f * (acc1[j][r] + acc1[4 + j][r]) + 0.5f, 0.f), 1.f) * hb[row * LDA + col];
            }
        if (RAWBAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        else __syncthreads();
        wave_gemm<1, PD2, LOADS, MFMA>(rhb, LDA, p2, KB, lane, acc2, sink);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r, col = 64 * w + 16 * j + (lane & 15);
                hb[row * LDA + col] = fmaxf(acc2[j][r], 0.f) * 0.5f + 0.5f * hb[row * LDA + col];
            }
        if (RAWBAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        else __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = sink + hb[tid];
}

template <typename K> float time_kernel(K k, int grid, const float* pk, float* out, int steps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, pk, out, steps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, pk, out, steps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / steps;
}

int main() {
    const size_t n = 3 * 256 * 256;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = 0.001f * (float)((i * 2654435761u) % 1000) - 0.5f;
    float *pk, *out;
    hipMalloc(&pk, n * 4); hipMalloc(&out, 64 * 256 * 4);
    hipMemcpy(pk, h.data(), n * 4, hipMemcpyHostToDevice);
    const int steps = 200;
    for (int grid : {1, 8, 32}) {
        printf("grid %2d  us/step: full %.2f | loads-only %.2f | mfma-only %.2f | full+rawbar %.2f | rawbar ring x2 %.2f | rawbar ring/2 %.2f\n", grid,
               time_kernel(probe<true, true, false, 8, 16>, grid, pk, out, steps),
               time_kernel(probe<true, false, false, 8, 16>, grid, pk, out, steps),
               time_kernel(probe<false, true, false, 8, 16>, grid, pk, out, steps),
               time_kernel(probe<true, true, true, 8, 16>, grid, pk, out, steps),
               time_kernel(probe<true, true, true, 16, 32>, grid, pk, out, steps),
               time_kernel(probe<true, true, true, 4, 8>, grid, pk, out, steps));
    }
    return 0;
}

// What does an IN-KERNEL exchange between the column-block workgroups of one row block cost on MI355X?
// (developer probe, GPU box:  hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier_probe.hip -o tools/bin/xcd_barrier_probe)
//
// Mirrors the scan's structure: G groups of W workgroups (256 threads), group g pinned to XCD g % 8 the same way
// rnn_step.hip places a row block.  Per iteration every workgroup publishes 1 KB (its 16 x 16 tile), arrives
// on the group's counter, waits until all W have arrived and then reads the group's W KB.  Two publish
// forms: (a) plain stores + agent-scope release fence, acquire fence after the wait; (b) write-through (sc1)
// stores drained with s_waitcnt, sc1 loads after the wait.  EVERY spin is bounded; a timed-out run reports it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void exchange_kernel(float* __restrict__ buf, unsigned* __restrict__ counters, int W, int G,
                                                       int iters, unsigned* __restrict__ timeouts, float* __restrict__ sink) {
    const int L = blockIdx.x, x = L & 7, s = L >> 3;
    const int j = s / W, wg = s - j * W;
    const int g = x + 8 * j;
    if (g >= G) return;
    const int tid = threadIdx.x;
    float* gbuf = buf + (size_t)g * 2 * W * 256;            // two alternating slabs of W KB per group
    unsigned* ctr = counters + g * 32;                       // own 128-byte line
    float acc = 0.f;
    __shared__ int ok;
    for (int it = 0; it < iters; ++it) {
        float* slab = gbuf + (size_t)(it & 1) * W * 256;
        const float v = (float)(it + wg) + acc * 1e-9f;
        if (MODE == 0) {
            slab[wg * 256 + tid] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        } else {
            __hip_atomic_store(slab + wg * 256 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 write-through
            __builtin_amdgcn_s_waitcnt(0);
        }
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(it + 1) * (unsigned)W;
            int spins = 0, good = 1;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > (1 << 22)) { good = 0; break; }                   // bounded: ~seconds at worst
                __builtin_amdgcn_s_sleep(1);
            }
            ok = good;
            if (!good) atomicAdd(timeouts, 1u);
        }
        __syncthreads();
        if (!ok) return;
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int k = tid; k < W * 256; k += 256) {
            if (MODE == 0) acc += slab[k];
            else acc += __hip_atomic_load(slab + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int iters = 400;
    float *buf, *sink; unsigned *ctr, *to;
    CK(hipMalloc(&buf, (size_t)32 * 2 * 64 * 256 * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&ctr, 32 * 32 * 4)); CK(hipMalloc(&to, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int W : {16, 48})
            for (int G : {1, 2, 8, 32}) {
                if ((long)G * W > 1536) continue;
                const int grid = 8 * W * ((G + 7) / 8);
                float best = 1e30f; unsigned tmo = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(ctr, 0, 32 * 32 * 4)); CK(hipMemset(to, 0, 4));
                    CK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(exchange_kernel<0>, dim3(grid), dim3(256), 0, 0, buf, ctr, W, G, iters, to, sink);
                    else hipLaunchKernelGGL(exchange_kernel<1>, dim3(grid), dim3(256), 0, 0, buf, ctr, W, G, iters, to, sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned t; CK(hipMemcpy(&t, to, 4, hipMemcpyDeviceToHost)); tmo += t;
                    if (ms < best) best = ms;
                }
                printf("%s  W=%2d workgroups/group  G=%2d groups : %.2f us per exchange%s\n", mode == 0 ? "plain+fences " : "sc1 write-thru",
                       W, G, best * 1000.f / iters, tmo ? "  (TIMEOUTS!)" : "");
            }
    return 0;
}

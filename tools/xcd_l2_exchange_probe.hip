// Can the 16 column-block workgroups of one row block exchange h / r*h INSIDE a kernel through their XCD's L2, and what does it cost?
// (developer probe, GPU box:  hipcc --offload-arch=gfx950 -O3 tools/xcd_l2_exchange_probe.hip -o tools/bin/xcd_l2_exchange_probe)
//
// G groups of W workgroups (256 threads); group g's workgroups have blockIdx % 8 == g % 8 (round-robin placement puts them on one
// XCD; every workgroup records its XCC_ID so the run says whether that held).  Per iteration each workgroup: loads the group's
// W KB of the previous iteration, derives its own 1 KB from them (a stale or torn read changes every later value), stores it,
// drains its stores (vmcnt 0), sets its flag; wave 0 polls the W flags; then everyone loads the W KB.
// MODE (publish / poll / read):
//   0  plain stores, flag plain store | poll: sc1 load           | read: sc1 loads                    (device-scope loads, no invalidate)
//   1  plain stores, flag plain store | poll: buffer_inv sc1 + plain load | read: buffer_inv sc1 + plain loads
//   2  sc1 write-through stores + flag | poll: sc1 load | read: sc1 loads                                (the round-1 recipe, for reference)
//   3  plain stores, flag plain store | poll: sc0 sc1 load (system scope) | read: sc0 sc1 loads
// Every value is checked against the closed form; every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE> __device__ __forceinline__ f32x4 ld4(const float* p) {
    f32x4 v;
    if (MODE == 0 || MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE> __device__ __forceinline__ unsigned ld1(const unsigned* p) {
    unsigned v;
    if (MODE == 0 || MODE == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE> __device__ __forceinline__ void st1(float* p, float v) {
    if (MODE == 2) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}

template <int MODE, int W>
__global__ __launch_bounds__(256) void exchange_kernel(float* __restrict__ buf, unsigned* __restrict__ flags, int G, int iters,
                                                       unsigned* __restrict__ errors, unsigned* __restrict__ xcc, unsigned long long* __restrict__ clk) {
    const int L = blockIdx.x, x = L & 7, s = L >> 3;
    const int j = s / W, wg = s - j * W;
    const int g = x + 8 * j;
    if (g >= G) return;
    const int tid = threadIdx.x;
    if (tid == 0) xcc[g * W + wg] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF;
    float* gbuf = buf + (size_t)g * 2 * W * 256;            // two alternating slabs of W KB per group
    unsigned* gfl = flags + g * 64;                          // the group's W flags share two 128-byte lines
    __shared__ int ok;
    unsigned bad = 0;
    float own = 1.0f + wg * 0.001f + tid * 0.0001f;          // iteration 0 value
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        float* slab = gbuf + (size_t)(it & 1) * W * 256;
        st1<MODE>(slab + wg * 256 + tid, own);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 64) {
            if (tid == 0) { asm volatile("global_store_dword %0, %1, off" :: "v"(gfl + wg), "v"((unsigned)(it + 1)) : "memory"); }
            int spins = 0, good = 1;
            while (true) {
                unsigned f = tid < W ? ld1<MODE>(gfl + tid) : 0xFFFFFFFFu;
                if (__all((int)(f >= (unsigned)(it + 1)))) break;
                if (++spins > (1 << 20)) { good = 0; break; }
            }
            if (tid == 0) ok = good;
        }
        __syncthreads();
        if (!ok) { if (tid == 0) atomicAdd(errors + 1, 1u); return; }
        if (MODE == 1) asm volatile("buffer_inv sc1" ::: "memory");
        // read the group's W KB: thread t reads 16 bytes of each of W/4 ... (W*256 floats / 256 threads = W floats = W/4 dwordx4)
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const int e = (q * 256 + tid) * 4;               // element index in the slab
            const f32x4 v = ld4<MODE>(slab + e);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ee = e + c, w2 = ee >> 8, t2 = ee & 255;
                // expected value of workgroup w2, thread t2 at iteration it (closed form below)
                const float expect = 1.0f + w2 * 0.001f + t2 * 0.0001f + it * 0.5f;
                if (v[c] != expect) ++bad;
                sum += v[c];
            }
        }
        own = 1.0f + wg * 0.001f + tid * 0.0001f + (it + 1) * 0.5f + (sum == 123456.f ? 1.f : 0.f);
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(errors, bad);
    if (tid == 0) clk[g * W + wg] = r1 - r0;
}

int main() {
    const int iters = 300, W = 16;
    float* buf; unsigned *flags, *err, *xcc; unsigned long long* clk;
    CK(hipMalloc(&buf, (size_t)64 * 2 * W * 256 * 4)); CK(hipMalloc(&flags, 64 * 64 * 4)); CK(hipMalloc(&err, 8)); CK(hipMalloc(&xcc, 64 * W * 4));
    CK(hipMalloc(&clk, 64 * W * 8));
    std::vector<unsigned> hx(64 * W); std::vector<unsigned long long> hc(64 * W);
    for (int mode = 0; mode < 4; ++mode)
        for (int G : {1, 4, 8, 32}) {
            const int grid = 8 * W * ((G + 7) / 8);
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(flags, 0, 64 * 64 * 4)); CK(hipMemset(err, 0, 8)); CK(hipMemset(buf, 0, (size_t)64 * 2 * W * 256 * 4));
                CK(hipDeviceSynchronize());
                switch (mode) {
                    case 0: hipLaunchKernelGGL((exchange_kernel<0, 16>), dim3(grid), dim3(256), 0, 0, buf, flags, G, iters, err, xcc, clk); break;
                    case 1: hipLaunchKernelGGL((exchange_kernel<1, 16>), dim3(grid), dim3(256), 0, 0, buf, flags, G, iters, err, xcc, clk); break;
                    case 2: hipLaunchKernelGGL((exchange_kernel<2, 16>), dim3(grid), dim3(256), 0, 0, buf, flags, G, iters, err, xcc, clk); break;
                    default: hipLaunchKernelGGL((exchange_kernel<3, 16>), dim3(grid), dim3(256), 0, 0, buf, flags, G, iters, err, xcc, clk); break;
                }
                CK(hipDeviceSynchronize());
                unsigned e[2]; CK(hipMemcpy(e, err, 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hx.data(), xcc, G * W * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), clk, G * W * 8, hipMemcpyDeviceToHost));
                int split = 0; double ns = 0;
                for (int g = 0; g < G; ++g) { for (int w = 1; w < W; ++w) if (hx[g * W + w] != hx[g * W]) { ++split; break; } }
                for (int i = 0; i < G * W; ++i) ns += hc[i] * 10.0; ns /= (G * W);
                if (rep == 1) printf("mode %d  G=%2d groups of %d: %.2f us per exchange, wrong values %u, timeouts %u, groups spanning XCDs %d\n", mode, G, W,
                                     ns / iters / 1000.0, e[0], e[1], split);
            }
        }
    return 0;
}

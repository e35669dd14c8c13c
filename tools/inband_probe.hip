// What is the floor of the cluster scans' in-kernel exchange, and which protocol reaches it?  (developer probe, GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/inband_probe.hip -o tools/bin/inband_probe && ./tools/bin/inband_probe)
//
// G groups of 16 workgroups (256 threads, 4 waves); a group's workgroups sit on one XCD (blockIdx % 8).  Iteration `it` of a group
// exchanges one [16 rows x 256 columns] fp32 slab (16 KB, its own address range, pre-filled with the sentinel 0xFFFFFFFF by the
// host): workgroup c stores columns [16c, 16c + 16) -- thread (row, col) one float, exactly what a scan step publishes -- and
// every WAVE then needs rows 0..15 x its K slice [64w, 64w + 64) (4 KB, four LDS-DMA instructions in full lines, as
// rnn_cluster_dev.h ld_rows_dma) before it can produce the next value.  Every loaded value is checked against the closed form.
// PROTOCOL:
//   0  flags (what the GRU scans did up to round 3): store, drain (vmcnt 0), barrier, flag; every wave polls the 16 flags
//      (sc1 dword loads), then loads its rows
//   1  in-band: store; every wave loads its rows until no word is the sentinel (re-issuing the DMA instructions that showed one)
//   2  in-band, two polls in flight (alternating LDS images, counted waits)
//   3  hint + in-band: store; every wave polls 16 single words (one per (producer, storing wave) of its slice) until none is the
//      sentinel, then loads its rows and validates them (re-polling if a word is still missing)
// and, first, the primitives on an idle chip: store -> ack, dword load round trip, 4-instruction DMA round trip (L2 hits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
constexpr unsigned SENT = 0xFFFFFFFFu;
constexpr int W = 16, COLS = 256, SLAB = 16 * COLS, GSTAMP = 32 * 16 * 4;

__device__ __forceinline__ unsigned ld1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st1(float* p, float v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ float expect(int it, int row, int col) { return 1.0f + 0.001f * col + 0.0625f * row + 0.5f * it; }

// rows 0..15 x [64 w, +64) of `slab` into the wave's LDS image (chunk c of row m at c ^ (m % 16)); instruction i = rows 4i..4i+3
// LAYOUT 1 (tile-major slab: producer c's 16 x 16 tile is 1 KB contiguous = 8 whole lines, written by ONE workgroup): instruction i = the
// tile of producer 4 w + i, lane l lands at chunk l = (row l / 4, piece l % 4) and fetches piece (l % 4) ^ ((row >> 2) & 3) of that row
template <int LAYOUT> __device__ __forceinline__ void dma_issue(const float* slab, int w, float* img, int lane, unsigned need) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((need >> i) & 1u) {
            const int rl = 4 * i + lane / 16;
            const int ch = (lane % 16) ^ (rl % 16);
            const int r1 = lane >> 2;
            const float* p = LAYOUT == 0 ? slab + rl * COLS + 64 * w + 4 * ch : slab + (4 * w + i) * 256 + r1 * 16 + 4 * ((lane & 3) ^ ((r1 >> 2) & 3));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(img + i * 256),
                                             16, 0, 16 /* sc1 */);
        }
    }
}
typedef __attribute__((address_space(3))) float lds_float;
typedef float f32x4 __attribute__((ext_vector_type(4)));
// the lane's 16 floats in MFMA operand order (row m = lane & 15, floats [16 q, +16) of the slice), by asm reads tied to their wait
template <int LAYOUT> __device__ __forceinline__ void img_read(f32x4 (&v)[4], const float* img, int lane) {
    const int m = lane & 15, q = lane >> 4;
    const unsigned src = (unsigned)(uintptr_t)(const lds_float*)(LAYOUT == 0 ? img + m * 64 : img + q * 256 + m * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pos = LAYOUT == 0 ? (q * 4 + j) ^ (m % 16) : j ^ ((m >> 2) & 3);
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(src + 16u * (unsigned)pos));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
}
// LAYOUT 2 (operand-major slab): wave w's instruction i is 1 KB contiguous -- lane (m, q) finds floats [64 w + 16 q + 4 i, + 4) of row m at
// ((4 w + i) 64 + lane) 16 bytes: four coalesced device-scope loads straight into the MFMA operand registers, no LDS staging at all
__device__ __forceinline__ void direct_read(f32x4 (&v)[4], const float* slab, int w, int lane) {
    const float* p = slab + ((size_t)(4 * w) * 64 + lane) * 4;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p) : "memory");
}
__device__ __forceinline__ bool has_sent(const f32x4 (&v)[4]) {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        bad = bad || __float_as_uint(v[j].x) == SENT || __float_as_uint(v[j].y) == SENT || __float_as_uint(v[j].z) == SENT || __float_as_uint(v[j].w) == SENT;
    return bad;
}
template <int LAYOUT> __device__ __forceinline__ unsigned need_of(unsigned long long bm) {
    unsigned need = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (bm & (LAYOUT == 0 ? 0x000F000F000F000Full << (4 * i) : 0xFFFFull << (16 * i))) need |= 1u << i;
    return need;
}

template <int PROTO, int LAYOUT, bool STAMP = false>
__global__ __launch_bounds__(256) void exchange_kernel(float* __restrict__ buf, unsigned* __restrict__ flags, int G, int iters, int extra_stores, int nload,
                                                       float* __restrict__ sink, unsigned* __restrict__ errors, unsigned long long* __restrict__ clk,
                                                       unsigned long long* __restrict__ polls) {
    const int L = blockIdx.x, x = L & 7, s = L >> 3;
    const int j = s / W, c = s - j * W;
    const int g = x + 8 * j;
    if (g >= G) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = tid >> 4, col = 16 * c + (tid & 15);
    float* gbuf = buf + (size_t)g * iters * SLAB;
    unsigned* gfl = flags + g * 64;
    __shared__ float stage[2][4][1024];
    unsigned bad = 0;
    unsigned long long npoll = 0;
    float own = expect(0, row, col);
    int dk = 0;                                                  // PROTO 2: DMA groups issued so far
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long sacc[5] = {0, 0, 0, 0, 0}, sprev = r0;
#define ST(i_) do { if (STAMP) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); sacc[i_] += t_ - sprev; sprev = t_; } } while (0)
    for (int it = 0; it < iters; ++it) {
        float* slab = gbuf + (size_t)it * SLAB;
        st1(LAYOUT == 0 ? slab + row * COLS + col : LAYOUT == 1 ? slab + c * 256 + row * 16 + (tid & 15)
                        : slab + (((c >> 2) * 4 + ((tid & 15) >> 2)) * 64 + (c & 3) * 16 + row) * 4 + (tid & 3), own);
        // the scans' other stores of a step (gate stash): younger than the exchange store, nobody waits for them on purpose
        for (int e = 0; e < extra_stores; ++e) st1(sink + ((size_t)(blockIdx.x * 4 + e) * iters + it) * 256 + tid, own);
        f32x4 v[4];
        int spins = 0;
        if (PROTO == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(0) : "memory");
            __syncthreads();
            if (tid == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(gfl + c), "v"((unsigned)(it + 1)) : "memory");
            while (true) {
                const unsigned f = lane < W ? ld1(gfl + lane) : 0xFFFFFFFFu;
                ++npoll;
                if (__all((int)(f >= (unsigned)(it + 1)))) break;
                if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(errors + 1, 1u); return; }
            }
            if (LAYOUT == 2) direct_read(v, slab, w, lane);
            else {
                dma_issue<LAYOUT>(slab, w, stage[0][w], lane, 15u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                img_read<LAYOUT>(v, stage[0][w], lane);
            }
        } else if (PROTO == 1 && LAYOUT == 2) {
            while (true) {
                direct_read(v, slab, w, lane);
                ++npoll;
                if (__ballot(has_sent(v)) == 0ull) break;
                if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(errors + 1, 1u); return; }
            }
        } else if (PROTO == 1) {
            unsigned need = 15u;
            if (w >= nload) { need = 0; }
            while (w < nload) {
                dma_issue<LAYOUT>(slab, w, stage[0][w], lane, need);
                ST(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ST(1);
                img_read<LAYOUT>(v, stage[0][w], lane);
                ++npoll;
                const unsigned long long bm = __ballot(has_sent(v));
                if (bm == 0ull) break;
                need = need_of<LAYOUT>(bm);
                if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(errors + 1, 1u); return; }
            }
        } else if (PROTO == 2) {
            dma_issue<LAYOUT>(slab, w, stage[dk & 1][w], lane, 15u); ++dk;
            while (true) {
                dma_issue<LAYOUT>(slab, w, stage[dk & 1][w], lane, 15u); ++dk;        // the next poll, already on its way
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              // the older one has landed
                img_read<LAYOUT>(v, stage[dk & 1][w], lane);                          // (dk - 2) & 1
                ++npoll;
                if (__ballot(has_sent(v)) == 0ull) break;
                if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(errors + 1, 1u); return; }
            }
        } else {
            // hint words: producer cc = 4 w + (lane >> 2), its storing wave (lane & 3) -> row 4 (lane & 3), column 16 cc
            const unsigned* hint = reinterpret_cast<const unsigned*>(LAYOUT == 0 ? slab + 4 * (lane & 3) * COLS + 16 * (4 * w + ((lane >> 2) & 3))
                                                                                : slab + (4 * w + ((lane >> 2) & 3)) * 256 + 4 * (lane & 3) * 16);
            while (true) {
                const unsigned f = lane < 16 ? ld1(hint) : 0u;
                ++npoll;
                if (__all((int)(f != SENT))) break;
                if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(errors + 1, 1u); return; }
            }
            unsigned need = 15u;
            while (true) {
                dma_issue<LAYOUT>(slab, w, stage[0][w], lane, need);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                img_read<LAYOUT>(v, stage[0][w], lane);
                const unsigned long long bm = __ballot(has_sent(v));
                if (bm == 0ull) break;
                ++npoll;
                need = need_of<LAYOUT>(bm);
                if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(errors + 1, 1u); return; }
            }
        }
        ST(2);
        float sum = 0.f;
        const int m = lane & 15, q = lane >> 4;
        if (PROTO == 1 && w >= nload) { for (int jj = 0; jj < 4; ++jj) for (int e = 0; e < 4; ++e) v[jj][e] = expect(it, m, 64 * w + 16 * q + 4 * jj + e); }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float vv[4] = {v[jj].x, v[jj].y, v[jj].z, v[jj].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (vv[e] != expect(it, m, 64 * w + 16 * q + 4 * jj + e)) ++bad;
                sum += vv[e];
            }
        }
        // the next value depends on what was loaded (a chain, like h_t on h_{t-1}); the cross-wave reduce of the scans' products
        ST(3);
        __shared__ float red[256];
        red[tid] = sum;
        __syncthreads();
        const float tot = red[(tid + 64) & 255] + red[(tid + 128) & 255];
        own = expect(it + 1, row, col) + (tot == 123456.f ? 1.f : 0.f);
        __syncthreads();
        ST(4);
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(errors, bad);
    if (lane == 0) { clk[(g * W + c) * 4 + w] = r1 - r0; polls[(g * W + c) * 4 + w] = npoll; }
    if (STAMP && g == 0 && c == 1 && tid == 0) for (int i = 0; i < 5; ++i) polls[GSTAMP + i] = sacc[i];
}

// primitives, one wave on an idle chip: [0] store -> ack, [1] dword load round trip (sc1), [2] 4 KB DMA round trip + LDS read, x 200 each
__global__ __launch_bounds__(64) void prim_kernel(float* buf, unsigned long long* out) {
    __shared__ float img[1024];
    const int lane = threadIdx.x;
    unsigned long long t0, acc[3] = {0, 0, 0};
    f32x4 v[4];
    for (int it = 0; it < 200; ++it) {
        float* slab = buf + (size_t)it * SLAB;
        t0 = __builtin_amdgcn_s_memrealtime();
        st1(slab + lane, 1.f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc[0] += __builtin_amdgcn_s_memrealtime() - t0;
        t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned f = ld1(reinterpret_cast<const unsigned*>(slab) + lane);
        acc[1] += __builtin_amdgcn_s_memrealtime() - t0;
        t0 = __builtin_amdgcn_s_memrealtime();
        dma_issue<0>(slab, 0, img, lane, 15u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        img_read<0>(v, img, lane);
        acc[2] += __builtin_amdgcn_s_memrealtime() - t0;
        if (f == 12345u && v[0].x == 7.f) st1(slab + 64, 2.f);
    }
    if (lane == 0) { out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2]; }
}

int main() {
    const int iters = 300, GMAX = 32;
    float *buf, *sink; unsigned *flags, *err; unsigned long long *clk, *polls;
    const size_t bufn = (size_t)GMAX * iters * SLAB;
    CK(hipMalloc(&buf, bufn * 4)); CK(hipMalloc(&flags, 64 * 64 * 4)); CK(hipMalloc(&err, 8));
    CK(hipMalloc(&clk, GMAX * W * 4 * 8)); CK(hipMalloc(&polls, (GMAX * W * 4 + 8) * 8));
    CK(hipMalloc(&sink, (size_t)8 * W * 4 * 4 * iters * 256 * 4));
    {
        unsigned long long* out; CK(hipMalloc(&out, 24));
        CK(hipMemset(buf, 0, bufn * 4)); CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(prim_kernel, dim3(1), dim3(64), 0, 0, buf, out); CK(hipDeviceSynchronize()); }
        unsigned long long h[3]; CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
        printf("primitives (one wave, idle chip, second pass = L2-resident lines): store -> ack %.0f ns, sc1 dword load %.0f ns, 4 KB LDS-DMA + read %.0f ns\n",
               h[0] * 10.0 / 200, h[1] * 10.0 / 200, h[2] * 10.0 / 200);
    }
    std::vector<unsigned long long> hc(GMAX * W * 4), hp(GMAX * W * 4);
    for (int layout = 0; layout < 3; ++layout)
    for (int extra : {0, 3})
        for (int proto = 0; proto < (layout == 2 ? 2 : 4); ++proto)
            for (int G : {1, 8, 32}) {
                const int grid = 8 * W * ((G + 7) / 8);
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemset(flags, 0, 64 * 64 * 4)); CK(hipMemset(err, 0, 8)); CK(hipMemset(buf, 0xFF, bufn * 4));
                    CK(hipDeviceSynchronize());
#define LAUNCH(P_, L_) hipLaunchKernelGGL((exchange_kernel<P_, L_>), dim3(grid), dim3(256), 0, 0, buf, flags, G, iters, extra, 4, sink, err, clk, polls)
                    if (layout == 2) { if (proto == 0) LAUNCH(0, 2); else LAUNCH(1, 2); }
                    else switch (proto * 2 + layout) {
                        case 0: LAUNCH(0, 0); break; case 1: LAUNCH(0, 1); break; case 2: LAUNCH(1, 0); break; case 3: LAUNCH(1, 1); break;
                        case 4: LAUNCH(2, 0); break; case 5: LAUNCH(2, 1); break; case 6: LAUNCH(3, 0); break; default: LAUNCH(3, 1); break;
                    }
                    CK(hipDeviceSynchronize());
                    unsigned e[2]; CK(hipMemcpy(e, err, 8, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hc.data(), clk, G * W * 4 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hp.data(), polls, G * W * 4 * 8, hipMemcpyDeviceToHost));
                    double ns = 0, pl = 0;
                    for (int i = 0; i < G * W * 4; ++i) { ns += hc[i] * 10.0; pl += hp[i]; }
                    ns /= (G * W * 4); pl /= (G * W * 4);
                    if (rep == 1) printf("layout %d  protocol %d  extra stores %d  G=%2d: %.3f us per exchange, %.2f polls per exchange, wrong values %u, timeouts %u\n", layout, proto, extra, G,
                                         ns / iters / 1000.0, pl / iters, e[0], e[1]);
                }
            }
    // in-band, one poll always enough (G = 1 and 8): where does the iteration go, and does it depend on how many waves load?
    for (int G : {1, 8, 32})
        for (int nload : {4, 2, 1, 0}) {
            const int grid = 8 * W * ((G + 7) / 8), extra = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(err, 0, 8)); CK(hipMemset(buf, 0xFF, bufn * 4)); CK(hipDeviceSynchronize());
                hipLaunchKernelGGL((exchange_kernel<1, 0, true>), dim3(grid), dim3(256), 0, 0, buf, flags, G, iters, extra, nload, sink, err, clk, polls);
                CK(hipDeviceSynchronize());
                unsigned e[2]; CK(hipMemcpy(e, err, 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hc.data(), clk, G * W * 4 * 8, hipMemcpyDeviceToHost));
                unsigned long long st[5]; CK(hipMemcpy(st, polls + GSTAMP, 40, hipMemcpyDeviceToHost));
                double ns = 0; for (int i = 0; i < G * W * 4; ++i) ns += hc[i] * 10.0; ns /= (G * W * 4);
                if (rep == 1) printf("in-band, G=%2d, %d loading waves per workgroup: %.3f us per exchange (stamped); store+DMA issue %.0f, vmcnt(0) %.0f, LDS read+check %.0f, "
                                     "validate %.0f, reduce+barriers %.0f ns; wrong %u timeouts %u\n", G, nload, ns / iters / 1000.0, st[0] * 10.0 / iters, st[1] * 10.0 / iters,
                                     st[2] * 10.0 / iters, st[3] * 10.0 / iters, st[4] * 10.0 / iters, e[0], e[1]);
            }
        }
    return 0;
}

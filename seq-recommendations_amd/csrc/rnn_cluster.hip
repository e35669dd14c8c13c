// Cluster form of the GRU scan: ONE launch per scan call.  The 16 session rows of a row block are owned by a GROUP of
// H/16 workgroups, one per 16 hidden columns, that stay resident for the whole scan with their slices of the recurrent
// kernel in registers and exchange h / r*h (forward) and d / [dpre_z|dpre_r] (BPTT) INSIDE the kernel:
//   producer: stores its 16 x 16 slice, drains the stores (vmcnt 0), workgroup barrier, one flag store;
//   consumer: every wave polls the group's H/16 flags with device-scope loads, then reads the rows (device scope).
// tools/xcd_l2_exchange_probe.hip prices one such exchange at ~1.1 us among 16 workgroups of one XCD (plain stores: the
// XCD's L2 is the meeting point) and ~1.8 us with write-through stores (correct at device scope wherever the workgroups
// sit), against ~2.9 us per dependent LAUNCH of the step-wise form (rnn_step.hip), which pays one per recurrent product.
// A group is dealt to one XCD by the same blockIdx -> XCD rule as the step-wise tiles; because that rule is a property
// of the dispatcher and not a guarantee, every workgroup publishes its XCC_ID with the first exchange (always write-through)
// and a group switches to plain stores only if all its members report the same XCD.
// Same arithmetic as the step-wise kernels, element for element (K split over the 4 waves, partial tiles summed in the same
// order), so the two forms agree bit for bit.  Step offsets travel in the kernel arguments (T <= CL_TMAX).
#include "common.h"
#include "rnn_cluster.h"
#include <cstdlib>
#include <map>
#include <mutex>

namespace {

constexpr int CL_TMAX = 159;
struct ClusterArgs {
    int H_real, T, n_groups, g_base;
    const float* XW; float* Hout; float* gates; float* aux;
    const float* dHout; float* dPre;
    const float* pk_a; const float* pk_b;      // packed B operands of the two products (rnn_step.hip layouts)
    unsigned* flags;                           // [groups][64]: words 0..31 phase counters, 32..63 XCC ids
    unsigned* error;                           // set when a bounded spin ran out
    unsigned epoch;
    // BPTT input gradient in parts (seqrec_dh_parts): dHout = slab 0, dh_ns slabs dh_stride floats apart, + dh_scale[q] * dh_add[dh_idx[q]]
    int dh_ns; long dh_stride; const float* dh_add; const int* dh_idx; const float* dh_scale; long dh_ld;
    int so[CL_TMAX + 1];
};

// Diagnostic build only (-DSEQREC_CLUSTER_STAMP, tools/cluster_stamps.py): workgroup (group 0, column block 1) sums the
// s_memrealtime (100 MHz) spent between marked points of a step; no stamp exists in the product build.
#ifdef SEQREC_CLUSTER_STAMP
__device__ unsigned long long g_cl_stamp[32];
#define CS_DECL unsigned long long cs_prev = __builtin_amdgcn_s_memrealtime(); unsigned long long cs_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; int cs_steps = 0
#define CS(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); cs_acc[i] += t_ - cs_prev; cs_prev = t_; } while (0)
#define CS_FLUSH(base_) do { if (gl == 0 && c == 1 && threadIdx.x == 0) { for (int i_ = 0; i_ < 12; ++i_) g_cl_stamp[(base_) + i_] = cs_acc[i_]; g_cl_stamp[(base_) + 12] = cs_steps; } } while (0)
#else
#define CS_DECL
#define CS(i)
#define CS_FLUSH(base_)
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_f32(float* p, float v, bool wt) {          // wt: write-through (device scope)
    if (wt) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_u32(unsigned* p, unsigned v, bool wt) {
    if (wt) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned ld_u32_dev(const unsigned* p) {            // device-scope load (bypasses the CU's L1)
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// The A rows of a product, coalesced: a wave needs rows 0..15 x its K slice [w*K/4, +K/4) -- 16 pieces of K bytes.  In MFMA
// operand order lane (row, q) reads K/16 consecutive floats of its row: every dwordx4 instruction then touches 16 rows x 4
// separate 16-byte pieces (64 requests, 32 half-used lines; the stamped build: 0.55-0.67 us per row load).  Here each row's
// slice is read by CONSECUTIVE lanes (full lines) straight into LDS by LDS-DMA (device scope), 16-byte chunk c of row m
// landing at chunk c ^ (m mod chunks) (the swizzle is applied on the source address: an LDS-DMA image is lane-linear), and
// the wave re-reads its own image in operand order, conflict-free.  Same values in the same registers as the direct form.
template <int K> __device__ __forceinline__ void ld_rows_dma(float (&a)[K / 16], const float* base, long row_stride, int nact,
                                                              int kslice0, float* lds_wave, int lane) {
    constexpr int SL = K / 4;                      // floats of a row's slice (64 at K = 256)
    constexpr int LPR = SL / 4;                    // 16-byte chunks per row slice = lanes per row: 16 at K = 256
    constexpr int RPI = 64 / LPR;                  // rows per DMA instruction: 4 at K = 256
    constexpr int NI = 16 / RPI;                   // DMA instructions: 4 at K = 256
    static_assert(LPR <= 64 && LPR >= 4, "slice fits a wave instruction");
    const int wl = __builtin_amdgcn_readfirstlane(0);
    (void)wl;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int rl = RPI * i + lane / LPR;                       // image row
        const int ch = (lane % LPR) ^ (rl % LPR);                  // source chunk that lands at position lane % LPR
        const float* p = base + (long)min(rl, nact - 1) * row_stride + kslice0 + 4 * ch;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(lds_wave + i * (RPI * SL)), 16, 0, 16 /* sc1 */);
    }
    const int m = lane & 15, q = lane >> 4;
    const float* src = lds_wave + m * SL;
#pragma unroll
    for (int j = 0; j < K / 64; ++j) {
        const int pos = (q * (K / 64) + j) ^ (m % LPR);
        const float4 t = *reinterpret_cast<const float4*>(src + 4 * pos);
        a[4 * j] = t.x; a[4 * j + 1] = t.y; a[4 * j + 2] = t.z; a[4 * j + 3] = t.w;
    }
}
// (A scalar-load poll -- s_load_dwordx16 glc, so that the poll leaves the vector-memory counter alone -- was measured at
// 2.6-3.0 us per wait against 0.35 us for the vector poll below: profiles/r02_v3_cluster_step_stamps.txt.)
// producer side with NB younger inline-asm stores allowed to stay in flight (they were issued AFTER the exchange stores;
// the counter is in order, so vmcnt(NB) says the exchange stores -- and everything older -- are done)
template <int NB> __device__ __forceinline__ void cl_publish_n(unsigned* myflag, unsigned value, bool wt) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NB) : "memory");
    __syncthreads();
    if (threadIdx.x == 0) st_u32(myflag, value, wt);
}

// producer side of an exchange: my stores are in L2 / memory, then the flag
__device__ __forceinline__ void cl_publish(unsigned* myflag, unsigned value, bool wt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) st_u32(myflag, value, wt);
}
// consumer side: every member's counter has reached `target` (wrap-safe); false = the bounded spin ran out.  Every wave
// polls for itself: no workgroup barrier between the flags and the wave's own row loads
#ifdef SEQREC_CLUSTER_SPINS          // diagnostic build (tools/cluster_spins.py): [0] waits, [1 + min(polls - 1, 6)] histogram of polls per wait
__device__ unsigned long long g_cl_spins[8];
#endif
template <int CB> __device__ __forceinline__ bool cl_wait_w(const unsigned* fl, unsigned target, unsigned* error) {
    const int lane = threadIdx.x & 63;
    int spins = 0;
    while (true) {
        const unsigned f = lane < CB ? ld_u32_dev(fl + lane) : target;
#ifdef SEQREC_CLUSTER_SPINS
        if (lane == 0 && __all((int)(f - target) >= 0)) { atomicAdd(&g_cl_spins[0], 1ull); atomicAdd(&g_cl_spins[1 + min(spins, 6)], 1ull); }
#endif
        if (__all((int)(f - target) >= 0)) return true;
        if (++spins > (1 << 22)) { if (lane == 0) atomicAdd(error, 1u); return false; }
    }
}
template <int CB> __device__ __forceinline__ bool cl_same_xcd(const unsigned* fl) {
    const unsigned mine = ld_u32_dev(fl + 32);
    bool same = true;
    for (int i = 1; i < CB; ++i) same = same && (ld_u32_dev(fl + 32 + i) == mine);
    return same;
}

// one or two 16x16 tile products with K split over the 4 waves (rnn_step.hip tile_16x16_reg, same order of sums)
template <int K, int NT>
__device__ __forceinline__ void cl_tiles(const float (&a)[K / 16], const float4 (&b0)[K / 64], const float4 (&b1)[K / 64],
                                         float* __restrict__ red, int tid, float& out0, float& out1) {
    const int lane = tid & 63, w = tid >> 6;
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f}, q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < K / 64; ++i) {
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 0], b0[i].x, p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 1], b0[i].y, p1, 0, 0, 0);
        if (NT == 2) {
            q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 0], b1[i].x, q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 1], b1[i].y, q1, 0, 0, 0);
        }
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 2], b0[i].z, p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 3], b0[i].w, p1, 0, 0, 0);
        if (NT == 2) {
            q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 2], b1[i].z, q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 3], b1[i].w, q1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[w * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = p0[r] + p1[r];
        if (NT == 2) red[1024 + w * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = q0[r] + q1[r];
    }
    __syncthreads();
    out0 = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
    if (NT == 2) out1 = (red[1024 + tid] + red[1280 + tid]) + (red[1536 + tid] + red[1792 + tid]);
}

// ------------------------------------------------------------------------------------------------------------------
// forward: per step  [z|r] = hs(xw + h_prev.U_zr) -> r*h_prev (exchange 1) -> h~ = act(xw_h + (r*h_prev).U_h),
//          h = z h_prev + (1-z) h~ (exchange 2).  Thread (row, col) keeps its h element in a register across steps.
// ------------------------------------------------------------------------------------------------------------------
template <int J, int ACT>
__global__ __launch_bounds__(256) void gru_cluster_fwd(ClusterArgs a) {
    constexpr int H = 64 * J, GH = 3 * H, CB = H / 16, NB = H / 64;
    const int L = blockIdx.x, x = L & 7, s = L >> 3, jj = s / CB, c = s - jj * CB;
    const int gl = x + 8 * jj;                       // group index inside this launch
    const int r0 = 16 * (a.g_base + gl);
    if (gl >= a.n_groups || r0 >= a.so[1] - a.so[0]) return;
    __shared__ float smem[1024 + 4 * 16 * (H / 4)];       // partial tiles of a product + the 4 waves' A-row images
    float* red = smem;
    float* stage = smem + 1024;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = tid >> 4, col = 16 * c + (tid & 15);
    float4 bz[NB], br[NB], bh[NB];
    {
        const float4* pa = reinterpret_cast<const float4*>(a.pk_a);
        const float4* pb = reinterpret_cast<const float4*>(a.pk_b);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            bz[i] = pa[((size_t)(c * 4 + w) * NB + i) * 64 + lane];
            br[i] = pa[((size_t)((CB + c) * 4 + w) * NB + i) * 64 + lane];
            bh[i] = pb[((size_t)(c * 4 + w) * NB + i) * 64 + lane];
        }
    }
    unsigned* fl = a.flags + (size_t)gl * 64;
    const unsigned base = a.epoch;
    if (tid == 0) st_u32(fl + 32 + c, (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF) + 1u, true);
    bool wt = true;                                   // write-through exchange stores until the group is known to share an XCD
    float hprev = 0.f;
    CS_DECL;
    // input projections of a step are requested one step ahead (after the flag store of the step before: nothing on the
    // exchange path waits for them)
    float n_xz = 0.f, n_xr = 0.f, n_xh = 0.f;
    auto prefetch_xw = [&](int t) {
        const int p0 = a.so[t], nact = min(16, a.so[t + 1] - p0 - r0);
        if (row < nact) { const float* xw = a.XW + ((long)p0 + r0 + row) * GH + col; n_xz = xw[0]; n_xr = xw[H]; n_xh = xw[2 * H]; }
    };
    prefetch_xw(0);
    for (int t = 0; t < a.T; ++t) {
        const int p0 = a.so[t], bt = a.so[t + 1] - p0;
        if (bt <= r0) break;
        const int nact = min(16, bt - r0);
        const bool ok = row < nact;
        const long q = (long)p0 + r0 + row;
        const bool more = t + 1 < a.T && a.so[t + 2] - a.so[t + 1] > r0;
        const float xz = n_xz, xr = n_xr, xh = n_xh;
        float accz = 0.f, accr = 0.f, acch = 0.f, dummy;
        float av[H / 16];
        CS(0);
        if (t > 0) {
            if (!cl_wait_w<CB>(fl, base + 2u * t, a.error)) return;
            CS(1);
            if (t == 1) wt = !cl_same_xcd<CB>(fl);
            ld_rows_dma<H>(av, a.Hout + ((long)a.so[t - 1] + r0) * H, H, nact, w * (H / 4), stage + w * (16 * H / 4), lane);
            CS(2);
            cl_tiles<H, 1>(av, br, br, red, tid, accr, dummy);        // r first: r * h_prev is what the others wait for
            CS(3);
        }
        const float r = hard_sigmoid(accr + xr);
        if (ok) st_f32(a.aux + q * H + col, r * hprev, wt);
        if (t > 0) {
            cl_publish_n<0>(fl + c, base + 2u * t + 1u, wt);
            CS(4);
            cl_tiles<H, 1>(av, bz, bz, red, tid, accz, dummy);        // z under the exchange (only the h update needs it)
            CS(5);
        }
        const float z = hard_sigmoid(accz + xz);
        if (t > 0) {
            if (!cl_wait_w<CB>(fl, base + 2u * t + 1u, a.error)) return;
            CS(6);
            ld_rows_dma<H>(av, a.aux + ((long)p0 + r0) * H, H, nact, w * (H / 4), stage + w * (16 * H / 4), lane);
            CS(7);
            __syncthreads();                                          // red: slower waves may still read the z product
            cl_tiles<H, 1>(av, bh, bh, red, tid, acch, dummy);
            CS(8);
        }
        const float hh = act_fwd<ACT>(acch + xh);
        float hn = z * hprev + (1.f - z) * hh;
        if (col >= a.H_real) hn = 0.f;
        if (ok) {
            st_f32(a.Hout + q * H + col, hn, wt);
            // the stash (z, r, h~ for the BPTT) rides behind the exchange store: in flight under the flag and the next poll
            st_f32(a.gates + q * GH + col, z, false);
            st_f32(a.gates + q * GH + H + col, r, false);
            st_f32(a.gates + q * GH + 2 * H + col, hh, false);
        }
        hprev = hn;
        if (more) {
            cl_publish_n<3>(fl + c, base + 2u * t + 2u, wt);
            prefetch_xw(t + 1);
        }
        CS(9);
#ifdef SEQREC_CLUSTER_STAMP
        if (t > 0) ++cs_steps;
#endif
    }
    CS_FLUSH(0);
}

// ------------------------------------------------------------------------------------------------------------------
// BPTT: per step (descending)  d = dh (1-z) act'(h~)  (exchange A)  ->  drh = d.U_h^T;  dpre_z, dpre_r  (exchange B)  ->
//       dh_prev = dh z + drh r + [dpre_z|dpre_r].U_zr^T, carried to step t-1 in a register of thread (row, col).
// Step 0 has no recurrent product and no exchange (h_prev = 0).
// ------------------------------------------------------------------------------------------------------------------
template <int J, int ACT>
__global__ __launch_bounds__(256) void gru_cluster_bwd(ClusterArgs a) {
    constexpr int H = 64 * J, GH = 3 * H, CB = H / 16, NB = H / 64;
    const int L = blockIdx.x, x = L & 7, s = L >> 3, jj = s / CB, c = s - jj * CB;
    const int gl = x + 8 * jj;
    const int r0 = 16 * (a.g_base + gl);
    if (gl >= a.n_groups || r0 >= a.so[1] - a.so[0]) return;
    __shared__ float smem[1024 + 4 * 16 * (2 * H / 4)];   // partial tiles of a product + the 4 waves' A-row images (K up to 2H)
    float* red = smem;
    float* stage = smem + 1024;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = tid >> 4, col = 16 * c + (tid & 15);
    float4 bh[NB], bzr[2 * NB];
    {
        const float4* pa = reinterpret_cast<const float4*>(a.pk_a);      // U_h^T, K = H
        const float4* pb = reinterpret_cast<const float4*>(a.pk_b);      // [U_z U_r]^T, K = 2H
#pragma unroll
        for (int i = 0; i < NB; ++i) bh[i] = pa[((size_t)(c * 4 + w) * NB + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) bzr[i] = pb[((size_t)(c * 4 + w) * (2 * NB) + i) * 64 + lane];
    }
    unsigned* fl = a.flags + (size_t)gl * 64;
    unsigned count = a.epoch;                         // this workgroup's published exchanges so far (same sequence in every member)
    if (tid == 0) st_u32(fl + 32 + c, (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF) + 1u, true);
    bool wt = true, first_x = true;
    int tg = 0;                                       // steps this row block is alive
    while (tg < a.T && a.so[tg + 1] - a.so[tg] > r0) ++tg;
    float carry = 0.f;
    // the element-wise operands of a step (dHout, z, r, h~, h_prev) are requested one step ahead: they do not depend on
    // the exchange, so their latency hides under the previous step's waits
    float n_dh = 0.f, n_z = 0.f, n_r = 0.f, n_hh = 0.f, n_h0 = 0.f;
    auto prefetch = [&](int t) {
        const int p0 = a.so[t], nact = min(16, a.so[t + 1] - p0 - r0);
        const long q = (long)p0 + r0 + (row < nact ? row : 0);
        float v = a.dHout[q * H + col];
        for (int sl = 1; sl < a.dh_ns; ++sl) v += a.dHout[(long)sl * a.dh_stride + q * H + col];      // slab order == the reduce launch
        if (a.dh_add) {
            const int ix = a.dh_idx[q];
            v += ix < 0 ? 0.f : (a.dh_scale ? a.dh_scale[q] : 1.f) * a.dh_add[(long)ix * a.dh_ld + col];
        }
        n_dh = v;
        n_z = a.gates[q * GH + col]; n_r = a.gates[q * GH + H + col]; n_hh = a.gates[q * GH + 2 * H + col];
        n_h0 = t > 0 ? a.Hout[((long)a.so[t - 1] + r0 + (row < nact ? row : 0)) * H + col] : 0.f;
    };
    if (tg > 0) prefetch(tg - 1);
    for (int t = tg - 1; t >= 0; --t) {
        const int p0 = a.so[t], bt = a.so[t + 1] - p0;
        const int bnext = t + 1 < a.T ? a.so[t + 2] - a.so[t + 1] : 0;
        const int nact = min(16, bt - r0);
        const bool ok = row < nact;
        const long q = (long)p0 + r0 + (ok ? row : 0);
        float dh = n_dh;
        if (r0 + row < bnext) dh += carry;
        const float z = n_z, r = n_r, hh = n_hh, h0 = n_h0;
        const float d = dh * (1.f - z) * act_grad<ACT>(hh);
        if (t == 0) {
            if (ok) {
                a.dPre[q * GH + col] = dh * (0.f - hh) * hard_sigmoid_grad(z);
                a.dPre[q * GH + H + col] = 0.f;
                a.dPre[q * GH + 2 * H + col] = d;
            }
            break;
        }
        if (ok) st_f32(a.dPre + q * GH + 2 * H + col, d, wt);
        cl_publish_n<0>(fl + c, ++count, wt);
        prefetch(t - 1);                              // behind the flag store: nothing on the exchange path is issued after it
        if (!cl_wait_w<CB>(fl, count, a.error)) return;
        if (first_x) { wt = !cl_same_xcd<CB>(fl); first_x = false; }
        float acc = 0.f, dummy;
        {
            float av[H / 16];
            ld_rows_dma<H>(av, a.dPre + ((long)p0 + r0) * GH + 2 * H, GH, nact, w * (H / 4), stage + w * (16 * 2 * H / 4), lane);
            cl_tiles<H, 1>(av, bh, bh, red, tid, acc, dummy);
        }
        const float dcar = dh * z + acc * r;
        if (ok) {
            st_f32(a.dPre + q * GH + col, dh * (h0 - hh) * hard_sigmoid_grad(z), wt);
            st_f32(a.dPre + q * GH + H + col, acc * h0 * hard_sigmoid_grad(r), wt);
        }
        cl_publish_n<0>(fl + c, ++count, wt);
        if (!cl_wait_w<CB>(fl, count, a.error)) return;
        float acc2 = 0.f;
        {
            float av2[2 * H / 16];
            ld_rows_dma<2 * H>(av2, a.dPre + ((long)p0 + r0) * GH, GH, nact, w * (2 * H / 4), stage + w * (16 * 2 * H / 4), lane);
            cl_tiles<2 * H, 1>(av2, bzr, bzr, red, tid, acc2, dummy);
        }
        carry = dcar + acc2;
    }
}

// per-stream flag buffers + epochs
struct FlagBuf { unsigned* flags; unsigned* error; unsigned epoch; };
std::map<hipStream_t, FlagBuf> g_flagbufs;
std::mutex g_flag_mu;
constexpr int CL_MAX_GROUPS = 64;

int get_flagbuf(hipStream_t st, int T, FlagBuf& out) {
    std::lock_guard<std::mutex> lk(g_flag_mu);
    auto it = g_flagbufs.find(st);
    if (it == g_flagbufs.end()) {
        FlagBuf fb{};
        hipError_t e = hipMalloc(&fb.flags, (CL_MAX_GROUPS * 64 + 64) * sizeof(unsigned));
        if (e != hipSuccess) return (int)e;
        e = hipMemset(fb.flags, 0, (CL_MAX_GROUPS * 64 + 64) * sizeof(unsigned));      // synchronous, once per stream
        if (e != hipSuccess) return (int)e;
        fb.error = fb.flags + CL_MAX_GROUPS * 64;
        fb.epoch = 16;
        it = g_flagbufs.emplace(st, fb).first;
    }
    FlagBuf& fb = it->second;
    if (fb.epoch > 0x70000000u) {                                                       // far from wrapping: start over
        const hipError_t e = hipMemsetAsync(fb.flags, 0, CL_MAX_GROUPS * 64 * sizeof(unsigned), st);
        if (e != hipSuccess) return (int)e;
        fb.epoch = 16;
    }
    out = fb;
    fb.epoch += 2u * (unsigned)T + 8u;
    return 0;
}

int g_cluster_override = -1;                  // seqrec_debug_scan_cluster(): tests compare the two forms in one process
bool cluster_enabled() {
    static const bool on = !(getenv("SEQREC_SCAN_CLUSTER") && atoi(getenv("SEQREC_SCAN_CLUSTER")) == 0);      // A/B switch
    return g_cluster_override >= 0 ? g_cluster_override != 0 : on;
}

template <int ACT> const void* fwd_kernel(int J) {
    switch (J) {
        case 1: return reinterpret_cast<const void*>(gru_cluster_fwd<1, ACT>);
        case 2: return reinterpret_cast<const void*>(gru_cluster_fwd<2, ACT>);
        case 4: return reinterpret_cast<const void*>(gru_cluster_fwd<4, ACT>);
        case 8: return reinterpret_cast<const void*>(gru_cluster_fwd<8, ACT>);
    }
    return nullptr;
}

}  // namespace

extern "C" void seqrec_debug_scan_cluster(int mode) { g_cluster_override = mode; }
#ifdef SEQREC_CLUSTER_SPINS
extern "C" void seqrec_debug_cluster_spins(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cl_spins), z, sizeof(z)); return; }
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cl_spins), sizeof(unsigned long long) * 8);
}
#endif
#ifdef SEQREC_CLUSTER_STAMP
extern "C" void seqrec_debug_cluster_stamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cl_stamp), sizeof(unsigned long long) * 32);
}
#endif

// Bounded spins that ran out since the stream's first cluster scan (0 in a healthy run): synchronises the stream.
extern "C" int seqrec_cluster_scan_errors(void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    unsigned* dev = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_flag_mu);
        auto it = g_flagbufs.find(st);
        if (it == g_flagbufs.end()) return 0;
        dev = it->second.error;
    }
    unsigned h = 0;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    if (hipMemcpy(&h, dev, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)(h > 0x7FFFFFFFu ? 0x7FFFFFFFu : h);
}

bool seqrec_cluster_gru_fwd(int act, int H, int H_real, int T, const int32_t* soh, const float* XW, float* Hout, float* gates,
                            float* aux, const float* upack, hipStream_t st, int* rc) {
    if (!cluster_enabled() || T > CL_TMAX || T < 1) return false;
    const int J = H / 64, CB = H / 16;
    const void* fn = act == 0 ? fwd_kernel<0>(J) : act == 1 ? fwd_kernel<1>(J) : fwd_kernel<2>(J);
    if (!fn) return false;
    const int B0 = soh[1] - soh[0];
    if (B0 <= 0) { *rc = 0; return true; }
    const int G = (B0 + 15) / 16;
    // residency: every workgroup of a launch must be able to be resident at once (2 per CU); larger batches go in slices of
    // row blocks (independent chains) on the same stream
    int gcap = (512 / CB) & ~7;
    if (gcap < 8) gcap = 8;
    if (gcap > CL_MAX_GROUPS) gcap = CL_MAX_GROUPS;
    ClusterArgs a = {};
    a.H_real = H_real; a.T = T; a.XW = XW; a.Hout = Hout; a.gates = gates; a.aux = aux;
    a.pk_a = upack; a.pk_b = upack + 2l * H * H;
    for (int t = 0; t <= T; ++t) a.so[t] = soh[t];
    for (int g0 = 0; g0 < G; g0 += gcap) {
        FlagBuf fb;
        if ((*rc = get_flagbuf(st, T, fb))) return true;
        a.flags = fb.flags; a.error = fb.error; a.epoch = fb.epoch;
        a.g_base = g0; a.n_groups = G - g0 < gcap ? G - g0 : gcap;
        const unsigned grid = 8u * CB * ((a.n_groups + 7) / 8);
        void* argv[1] = {&a};
        const hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(256), argv, 0, st);
        if (e != hipSuccess) { *rc = (int)e; return true; }
    }
    *rc = 0;
    return true;
}

namespace {
template <int ACT> const void* bwd_kernel(int J) {
    switch (J) {
        case 1: return reinterpret_cast<const void*>(gru_cluster_bwd<1, ACT>);
        case 2: return reinterpret_cast<const void*>(gru_cluster_bwd<2, ACT>);
        case 4: return reinterpret_cast<const void*>(gru_cluster_bwd<4, ACT>);
        case 8: return reinterpret_cast<const void*>(gru_cluster_bwd<8, ACT>);
    }
    return nullptr;
}
}  // namespace

bool seqrec_cluster_gru_bwd(int act, int H, int H_real, int T, const int32_t* soh, const float* dHout, const float* Hout,
                            const float* gates, const float* aux, float* dPre, const float* upack, hipStream_t st, int* rc,
                            const seqrec_dh_parts* parts) {
    (void)aux;
    if (!cluster_enabled() || T > CL_TMAX || T < 1) return false;
    const int J = H / 64, CB = H / 16;
    const void* fn = act == 0 ? bwd_kernel<0>(J) : act == 1 ? bwd_kernel<1>(J) : bwd_kernel<2>(J);
    if (!fn) return false;
    const int B0 = soh[1] - soh[0];
    if (B0 <= 0) { *rc = 0; return true; }
    const int G = (B0 + 15) / 16;
    int gcap = (512 / CB) & ~7;
    if (gcap < 8) gcap = 8;
    if (gcap > CL_MAX_GROUPS) gcap = CL_MAX_GROUPS;
    ClusterArgs a = {};
    a.H_real = H_real; a.T = T; a.dHout = dHout; a.Hout = const_cast<float*>(Hout); a.gates = const_cast<float*>(gates); a.dPre = dPre;
    if (parts) {
        a.dHout = parts->slabs; a.dh_ns = parts->n_slabs; a.dh_stride = (long)parts->slab_stride;
        a.dh_add = parts->add_table; a.dh_idx = parts->add_index; a.dh_scale = parts->add_scale; a.dh_ld = (long)parts->add_ld;
    }
    a.pk_a = upack + 3l * H * H; a.pk_b = upack + 4l * H * H;
    for (int t = 0; t <= T; ++t) a.so[t] = soh[t];
    for (int g0 = 0; g0 < G; g0 += gcap) {
        FlagBuf fb;
        if ((*rc = get_flagbuf(st, T, fb))) return true;
        a.flags = fb.flags; a.error = fb.error; a.epoch = fb.epoch;
        a.g_base = g0; a.n_groups = G - g0 < gcap ? G - g0 : gcap;
        const unsigned grid = 8u * CB * ((a.n_groups + 7) / 8);
        void* argv[1] = {&a};
        const hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(256), argv, 0, st);
        if (e != hipSuccess) { *rc = (int)e; return true; }
    }
    *rc = 0;
    return true;
}

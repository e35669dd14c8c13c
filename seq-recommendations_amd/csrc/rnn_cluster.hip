// Cluster form of the GRU scan: ONE launch per scan call.  The 16 session rows of a row block are owned by a GROUP of
// H/16 workgroups, one per 16 hidden columns, that stay resident for the whole scan with their slices of the recurrent
// kernel in registers and exchange h / r*h (forward) and d / [dpre_z|dpre_r] (BPTT) INSIDE the kernel:
//   producer: stores its 16 x 16 slice, drains the stores (vmcnt 0), workgroup barrier, one flag store;
//   consumer: wave 0 polls the group's H/16 flags with device-scope loads, workgroup barrier, device-scope loads of the rows.
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
    int so[CL_TMAX + 1];
};

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
// N consecutive floats with device-scope loads; the caller waits (cl_wait_loads) before the first use
template <int N> __device__ __forceinline__ void ld_vec_dev(float (&a)[N], const float* p) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        f32x4v v;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p + 4 * i) : "memory");
        a[4 * i + 0] = v[0]; a[4 * i + 1] = v[1]; a[4 * i + 2] = v[2]; a[4 * i + 3] = v[3];
    }
}
__device__ __forceinline__ void cl_wait_loads() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int K> __device__ __forceinline__ int cl_koff(int lane, int w) { return w * (K / 4) + (lane >> 4) * (K / 16); }

// producer side of an exchange: my stores are in L2 / memory, then the flag
__device__ __forceinline__ void cl_publish(unsigned* myflag, unsigned value, bool wt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) st_u32(myflag, value, wt);
}
// consumer side: every member's counter has reached `target` (wrap-safe); false = the bounded spin ran out
template <int CB> __device__ __forceinline__ bool cl_wait(const unsigned* fl, unsigned target, int* ok_s, unsigned* error) {
    if (threadIdx.x < 64) {
        int good = 1, spins = 0;
        while (true) {
            const unsigned f = (int)threadIdx.x < CB ? ld_u32_dev(fl + threadIdx.x) : target;
            if (__all((int)(f - target) >= 0)) break;
            if (++spins > (1 << 22)) { good = 0; break; }
        }
        if (threadIdx.x == 0) { *ok_s = good; if (!good) atomicAdd(error, 1u); }
    }
    __syncthreads();
    return *ok_s != 0;
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
    __shared__ float red[2 * 1024];
    __shared__ int ok_s;
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
    const int koff = cl_koff<H>(lane, w);
    for (int t = 0; t < a.T; ++t) {
        const int p0 = a.so[t], bt = a.so[t + 1] - p0;
        if (bt <= r0) break;
        const int nact = min(16, bt - r0);
        const bool ok = row < nact;
        const long q = (long)p0 + r0 + row;
        const int arow = min(lane & 15, nact - 1);
        float xz = 0.f, xr = 0.f, xh = 0.f;
        if (ok) { const float* xw = a.XW + q * GH + col; xz = xw[0]; xr = xw[H]; xh = xw[2 * H]; }
        float accz = 0.f, accr = 0.f, acch = 0.f;
        float av[H / 16];
        if (t > 0) {
            if (!cl_wait<CB>(fl, base + 2u * t, &ok_s, a.error)) return;
            if (t == 1) wt = !cl_same_xcd<CB>(fl);
            ld_vec_dev(av, a.Hout + ((long)a.so[t - 1] + r0 + arow) * H + koff);
            cl_wait_loads();
            cl_tiles<H, 2>(av, bz, br, red, tid, accz, accr);
        }
        const float z = hard_sigmoid(accz + xz), r = hard_sigmoid(accr + xr);
        if (ok) {
            a.gates[q * GH + col] = z;
            a.gates[q * GH + H + col] = r;
            st_f32(a.aux + q * H + col, r * hprev, wt);
        }
        if (t > 0) {
            cl_publish(fl + c, base + 2u * t + 1u, wt);
            if (!cl_wait<CB>(fl, base + 2u * t + 1u, &ok_s, a.error)) return;
            ld_vec_dev(av, a.aux + ((long)p0 + r0 + arow) * H + koff);
            cl_wait_loads();
            float dummy;
            cl_tiles<H, 1>(av, bh, bh, red, tid, acch, dummy);
        }
        const float hh = act_fwd<ACT>(acch + xh);
        float hn = z * hprev + (1.f - z) * hh;
        if (col >= a.H_real) hn = 0.f;
        if (ok) {
            st_f32(a.Hout + q * H + col, hn, wt);
            a.gates[q * GH + 2 * H + col] = hh;
        }
        hprev = hn;
        if (t + 1 < a.T && a.so[t + 2] - a.so[t + 1] > r0) cl_publish(fl + c, base + 2u * t + 2u, wt);
    }
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
    __shared__ float red[1024];
    __shared__ int ok_s;
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
    const int koff1 = cl_koff<H>(lane, w), koff2 = cl_koff<2 * H>(lane, w);
    // the element-wise operands of a step (dHout, z, r, h~, h_prev) are requested one step ahead: they do not depend on
    // the exchange, so their latency hides under the previous step's waits
    float n_dh = 0.f, n_z = 0.f, n_r = 0.f, n_hh = 0.f, n_h0 = 0.f;
    auto prefetch = [&](int t) {
        const int p0 = a.so[t], nact = min(16, a.so[t + 1] - p0 - r0);
        const long q = (long)p0 + r0 + (row < nact ? row : 0);
        n_dh = a.dHout[q * H + col];
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
        const int arow = min(lane & 15, nact - 1);
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
        prefetch(t - 1);
        cl_publish(fl + c, ++count, wt);
        if (!cl_wait<CB>(fl, count, &ok_s, a.error)) return;
        if (first_x) { wt = !cl_same_xcd<CB>(fl); first_x = false; }
        float acc = 0.f, dummy;
        {
            float av[H / 16];
            ld_vec_dev(av, a.dPre + ((long)p0 + r0 + arow) * GH + 2 * H + koff1);
            cl_wait_loads();
            cl_tiles<H, 1>(av, bh, bh, red, tid, acc, dummy);
        }
        const float dcar = dh * z + acc * r;
        if (ok) {
            st_f32(a.dPre + q * GH + col, dh * (h0 - hh) * hard_sigmoid_grad(z), wt);
            st_f32(a.dPre + q * GH + H + col, acc * h0 * hard_sigmoid_grad(r), wt);
        }
        cl_publish(fl + c, ++count, wt);
        if (!cl_wait<CB>(fl, count, &ok_s, a.error)) return;
        float acc2 = 0.f;
        {
            float av2[2 * H / 16];
            ld_vec_dev(av2, a.dPre + ((long)p0 + r0 + arow) * GH + koff2);
            cl_wait_loads();
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
                            const float* gates, const float* aux, float* dPre, const float* upack, hipStream_t st, int* rc) {
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

// Cluster form of the recurrent scans (GRU here; LSTM and SimpleRNN in rnn_cluster2.hip): ONE launch per scan call.  The 16 session rows of a row block are owned by a GROUP of
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
#include "rnn_cluster_dev.h"
#include <cstdlib>
#include <map>
#include <mutex>

using namespace seqrec_cluster;

#ifdef SEQREC_CLUSTER_SPINS
__device__ unsigned long long seqrec_cluster::g_cl_spins[8];
#endif

namespace {

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
    cl_red_store(red + w * 256, lane, p0, p1);
    if (NT == 2) cl_red_store(red + 1024 + w * 256, lane, q0, q1);
    __syncthreads();
    const float* rt = red + cl_red_r(tid);
    out0 = (rt[0] + rt[256]) + (rt[512] + rt[768]);
    if (NT == 2) out1 = (rt[1024] + rt[1280]) + (rt[1536] + rt[1792]);
}
template <int N> __device__ __forceinline__ void mul_vec(float (&o)[N], const float (&a)[N], const float (&m)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = a[i] * m[i];
}

// ------------------------------------------------------------------------------------------------------------------
// forward: per step  [z|r] = hs(xw + h_prev.U_zr) -> r*h_prev (exchange 1) -> h~ = act(xw_h + (r*h_prev).U_h),
//          h = z h_prev + (1-z) h~ (exchange 2).  Thread (row, col) keeps its h element in a register across steps.
// ------------------------------------------------------------------------------------------------------------------
template <int J, int ACT, bool RD>
__global__ __launch_bounds__(256) void gru_cluster_fwd(ClusterArgs a) {
    constexpr int H = 64 * J, GH = 3 * H, CB = H / 16, NB = H / 64;
    const int L = blockIdx.x, x = L & 7, s = L >> 3, jj = s / CB, c = s - jj * CB;
    const int gl = x + 8 * jj;                       // group index inside this launch
    const int r0 = 16 * (a.g_base + gl);
    if (gl >= a.n_groups || r0 >= a.so[1] - a.so[0]) return;
    __shared__ __attribute__((aligned(16))) float smem[1024 + 4 * 16 * (H / 4)];       // partial tiles of a product + the 4 waves' A-row images
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
    // recurrent dropout (Keras recurrent_dropout: one mask per gate and session, fixed over time): the lane's multipliers of
    // its A elements stay in registers for the whole scan (rnn_step.hip mask_vec: same products)
    [[maybe_unused]] float mz[RD ? H / 16 : 1], mr[RD ? H / 16 : 1], mh[RD ? H / 16 : 1];
    if constexpr (RD) {
        const int srow = min(r0 + (lane & 15), a.B - 1), koff = w * (H / 4) + (lane >> 4) * (H / 16);
        ld_mask(mz, a.rmask, a.B, H, 0, srow, koff);
        ld_mask(mr, a.rmask, a.B, H, 1, srow, koff);
        ld_mask(mh, a.rmask, a.B, H, 2, srow, koff);
    }
    unsigned* fl = a.flags + (size_t)gl * 64;
    const unsigned base = a.epoch;
    if (tid == 0) st_u32(fl + 32 + c, xcc_id() + 1u, true);
    bool wt = true;                                   // write-through exchange stores until the group is known to share an XCD
    float hprev = 0.f;
    CS_DECL;
    // input projections of a step are requested one step ahead (after the flag store of the step before: nothing on the
    // exchange path waits for them)
    float n_xz = 0.f, n_xr = 0.f, n_xh = 0.f;
    auto prefetch_xw = [&](int p0, int p1) {                      // the step whose tokens are [p0, p1)
        const int nact = min(16, p1 - p0 - r0);
        if (row < nact) { const float* xw = a.XW + ((long)p0 + r0 + row) * GH + col; n_xz = xw[0]; n_xr = xw[H]; n_xh = xw[2 * H]; }
    };
    StepWindow sw;                                                // step offsets in registers (rnn_cluster_dev.h)
    sw.init_up(a.so, a.T);
    prefetch_xw(sw.s0, sw.s1);
    for (int t = 0; t < a.T; ++t) {
        const int p0 = sw.s0, bt = sw.s1 - p0;
        if (bt <= r0) break;
        sw.request_up(a.so, a.T, t);
        const int nact = min(16, bt - r0);
        const bool ok = row < nact;
        const long q = (long)p0 + r0 + row;
        const bool more = sw.s2 - sw.s1 > r0;
        const float xz = n_xz, xr = n_xr, xh = n_xh;
        float accz = 0.f, accr = 0.f, acch = 0.f, dummy;
        float av[H / 16];
        [[maybe_unused]] float am[RD ? H / 16 : 1];
        CS(0);
        if (t > 0) {
            // a wait that runs out: the step's output is poisoned (nothing downstream may look plausible) and the wave leaves
            if (!cl_wait_w<CB>(fl, base + 2u * t, a.error, a.spin_limit)) { if (ok) a.Hout[q * H + col] = __builtin_nanf(""); return; }
            CS(1);
            if (t == 1) wt = !cl_same_xcd<CB>(fl);
            ld_rows_dma<H>(av, a.Hout + ((long)sw.prev + r0) * H, H, nact, w * (H / 4), stage + w * (16 * H / 4), lane);
            CS(2);
            if constexpr (RD) { mul_vec(am, av, mr); cl_tiles<H, 1>(am, br, br, red, tid, accr, dummy); }
            else cl_tiles<H, 1>(av, br, br, red, tid, accr, dummy);        // r first: r * h_prev is what the others wait for
            CS(3);
        }
        const float r = hard_sigmoid(accr + xr);
        if (ok) st_f32(a.aux + q * H + col, r * hprev, wt);
        if (t > 0) {
            cl_publish_n<0>(fl + c, base + 2u * t + 1u, wt);
            CS(4);
            if constexpr (RD) { mul_vec(am, av, mz); cl_tiles<H, 1>(am, bz, bz, red, tid, accz, dummy); }
            else cl_tiles<H, 1>(av, bz, bz, red, tid, accz, dummy);        // z under the exchange (only the h update needs it)
            CS(5);
        }
        const float z = hard_sigmoid(accz + xz);
        if (t > 0) {
            if (!cl_wait_w<CB>(fl, base + 2u * t + 1u, a.error, a.spin_limit)) { if (ok) a.Hout[q * H + col] = __builtin_nanf(""); return; }
            CS(6);
            ld_rows_dma<H>(av, a.aux + ((long)p0 + r0) * H, H, nact, w * (H / 4), stage + w * (16 * H / 4), lane);
            CS(7);
            __syncthreads();                                          // red: slower waves may still read the z product
            if constexpr (RD) { mul_vec(am, av, mh); cl_tiles<H, 1>(am, bh, bh, red, tid, acch, dummy); }
            else cl_tiles<H, 1>(av, bh, bh, red, tid, acch, dummy);
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
            prefetch_xw(sw.s1, sw.s2);
        }
        sw.advance_up();
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
template <int J, int ACT, bool RD>
__global__ __launch_bounds__(256) void gru_cluster_bwd(ClusterArgs a) {
    constexpr int H = 64 * J, GH = 3 * H, CB = H / 16, NB = H / 64;
    const int L = blockIdx.x, x = L & 7, s = L >> 3, jj = s / CB, c = s - jj * CB;
    const int gl = x + 8 * jj;
    const int r0 = 16 * (a.g_base + gl);
    if (gl >= a.n_groups || r0 >= a.so[1] - a.so[0]) return;
    __shared__ __attribute__((aligned(16))) float smem[1024 + 4 * 16 * (2 * H / 4)];   // partial tiles of a product + the 4 waves' A-row images (K up to 2H)
    float* red = smem;
    float* stage = smem + 1024;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = tid >> 4, col = 16 * c + (tid & 15);
    float4 bh[NB], bzr[1][2 * NB];
    {
        const float4* pa = reinterpret_cast<const float4*>(a.pk_a);      // U_h^T, K = H
        const float4* pb = reinterpret_cast<const float4*>(a.pk_b);      // [U_z U_r]^T, K = 2H
#pragma unroll
        for (int i = 0; i < NB; ++i) bh[i] = pa[((size_t)(c * 4 + w) * NB + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) bzr[0][i] = pb[((size_t)(c * 4 + w) * (2 * NB) + i) * 64 + lane];
    }
    // recurrent dropout: the products come back through the masks of the OUTPUT element (row, col) -- d(r h m_h) -> d(r h),
    // and the [z z r r] wave halves of the K = 2H product each through their gate's mask (rnn_step.hip gru_step_bwd)
    [[maybe_unused]] float m_z = 1.f, m_r = 1.f, m_h = 1.f;
    if constexpr (RD) {
        const long srow = min(r0 + row, a.B - 1);
        m_z = a.rmask[(0L * a.B + srow) * H + col]; m_r = a.rmask[(1L * a.B + srow) * H + col]; m_h = a.rmask[(2L * a.B + srow) * H + col];
    }
    unsigned* fl = a.flags + (size_t)gl * 64;
    unsigned count = a.epoch;                         // this workgroup's published exchanges so far (same sequence in every member)
    if (tid == 0) st_u32(fl + 32 + c, xcc_id() + 1u, true);
    bool wt = true, first_x = true;
    int tg = 0;                                       // steps this row block is alive
    tg = cl_alive_steps(a.so, a.T, r0);
    float carry = 0.f;
    // the element-wise operands of a step (dHout, z, r, h~, h_prev) are requested one step ahead: they do not depend on
    // the exchange, so their latency hides under the previous step's waits
    float n_dh = 0.f, n_z = 0.f, n_r = 0.f, n_hh = 0.f, n_h0 = 0.f;
    auto prefetch = [&](int t, int p0, int p1, int pm1) {          // step t: tokens [p0, p1), step t - 1 starts at pm1
        const int nact = min(16, p1 - p0 - r0);
        const long q = (long)p0 + r0 + (row < nact ? row : 0);
        float v = a.dHout[q * H + col];
        for (int sl = 1; sl < a.dh_ns; ++sl) v += a.dHout[(long)sl * a.dh_stride + q * H + col];      // slab order == the reduce launch
        if (a.dh_add) {
            const int ix = a.dh_idx[q];
            v += ix < 0 ? 0.f : (a.dh_scale ? a.dh_scale[q] : 1.f) * a.dh_add[(long)ix * a.dh_ld + col];
        }
        n_dh = v;
        n_z = a.gates[q * GH + col]; n_r = a.gates[q * GH + H + col]; n_hh = a.gates[q * GH + 2 * H + col];
        n_h0 = t > 0 ? a.Hout[((long)pm1 + r0 + (row < nact ? row : 0)) * H + col] : 0.f;
    };
    StepWindow sw;
    sw.init_down(a.so, a.T, tg > 0 ? tg - 1 : 0);
    if (tg > 0) prefetch(tg - 1, sw.s0, sw.s1, sw.prev);
    CS_DECL;
    for (int t = tg - 1; t >= 0; --t, sw.advance_down()) {
        CS(0);
        const int p0 = sw.s0, bt = sw.s1 - p0;
        const int bnext = sw.s2 - sw.s1;                             // 0 behind the last step (entries past T read as so[T])
        sw.request_down(a.so, t);
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
        CS(1);
        cl_publish_n<0>(fl + c, ++count, wt);
        CS(2);
        prefetch(t - 1, sw.prev, sw.s0, sw.nxt);      // behind the flag store: nothing on the exchange path is issued after it
        CS(3);
        // a wait that runs out: this step's gradient is poisoned (the norm then is not finite and the update refuses it)
        if (!cl_wait_w<CB>(fl, count, a.error, a.spin_limit)) { if (ok) a.dPre[q * GH + col] = __builtin_nanf(""); return; }
        CS(4);
        if (first_x) { wt = !cl_same_xcd<CB>(fl); first_x = false; }
        float acc = 0.f, dummy;
        {
            float av[H / 16];
            ld_rows_dma<H>(av, a.dPre + ((long)p0 + r0) * GH + 2 * H, GH, nact, w * (H / 4), stage + w * (16 * 2 * H / 4), lane);
            CS(5);
            cl_tiles<H, 1>(av, bh, bh, red, tid, acc, dummy);
        }
        CS(6);
        if constexpr (RD) acc *= m_h;
        const float dcar = dh * z + acc * r;
        if (ok) {
            st_f32(a.dPre + q * GH + col, dh * (h0 - hh) * hard_sigmoid_grad(z), wt);
            st_f32(a.dPre + q * GH + H + col, acc * h0 * hard_sigmoid_grad(r), wt);
        }
        cl_publish_n<0>(fl + c, ++count, wt);
        CS(7);
        if (!cl_wait_w<CB>(fl, count, a.error, a.spin_limit)) { if (ok) a.dPre[q * GH + col] = __builtin_nanf(""); return; }
        CS(8);
        float acc2[1] = {0.f}, acc2b[1] = {0.f};
        {
            float av2[2 * H / 16];
            ld_rows_dma<2 * H>(av2, a.dPre + ((long)p0 + r0) * GH, GH, nact, w * (2 * H / 4), stage + w * (16 * 2 * H / 4), lane);
            CS(9);
            cl_tiles_n<2 * H, 1, RD>(av2, bzr, red, tid, acc2, acc2b);
        }
        if constexpr (RD) carry = dcar + (acc2[0] * m_z + acc2b[0] * m_r);
        else carry = dcar + acc2[0];
        CS(10);
        CS_STEP();
    }
    CS_FLUSH(13);
}

}  // namespace

// ---- host side shared by every cluster scan ------------------------------------------------------------------------
namespace seqrec_cluster {
namespace {
std::map<hipStream_t, FlagBuf> g_flagbufs;
std::mutex g_flag_mu;
int g_cluster_override = -1;                  // seqrec_debug_scan_cluster(): tests compare the two forms in one process
int g_spin_override = 0;                      // seqrec_debug_cluster_spin_limit(): tests force a timeout
std::map<const void*, int> g_wg_per_cu;       // kernel -> resident workgroups per CU (occupancy query, once per kernel)
int g_cus = 0;
}  // namespace

// Hidden state of the library, part 1 of 2 (the other is the launch-graph cache of rnn_step.hip): 16.25 KB of exchange
// flags per stream, allocated on the stream's first cluster scan, epoch-numbered so that calls need no reset, freed by
// seqrec_release_stream.
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

bool cluster_enabled() {
    static const bool on = seqrec_env("SEQREC_SCAN_CLUSTER", 1) != 0;      // A/B switch
    return g_cluster_override >= 0 ? g_cluster_override != 0 : on;
}
int cluster_spin_limit() { return g_spin_override > 0 ? g_spin_override : CL_SPIN_LIMIT; }

// The in-kernel waits need every workgroup of a launch resident at once.  Groups are dealt to the XCDs round-robin
// (group g -> XCD g mod 8, all its CB column-block workgroups with it), so the bound is per XCD: what the occupancy
// query says one CU holds of this kernel x the CUs of an XCD, in whole groups.  0 = not even one group per XCD fits.
// (Other processes or streams on the same GPU can still take the CUs away: the waits are bounded and report.)
int cluster_group_cap(const void* kernel, int CB) {
    std::lock_guard<std::mutex> lk(g_flag_mu);
    if (g_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        g_cus = n;
    }
    auto it = g_wg_per_cu.find(kernel);
    if (it == g_wg_per_cu.end()) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, 0) != hipSuccess || nb < 0) nb = 0;
        it = g_wg_per_cu.emplace(kernel, nb).first;
    }
    const int per_xcd = (it->second * (g_cus / 8)) / CB;           // whole groups one XCD can hold
    int cap = 8 * per_xcd;
    if (cap > CL_MAX_GROUPS) cap = CL_MAX_GROUPS;
    return cap;
}

// One launch holds up to CL_MAX_GROUPS row blocks even when they cannot all be resident at once (LSTM-512: 2 workgroups per CU =
// 2 groups per XCD, a 512-session batch has 4 per XCD).  That is safe because of the ORDER, not the count: group g sits on XCD
// g mod 8 with all its members, and the workgroups an XCD receives are, in dispatch order, ALL members of its first group, then
// all of its second, ... -- so at any time an XCD holds complete groups, which finish on their own, and at most ONE group that is
// still arriving, whose resident members wait (bounded) for slots that the complete groups free.  Row blocks are sorted by
// length: the short blocks at the back of the order run in the slots the medium ones leave, beside the longest block of their
// XCD, instead of in a second launch behind it (c4: the 16 shortest row blocks used to cost a launch and 2-4 steps per scan).
// cluster_group_cap() still has to say that at least one group per XCD fits.
int launch_sliced(const void* fn, ClusterArgs& a, int CB, int G, int T, hipStream_t st) {
    const int gcap = cluster_group_cap(fn, CB);
    if (gcap < 1) return SEQREC_E_UNSUPPORTED;
    a.spin_limit = cluster_spin_limit();
    for (int g0 = 0; g0 < G; g0 += CL_MAX_GROUPS) {
        FlagBuf fb;
        const int rc = get_flagbuf(st, T, fb);
        if (rc) return rc;
        a.flags = fb.flags; a.error = fb.error; a.epoch = fb.epoch;
        a.g_base = g0; a.n_groups = G - g0 < CL_MAX_GROUPS ? G - g0 : CL_MAX_GROUPS;
        const unsigned grid = 8u * CB * ((a.n_groups + 7) / 8);
        void* argv[1] = {&a};
        const hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(256), argv, 0, st);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
}  // namespace seqrec_cluster

extern "C" void seqrec_debug_scan_cluster(int mode) { seqrec_cluster::g_cluster_override = mode; }
extern "C" void seqrec_debug_cluster_spin_limit(int polls) { seqrec_cluster::g_spin_override = polls; }
#ifdef SEQREC_CLUSTER_SPINS
extern "C" void seqrec_debug_cluster_spins(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(seqrec_cluster::g_cl_spins), z, sizeof(z)); return; }
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(seqrec_cluster::g_cl_spins), sizeof(unsigned long long) * 8);
}
#endif
#ifdef SEQREC_CLUSTER_STAMP
extern "C" void seqrec_debug_cluster_stamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(seqrec_cluster::g_cl_stamp), sizeof(unsigned long long) * 32);
}
#endif

// Bounded spins that ran out since the stream's first cluster scan (0 in a healthy run): synchronises the stream.
extern "C" int seqrec_cluster_scan_errors(void* stream) {
    using namespace seqrec_cluster;
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
// clears the counter (after the caller has reported it)
extern "C" int seqrec_cluster_scan_errors_reset(void* stream) {
    using namespace seqrec_cluster;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lk(g_flag_mu);
    auto it = g_flagbufs.find(st);
    if (it == g_flagbufs.end()) return 0;
    return (int)hipMemsetAsync(it->second.error, 0, sizeof(unsigned), st);
}
void seqrec_cluster_release_stream(hipStream_t st) {
    using namespace seqrec_cluster;
    std::lock_guard<std::mutex> lk(g_flag_mu);
    auto it = g_flagbufs.find(st);
    if (it == g_flagbufs.end()) return;
    (void)hipFree(it->second.flags);
    g_flagbufs.erase(it);
}

// ---- GRU dispatch ------------------------------------------------------------------------------------------------------
namespace {
template <int ACT, bool RD> const void* gru_fwd_kernel(int J) {
    switch (J) {
        case 1: return reinterpret_cast<const void*>(gru_cluster_fwd<1, ACT, RD>);
        case 2: return reinterpret_cast<const void*>(gru_cluster_fwd<2, ACT, RD>);
        case 4: return reinterpret_cast<const void*>(gru_cluster_fwd<4, ACT, RD>);
        case 8: return reinterpret_cast<const void*>(gru_cluster_fwd<8, ACT, RD>);
    }
    return nullptr;
}
template <int ACT, bool RD> const void* gru_bwd_kernel(int J) {
    switch (J) {
        case 1: return reinterpret_cast<const void*>(gru_cluster_bwd<1, ACT, RD>);
        case 2: return reinterpret_cast<const void*>(gru_cluster_bwd<2, ACT, RD>);
        case 4: return reinterpret_cast<const void*>(gru_cluster_bwd<4, ACT, RD>);
        case 8: return reinterpret_cast<const void*>(gru_cluster_bwd<8, ACT, RD>);
    }
    return nullptr;
}
}  // namespace

bool seqrec_cluster_other_fwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* soh, const float* XW, float* Hout,
                              float* gates, float* aux, const float* upack, const float* rmask, hipStream_t st, int* rc);
bool seqrec_cluster_other_bwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* soh, const float* dHout,
                              const float* Hout, const float* gates, const float* aux, float* dPre, const float* upack,
                              const float* rmask, hipStream_t st, int* rc);

bool seqrec_cluster_fwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* soh, const float* XW, float* Hout,
                        float* gates, float* aux, const float* upack, const float* rmask, hipStream_t st, int* rc) {
    using namespace seqrec_cluster;
    if (!cluster_enabled() || T > CL_TMAX || T < 1) return false;
    if (cell != SEQREC_CELL_GRU) return seqrec_cluster_other_fwd(cell, act, H, H_real, T, B, soh, XW, Hout, gates, aux, upack, rmask, st, rc);
    const int J = H / 64, CB = H / 16;
    const void* fn = rmask ? (act == 0 ? gru_fwd_kernel<0, true>(J) : act == 1 ? gru_fwd_kernel<1, true>(J) : gru_fwd_kernel<2, true>(J))
                           : (act == 0 ? gru_fwd_kernel<0, false>(J) : act == 1 ? gru_fwd_kernel<1, false>(J) : gru_fwd_kernel<2, false>(J));
    if (!fn || cluster_group_cap(fn, CB) < 1) return false;
    const int B0 = soh[1] - soh[0];
    if (B0 <= 0) { *rc = 0; return true; }
    ClusterArgs a = {};
    a.H_real = H_real; a.T = T; a.XW = XW; a.Hout = Hout; a.gates = gates; a.aux = aux; a.rmask = rmask; a.B = B;
    a.pk_a = upack; a.pk_b = upack + 2l * H * H;
    for (int t = 0; t <= T; ++t) a.so[t] = soh[t];
    *rc = launch_sliced(fn, a, CB, (B0 + 15) / 16, T, st);
    return true;
}

bool seqrec_cluster_bwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* soh, const float* dHout, const float* Hout,
                        const float* gates, const float* aux, float* dPre, const float* upack, const float* rmask, hipStream_t st,
                        int* rc, const seqrec_dh_parts* parts) {
    using namespace seqrec_cluster;
    if (!cluster_enabled() || T > CL_TMAX || T < 1) return false;
    if (cell != SEQREC_CELL_GRU) {
        if (parts) return false;
        return seqrec_cluster_other_bwd(cell, act, H, H_real, T, B, soh, dHout, Hout, gates, aux, dPre, upack, rmask, st, rc);
    }
    const int J = H / 64, CB = H / 16;
    const void* fn = rmask ? (act == 0 ? gru_bwd_kernel<0, true>(J) : act == 1 ? gru_bwd_kernel<1, true>(J) : gru_bwd_kernel<2, true>(J))
                           : (act == 0 ? gru_bwd_kernel<0, false>(J) : act == 1 ? gru_bwd_kernel<1, false>(J) : gru_bwd_kernel<2, false>(J));
    if (!fn || cluster_group_cap(fn, CB) < 1) return false;
    const int B0 = soh[1] - soh[0];
    if (B0 <= 0) { *rc = 0; return true; }
    ClusterArgs a = {};
    a.H_real = H_real; a.T = T; a.dHout = dHout; a.Hout = const_cast<float*>(Hout); a.gates = const_cast<float*>(gates); a.dPre = dPre;
    a.aux = const_cast<float*>(aux); a.rmask = rmask; a.B = B;
    if (parts) {
        a.dHout = parts->slabs; a.dh_ns = parts->n_slabs; a.dh_stride = (long)parts->slab_stride;
        a.dh_add = parts->add_table; a.dh_idx = parts->add_index; a.dh_scale = parts->add_scale; a.dh_ld = (long)parts->add_ld;
    }
    a.pk_a = upack + 3l * H * H; a.pk_b = upack + 4l * H * H;
    for (int t = 0; t <= T; ++t) a.so[t] = soh[t];
    *rc = launch_sliced(fn, a, CB, (B0 + 15) / 16, T, st);
    return true;
}

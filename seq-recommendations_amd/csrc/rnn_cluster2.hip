// Cluster form of the LSTM and SimpleRNN scans -- the only cells the reference builds (model.py:344-352), with its
// recurrent_dropout (model.py:346,351).  Protocol, placement and residency rules: rnn_cluster.hip.  Both cells need ONE
// in-kernel exchange per time step each way:
//   forward   h_{t-1} (all four LSTM gates read it); thread (row, col) keeps c_t / nothing else across steps
//   BPTT      the step's pre-activation gradients dPre_t [16 x G H] (they are the kernel's output anyway);
//             thread (row, col) carries dh and dc to step t-1 in registers
// Arithmetic = the step-wise kernels of rnn_step.hip element for element (K split over the 4 waves in the same
// K-permuted order, partial tiles summed in the same order, the cell formulas from ONE shared definition), so the two
// forms agree bit for bit in Hout / gates / c and to the last bits in dPre.
// LSTM BPTT: K = 4H (2048 at H = 512), the wave's slice of a dPre row is H floats -- too long for registers next to the
// wave's slice of U^T (128 VGPRs at H = 512).  It is streamed in PIECES of 16 floats per lane (4 KB per wave) through a
// four-slot LDS-DMA ring: three pieces are in flight while the 16 MFMAs of the fourth run.
#include "rnn_cluster_dev.h"
#include <type_traits>

using namespace seqrec_cluster;

namespace {

#define CL_PROLOGUE(CBV)                                                                              \
    const int L = blockIdx.x, x = L & 7, s = L >> 3, jj = s / (CBV), c = s - jj * (CBV);              \
    const int gl = x + 8 * jj;                                                                        \
    const int r0 = 16 * (a.g_base + gl);                                                              \
    if (gl >= a.n_groups || r0 >= a.so[1] - a.so[0]) return;                                          \
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;                                       \
    const int row = tid >> 4, col = 16 * c + (tid & 15);                                              \
    unsigned* fl = a.flags + (size_t)gl * 64;                                                         \
    if (tid == 0) st_u32(fl + 32 + c, xcc_id() + 1u, true);                                          \
    bool wt = true /* write-through exchange stores until the group is known to share an XCD */

template <int N> __device__ __forceinline__ void mul_vec(float (&o)[N], const float (&a)[N], const float (&m)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = a[i] * m[i];
}
template <int I, int N, class F> __device__ __forceinline__ void cl_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); cl_static_for<I + 1, N>(f); }
}

// ------------------------------------------------------------------------------------------------------------------
// SimpleRNN forward: h_t = act(xw_t + h_{t-1} . U)      (rnn_step.hip srnn_step_fwd)
// ------------------------------------------------------------------------------------------------------------------
template <int J, int ACT, bool RD>
__global__ __launch_bounds__(256) void srnn_cluster_fwd(ClusterArgs a) {
    constexpr int H = 64 * J, CB = H / 16, NB = H / 64;
    CL_PROLOGUE(CB);
    __shared__ __attribute__((aligned(16))) float smem[1024 + 4 * 16 * (H / 4)];
    float* red = smem;
    float* stage = smem + 1024 + w * (16 * H / 4);
    float4 b[1][NB];
    {
        const float4* pa = reinterpret_cast<const float4*>(a.pk_a);
#pragma unroll
        for (int i = 0; i < NB; ++i) b[0][i] = pa[((size_t)(c * 4 + w) * NB + i) * 64 + lane];
    }
    [[maybe_unused]] float mk[RD ? H / 16 : 1];
    if constexpr (RD) ld_mask(mk, a.rmask, a.B, H, 0, min(r0 + (lane & 15), a.B - 1), w * (H / 4) + (lane >> 4) * (H / 16));
    const unsigned base = a.epoch;
    float n_x = 0.f;
    auto prefetch_xw = [&](int p0, int p1) {                      // the step whose tokens are [p0, p1)
        const int nact = min(16, p1 - p0 - r0);
        if (row < nact) n_x = a.XW[((long)p0 + r0 + row) * H + col];
    };
    StepWindow sw;                                                // step offsets in registers (rnn_cluster_dev.h)
    sw.init_up(a.so, a.T);
    prefetch_xw(sw.s0, sw.s1);
    for (int t = 0; t < a.T; ++t, sw.advance_up()) {
        const int p0 = sw.s0, bt = sw.s1 - p0;
        if (bt <= r0) break;
        sw.request_up(a.so, a.T, t);
        const int nact = min(16, bt - r0);
        const bool ok = row < nact;
        const long q = (long)p0 + r0 + row;
        const bool more = sw.s2 - sw.s1 > r0;
        const float xw = n_x;
        float acc[1] = {0.f}, unused[1];
        if (t > 0) {
            if (!cl_wait_w<CB>(fl, base + (unsigned)t, a.error, a.spin_limit)) { if (ok) a.Hout[q * H + col] = __builtin_nanf(""); return; }
            if (t == 1) wt = !cl_same_xcd<CB>(fl);
            float av[H / 16];
            ld_rows_dma<H>(av, a.Hout + ((long)sw.prev + r0) * H, H, nact, w * (H / 4), stage, lane);
            if constexpr (RD) mul_vec(av, av, mk);
            cl_tiles_n<H, 1>(av, b, red, tid, acc, unused);
        }
        float y = act_fwd<ACT>(acc[0] + xw);
        if (col >= a.H_real) y = 0.f;
        if (ok) st_f32(a.Hout + q * H + col, y, wt);
        if (more) {
            cl_publish_n<0>(fl + c, base + (unsigned)t + 1u, wt);     // (its barrier also orders this step's reads of `red` before the next writes)
            prefetch_xw(sw.s1, sw.s2);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// SimpleRNN BPTT: dPre_t = dh act'(h_t);  dh_{t-1} += dPre_t . U^T        (pointwise_bwd_step + gemm_bwd_step<H>)
// ------------------------------------------------------------------------------------------------------------------
template <int J, int ACT, bool RD>
__global__ __launch_bounds__(256) void srnn_cluster_bwd(ClusterArgs a) {
    constexpr int H = 64 * J, CB = H / 16, NB = H / 64;
    CL_PROLOGUE(CB);
    __shared__ __attribute__((aligned(16))) float smem[1024 + 4 * 16 * (H / 4)];
    float* red = smem;
    float* stage = smem + 1024 + w * (16 * H / 4);
    float4 b[1][NB];
    {
        const float4* pb = reinterpret_cast<const float4*>(a.pk_b);      // U^T, K = H
#pragma unroll
        for (int i = 0; i < NB; ++i) b[0][i] = pb[((size_t)(c * 4 + w) * NB + i) * 64 + lane];
    }
    [[maybe_unused]] float m_o = 1.f;
    if constexpr (RD) m_o = a.rmask[(long)min(r0 + row, a.B - 1) * H + col];
    unsigned count = a.epoch;
    bool first_x = true;
    int tg = 0;
    tg = cl_alive_steps(a.so, a.T, r0);
    float carry = 0.f, n_dh = 0.f, n_h = 0.f;
    auto prefetch = [&](int t, int p0, int p1, int pm1) {          // step t: tokens [p0, p1), step t - 1 starts at pm1
        const int nact = min(16, p1 - p0 - r0); (void)t; (void)pm1;
        const long q = (long)p0 + r0 + (row < nact ? row : 0);
        n_dh = a.dHout[q * H + col]; n_h = a.Hout[q * H + col];
    };
    StepWindow sw;
    sw.init_down(a.so, a.T, tg > 0 ? tg - 1 : 0);
    if (tg > 0) prefetch(tg - 1, sw.s0, sw.s1, sw.prev);
    for (int t = tg - 1; t >= 0; --t, sw.advance_down()) {
        const int p0 = sw.s0, bt = sw.s1 - p0;
        const int bnext = sw.s2 - sw.s1;                             // 0 behind the last step (entries past T read as so[T])
        sw.request_down(a.so, t);
        const int nact = min(16, bt - r0);
        const bool ok = row < nact;
        const long q = (long)p0 + r0 + (ok ? row : 0);
        float dh = n_dh;
        if (r0 + row < bnext) dh += carry;
        const float dp = dh * act_grad<ACT>(n_h);
        if (t == 0) { if (ok) a.dPre[q * H + col] = dp; break; }
        if (ok) st_f32(a.dPre + q * H + col, dp, wt);
        cl_publish_n<0>(fl + c, ++count, wt);
        prefetch(t - 1, sw.prev, sw.s0, sw.nxt);
        if (!cl_wait_w<CB>(fl, count, a.error, a.spin_limit)) { if (ok) a.dPre[q * H + col] = __builtin_nanf(""); return; }
        if (first_x) { wt = !cl_same_xcd<CB>(fl); first_x = false; }
        float acc[1] = {0.f}, unused[1];
        {
            float av[H / 16];
            ld_rows_dma<H>(av, a.dPre + ((long)p0 + r0) * H, H, nact, w * (H / 4), stage, lane);
            cl_tiles_n<H, 1>(av, b, red, tid, acc, unused);
        }
        carry = RD ? acc[0] * m_o : acc[0];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LSTM forward: [i|f|c~|o] = xw_t + h_{t-1} . U ;  c_t = f c_{t-1} + i c~ ;  h_t = o act(c_t)      (lstm_step_fwd)
// The wave's K slice of h_{t-1} feeds all four gate tiles (8 independent accumulators).
// ------------------------------------------------------------------------------------------------------------------
template <int J, int ACT, bool RD>
__global__ __launch_bounds__(256, 2) void lstm_cluster_fwd(ClusterArgs a) {
    constexpr int H = 64 * J, GH = 4 * H, CB = H / 16, NB = H / 64;
    CL_PROLOGUE(CB);
    __shared__ __attribute__((aligned(16))) float smem[4096 + 4 * 16 * (H / 4)];       // 4 gate tiles x 4 waves of partial sums + the waves' A-row images
    float* red = smem;
    float* stage = smem + 4096 + w * (16 * H / 4);
    // recurrent dropout: the lane's multipliers of its H/16 A elements, per gate.  A multiplier is 0 or 1/(1-p) (inverted dropout,
    // seqrec_dropout_mask), so a gate's H/16 <= 32 of them are ONE bit word + the keep value instead of H/16 registers (round 4:
    // 4 x 32 floats next to the 128 registers of the U slice pushed the H = 512 kernel out of the cluster form); the product
    // a * (bit ? keep : 0) is the multiply the float mask gave, bit for bit.
    [[maybe_unused]] unsigned mbits[RD ? 4 : 1];
    [[maybe_unused]] float keepv = 0.f;
    if constexpr (RD) {
        static_assert(H / 16 <= 32, "one bit word per gate");
        const int srow = min(r0 + (lane & 15), a.B - 1), koff = w * (H / 4) + (lane >> 4) * (H / 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float m[H / 16];
            ld_mask(m, a.rmask, a.B, H, g, srow, koff);
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < H / 16; ++i) { if (m[i] != 0.f) { bits |= 1u << i; keepv = m[i]; } }
            mbits[g] = bits;
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // the mask words are built before the 128 registers of the U slice become live
    float4 b[4][NB];
    {
        const float4* pa = reinterpret_cast<const float4*>(a.pk_a);      // packed [U_i U_f U_c U_o], N = 4H: gate g's column block = g CB + c
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < NB; ++i) b[g][i] = pa[((size_t)((g * CB + c) * 4 + w) * NB + i) * 64 + lane];
    }
    const unsigned base = a.epoch;
    float cprev = 0.f;
    CS_DECL;
    float n_x[4] = {0.f, 0.f, 0.f, 0.f};
    auto prefetch_xw = [&](int p0, int p1) {                      // the step whose tokens are [p0, p1)
        const int nact = min(16, p1 - p0 - r0);
        if (row < nact) {
            const float* xw = a.XW + ((long)p0 + r0 + row) * GH + col;
#pragma unroll
            for (int g = 0; g < 4; ++g) n_x[g] = xw[g * H];
        }
    };
    StepWindow sw;                                                // step offsets in registers (rnn_cluster_dev.h)
    sw.init_up(a.so, a.T);
    prefetch_xw(sw.s0, sw.s1);
    for (int t = 0; t < a.T; ++t, sw.advance_up()) {
        const int p0 = sw.s0, bt = sw.s1 - p0;
        if (bt <= r0) break;
        sw.request_up(a.so, a.T, t);
        const int nact = min(16, bt - r0);
        const bool ok = row < nact;
        const long q = (long)p0 + r0 + row;
        const bool more = sw.s2 - sw.s1 > r0;
        float xw[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xw[g] = n_x[g];
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, unused[4];
        CS(0);
        if (t > 0) {
            if (!cl_wait_w<CB>(fl, base + (unsigned)t, a.error, a.spin_limit)) { if (ok) a.Hout[q * H + col] = __builtin_nanf(""); return; }
            CS(1);
            if (t == 1) wt = !cl_same_xcd<CB>(fl);
            float av[H / 16];
            ld_rows_dma<H>(av, a.Hout + ((long)sw.prev + r0) * H, H, nact, w * (H / 4), stage, lane);
            CS(2);
            if constexpr (RD) {
                // every gate reads h_{t-1} through its own mask: four masked copies of the A operand, one tile each
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g) __syncthreads();                          // red of the previous gate has been read
                    // the mask word is made opaque per step: seen as loop-invariant, its 4 x H/16 multipliers are hoisted out of the
                    // time loop and become 128 live registers (spilled) again
                    asm volatile("" : "+v"(mbits[g]));
                    acc[g] = cl_tile_bitmasked<H>(av, mbits[g], keepv, b[g], red, tid);
                }
            } else {
                cl_tiles_n<H, 4>(av, b, red, tid, acc, unused);
            }
            CS(3);
        }
        float gi, gf, gg, go, cn, hn;
        lstm_cell_fwd<ACT>(acc[0] + xw[0], acc[1] + xw[1], acc[2] + xw[2], acc[3] + xw[3], cprev, col < a.H_real, gi, gf, gg, go, cn, hn);
        if (ok) {
            st_f32(a.Hout + q * H + col, hn, wt);
            // c and the gate stash (for the BPTT) ride behind the exchange store
            st_f32(a.aux + q * H + col, cn, false);
            st_f32(a.gates + q * GH + col, gi, false);
            st_f32(a.gates + q * GH + H + col, gf, false);
            st_f32(a.gates + q * GH + 2 * H + col, gg, false);
            st_f32(a.gates + q * GH + 3 * H + col, go, false);
        }
        cprev = cn;
        if (more) {
            cl_publish_n<5>(fl + c, base + (unsigned)t + 1u, wt);
            prefetch_xw(sw.s1, sw.s2);
        }
        CS(4);
        if (t > 0) CS_STEP();
    }
    CS_FLUSH(0);
}

// ------------------------------------------------------------------------------------------------------------------
// LSTM BPTT: per step (descending) the four pre-activation gradients of (row, col) from dh, dc and the stash (exchange),
// then dh_{t-1}[row, col] = dPre_t[row, :] . U^T[:, col] with K = 4H split over the waves as [i f c o]
// (pointwise_bwd_step<LSTM> + gemm_bwd_step<4H>).  dh and dc for step t-1 stay in registers of thread (row, col).
// ------------------------------------------------------------------------------------------------------------------
template <int J, int ACT, bool RD>
__global__ __launch_bounds__(256, 2) void lstm_cluster_bwd(ClusterArgs a) {
    constexpr int H = 64 * J, GH = 4 * H, CB = H / 16, K = 4 * H;
    constexpr int SLQ = K / 16;                          // floats of a lane's run of its A row: H / 4
    constexpr int CHF = SLQ < 16 ? SLQ : 16;             // floats per lane and piece (64-byte source segments)
    constexpr int NPC = SLQ / CHF;                       // pieces per step: 1, 2, 4, 8 at H = 64 ... 512
    constexpr int RING = NPC < 4 ? NPC : 4;              // ring slots: up to 3 pieces in flight ahead of the one in the MFMAs
    constexpr int NI = 16 / (64 / CHF);                  // DMA instructions per piece
    constexpr int IMG = 16 * 4 * CHF;                    // floats of a piece image
    CL_PROLOGUE(CB);
    __shared__ __attribute__((aligned(16))) float smem[1024 + 4 * RING * IMG];
    float* red = smem;
    float* ring = smem + 1024 + w * (RING * IMG);
    float4 b[K / 64];
    {
        const float4* pb = reinterpret_cast<const float4*>(a.pk_b);      // U^T, K = 4H: wave w owns gate block w
#pragma unroll
        for (int i = 0; i < K / 64; ++i) b[i] = pb[((size_t)(c * 4 + w) * (K / 64) + i) * 64 + lane];
    }
    [[maybe_unused]] float m_g[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (RD) {
        const long srow = min(r0 + row, a.B - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) m_g[g] = a.rmask[((long)g * a.B + srow) * H + col];
    }
    unsigned count = a.epoch;
    bool first_x = true;
    int tg = 0;
    tg = cl_alive_steps(a.so, a.T, r0);
    float dh_carry = 0.f, dc_carry = 0.f;
    // element-wise operands of a step, requested one step ahead (they do not depend on the exchange)
    float n_dh = 0.f, n_g[4] = {0.f, 0.f, 0.f, 0.f}, n_cn = 0.f, n_cp = 0.f;
    auto prefetch = [&](int t, int p0, int p1, int pm1) {          // step t: tokens [p0, p1), step t - 1 starts at pm1
        const int nact = min(16, p1 - p0 - r0); (void)t; (void)pm1;
        const int rr = row < nact ? row : 0;
        const long q = (long)p0 + r0 + rr;
        n_dh = a.dHout[q * H + col];
#pragma unroll
        for (int g = 0; g < 4; ++g) n_g[g] = a.gates[q * GH + g * H + col];
        n_cn = a.aux[q * H + col];
        n_cp = t > 0 ? a.aux[((long)pm1 + r0 + rr) * H + col] : 0.f;
    };
    StepWindow sw;
    sw.init_down(a.so, a.T, tg > 0 ? tg - 1 : 0);
    if (tg > 0) prefetch(tg - 1, sw.s0, sw.s1, sw.prev);
    const int kslice0 = w * (K / 4);
    CS_DECL;
    for (int t = tg - 1; t >= 0; --t, sw.advance_down()) {
        CS(0);
        const int p0 = sw.s0, bt = sw.s1 - p0;
        const int bnext = sw.s2 - sw.s1;                             // 0 behind the last step (entries past T read as so[T])
        sw.request_down(a.so, t);
        const int nact = min(16, bt - r0);
        const bool ok = row < nact;
        const long q = (long)p0 + r0 + (ok ? row : 0);
        const bool carried = r0 + row < bnext;
        const float dh = carried ? n_dh + dh_carry : n_dh;
        float di, df, dg, dout, dcp;
        lstm_cell_bwd<ACT>(dh, carried ? dc_carry : 0.f, n_g[0], n_g[1], n_g[2], n_g[3], n_cn, n_cp, di, df, dg, dout, dcp);
        dc_carry = dcp;
        if (t == 0) {
            if (ok) {
                float* o = a.dPre + q * GH + col;
                o[0] = di; o[H] = df; o[2 * H] = dg; o[3 * H] = dout;
            }
            break;
        }
        if (ok) {
            st_f32(a.dPre + q * GH + col, di, wt);
            st_f32(a.dPre + q * GH + H + col, df, wt);
            st_f32(a.dPre + q * GH + 2 * H + col, dg, wt);
            st_f32(a.dPre + q * GH + 3 * H + col, dout, wt);
        }
        CS(1);
        cl_publish_n<0>(fl + c, ++count, wt);
        prefetch(t - 1, sw.prev, sw.s0, sw.nxt);
        CS(2);
        if (!cl_wait_w<CB>(fl, count, a.error, a.spin_limit)) { if (ok) a.dPre[q * GH + col] = __builtin_nanf(""); return; }
        CS(3);
        if (first_x) { wt = !cl_same_xcd<CB>(fl); first_x = false; }
        // dPre rows of the step, wave w's gate block, in pieces through the two-deep ring
        const float* rows = a.dPre + ((long)p0 + r0) * GH;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // RING - 1 pieces ahead: the wave's slice of a step is H x 16 rows x 4 B = 32 KB at H = 512 and an L2 round trip is
        // ~0.9 us -- with one piece ahead the loop ran at the latency of a piece per piece (4.7 us per step), not at the MFMAs' pace
#pragma unroll
        for (int pc = 0; pc < RING - 1 && pc < NPC; ++pc) dma_piece_issue<SLQ, CHF>(rows, GH, nact, kslice0, pc, ring + pc * IMG, lane);
        cl_static_for<0, NPC>([&](auto PC) {
            constexpr int pc = decltype(PC)::value;
            // slot (pc + RING - 1) % RING was read by piece pc - 1: its LDS reads have returned (dma_piece_read waits for them)
            if constexpr (pc + RING - 1 < NPC) dma_piece_issue<SLQ, CHF>(rows, GH, nact, kslice0, pc + RING - 1, ring + ((pc + RING - 1) % RING) * IMG, lane);
            // the counter is in order and only DMA loads are outstanding here (the polls drained everything older): "all but
            // the younger pieces" means piece pc has landed
            constexpr int younger = (pc + RING - 1 < NPC ? pc + RING - 1 : NPC - 1) - pc;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(younger * NI) : "memory");
            f32x4 av[CHF / 4];
            dma_piece_read<CHF>(av, ring + (pc % RING) * IMG, lane);
#pragma unroll
            for (int i = 0; i < CHF / 4; ++i) {
                const float4 bb = b[pc * (CHF / 4) + i];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][0], bb.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][1], bb.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][2], bb.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][3], bb.w, acc1, 0, 0, 0);
            }
        });
        CS(4);
        cl_red_store(red + w * 256, lane, acc0, acc1);
        __syncthreads();
        {
            const float* rt = red + cl_red_r(tid);
            if constexpr (RD) {
                float v = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) v += rt[g * 256] * m_g[g];
                dh_carry = v;
            } else {
                dh_carry = (rt[0] + rt[256]) + (rt[512] + rt[768]);
            }
        }
        // (the next write of `red` is behind the next step's publish barrier)
        CS(5);
        CS_STEP();
    }
    CS_FLUSH(16);
}

#define KERNEL_TABLE(NAME, KERN)                                                                                     \
    const void* NAME(int J, int act, bool rd) {                                                                      \
        switch ((J * 4 + act) * 2 + (rd ? 1 : 0)) {                                                                  \
            case (1 * 4 + 0) * 2: return reinterpret_cast<const void*>(KERN<1, 0, false>);                           \
            case (1 * 4 + 1) * 2: return reinterpret_cast<const void*>(KERN<1, 1, false>);                           \
            case (1 * 4 + 2) * 2: return reinterpret_cast<const void*>(KERN<1, 2, false>);                           \
            case (2 * 4 + 0) * 2: return reinterpret_cast<const void*>(KERN<2, 0, false>);                           \
            case (2 * 4 + 1) * 2: return reinterpret_cast<const void*>(KERN<2, 1, false>);                           \
            case (2 * 4 + 2) * 2: return reinterpret_cast<const void*>(KERN<2, 2, false>);                           \
            case (4 * 4 + 0) * 2: return reinterpret_cast<const void*>(KERN<4, 0, false>);                           \
            case (4 * 4 + 1) * 2: return reinterpret_cast<const void*>(KERN<4, 1, false>);                           \
            case (4 * 4 + 2) * 2: return reinterpret_cast<const void*>(KERN<4, 2, false>);                           \
            case (8 * 4 + 0) * 2: return reinterpret_cast<const void*>(KERN<8, 0, false>);                           \
            case (8 * 4 + 1) * 2: return reinterpret_cast<const void*>(KERN<8, 1, false>);                           \
            case (8 * 4 + 2) * 2: return reinterpret_cast<const void*>(KERN<8, 2, false>);                           \
            case (1 * 4 + 0) * 2 + 1: return reinterpret_cast<const void*>(KERN<1, 0, true>);                        \
            case (1 * 4 + 1) * 2 + 1: return reinterpret_cast<const void*>(KERN<1, 1, true>);                        \
            case (1 * 4 + 2) * 2 + 1: return reinterpret_cast<const void*>(KERN<1, 2, true>);                        \
            case (2 * 4 + 0) * 2 + 1: return reinterpret_cast<const void*>(KERN<2, 0, true>);                        \
            case (2 * 4 + 1) * 2 + 1: return reinterpret_cast<const void*>(KERN<2, 1, true>);                        \
            case (2 * 4 + 2) * 2 + 1: return reinterpret_cast<const void*>(KERN<2, 2, true>);                        \
            case (4 * 4 + 0) * 2 + 1: return reinterpret_cast<const void*>(KERN<4, 0, true>);                        \
            case (4 * 4 + 1) * 2 + 1: return reinterpret_cast<const void*>(KERN<4, 1, true>);                        \
            case (4 * 4 + 2) * 2 + 1: return reinterpret_cast<const void*>(KERN<4, 2, true>);                        \
            case (8 * 4 + 0) * 2 + 1: return RD8 ? reinterpret_cast<const void*>(KERN<8, 0, RD8>) : nullptr;         \
            case (8 * 4 + 1) * 2 + 1: return RD8 ? reinterpret_cast<const void*>(KERN<8, 1, RD8>) : nullptr;         \
            case (8 * 4 + 2) * 2 + 1: return RD8 ? reinterpret_cast<const void*>(KERN<8, 2, RD8>) : nullptr;         \
        }                                                                                                            \
        return nullptr;                                                                                              \
    }
// (round 3 kept the H = 512 forward LSTM with recurrent dropout out of the cluster form: 128 mask + 128 kernel registers per lane;
// round 4 holds the masks as one bit word per gate -- lstm_cluster_fwd -- and every kernel has every form)
#define RD8 true
KERNEL_TABLE(srnn_fwd_kernel, srnn_cluster_fwd)
KERNEL_TABLE(srnn_bwd_kernel, srnn_cluster_bwd)
KERNEL_TABLE(lstm_bwd_kernel, lstm_cluster_bwd)
KERNEL_TABLE(lstm_fwd_kernel, lstm_cluster_fwd)
#undef RD8

}  // namespace

#ifdef SEQREC_CLUSTER_STAMP
extern "C" void seqrec_debug_cluster_stamps2(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(seqrec_cluster::g_cl_stamp), sizeof(unsigned long long) * 32);
}
#endif

bool seqrec_cluster_other_fwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* soh, const float* XW, float* Hout,
                              float* gates, float* aux, const float* upack, const float* rmask, hipStream_t st, int* rc) {
    const int J = H / 64, CB = H / 16;
    const bool lstm = cell == SEQREC_CELL_LSTM;
    const void* fn = lstm ? lstm_fwd_kernel(J, act, rmask != nullptr) : srnn_fwd_kernel(J, act, rmask != nullptr);
    if (!fn || cluster_group_cap(fn, CB) < 1) return false;
    // an LSTM forward with dropout at H = 512 would be followed by a cluster BPTT: both forms are the same arithmetic
    const int B0 = soh[1] - soh[0];
    if (B0 <= 0) { *rc = 0; return true; }
    ClusterArgs a = {};
    a.H_real = H_real; a.T = T; a.XW = XW; a.Hout = Hout; a.gates = gates; a.aux = aux; a.rmask = rmask; a.B = B;
    a.pk_a = upack;
    for (int t = 0; t <= T; ++t) a.so[t] = soh[t];
    *rc = launch_sliced(fn, a, CB, (B0 + 15) / 16, T, st);
    return true;
}

bool seqrec_cluster_other_bwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* soh, const float* dHout,
                              const float* Hout, const float* gates, const float* aux, float* dPre, const float* upack,
                              const float* rmask, hipStream_t st, int* rc) {
    const int J = H / 64, CB = H / 16;
    const bool lstm = cell == SEQREC_CELL_LSTM;
    const void* fn = lstm ? lstm_bwd_kernel(J, act, rmask != nullptr) : srnn_bwd_kernel(J, act, rmask != nullptr);
    if (!fn || cluster_group_cap(fn, CB) < 1) return false;
    const int B0 = soh[1] - soh[0];
    if (B0 <= 0) { *rc = 0; return true; }
    ClusterArgs a = {};
    a.H_real = H_real; a.T = T; a.dHout = dHout; a.Hout = const_cast<float*>(Hout); a.gates = const_cast<float*>(gates);
    a.aux = const_cast<float*>(aux); a.dPre = dPre; a.rmask = rmask; a.B = B;
    a.pk_b = upack + (lstm ? 4l : 1l) * H * H;
    for (int t = 0; t <= T; ++t) a.so[t] = soh[t];
    *rc = launch_sliced(fn, a, CB, (B0 + 15) / 16, T, st);
    return true;
}

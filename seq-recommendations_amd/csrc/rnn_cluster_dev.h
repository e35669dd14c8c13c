// Device-side pieces shared by the cluster scans (rnn_cluster.hip: GRU; rnn_cluster2.hip: LSTM, SimpleRNN) and the host
// helpers that launch them.  See rnn_cluster.hip for the exchange protocol.
#pragma once
#include "common.h"
#include "rnn_cluster.h"

namespace seqrec_cluster {

constexpr int CL_TMAX = 159;
constexpr int CL_MAX_GROUPS = 64;
constexpr int CL_SPIN_LIMIT = 1 << 22;

struct ClusterArgs {
    int H_real, T, n_groups, g_base;
    const float* XW; float* Hout; float* gates; float* aux;
    const float* dHout; float* dPre;
    const float* pk_a; const float* pk_b;      // packed B operands of the two products (rnn_step.hip layouts)
    unsigned* flags;                           // [groups][64]: words 0..31 phase counters, 32..63 XCC ids
    unsigned* error;                           // counts the bounded spins that ran out
    unsigned epoch;
    int spin_limit;                            // polls per wait before the wave gives up (CL_SPIN_LIMIT; tests lower it)
    const float* rmask; int B;                 // recurrent-dropout multipliers [G][B][H] of the sorted session rows, or null
    // BPTT input gradient in parts (seqrec_dh_parts): dHout = slab 0, dh_ns slabs dh_stride floats apart, + dh_scale[q] * dh_add[dh_idx[q]]
    int dh_ns; long dh_stride; const float* dh_add; const int* dh_idx; const float* dh_scale; long dh_ld;
    int so[CL_TMAX + 1];
};

// per-stream flag buffers + epochs (rnn_cluster.hip)
struct FlagBuf { unsigned* flags; unsigned* error; unsigned epoch; };
int get_flagbuf(hipStream_t st, int T, FlagBuf& out);
bool cluster_enabled();
int cluster_spin_limit();
// groups one launch may hold so that ALL its workgroups are resident at once (0: the kernel does not fit -> step-wise form)
int cluster_group_cap(const void* kernel, int CB);
// the launch loop shared by every cluster scan: slices of at most cluster_group_cap() row blocks on the same stream
int launch_sliced(const void* fn, ClusterArgs& a, int CB, int n_row_blocks, int T, hipStream_t st);

// Diagnostic build only (-DSEQREC_CLUSTER_STAMP, tools/cluster_stamps.py): workgroup (group 0, column block 1) sums the
// s_memrealtime (100 MHz) spent between marked points of a step; no stamp exists in the product build.
#ifdef SEQREC_CLUSTER_STAMP
static __device__ unsigned long long g_cl_stamp[32];      // one per translation unit (rnn_cluster.hip / rnn_cluster2.hip)
#define CS_DECL unsigned long long cs_prev = __builtin_amdgcn_s_memrealtime(); unsigned long long cs_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; int cs_steps = 0
#define CS(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); cs_acc[i] += t_ - cs_prev; cs_prev = t_; } while (0)
#define CS_STEP() ++cs_steps
#define CS_FLUSH(base_) do { if (gl == 0 && a.g_base == 0 && c == 1 && threadIdx.x == 0) { for (int i_ = 0; i_ < 12; ++i_) seqrec_cluster::g_cl_stamp[(base_) + i_] = cs_acc[i_]; seqrec_cluster::g_cl_stamp[(base_) + 12] = cs_steps; } } while (0)
#else
#define CS_DECL
#define CS(i)
#define CS_STEP()
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
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF; }

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
// The same in PIECES, for slices too long to sit in registers / one LDS image at once (LSTM BPTT: K = 4H, the wave's slice is
// H floats per row).  In operand order lane (row, q) owns SLQ consecutive floats of its row, [kslice0 + q SLQ, + SLQ); piece
// `pc` is floats [pc CHF, (pc + 1) CHF) of every lane's run -- the A operands of MFMAs pc CHF .. (pc + 1) CHF - 1 of the
// wave.  issue: 16 rows x 4 runs x CHF floats by LDS-DMA (full 128-byte lines at CHF = 32) into `img` (16 x 4 CHF floats),
// 16-byte chunk c of row m at chunk c ^ (m mod chunks); read: the lane's CHF floats back in operand order.
template <int SLQ, int CHF> __device__ __forceinline__ void dma_piece_issue(const float* base, long row_stride, int nact, int kslice0,
                                                                             int pc, float* img, int lane) {
    constexpr int LPR = CHF;                       // 16-byte chunks per image row: 4 runs x CHF / 4
    constexpr int RPI = 64 / LPR;                  // rows per DMA instruction
    constexpr int NI = 16 / RPI;
    static_assert(LPR <= 64 && LPR >= 4 && (CHF % 4) == 0, "piece fits a wave instruction");
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int rl = RPI * i + lane / LPR;
        const int ch = (lane % LPR) ^ (rl % LPR);                  // image chunk -> (run, 16-byte unit inside the run's piece)
        const int run = ch / (CHF / 4), u = ch % (CHF / 4);
        const float* p = base + (long)min(rl, nact - 1) * row_stride + kslice0 + run * SLQ + pc * CHF + 4 * u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(img + i * (RPI * 4 * CHF)), 16, 0, 16 /* sc1 */);
    }
}
// The reads are inline asm: hipcc otherwise puts s_waitcnt vmcnt(0) in front of every ds_read that follows an LDS-DMA (it
// cannot see that the ring slots differ) and the piece in flight would be drained before the piece at hand is used.  The
// wait for the reads is tied to the registers it covers, so no consumer can be scheduled in front of it.
typedef __attribute__((address_space(3))) float cl_lds_float;
__device__ __forceinline__ unsigned cl_lds_addr(const float* p) { return (unsigned)(uintptr_t)(const cl_lds_float*)p; }
template <int CHF> __device__ __forceinline__ void dma_piece_read(f32x4 (&v)[CHF / 4], const float* img, int lane) {
    constexpr int LPR = CHF;
    const int m = lane & 15, q = lane >> 4;
    const unsigned src = cl_lds_addr(img + m * (4 * CHF));
#pragma unroll
    for (int j = 0; j < CHF / 4; ++j) {
        const int pos = (q * (CHF / 4) + j) ^ (m % LPR);
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(src + 16u * (unsigned)pos));
    }
    if constexpr (CHF / 4 == 8)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
    else if constexpr (CHF / 4 == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
    else {
        static_assert(CHF / 4 == 8 || CHF / 4 == 4, "piece of 16 or 32 floats per lane");
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
// consumer side: every member's counter has reached `target` (wrap-safe); false = the bounded spin ran out (counted in
// *error; the caller poisons its outputs and leaves).  Every wave polls for itself: no workgroup barrier between the flags
// and the wave's own row loads
#ifdef SEQREC_CLUSTER_SPINS          // diagnostic build (tools/cluster_spins.py): [0] waits, [1 + min(polls - 1, 6)] histogram of polls per wait
extern __device__ unsigned long long g_cl_spins[8];
#endif
template <int CB> __device__ __forceinline__ bool cl_wait_w(const unsigned* fl, unsigned target, unsigned* error, int spin_limit) {
    const int lane = threadIdx.x & 63;
    int spins = 0;
    while (true) {
        const unsigned f = lane < CB ? ld_u32_dev(fl + lane) : target;
#ifdef SEQREC_CLUSTER_SPINS
        if (lane == 0 && __all((int)(f - target) >= 0)) { atomicAdd(&g_cl_spins[0], 1ull); atomicAdd(&g_cl_spins[1 + min(spins, 6)], 1ull); }
#endif
        if (__all((int)(f - target) >= 0)) return true;
        if (++spins > spin_limit) { if (lane == 0) atomicAdd(error, 1u); return false; }
    }
}
template <int CB> __device__ __forceinline__ bool cl_same_xcd(const unsigned* fl) {          // one poll: lane i reads member i's XCC id
    const int lane = threadIdx.x & 63;
    const unsigned mine = ld_u32_dev(fl + 32 + (lane < CB ? lane : 0));
    return __all((int)(mine == (unsigned)__builtin_amdgcn_readfirstlane((int)mine)));
}

// Steps a row block is alive: the first t with so[t + 1] - so[t] <= r0 (step sizes never grow: the batcher sorts sessions by
// length).  By bisection -- a linear walk is one dependent scalar load per step, ~3 us in front of the longest block's BPTT.
__device__ __forceinline__ int cl_alive_steps(const int* so, int T, int r0) {
    int lo = 0, hi = T;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (so[mid + 1] - so[mid] > r0) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Step offsets of the scan, carried in registers: so[t], so[t + 1], so[t + 2] of the step at hand, the next one requested a step
// ahead -- a scalar load from the kernel arguments and its wait at every loop top (and another for so[t - 1] in front of the row
// load) was ~0.25 us of a 3.5 us step.  Entries past T read as so[T] (an empty step).
struct StepWindow {
    int prev, s0, s1, s2, nxt, pend;       // so[t - 1], so[t], so[t + 1], so[t + 2]; descending only: nxt = so[t - 2]; pend: the next window's new entry
    __device__ __forceinline__ void init_up(const int* so, int T) { prev = so[0]; s0 = so[0]; s1 = so[min(1, T)]; s2 = so[min(2, T)]; nxt = 0; pend = 0; }
    __device__ __forceinline__ void request_up(const int* so, int T, int t) { pend = so[min(t + 3, T)]; }
    __device__ __forceinline__ void advance_up() { prev = s0; s0 = s1; s1 = s2; s2 = pend; }
    __device__ __forceinline__ void init_down(const int* so, int T, int t) {
        prev = so[max(t - 1, 0)]; s0 = so[t]; s1 = so[min(t + 1, T)]; s2 = so[min(t + 2, T)]; nxt = so[max(t - 2, 0)]; pend = 0;
    }
    __device__ __forceinline__ void request_down(const int* so, int t) { pend = so[max(t - 3, 0)]; }
    __device__ __forceinline__ void advance_down() { s2 = s1; s1 = s0; s0 = prev; prev = nxt; nxt = pend; }
};

// The cross-wave reduce buffer of a 16x16 tile: wave w's partial tile at red[w * 256 ...].  An MFMA 16x16x4 lane (q = lane >> 4,
// n = lane & 15) holds C[4q + r][n], r = 0..3: the four are stored as ONE 16-byte write at [q][n][r] (round 4; four 4-byte writes
// before -- 16 -> 4 LDS write instructions per wave in the LSTM forward step), and thread tid = (row, col) = (tid >> 4, tid & 15)
// reads its element at cl_red_r(tid).  A wave's 64 reads cover 64 consecutive floats: conflict-free, like the writes.
__device__ __forceinline__ int cl_red_w(int lane) { return ((lane >> 4) * 16 + (lane & 15)) * 4; }
__device__ __forceinline__ int cl_red_r(int tid) { return ((tid >> 6) * 16 + (tid & 15)) * 4 + ((tid >> 4) & 3); }
__device__ __forceinline__ void cl_red_store(float* red_wave, int lane, const f32x4& a, const f32x4& b) {
    *reinterpret_cast<f32x4*>(red_wave + cl_red_w(lane)) = f32x4{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
}

// NT 16x16 tile products (one shared A operand, NT packed B operands) with K split over the 4 waves (rnn_step.hip
// tile_16x16_reg: same order of sums per tile).  red: NT x 4 x 256 floats.  SPLIT: out[t] = the sum of waves 0-1, out2[t] =
// the sum of waves 2-3 (the caller weights the halves); else out[t] = (w0 + w1) + (w2 + w3).
template <int K, int NT, bool SPLIT = false>
__device__ __forceinline__ void cl_tiles_n(const float (&a)[K / 16], const float4 (&b)[NT][K / 64], float* __restrict__ red, int tid,
                                           float (&out)[NT], float (&out2)[NT]) {
    const int lane = tid & 63, w = tid >> 6;
    f32x4 p0[NT], p1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { p0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; p1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < K / 64; ++i) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            p0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 0], b[t][i].x, p0[t], 0, 0, 0);
            p1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 1], b[t][i].y, p1[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            p0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 2], b[t][i].z, p0[t], 0, 0, 0);
            p1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 3], b[t][i].w, p1[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) cl_red_store(red + t * 1024 + w * 256, lane, p0[t], p1[t]);
    __syncthreads();
    const int rp = cl_red_r(tid);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float* rt = red + t * 1024 + rp;
        if (SPLIT) { out[t] = rt[0] + rt[256]; out2[t] = rt[512] + rt[768]; }
        else out[t] = (rt[0] + rt[256]) + (rt[512] + rt[768]);
    }
}

// ONE tile product with the A operand multiplied by a bit mask on the way in: a[i] * (bit i of `bits` ? keep : 0) -- the masked
// copy of the operand is never materialised (4 live values at a time instead of K/16: the LSTM-512 forward under recurrent
// dropout has no registers for it next to its 128 of U).  Same MFMAs in the same order as cl_tiles_n<K, 1> on the masked copy.
template <int K>
__device__ __forceinline__ float cl_tile_bitmasked(const float (&a)[K / 16], unsigned bits, float keep, const float4 (&b)[K / 64],
                                                   float* __restrict__ red, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < K / 64; ++i) {
        __builtin_amdgcn_sched_barrier(0);         // keep the masked values just-in-time: hoisted, they are K/16 live registers again
        // sign-extended 1-bit field = all ones or zero: (a AND it) * keep is a * keep or (+0) * keep -- one v_bfe_i32, one v_and, one
        // v_mul per element, no condition register.  (A dropped element enters the MFMA as +0 where the float mask gave +-0: the
        // accumulators start at +0, so no sum can tell.)
        const float a0 = __uint_as_float(__float_as_uint(a[4 * i + 0]) & (unsigned)__builtin_amdgcn_sbfe((int)bits, 4 * i + 0, 1)) * keep;
        const float a1 = __uint_as_float(__float_as_uint(a[4 * i + 1]) & (unsigned)__builtin_amdgcn_sbfe((int)bits, 4 * i + 1, 1)) * keep;
        const float a2 = __uint_as_float(__float_as_uint(a[4 * i + 2]) & (unsigned)__builtin_amdgcn_sbfe((int)bits, 4 * i + 2, 1)) * keep;
        const float a3 = __uint_as_float(__float_as_uint(a[4 * i + 3]) & (unsigned)__builtin_amdgcn_sbfe((int)bits, 4 * i + 3, 1)) * keep;
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[i].x, p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[i].y, p1, 0, 0, 0);
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b[i].z, p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b[i].w, p1, 0, 0, 0);
    }
    cl_red_store(red + w * 256, lane, p0, p1);
    __syncthreads();
    const float* rt = red + cl_red_r(tid);
    return (rt[0] + rt[256]) + (rt[512] + rt[768]);
}

// recurrent-dropout multipliers of gate g for the lane's A elements (operand order: K/16 consecutive floats at koff)
template <int N> __device__ __forceinline__ void ld_mask(float (&m)[N], const float* __restrict__ rmask, int B, int H, int g, int srow, int koff) {
    const float4* p = reinterpret_cast<const float4*>(rmask + ((long)g * B + srow) * H + koff);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { const float4 v = p[i]; m[4 * i] = v.x; m[4 * i + 1] = v.y; m[4 * i + 2] = v.z; m[4 * i + 3] = v.w; }
}

}  // namespace seqrec_cluster

// LSTM cell arithmetic shared by the step-wise and the cluster kernels (ONE definition, so that both forms round alike)
template <int ACT>
__device__ __forceinline__ void lstm_cell_fwd(float pi, float pf, float pc, float po, float cp, bool live, float& gi, float& gf, float& gg,
                                              float& go, float& c, float& h, int rt = 0) {
#pragma clang fp contract(off)      // a*b + c*d may contract either way: without this the two forms round c differently
    gi = hard_sigmoid(pi); gf = hard_sigmoid(pf); gg = act_fwd<ACT>(pc, rt); go = hard_sigmoid(po);
    c = gf * cp + gi * gg;
    h = go * act_fwd<ACT>(c, rt);
    if (!live) { c = 0.f; h = 0.f; }
}
// one (row, hidden column) of the LSTM BPTT step: the four pre-activation gradients and dc for the previous token
template <int ACT>
__device__ __forceinline__ void lstm_cell_bwd(float dh, float dcin, float gi, float gf, float gg, float go, float cn, float cp,
                                              float& di, float& df, float& dg, float& dout, float& dc_prev, int rt = 0) {
#pragma clang fp contract(off)
    const float ac = act_fwd<ACT>(cn, rt);
    const float dct = dcin + dh * go * act_grad<ACT>(ac, rt);
    di = dct * gg * hard_sigmoid_grad(gi);
    df = dct * cp * hard_sigmoid_grad(gf);
    dg = dct * gi * act_grad<ACT>(gg, rt);
    dout = dh * ac * hard_sigmoid_grad(go);
    dc_prev = dct * gf;
}

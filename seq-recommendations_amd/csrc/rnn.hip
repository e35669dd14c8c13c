// Recurrent scan (forward + BPTT) over the ragged, length-sorted, time-major packed batch.
// Replaces Keras' masked K.rnn over SimpleRNN / LSTM (model.py:248-255,344-369) and its Theano
// autodiff; GRU (Keras 2.0 equations, reset BEFORE the recurrent matmul) is the build's extension.
//
// Design (MI355X): sessions are independent, so the batch is cut into row blocks of 16 sessions
// (one MFMA M-tile) and ONE persistent workgroup per block walks all of its time steps -- no
// inter-workgroup synchronisation, no per-step launch.  Because sessions are sorted by length the
// active rows of a block at step t are a prefix, and a block stops at its own longest session.
//   * hidden state h (and the A operands r*h / dpre) live in LDS across steps ([16][K+2] floats,
//     stride = 2 mod 32 -> conflict-free MFMA A-fragment reads);
//   * wave w of the 4 owns hidden columns [16J*w, 16J*(w+1)) of EVERY gate (J = H/64), so all gate
//     arithmetic of a unit is lane-local in the accumulators (C/D map of v_mfma_f32_16x16x4_f32:
//     col = lane&15, row = 4*(lane>>4)+reg) and cell state / z / carried gradients stay in registers;
//   * the recurrent kernel U is re-laid-out once per update (seqrec_rnn_pack_u; "packed":
//     [wave][k/4][col group][lane][4])
//     so that a wave streams its B fragments from L2 with fully coalesced 16-byte loads straight
//     into registers (software ring, 16 loads in flight per lane); U never touches LDS.
// Arithmetic: exact fp32 on the f32-input MFMA (bitwise an fmaf chain over k).
//
// Roofline: MFMA (157.3 TFLOP/s fp32); algorithmic flops fwd = 2*G*H*H per token, bwd = same for
// dh (the dU GEMM runs separately in gemm.hip).  The scan is latency-bound by its T dependent
// steps; per step a workgroup streams the whole of U (4*G*H*H bytes) from L2.
#include "common.h"

namespace {

struct RnnArgs {
    const int* step_off;
    int T, B, H_real, act;
    const float* XW;
    float* Hout;
    float* gates;
    float* aux;
    const float* pk0;
    const float* pk1;
    // backward
    const float* dHout;
    const float* HoutR;
    const float* gatesR;
    const float* auxR;
    float* dPre;
};

// Branch-free guarded global access: raw buffer ops drop (stores) or zero (loads) any lane whose
// byte offset is >= num_records, so inactive rows of a 16-row block get the INVALID offset
// instead of an exec-mask branch around every access (hipcc would otherwise emit one
// s_cbranch_execz per guarded load and serialise their latencies).  The per-step base goes into the
// scalar offset.  All buffers of one launch are < 2 GiB (host check).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int INVALID_OFF = 0x7FFFFFF0;
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, INVALID_OFF, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, int voff, int soff) {
#ifdef SEQREC_PROBE_NO_LOADS      // timing-only ablation builds (tools/), never shipped
    return 0.5f;
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
#endif
}
__device__ __forceinline__ void bstore(rsrc_t r, int voff, int soff, float v) {
#ifdef SEQREC_PROBE_NO_STORES
    asm volatile("" ::"v"(v));
#else
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
#endif
}

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<4> { typedef float4 type; };
__device__ __forceinline__ float vget(const float& v, int) { return v; }
__device__ __forceinline__ float vget(const float2& v, int e) { return e == 0 ? v.x : v.y; }
__device__ __forceinline__ float vget(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// acc[NCB] (16 x 16*NCB per wave) += A(16 x 4*KB, LDS, row stride lda) . Bpacked
template <int NCB>
struct WaveGemm {
    static constexpr int VEC = NCB >= 4 ? 4 : NCB;
    static constexpr int NCG = NCB / VEC;
    static constexpr int PDK = (16 / NCG) > 0 ? (16 / NCG) : 1;   // k-blocks kept in flight
    typedef typename VecT<VEC>::type V;
    V ring[PDK][NCG];

    __device__ __forceinline__ void preload(const float* __restrict__ pk, int lane) {
        const V* p = reinterpret_cast<const V*>(pk) + lane;
#pragma unroll
        for (int i = 0; i < PDK; ++i)
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) ring[i][cg] = p[(i * NCG + cg) * 64];
    }

    // KB must be a multiple of PDK (host checks)
    __device__ __forceinline__ void run(const float* __restrict__ ldsA, int lda, const float* __restrict__ pk,
                                        int KB, int lane, f32x4 (&acc)[NCB]) {
        const V* p = reinterpret_cast<const V*>(pk) + lane;
        const float* ap = ldsA + (lane & 15) * lda + (lane >> 4);
        int kb0 = 0;
#pragma unroll 1
        for (; kb0 + PDK < KB; kb0 += PDK) {
#pragma unroll
            for (int i = 0; i < PDK; ++i) {
                const int kb = kb0 + i;
                const float a = ap[4 * kb];
                V b[NCG];
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg) {
                    b[cg] = ring[i][cg];
                    ring[i][cg] = p[((kb + PDK) * NCG + cg) * 64];
                }
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        acc[cg * VEC + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vget(b[cg], e), acc[cg * VEC + e], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < PDK; ++i) {
            const float a = ap[4 * (kb0 + i)];
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    acc[cg * VEC + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vget(ring[i][cg], e), acc[cg * VEC + e], 0, 0, 0);
        }
    }
};

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS ops (lgkmcnt), not
// for its global stores/loads (vmcnt) -- the stash stores of a step and the next phase's U
// prefetch stay in flight across it (a __syncthreads() would drain them: +~4 us per step).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int N> __device__ __forceinline__ void zero_acc(f32x4 (&acc)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------
// pack U into per-wave MFMA B-fragment order.
//   mode 0 (forward):   B[k][(gi, hc)] = U[k*ldu + gates[gi]*H + hc],            K = H
//   mode 1 (backward):  B[k = gi*H + jj][hc] = U[hc*ldu + gates[gi]*H + jj],     K = ng*H
// out[(((w*KB + kb)*NCG + cg)*64 + l)*VEC + e], col block cb = cg*VEC + e, k = 4*kb + (l>>4)
// ------------------------------------------------------------------------------------------
struct PackJob { int mode, ng, g[4]; long out_off; };
struct PackArgs { const float* U; float* out; int ldu, H, njobs; PackJob job[4]; };

__global__ void pack_u_kernel(PackArgs pa) {
    const PackJob jb = pa.job[blockIdx.y];
    const float* __restrict__ U = pa.U;
    float* __restrict__ out = pa.out + jb.out_off;
    const int H = pa.H, ldu = pa.ldu, mode = jb.mode, ng = jb.ng;
    const int J = H / 64;
    const int NCB = mode == 0 ? ng * J : J;
    const int VEC = NCB >= 4 ? 4 : NCB;
    const int NCG = NCB / VEC;
    const int K = mode == 0 ? H : ng * H;
    const int KB = K / 4;
    const long total = (long)K * NCB * 16 * 4;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        long q = o;
        const int e = (int)(q % VEC); q /= VEC;
        const int l = (int)(q % 64); q /= 64;
        const int cg = (int)(q % NCG); q /= NCG;
        const int kb = (int)(q % KB); q /= KB;
        const int w = (int)q;
        const int cb = cg * VEC + e;
        const int k = 4 * kb + (l >> 4);
        const int cc = l & 15;
        float v;
        if (mode == 0) {
            const int gi = cb / J, j = cb % J;
            const int hc = 16 * J * w + 16 * j + cc;
            v = U[(long)k * ldu + jb.g[gi] * H + hc];
        } else {
            const int hc = 16 * J * w + 16 * cb + cc;
            const int gi = k / H, jj = k % H;
            v = U[(long)hc * ldu + jb.g[gi] * H + jj];
        }
        out[o] = v;
    }
}

#define ROWCOL_SETUP()                                         \
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6; \
    const int cc = lane & 15, rq = lane >> 4;                  \
    const int r0 = blockIdx.x * 16;                            \
    const int cbase = 16 * J * w + cc;

// per-lane byte offsets of accumulator register r (row 4*rq + r) in a [*, ld] float buffer
#define ROW_OFFSETS(NAME, LD)                                                      \
    int NAME[4];                                                                   \
    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                  \
        NAME[r] = (4 * rq + r) < nact ? ((4 * rq + r) * (LD) + cbase) * 4 : INVALID_OFF;

// ------------------------------------------------------------------------------------------
// SimpleRNN:  h = act(xw + h_prev . U)
// ------------------------------------------------------------------------------------------
template <int J, int ACT>
__global__ __launch_bounds__(256) void srnn_fwd_kernel(RnnArgs a) {
    constexpr int H = 64 * J, LDA = H + 2, KB = H / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ROWCOL_SETUP();
    float* hb0 = smem;
    float* hb1 = smem + 16 * LDA;
    for (int i = tid; i < 32 * LDA; i += 256) smem[i] = 0.f;
    __syncthreads();
    const float* pk = a.pk0 + (size_t)w * H * (16 * J);
    const rsrc_t rXW = mk_rsrc(a.XW), rH = mk_rsrc(a.Hout);
    WaveGemm<J> gm;
    int cur = 0;
    for (int t = 0; t < a.T; ++t) {
        const int o0 = a.step_off[t], bt = a.step_off[t + 1] - o0;
        if (bt <= r0) break;
        const int nact = min(16, bt - r0);
        const int so = (o0 + r0) * H * 4;
        ROW_OFFSETS(vo, H);
        gm.preload(pk, lane);
        float x[J][4];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[j][r] = bload(rXW, vo[r] + 64 * j, so);
        float* hc_ = cur ? hb1 : hb0;
        float* hn_ = cur ? hb0 : hb1;
        f32x4 acc[J];
        zero_acc(acc);
        gm.run(hc_, LDA, pk, KB, lane, acc);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r, col = cbase + 16 * j;
                float y = act_fwd<ACT>(acc[j][r] + x[j][r]);
                if (col >= a.H_real) y = 0.f;
                hn_[row * LDA + col] = y;
                bstore(rH, vo[r] + 64 * j, so, y);
            }
        lds_barrier();
        cur ^= 1;
    }
}

template <int J, int ACT>
__global__ __launch_bounds__(256) void srnn_bwd_kernel(RnnArgs a) {
    constexpr int H = 64 * J, LDP = H + 2, KB = H / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ROWCOL_SETUP();
    float* dp = smem;
    const float* pk = a.pk0 + (size_t)w * H * (16 * J);
    const rsrc_t rDH = mk_rsrc(a.dHout), rH = mk_rsrc(a.HoutR), rDP = mk_rsrc(a.dPre);
    WaveGemm<J> gm;
    float dhc[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) dhc[j][r] = 0.f;
    int tb = 0;
    while (tb < a.T && a.step_off[tb + 1] - a.step_off[tb] > r0) ++tb;
    for (int t = tb - 1; t >= 0; --t) {
        const int o0 = a.step_off[t], bt = a.step_off[t + 1] - o0;
        const int nact = min(16, bt - r0);
        const int so = (o0 + r0) * H * 4;
        ROW_OFFSETS(vo, H);
        gm.preload(pk, lane);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r, col = cbase + 16 * j;
                const float dh = dhc[j][r] + bload(rDH, vo[r] + 64 * j, so);
                const float d = (row < nact) ? dh * act_grad<ACT>(bload(rH, vo[r] + 64 * j, so)) : 0.f;
                bstore(rDP, vo[r] + 64 * j, so, d);
                dp[row * LDP + col] = d;
            }
        lds_barrier();
        f32x4 acc[J];
        zero_acc(acc);
        gm.run(dp, LDP, pk, KB, lane, acc);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) dhc[j][r] = acc[j][r];
        lds_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// LSTM (gate order i,f,c,o; hard_sigmoid gates; act on candidate and on c)
// ------------------------------------------------------------------------------------------
template <int J, int ACT>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(RnnArgs a) {
    constexpr int H = 64 * J, LDA = H + 2, KB = H / 4, GH = 4 * H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ROWCOL_SETUP();
    float* hb0 = smem;
    float* hb1 = smem + 16 * LDA;
    for (int i = tid; i < 32 * LDA; i += 256) smem[i] = 0.f;
    __syncthreads();
    const float* pk = a.pk0 + (size_t)w * H * (64 * J);
    const rsrc_t rXW = mk_rsrc(a.XW), rH = mk_rsrc(a.Hout), rG = mk_rsrc(a.gates), rC = mk_rsrc(a.aux);
    WaveGemm<4 * J> gm;
    float cst[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cst[j][r] = 0.f;
    int cur = 0;
    for (int t = 0; t < a.T; ++t) {
        const int o0 = a.step_off[t], bt = a.step_off[t + 1] - o0;
        if (bt <= r0) break;
        const int nact = min(16, bt - r0);
        const int soG = (o0 + r0) * GH * 4, soH = (o0 + r0) * H * 4;
        ROW_OFFSETS(voG, GH);
        ROW_OFFSETS(voH, H);
        gm.preload(pk, lane);
        float x[4][J][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[g][j][r] = bload(rXW, voG[r] + (g * H + 16 * j) * 4, soG);
        float* hc_ = cur ? hb1 : hb0;
        float* hn_ = cur ? hb0 : hb1;
        f32x4 acc[4 * J];
        zero_acc(acc);
        gm.run(hc_, LDA, pk, KB, lane, acc);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r, col = cbase + 16 * j;
                const float gi = hard_sigmoid(acc[0 * J + j][r] + x[0][j][r]);
                const float gf = hard_sigmoid(acc[1 * J + j][r] + x[1][j][r]);
                const float gg = act_fwd<ACT>(acc[2 * J + j][r] + x[2][j][r]);
                const float go = hard_sigmoid(acc[3 * J + j][r] + x[3][j][r]);
                float c = gf * cst[j][r] + gi * gg;
                float h = go * act_fwd<ACT>(c);
                if (col >= a.H_real) { c = 0.f; h = 0.f; }
                cst[j][r] = c;
                hn_[row * LDA + col] = h;
                bstore(rH, voH[r] + 64 * j, soH, h);
                bstore(rC, voH[r] + 64 * j, soH, c);
                bstore(rG, voG[r] + (0 * H + 16 * j) * 4, soG, gi);
                bstore(rG, voG[r] + (1 * H + 16 * j) * 4, soG, gf);
                bstore(rG, voG[r] + (2 * H + 16 * j) * 4, soG, gg);
                bstore(rG, voG[r] + (3 * H + 16 * j) * 4, soG, go);
            }
        lds_barrier();
        cur ^= 1;
    }
}

template <int J, int ACT>
__global__ __launch_bounds__(256) void lstm_bwd_kernel(RnnArgs a) {
    constexpr int H = 64 * J, GH = 4 * H, LDP = GH + 2, KB = GH / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ROWCOL_SETUP();
    float* dp = smem;
    const float* pk = a.pk0 + (size_t)w * GH * (16 * J);
    const rsrc_t rDH = mk_rsrc(a.dHout), rG = mk_rsrc(a.gatesR), rC = mk_rsrc(a.auxR), rDP = mk_rsrc(a.dPre);
    WaveGemm<J> gm;
    float dhc[J][4], dcc[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { dhc[j][r] = 0.f; dcc[j][r] = 0.f; }
    int tb = 0;
    while (tb < a.T && a.step_off[tb + 1] - a.step_off[tb] > r0) ++tb;
    for (int t = tb - 1; t >= 0; --t) {
        const int o0 = a.step_off[t], bt = a.step_off[t + 1] - o0;
        const int nact = min(16, bt - r0);
        const int soG = (o0 + r0) * GH * 4, soH = (o0 + r0) * H * 4;
        const int soP = t > 0 ? (a.step_off[t - 1] + r0) * H * 4 : 0;
        ROW_OFFSETS(voG, GH);
        ROW_OFFSETS(voH, H);
        gm.preload(pk, lane);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r, col = cbase + 16 * j;
                const float gi = bload(rG, voG[r] + (0 * H + 16 * j) * 4, soG);
                const float gf = bload(rG, voG[r] + (1 * H + 16 * j) * 4, soG);
                const float gg = bload(rG, voG[r] + (2 * H + 16 * j) * 4, soG);
                const float go = bload(rG, voG[r] + (3 * H + 16 * j) * 4, soG);
                const float cn = bload(rC, voH[r] + 64 * j, soH);
                const float cp = bload(rC, t > 0 ? voH[r] + 64 * j : INVALID_OFF, soP);
                const float dh = dhc[j][r] + bload(rDH, voH[r] + 64 * j, soH);
                const float ac = act_fwd<ACT>(cn);
                const float dct = dcc[j][r] + dh * go * act_grad<ACT>(ac);
                const float dpi = dct * gg * hard_sigmoid_grad(gi);
                const float dpf = dct * cp * hard_sigmoid_grad(gf);
                const float dpc = dct * gi * act_grad<ACT>(gg);
                const float dpo = dh * ac * hard_sigmoid_grad(go);
                dcc[j][r] = (row < nact) ? dct * gf : 0.f;
                bstore(rDP, voG[r] + (0 * H + 16 * j) * 4, soG, dpi);
                bstore(rDP, voG[r] + (1 * H + 16 * j) * 4, soG, dpf);
                bstore(rDP, voG[r] + (2 * H + 16 * j) * 4, soG, dpc);
                bstore(rDP, voG[r] + (3 * H + 16 * j) * 4, soG, dpo);
                float* d = dp + row * LDP + col;
                d[0] = dpi; d[H] = dpf; d[2 * H] = dpc; d[3 * H] = dpo;
            }
        lds_barrier();
        f32x4 acc[J];
        zero_acc(acc);
        gm.run(dp, LDP, pk, KB, lane, acc);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) dhc[j][r] = acc[j][r];
        lds_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// GRU (Keras 2.0: z,r,h; h~ = act(x_h + (r*h_prev).U_h); h = z*h_prev + (1-z)*h~)
// ------------------------------------------------------------------------------------------
template <int J, int ACT>
__global__ __launch_bounds__(256) void gru_fwd_kernel(RnnArgs a) {
    constexpr int H = 64 * J, LDA = H + 2, KB = H / 4, GH = 3 * H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ROWCOL_SETUP();
    float* hb = smem;
    float* rhb = smem + 16 * LDA;
    for (int i = tid; i < 32 * LDA; i += 256) smem[i] = 0.f;
    __syncthreads();
    const float* pk_zr = a.pk0 + (size_t)w * H * (32 * J);
    const float* pk_h = a.pk1 + (size_t)w * H * (16 * J);
    const rsrc_t rXW = mk_rsrc(a.XW), rH = mk_rsrc(a.Hout), rG = mk_rsrc(a.gates), rA = mk_rsrc(a.aux);
    WaveGemm<2 * J> g1;
    WaveGemm<J> g2;
    // vmcnt retires in issue order and counts stores: a load issued BEHIND a step's stash stores
    // cannot be waited for before those stores are acknowledged.  So each GEMM's first U fragments
    // are requested before the preceding epilogue issues its stores.
    g1.preload(pk_zr, lane);
    for (int t = 0; t < a.T; ++t) {
        const int o0 = a.step_off[t], bt = a.step_off[t + 1] - o0;
        if (bt <= r0) break;
        const int nact = min(16, bt - r0);
        const int soG = (o0 + r0) * GH * 4, soH = (o0 + r0) * H * 4;
        ROW_OFFSETS(voG, GH);
        ROW_OFFSETS(voH, H);
        float x[3][J][4];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[g][j][r] = bload(rXW, voG[r] + (g * H + 16 * j) * 4, soG);
        f32x4 acc1[2 * J];
        zero_acc(acc1);
        g1.run(hb, LDA, pk_zr, KB, lane, acc1);
        g2.preload(pk_h, lane);
        float zr[J][4], hp[J][4];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r, col = cbase + 16 * j;
                const float z = hard_sigmoid(acc1[j][r] + x[0][j][r]);
                const float rr = hard_sigmoid(acc1[J + j][r] + x[1][j][r]);
                const float h0 = hb[row * LDA + col];
                const float rh = rr * h0;
                zr[j][r] = z;
                hp[j][r] = h0;
                rhb[row * LDA + col] = rh;
                bstore(rG, voG[r] + (16 * j) * 4, soG, z);
                bstore(rG, voG[r] + (H + 16 * j) * 4, soG, rr);
                bstore(rA, voH[r] + 64 * j, soH, rh);
            }
        lds_barrier();
        f32x4 acc2[J];
        zero_acc(acc2);
        g2.run(rhb, LDA, pk_h, KB, lane, acc2);
        g1.preload(pk_zr, lane);          // next step's first fragments, ahead of this step's stores
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r, col = cbase + 16 * j;
                const float hh = act_fwd<ACT>(acc2[j][r] + x[2][j][r]);
                float hn = zr[j][r] * hp[j][r] + (1.f - zr[j][r]) * hh;
                if (col >= a.H_real) hn = 0.f;
                hb[row * LDA + col] = hn;
                bstore(rH, voH[r] + 64 * j, soH, hn);
                bstore(rG, voG[r] + (2 * H + 16 * j) * 4, soG, hh);
            }
        lds_barrier();
    }
}

template <int J, int ACT>
__global__ __launch_bounds__(256) void gru_bwd_kernel(RnnArgs a) {
    constexpr int H = 64 * J, GH = 3 * H, LD1 = H + 2, LD2 = 2 * H + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ROWCOL_SETUP();
    float* dph = smem;                 // [16][H+2]   dpre_h           (A of GEMM 1)
    float* dpzr = smem + 16 * LD1;     // [16][2H+2]  dpre_z | dpre_r  (A of GEMM 2)
    const float* pk_hT = a.pk0 + (size_t)w * H * (16 * J);
    const float* pk_zrT = a.pk1 + (size_t)w * (2 * H) * (16 * J);
    const rsrc_t rDH = mk_rsrc(a.dHout), rH = mk_rsrc(a.HoutR), rG = mk_rsrc(a.gatesR), rDP = mk_rsrc(a.dPre);
    WaveGemm<J> g1, g2;
    float dhc[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) dhc[j][r] = 0.f;
    int tb = 0;
    while (tb < a.T && a.step_off[tb + 1] - a.step_off[tb] > r0) ++tb;
    // stash values of the step about to be processed (loaded one step ahead, under GEMM 2)
    float lz[J][4], lr[J][4], lhh[J][4], lh0[J][4], ldh[J][4];
    auto load_step = [&](int t) {
        const int o0 = a.step_off[t], bt = a.step_off[t + 1] - o0;
        const int nact = min(16, bt - r0);
        const int soG = (o0 + r0) * GH * 4, soH = (o0 + r0) * H * 4;
        const int soP = t > 0 ? (a.step_off[t - 1] + r0) * H * 4 : 0;
        ROW_OFFSETS(voG, GH);
        ROW_OFFSETS(voH, H);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lz[j][r] = bload(rG, voG[r] + (16 * j) * 4, soG);
                lr[j][r] = bload(rG, voG[r] + (H + 16 * j) * 4, soG);
                lhh[j][r] = bload(rG, voG[r] + (2 * H + 16 * j) * 4, soG);
                lh0[j][r] = bload(rH, t > 0 ? voH[r] + 64 * j : INVALID_OFF, soP);
                ldh[j][r] = bload(rDH, voH[r] + 64 * j, soH);
            }
    };
    if (tb > 0) load_step(tb - 1);
    g1.preload(pk_hT, lane);
    for (int t = tb - 1; t >= 0; --t) {
        const int o0 = a.step_off[t], bt = a.step_off[t + 1] - o0;
        const int nact = min(16, bt - r0);
        const int soG = (o0 + r0) * GH * 4;
        ROW_OFFSETS(voG, GH);
        float zv[J][4], rv[J][4], hpv[J][4], dzv[J][4], dcar[J][4];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r, col = cbase + 16 * j;
                const float z = lz[j][r], rr = lr[j][r], hh = lhh[j][r], h0 = lh0[j][r];
                const float dh = (row < nact) ? dhc[j][r] + ldh[j][r] : 0.f;
                const float d = dh * (1.f - z) * act_grad<ACT>(hh);
                zv[j][r] = z; rv[j][r] = rr; hpv[j][r] = h0;
                dzv[j][r] = dh * (h0 - hh);
                dcar[j][r] = dh * z;
                bstore(rDP, voG[r] + (2 * H + 16 * j) * 4, soG, d);
                dph[row * LD1 + col] = d;
            }
        lds_barrier();
        f32x4 acc1[J];
        zero_acc(acc1);
        g1.run(dph, LD1, pk_hT, H / 4, lane, acc1);
        g2.preload(pk_zrT, lane);
        if (t > 0) load_step(t - 1);      // next step's stash, in flight under GEMM 2
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r, col = cbase + 16 * j;
                const float drh = acc1[j][r];
                const float dr = drh * hpv[j][r];
                dcar[j][r] += drh * rv[j][r];
                const float dpz = dzv[j][r] * hard_sigmoid_grad(zv[j][r]);
                const float dpr = dr * hard_sigmoid_grad(rv[j][r]);
                dpzr[row * LD2 + col] = dpz;
                dpzr[row * LD2 + H + col] = dpr;
                bstore(rDP, voG[r] + (16 * j) * 4, soG, dpz);
                bstore(rDP, voG[r] + (H + 16 * j) * 4, soG, dpr);
            }
        lds_barrier();
        f32x4 acc2[J];
        zero_acc(acc2);
        g2.run(dpzr, LD2, pk_zrT, (2 * H) / 4, lane, acc2);
        g1.preload(pk_hT, lane);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) dhc[j][r] = dcar[j][r] + acc2[j][r];
        // the next iteration's first LDS write (dph) is fenced from this GEMM-2's reads of dpzr by
        // its own barrier; dph itself was last read before the barrier above.
    }
}

int set_lds(const void* fn, size_t bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

#define LAUNCH_JA(KERNEL, JJ, AA)                                                              \
    { rc__ = set_lds((const void*)KERNEL<JJ, AA>, lds__);                                      \
      if (!rc__) hipLaunchKernelGGL((KERNEL<JJ, AA>), grid, dim3(256), lds__, st, a); }
#define DISPATCH_A(KERNEL, JJ)                                                                 \
    switch (act) {                                                                             \
        case SEQREC_ACT_RELU: LAUNCH_JA(KERNEL, JJ, SEQREC_ACT_RELU) break;                     \
        case SEQREC_ACT_TANH: LAUNCH_JA(KERNEL, JJ, SEQREC_ACT_TANH) break;                     \
        default: LAUNCH_JA(KERNEL, JJ, SEQREC_ACT_LINEAR) break;                                \
    }
#define DISPATCH_J(KERNEL, LDS_BYTES)                                                          \
    do {                                                                                       \
        const size_t lds__ = (LDS_BYTES);                                                      \
        int rc__ = 0;                                                                          \
        switch (J) {                                                                           \
            case 1: DISPATCH_A(KERNEL, 1) break;                                               \
            case 2: DISPATCH_A(KERNEL, 2) break;                                               \
            case 4: DISPATCH_A(KERNEL, 4) break;                                               \
            case 8: DISPATCH_A(KERNEL, 8) break;                                               \
            default: return SEQREC_E_SHAPE;                                                    \
        }                                                                                      \
        if (rc__) return rc__;                                                                 \
        SEQREC_LAUNCH_CHECK();                                                                 \
    } while (0)

bool check_common(int cell, int act, int H, int H_real, int T, int B) {
    if (cell < 0 || cell > 2 || act < 0 || act > 2) return false;
    if (!(H == 64 || H == 128 || H == 256 || H == 512)) return false;
    if (H_real < 1 || H_real > H || T < 0 || B < 0) return false;
    if ((long)B * T * 4 * H * 4 >= 0x7FFFFFF0L) return false;      // every activation buffer < 2 GiB (buffer offsets)
    return true;
}

}  // namespace

extern "C" int64_t seqrec_rnn_upack_floats(int cell, int H) {
    const int G = cell == SEQREC_CELL_LSTM ? 4 : (cell == SEQREC_CELL_GRU ? 3 : 1);
    // forward layouts, then backward (transposed) layouts; GRU: + U_h^T once more in the full-K-per-wave order of the
    // step-wise scan's wide BPTT tile (rnn_step.hip gru_step_bwd0_wide)
    return (int64_t)2 * G * H * H + (cell == SEQREC_CELL_GRU ? (int64_t)H * H : 0);
}

extern "C" int seqrec_rnn_pack_u(int cell, int H, const float* U, float* upack, void* stream) {
    if (cell < 0 || cell > 2 || !(H == 64 || H == 128 || H == 256 || H == 512)) return SEQREC_E_SHAPE;
    if (!U || !upack) return SEQREC_E_ARG;
    const long HH = (long)H * H;
    PackArgs pa = {};
    pa.U = U; pa.out = upack; pa.H = H;
    auto job = [&](int i, int mode, int ng, int g0, int g1, int g2, int g3, long off) {
        pa.job[i].mode = mode; pa.job[i].ng = ng;
        pa.job[i].g[0] = g0; pa.job[i].g[1] = g1; pa.job[i].g[2] = g2; pa.job[i].g[3] = g3;
        pa.job[i].out_off = off;
    };
    if (cell == SEQREC_CELL_SIMPLERNN) {
        pa.ldu = H; pa.njobs = 2;
        job(0, 0, 1, 0, 0, 0, 0, 0);
        job(1, 1, 1, 0, 0, 0, 0, HH);
    } else if (cell == SEQREC_CELL_LSTM) {
        pa.ldu = 4 * H; pa.njobs = 2;
        job(0, 0, 4, 0, 1, 2, 3, 0);
        job(1, 1, 4, 0, 1, 2, 3, 4 * HH);
    } else {
        pa.ldu = 3 * H; pa.njobs = 4;
        job(0, 0, 2, 0, 1, 0, 0, 0);          // fwd  [z r]
        job(1, 0, 1, 2, 0, 0, 0, 2 * HH);     // fwd  h
        job(2, 1, 1, 2, 0, 0, 0, 3 * HH);     // bwd  U_h^T
        job(3, 1, 2, 0, 1, 0, 0, 4 * HH);     // bwd  [U_z U_r]^T
    }
    hipLaunchKernelGGL(pack_u_kernel, dim3(256, pa.njobs), dim3(256), 0, as_stream(stream), pa);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int seqrec_rnn_fwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* step_off,
                              const float* XW, float* Hout, float* gates, float* aux,
                              const float* upack, void* stream) {
    if (!check_common(cell, act, H, H_real, T, B)) return SEQREC_E_SHAPE;
    if (T == 0 || B == 0) return 0;
    if (!step_off || !XW || !Hout || !upack) return SEQREC_E_ARG;
    if (cell != SEQREC_CELL_SIMPLERNN && (!gates || !aux)) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    const int J = H / 64;
    RnnArgs a = {};
    a.step_off = step_off; a.T = T; a.B = B; a.H_real = H_real; a.act = act;
    a.XW = XW; a.Hout = Hout; a.gates = gates; a.aux = aux;
    a.pk0 = upack;
    dim3 grid((B + 15) / 16);
    if (cell == SEQREC_CELL_SIMPLERNN) {
        DISPATCH_J(srnn_fwd_kernel, (size_t)32 * (H + 2) * 4);
    } else if (cell == SEQREC_CELL_LSTM) {
        DISPATCH_J(lstm_fwd_kernel, (size_t)32 * (H + 2) * 4);
    } else {
        a.pk1 = upack + (size_t)2 * H * H;
        DISPATCH_J(gru_fwd_kernel, (size_t)32 * (H + 2) * 4);
    }
    return 0;
}

extern "C" int seqrec_rnn_bwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* step_off,
                              const float* dHout, const float* Hout, const float* gates, const float* aux,
                              float* dPre, const float* upack, void* stream) {
    if (!check_common(cell, act, H, H_real, T, B)) return SEQREC_E_SHAPE;
    if (T == 0 || B == 0) return 0;
    if (!step_off || !dHout || !Hout || !dPre || !upack) return SEQREC_E_ARG;
    if (cell != SEQREC_CELL_SIMPLERNN && (!gates || !aux)) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    const int J = H / 64;
    const int G = cell == SEQREC_CELL_LSTM ? 4 : (cell == SEQREC_CELL_GRU ? 3 : 1);
    RnnArgs a = {};
    a.step_off = step_off; a.T = T; a.B = B; a.H_real = H_real; a.act = act;
    a.dHout = dHout; a.HoutR = Hout; a.gatesR = gates; a.auxR = aux; a.dPre = dPre;
    a.pk0 = upack + (size_t)G * H * H;
    dim3 grid((B + 15) / 16);
    if (cell == SEQREC_CELL_SIMPLERNN) {
        DISPATCH_J(srnn_bwd_kernel, (size_t)16 * (H + 2) * 4);
    } else if (cell == SEQREC_CELL_LSTM) {
        DISPATCH_J(lstm_bwd_kernel, (size_t)16 * (4 * H + 2) * 4);
    } else {
        a.pk1 = a.pk0 + (size_t)H * H;
        DISPATCH_J(gru_bwd_kernel, (size_t)16 * (3 * H + 4) * 4);
    }
    return 0;
}

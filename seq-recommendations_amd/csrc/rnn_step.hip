// Step-wise form of the GRU scan: one small whole-chip launch per recurrent GEMM instead of one
// persistent workgroup per 16 sessions (rnn.hip).
//
// Why both exist (measured on MI355X, H = 256): the persistent form is bound by the f32 MFMA time of
// ONE CU per 16 sessions (768 MFMAs/wave/step = 12 us) plus its epilogue traffic (17-21 us/step); the
// critical path of an MSNBC-shaped batch is its longest session (T ~ 30-46 steps) while only the first
// row block is still alive.  Cutting a step into launches spreads each recurrent GEMM over
// (B_t/16) x (N/64) workgroups, so a step costs two dependent launches of ~1 us of MFMA each; the
// kernel boundary (1.7-2.4 us) is the price, and it also gives the h / r*h exchange between column
// slices for free (no in-kernel inter-workgroup protocol, no residency assumption).
//
// Workgroup = 256 threads: 16 session rows x 16 output columns with K split over the 4 waves;
// BOTH operands go straight from global memory into MFMA operand registers: the A rows (h_prev / r*h /
// dpre, or the BPTT factor d recomputed on the fly) through a K-permuted packing of U that lets lane
// (row, q) of wave w own K/16 consecutive k of its row (one or two 16-byte loads), the B operand (U slice)
// from a per-(column block, wave) packed layout; every load of the kernel is issued before the first MFMA
// waits, partial tiles meet in LDS.  Only the LSTM forward step, whose four gate waves share one A tile,
// stages h_prev in LDS.  Exact fp32 (v_mfma_f32_16x16x4_f32).  Needs the step offsets on the HOST to size
// the launches.
#include "common.h"
#include "rnn_cluster.h"
#include "rnn_cluster_dev.h"
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int INVALID_OFF = 0x7FFFFFF0;
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, INVALID_OFF, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
#ifndef SEQREC_STORE_AUX
#define SEQREC_STORE_AUX 0       // cache policy of the step kernels' output stores (tuning builds: 2 nt, 16 sc1, 17 sc0 sc1)
#endif
__device__ __forceinline__ void bstore(rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, SEQREC_STORE_AUX);
}

struct StepArgs {
    int H, H_real;
    int p0, pprev0, bt, bnext, first;        // token offset of this step / previous step, rows, rows of step t+1
    int tag;                                  // launch index inside the scan call (orders the nodes of a captured graph)
    const float* XW; float* Hout; float* gates; float* aux;
    const float* pk;                          // packed B for this launch
    const float* dHout; float* dPre; float* dHc; float* tmpc;
    const float* rmask;                       // recurrent-dropout multipliers [G][B][H] (sorted session rows) or null
    int B;
    int cbn, xcd;                             // column blocks of this launch; XCD-aware tile placement on/off
    int act_rt;                               // SEQREC_ACT_* of the call: read by the SEQREC_ACT_OTHER instances only (common.h act_fwd)
};

// Diagnostic build only (-DSEQREC_STAMP, tools/stamp_probe.py): s_memtime stamps of ONE workgroup per launch (row
// block 0, column block 1, wave 0) into a device array of the code object; no stamp exists in the product build.
#ifdef SEQREC_STAMP
__device__ unsigned long long g_stamp[256 * 8];
#define STAMP(i)                                                                                              \
    do {                                                                                                      \
        if (stamp_me) {                                                                                       \
            unsigned long long t_;                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            if (threadIdx.x == 0) g_stamp[(a.tag & 255) * 8 + (i)] = t_;                                      \
        }                                                                                                     \
    } while (0)
#define STAMP_REAL(i)                                                                                         \
    do {                                                                                                      \
        if (stamp_me) {                                                                                       \
            unsigned long long t_;                                                                            \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
            if (threadIdx.x == 0) g_stamp[(a.tag & 255) * 8 + (i)] = t_;                                      \
        }                                                                                                     \
    } while (0)
#else
#define STAMP(i)
#define STAMP_REAL(i)
#endif

// Tile placement (speed only, never correctness): the launch is 1-D and workgroup ids are dealt
// round-robin to the 8 XCDs, each with its own L2.  Row block rb is pinned to XCD rb % 8 with ALL its
// column blocks, at every step and in both directions: the h / r*h / dpre rows one launch writes are
// then still in the L2 of the XCD whose workgroups read them in the next launch (instead of being
// fetched by up to 8 L2s from the Infinity Cache), and every XCD keeps the packed U resident.
// Grid = 8 * cbn * ceil(rb / 8); workgroups of XCDs without a row block exit at once.
__device__ __forceinline__ bool tile_of(const StepArgs& a, int& r0, int& cb) {
#ifdef SEQREC_SCAN_EMPTY        // timing-only build (tools/bench_scan.py): every workgroup exits at once, which
    return false;               // leaves the launch chain itself -- same launches, same grids -- to be timed
#endif
    const int L = blockIdx.x;
    int rbk;
    if (a.xcd) {
        const int x = L & 7, s = L >> 3;
        const int j = s / a.cbn;
        cb = s - j * a.cbn;
        rbk = x + 8 * j;
    } else {                                  // plain order: row blocks fastest
        const int rbn = gridDim.x / a.cbn;
        cb = L / rbn;
        rbk = L - cb * rbn;
    }
    r0 = rbk * 16;
    return r0 < a.bt;
}

// Workgroup = 16 session rows x 16 output columns; the 4 waves split K (wave w owns k-blocks
// [w*K/16, (w+1)*K/16)), partial 16x16 tiles are summed through LDS and every thread finishes ONE
// output element (row = tid>>4, col = tid&15), whose epilogue inputs were requested at kernel entry.
// Per launch the dependent chain is: loads (one L2 round trip) -> 16..32 MFMAs -> LDS reduce -> store.
//
// packed[((cb*4 + w)*(K/64) + i)*64 + lane] (float4): element e <-> MFMA m = 4*i + e of wave w,
//   value = B[k(w, m, lane>>4)][16*cb + (lane&15)],  k = w*(K/4) + q*(K/16) + m  -- lane (row, q) of wave w owns
//   K/16 CONSECUTIVE k of its A row and loads them straight from global memory (every accepted shape: K <= 2048)
// mode 0: B[k][n] = U[k*ldu + coff + n]           (forward:  K = H)
// mode 1: B[k][n] = U[n*ldu + coff + k]           (backward: transposed; K = H, 2H or 4H)
// wide != 0: the full-K-per-wave order of the wide tiles: out[((cb*(K/16) + i)*64 + lane)] (float4), element e <-> MFMA
//   m = 4*i + e of the wave that owns column block cb, k = q*(K/4) + m -- lane (row, q) owns K/4 consecutive k of its A row
struct PackStepJob { int coff, K, N, mode; long off; int wide; };
struct PackStepArgs { const float* U; float* out; int ldu; PackStepJob job[5]; };
// the step's shared negatives ride in the same launch (grid row `njobs`): draw k (oracle/rng.py counter RNG, exactly
// seqrec_sample_gather), copy of its table row, its log-Q -- two ~5 us launches of a training step become one
struct SampleJob { uint64_t key, step; int K, V, width, njobs; const uint32_t* thresh; const int* alias; const float* table;
                   const float* logq; int* neg; float* rows; float* lq; };
__device__ __forceinline__ void sample_rows(const SampleJob& sj) {
    const int lane = threadIdx.x & 63;
    for (int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < sj.K; k += gridDim.x * (blockDim.x >> 6)) {
        const uint64_t r = rand64(sj.key, sj.step * (uint64_t)sj.K + (uint64_t)k);
        const uint64_t hi = r >> 32;
        const uint32_t lo = (uint32_t)(r & 0xFFFFFFFFu);
        const int j = (int)((hi * (uint64_t)sj.V) >> 32);
        const int id = lo < sj.thresh[j] ? j : sj.alias[j];
        const float* src = sj.table + (long)id * sj.width;
        float* dst = sj.rows + (long)k * sj.width;
        if ((sj.width & 3) == 0) {
            for (int c = lane; c < sj.width / 4; c += 64) reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(src)[c];
        } else {
            for (int c = lane; c < sj.width; c += 64) dst[c] = src[c];
        }
        if (lane == 0) {
            sj.neg[k] = id;
            if (sj.lq) sj.lq[k] = sj.logq[id];
        }
    }
}
// ... and so does the NEXT thing a training step needs besides the weights: its batch.  Grid row `bj.row` gathers ids / targets / prev
// links of the ragged batch from the HBM-resident data set (seqrec_pack_batch_host: step offsets and session indices in the kernel
// arguments) -- the step's three openers (batch, U re-pack, negatives) in ONE launch (round 4: 6.6 + 8.9 us were two).
constexpr int PACK_MERGED_MAX = 640;           // B + T + 1 ints of a batch that rides here (the kernel-argument segment is 4 KB)
struct BatchJob { const int* flat; const long* starts; int B, T, row, gbx; int* sess_out; int* step_off_out; int* ids; int* tgt; int* prev;
                  int v[PACK_MERGED_MAX]; };
__device__ __forceinline__ void batch_rows(const BatchJob& h) {
    if ((int)blockIdx.x >= h.T * h.gbx) return;
    const int t = blockIdx.x / h.gbx;
    const int r = (blockIdx.x - t * h.gbx) * blockDim.x + threadIdx.x;
    if (t == 0) {
        if (r < h.B) h.sess_out[r] = h.v[h.T + 1 + r];
        if (r <= h.T) h.step_off_out[r] = h.v[r];
    }
    const int p0 = h.v[t];
    if (r >= h.v[t + 1] - p0) return;
    const long base = h.starts[h.v[h.T + 1 + r]] + t;
    const int p = p0 + r;
    h.ids[p] = h.flat[base];
    h.tgt[p] = h.flat[base + 1];
    h.prev[p] = t > 0 ? h.v[t - 1] + r : -1;
}
__device__ __forceinline__ void pack_rows(const PackStepArgs& pa);
__global__ void pack_step_batch_kernel(PackStepArgs pa, SampleJob sj, BatchJob bj) {
    if ((int)blockIdx.y == bj.row) { batch_rows(bj); return; }
    if ((int)blockIdx.y == sj.njobs) { sample_rows(sj); return; }
    pack_rows(pa);
}
__global__ void pack_step_kernel(PackStepArgs pa, SampleJob sj) {
    if ((int)blockIdx.y == sj.njobs) { sample_rows(sj); return; }      // njobs < 0: no sampling row in this launch
    pack_rows(pa);
}
__device__ __forceinline__ void pack_rows(const PackStepArgs& pa) {
    const PackStepJob jb = pa.job[blockIdx.y];
    const float* __restrict__ U = pa.U;
    float* __restrict__ out = pa.out + jb.off;
    const int ldu = pa.ldu, coff = jb.coff, K = jb.K, N = jb.N, mode = jb.mode;
    const long total = (long)K * N;
    const int G4 = K / 64;                 // float4 groups per wave
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        long q = o;
        const int e = (int)(q & 3); q >>= 2;
        const int l = (int)(q & 63); q >>= 6;
        int k, cb;
        if (jb.wide) {
            const int i = (int)(q % (K / 16)); q /= (K / 16);
            cb = (int)q;
            k = (l >> 4) * (K / 4) + 4 * i + e;
        } else {
            const int i = (int)(q % G4); q /= G4;
            const int w = (int)(q & 3); q >>= 2;
            cb = (int)q;
            k = w * (K / 4) + (l >> 4) * (K / 16) + 4 * i + e;
        }
        const int n = 16 * cb + (l & 15);
        out[o] = mode == 0 ? U[(long)k * ldu + coff + n] : U[(long)n * ldu + coff + k];
    }
}

// ---- register-operand tile product: no LDS staging and no barrier in front of the MFMAs.
// lane (row = lane & 15, q = lane >> 4) of wave w owns k in [w*K/4 + q*K/16, +K/16) of its A row.
template <int K> __device__ __forceinline__ int a_koff(int lane, int w) { return w * (K / 4) + (lane >> 4) * (K / 16); }
// (plain 16-byte global loads from a CLAMPED row pointer: rows beyond the block's last session read a
//  valid row -- they only feed output rows that are never stored.  hipcc lowers the b128 raw-buffer
//  builtin to a single dword load on this toolchain, so the branch-free buffer form is not available.)
template <int N>      // N (multiple of 4) consecutive floats
__device__ __forceinline__ void gload_vec(float (&a)[N], const float* __restrict__ p) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const float4 v = reinterpret_cast<const float4*>(p)[i];
        a[4 * i + 0] = v.x; a[4 * i + 1] = v.y; a[4 * i + 2] = v.z; a[4 * i + 3] = v.w;
    }
}
template <int N>
__device__ __forceinline__ void gstore_vec(const float (&a)[N], float* __restrict__ p) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) reinterpret_cast<float4*>(p)[i] = make_float4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
}
template <int K>
__device__ __forceinline__ float tile_16x16_reg(const float (&a)[K / 16], const float4 (&b)[K / 64], float* __restrict__ red,
                                                int tid) {
    const int lane = tid & 63, w = tid >> 6;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);      // every load of the kernel is in flight before the first MFMA waits
#pragma unroll
    for (int i = 0; i < K / 64; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 0], b[i].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 1], b[i].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 2], b[i].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 3], b[i].w, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc0[r] + acc1[r];
    __syncthreads();
    return (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
}
// recurrent-dropout multipliers of gate g for the lane's A elements (sorted session row srow)
template <int N>
__device__ __forceinline__ void mask_vec(float (&a)[N], const float* __restrict__ rmask, int B, int H, int g, int srow, int koff,
                                         bool ok) {
    if (!ok) return;
    const float4* m = reinterpret_cast<const float4*>(rmask + ((long)g * B + srow) * H + koff);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const float4 v = m[i];
        a[4 * i + 0] *= v.x; a[4 * i + 1] *= v.y; a[4 * i + 2] *= v.z; a[4 * i + 3] *= v.w;
    }
}

__device__ __forceinline__ float4 mask4(const float* __restrict__ rmask, int B, int H, int g, int srow, int c4) {
    return reinterpret_cast<const float4*>(rmask + ((long)g * B + srow) * H)[c4];
}

template <int J, int ACT, int PHASE>
__global__ __launch_bounds__(256) void gru_step_fwd(StepArgs a_in) {
    const StepArgs& a = a_in;
    int r0, cb;
    if (!tile_of(a, r0, cb)) return;
    // PHASE 0: [z|r] = hs(xw + h_prev.U_zr), r*h_prev        grid (rows/16, 2H/16)
    // PHASE 1: h~ = act(xw_h + (r*h_prev).U_h), h = z h_prev + (1-z) h~   grid (rows/16, H/16)
    constexpr int H = 64 * J, GH = 3 * H;
    __shared__ float red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    [[maybe_unused]] const bool stamp_me = r0 == 0 && cb == 1 && w == 0;
    STAMP_REAL(6);
    STAMP(0);
    const int nact = min(16, a.bt - r0);
    const int row = tid >> 4, col = 16 * cb + (tid & 15);     // this thread's output element
    const bool ok = row < nact;
    const rsrc_t rXW = mk_rsrc(a.XW), rH = mk_rsrc(a.Hout), rG = mk_rsrc(a.gates), rA = mk_rsrc(a.aux);
    const int soG = (a.p0 + r0) * GH * 4, soH = (a.p0 + r0) * H * 4, soP = (a.pprev0 + r0) * H * 4;
    const int vg = ok ? (row * GH + col) * 4 : INVALID_OFF;
    const int vh = ok ? (row * H + col) * 4 : INVALID_OFF;
    // all loads issued up front in the order their data is needed: A rows, packed U slice, epilogue operands
    const int mg = PHASE == 1 ? 2 : (col >= H ? 1 : 0);      // gate whose recurrent-dropout mask applies to A
    float4 b[H / 64];
    float av[H / 16];
    const int arow = min(lane & 15, nact - 1), koff = a_koff<H>(lane, w);
    if (!a.first) {
        const float4* pk = reinterpret_cast<const float4*>(a.pk) + (size_t)(cb * 4 + w) * (H / 64) * 64 + lane;
#pragma unroll
        for (int i = 0; i < H / 64; ++i) b[i] = pk[i * 64];
        // A rows straight into MFMA operand registers: PHASE 0 -> h_prev = Hout[prev step], PHASE 1 -> r*h_prev = aux[this step]
        gload_vec(av, (PHASE == 0 ? a.Hout + (long)(a.pprev0 + r0 + arow) * H : a.aux + (long)(a.p0 + r0 + arow) * H) + koff);
    }
    const float xw = bload(rXW, PHASE == 0 ? vg : vg + 2 * H * 4, soG);
    float zg = 0.f, h0 = 0.f;
    if (PHASE == 1) {
        zg = bload(rG, vg, soG);
        h0 = bload(rH, a.first ? INVALID_OFF : vh, soP);
    } else if (col >= H) {
        h0 = bload(rH, (a.first || !ok) ? INVALID_OFF : (row * H + (col - H)) * 4, soP);   // h_prev for r * h_prev
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
    float acc = 0.f;
    if (!a.first) {
        if (a.rmask) mask_vec(av, a.rmask, a.B, H, mg, r0 + arow, koff, true);
#ifdef SEQREC_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(2);
#endif
        acc = tile_16x16_reg<H>(av, b, red, tid);
    }
    STAMP(3);
    if (PHASE == 0) {
        const float g = hard_sigmoid(acc + xw);
        bstore(rG, vg, soG, g);
        if (col >= H) bstore(rA, ok ? (row * H + (col - H)) * 4 : INVALID_OFF, soH, g * h0);   // publish r * h_prev
    } else {
        const float hh = act_fwd<ACT>(acc + xw, a.act_rt);
        float hn = zg * h0 + (1.f - zg) * hh;
        if (col >= a.H_real) hn = 0.f;
        bstore(rH, vh, soH, hn);
        bstore(rG, vg + 2 * H * 4, soG, hh);
    }
#ifdef SEQREC_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP(4);
    STAMP_REAL(7);
}

template <int J, int ACT, int PHASE>
__global__ __launch_bounds__(256) void gru_step_bwd(StepArgs a_in) {
    const StepArgs& a = a_in;
    int r0, cb;
    if (!tile_of(a, r0, cb)) return;
    // PHASE 0 (grid rows/16 x H/16): d = dh (1-z) act'(h~) for the whole row -> LDS; drh = d . U_h^T (own cols);
    //          dpre_z, dpre_r, dpre_h -> dPre;  dcar = dh z + drh r -> tmpc
    // PHASE 1 (grid rows/16 x H/16, skipped at t = 0): dh_prev = tmpc + [dpre_z|dpre_r] . U_zr^T -> dHc[prev token]
    constexpr int H = 64 * J, GH = 3 * H;
    constexpr int K = PHASE == 0 ? H : 2 * H;
    __shared__ float red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    [[maybe_unused]] const bool stamp_me = r0 == 0 && cb == 1 && w == 0;
    STAMP_REAL(6);
    STAMP(0);
    const int nact = min(16, a.bt - r0);
    const int row = tid >> 4, col = 16 * cb + (tid & 15);
    const bool ok = row < nact;
    const rsrc_t rDH = mk_rsrc(a.dHout), rH = mk_rsrc(a.Hout), rG = mk_rsrc(a.gates), rDP = mk_rsrc(a.dPre),
                 rC = mk_rsrc(a.dHc), rT = mk_rsrc(a.tmpc);
    const int soG = (a.p0 + r0) * GH * 4, soH = (a.p0 + r0) * H * 4, soP = (a.pprev0 + r0) * H * 4;
    const int vg = ok ? (row * GH + col) * 4 : INVALID_OFF;
    const int vh = ok ? (row * H + col) * 4 : INVALID_OFF;
    // Every load of the kernel is ISSUED before the first wait: the packed U slice (pinned by its own scheduling
    // barrier -- hipcc otherwise fetches it only after d has been computed from the A-side rows: two memory
    // latencies in series in a launch that is all latency), then the A-side rows, then the epilogue operands.
    const long pt = (long)a.p0 + r0;
    const bool aok = (lane & 15) < nact;
    const int arow = min(lane & 15, nact - 1), koff = a_koff<K>(lane, w);
    const long q = pt + arow;
    float av[K / 16];
    float dh[PHASE == 0 ? K / 16 : 1], zz[PHASE == 0 ? K / 16 : 1], hh[PHASE == 0 ? K / 16 : 1];
    // PHASE 0: rows whose session goes on at step t+1 ("carried", row < bnext) find d = dh (1-z) act'(h~) READY in dPre:
    // the thread of PHASE 1 (step t+1) that finished dh_prev[row, col] also finished d[row, col] (one more element-wise
    // product in its epilogue).  Only the rows of sessions that END here compute d themselves, from dHout alone.  The
    // stamped build showed why it matters: this launch spent 1.3 us ISSUING its 26 loads per wave (86 KB per workgroup
    // through one CU's 64 B/clk vector-memory path, every one of the 16 column-block workgroups of a row block
    // re-reading the same four A-side arrays) against 0.24 us in the forward step; now 38 KB, like the forward step.
    const bool carried = PHASE == 0 && (r0 + arow) < a.bnext;
    float4 b[K / 64];
    const float4* pk = reinterpret_cast<const float4*>(a.pk) + (size_t)(cb * 4 + w) * (K / 64) * 64 + lane;
#pragma unroll
    for (int i = 0; i < K / 64; ++i) b[i] = pk[i * 64];
    __builtin_amdgcn_sched_barrier(0);       // the U slice is requested before anything that could wait
    if constexpr (PHASE == 0) {
        // d is loaded by EVERY lane (a row that ends here reads a slot it is about to overwrite: valid memory, value unused):
        // a load inside the `carried` branch makes hipcc merge the two definitions of av with waits in front of the
        // remaining loads
        gload_vec(av, a.dPre + q * GH + 2 * H + koff);
        if (!carried) {
            gload_vec(dh, a.dHout + q * H + koff);
            gload_vec(zz, a.gates + q * GH + koff);
            gload_vec(hh, a.gates + q * GH + 2 * H + koff);
        }
    } else {
        gload_vec(av, a.dPre + q * GH + koff);
    }
    float e_dh = 0.f, e_z = 0.f, e_r = 0.f, e_hh = 0.f, e_h0 = 0.f, e_t = 0.f, p_dh = 0.f, p_z = 0.f, p_hh = 0.f;
    const int soGp = (a.pprev0 + r0) * GH * 4;
    if (PHASE == 0) {
        e_dh = bload(rDH, vh, soH) + bload(rC, (ok && r0 + row < a.bnext) ? vh : INVALID_OFF, soH);
        e_z = bload(rG, vg, soG);
        e_r = bload(rG, vg + H * 4, soG);
        e_hh = bload(rG, vg + 2 * H * 4, soG);
        e_h0 = bload(rH, a.first ? INVALID_OFF : vh, soP);
    } else {
        e_t = bload(rT, vh, soH);
        p_dh = bload(rDH, vh, soP);                       // step t-1: its dHout, z and h~ for d[row, col]
        p_z = bload(rG, vg, soGp);
        p_hh = bload(rG, vg + 2 * H * 4, soGp);
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
#ifdef SEQREC_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(2);
#endif
    if constexpr (PHASE == 0) {
        if (!carried) {
#pragma unroll
            for (int j = 0; j < K / 16; ++j) av[j] = dh[j] * (1.f - zz[j]) * act_grad<ACT>(hh[j], a.act_rt);
            if (cb == 0 && aok) gstore_vec(av, a.dPre + q * GH + 2 * H + koff);
        }
    }
    STAMP(5);
    float acc = tile_16x16_reg<K>(av, b, red, tid);
    STAMP(3);
    const int srow = min(r0 + row, a.B - 1);
    if (PHASE == 0) {
        if (a.rmask) acc *= a.rmask[((long)2 * a.B + srow) * H + col];          // d(r*h*m2) -> d(r*h)
        bstore(rDP, vg, soG, e_dh * (e_h0 - e_hh) * hard_sigmoid_grad(e_z));
        bstore(rDP, vg + H * 4, soG, acc * e_h0 * hard_sigmoid_grad(e_r));
        bstore(rT, vh, soH, e_dh * e_z + acc * e_r);
    } else {
        if (a.rmask) {   // K = 2H is split over the waves as [z z r r]: the two halves carry different masks
            acc = (red[tid] + red[256 + tid]) * a.rmask[((long)0 * a.B + srow) * H + col] +
                  (red[512 + tid] + red[768 + tid]) * a.rmask[((long)1 * a.B + srow) * H + col];
        }
        const float dcar = e_t + acc;
        bstore(rC, vh, soP, dcar);
        bstore(rDP, vg + 2 * H * 4, soGp, (p_dh + dcar) * (1.f - p_z) * act_grad<ACT>(p_hh, a.act_rt));      // d of step t-1, ready for its PHASE 0
    }
#ifdef SEQREC_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP(4);
    STAMP_REAL(7);
}

// ---------------------------------------------------------------------------------------------
// GRU, time step 0: h_prev = 0, so nothing crosses columns -- the step is pointwise.  One launch instead of the two
// widest forward launches of the scan, and the widest BPTT launch without its (all-zero-weighted) recurrent product:
//   forward   z = hs(xw_z), r = hs(xw_r), h~ = act(xw_h), h = (1 - z) h~, r*h_prev = 0
//   backward  d = dh (1 - z) act'(h~) = dpre_h;  dpre_z = -dh h~ hs'(z);  dpre_r = 0   (no carry leaves step 0)
// One thread per (session row, hidden column); grid = ceil(bt * H / 256).
// ---------------------------------------------------------------------------------------------
template <int ACT>
__global__ void gru_first_step_fwd(StepArgs a_in) {
    const StepArgs& a = a_in;
    const int H = a.H;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)a.bt * H) return;
    const int row = (int)(e / H), col = (int)(e % H);
    const long q = (long)a.p0 + row, GH = 3L * H;
    const float* xw = a.XW + q * GH + col;
    const float z = hard_sigmoid(xw[0]), r = hard_sigmoid(xw[H]), hh = act_fwd<ACT>(xw[2 * H], a.act_rt);
    float* g = a.gates + q * GH + col;
    g[0] = z; g[H] = r; g[2 * H] = hh;
    a.aux[q * H + col] = 0.f;
    a.Hout[q * H + col] = col >= a.H_real ? 0.f : (1.f - z) * hh;
}
template <int ACT>
__global__ void gru_first_step_bwd(StepArgs a_in) {
    const StepArgs& a = a_in;
    const int H = a.H;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)a.bt * H) return;
    const int row = (int)(e / H), col = (int)(e % H);
    const long q = (long)a.p0 + row, GH = 3L * H;
    float dh = a.dHout[q * H + col];
    if (row < a.bnext) dh += a.dHc[q * H + col];
    const float* g = a.gates + q * GH + col;
    const float z = g[0], hh = g[2 * H];
    float* o = a.dPre + q * GH + col;
    o[0] = dh * (0.f - hh) * hard_sigmoid_grad(z);
    o[H] = 0.f;
    o[2 * H] = dh * (1.f - z) * act_grad<ACT>(hh, a.act_rt);
}

// ---------------------------------------------------------------------------------------------
// Wide BPTT phase 0 (launches with more than 8 row blocks): workgroup = 16 session rows x 64 columns, each wave owns
// 16 columns over the FULL K (64 MFMAs, no split-K, no LDS reduce).  The A operand d = dh (1-z) act'(h~) is computed
// ONCE per workgroup (every thread 16 elements), stored by column group 0, and staged in LDS [16][K+4] -- in the
// narrow tile each of a row block's 16 column-block workgroups recomputes it from 64 KB of loads, which is what the
// issue stalls of the wide launches scale with (profiles/r02_v1_c3_scan_widths_pmc.json).  Same arithmetic order per
// output element?  No: K is summed in one chain per wave instead of four partial chains -> results differ from the
// narrow tile in the last bits (both are exact-fp32 FMA chains; parity tests cover both).
// packed: the `wide` order of pack_step_kernel.
// ---------------------------------------------------------------------------------------------
template <int J, int ACT>
__global__ __launch_bounds__(256) void gru_step_bwd0_wide(StepArgs a_in) {
    const StepArgs& a = a_in;
    int r0, cbw;
    if (!tile_of(a, r0, cbw)) return;
    constexpr int H = 64 * J, GH = 3 * H, K = H, LDA = K + 4;
    __shared__ float ab[16 * LDA];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nact = min(16, a.bt - r0);
    const long pt = (long)a.p0 + r0;
    // the wave's U slice: 16 columns x full K
    float4 b[K / 16];
    const float4* pk = reinterpret_cast<const float4*>(a.pk) + (size_t)(cbw * 4 + w) * (K / 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < K / 16; ++i) b[i] = pk[i * 64];
    __builtin_amdgcn_sched_barrier(0);
    // d, cooperatively: thread (row = tid >> 4, seg = tid & 15) owns k in [seg*K/16, +K/16)
    constexpr int KT = K / 16;
    const int drow = tid >> 4, dk0 = (tid & 15) * KT;
    const bool dok = drow < nact;
    const long dq = pt + min(drow, nact - 1);
    const bool carried = (r0 + min(drow, nact - 1)) < a.bnext;      // d of a carried row was finished by PHASE 1 of step t+1
    float dv[KT], dh[KT], zz[KT], hh[KT];
    gload_vec(dv, a.dPre + dq * GH + 2 * H + dk0);                  // every thread (see gru_step_bwd)
    if (!carried) {
        gload_vec(dh, a.dHout + dq * H + dk0);
        gload_vec(zz, a.gates + dq * GH + dk0);
        gload_vec(hh, a.gates + dq * GH + 2 * H + dk0);
    }
    // epilogue operands of this lane's 4 output elements: rows 4q + r, column col
    const int q = lane >> 4, col = 64 * cbw + 16 * w + (lane & 15);
    const rsrc_t rDH = mk_rsrc(a.dHout), rH = mk_rsrc(a.Hout), rG = mk_rsrc(a.gates), rDP = mk_rsrc(a.dPre),
                 rC = mk_rsrc(a.dHc), rT = mk_rsrc(a.tmpc);
    const int soG = (a.p0 + r0) * GH * 4, soH = (a.p0 + r0) * H * 4, soP = (a.pprev0 + r0) * H * 4;
    float e_dh[4], e_z[4], e_r[4], e_hh[4], e_h0[4];
    int vg[4], vh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;
        const bool ok = row < nact;
        vg[r] = ok ? (row * GH + col) * 4 : INVALID_OFF;
        vh[r] = ok ? (row * H + col) * 4 : INVALID_OFF;
        e_dh[r] = bload(rDH, vh[r], soH) + bload(rC, (ok && r0 + row < a.bnext) ? vh[r] : INVALID_OFF, soH);
        e_z[r] = bload(rG, vg[r], soG);
        e_r[r] = bload(rG, vg[r] + H * 4, soG);
        e_hh[r] = bload(rG, vg[r] + 2 * H * 4, soG);
        e_h0[r] = bload(rH, a.first ? INVALID_OFF : vh[r], soP);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        if (!carried) {
#pragma unroll
            for (int j = 0; j < KT; ++j) dv[j] = dh[j] * (1.f - zz[j]) * act_grad<ACT>(hh[j], a.act_rt);
            if (cbw == 0 && dok) gstore_vec(dv, a.dPre + (pt + drow) * GH + 2 * H + dk0);
        }
        float* o = ab + drow * LDA + dk0;
#pragma unroll
        for (int j = 0; j < KT / 4; ++j) reinterpret_cast<float4*>(o)[j] = make_float4(dv[4 * j], dv[4 * j + 1], dv[4 * j + 2], dv[4 * j + 3]);
    }
    __syncthreads();
    // A operand: lane (row, q) owns k in [q*K/4, +K/4) of its row
    const float* ap = ab + (lane & 15) * LDA + q * (K / 4);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < K / 16; ++i) {
        const float4 av = reinterpret_cast<const float4*>(ap)[i];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[i].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[i].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[i].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[i].w, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float acc = acc0[r] + acc1[r];
        if (a.rmask) acc *= a.rmask[((long)2 * a.B + min(r0 + 4 * q + r, a.B - 1)) * H + col];      // d(r*h*m2) -> d(r*h)
        bstore(rDP, vg[r], soG, e_dh[r] * (e_h0[r] - e_hh[r]) * hard_sigmoid_grad(e_z[r]));
        bstore(rDP, vg[r] + H * 4, soG, acc * e_h0[r] * hard_sigmoid_grad(e_r[r]));
        bstore(rT, vh[r], soH, e_dh[r] * e_z[r] + acc * e_r[r]);
    }
}

// ---------------------------------------------------------------------------------------------
// LSTM forward step (ONE launch per step): workgroup = 16 rows x 16 hidden columns; K is split over the 4 waves like
// every other step kernel (the wave's K slice of h_prev straight into MFMA operand registers) and feeds the FOUR gate
// tiles -- 8 independent accumulators per wave; partial tiles meet in LDS and thread (row, col) finishes c and h.
// The tile products and the cell arithmetic are the cluster kernel's own functions (rnn_cluster_dev.h): the two forms
// of the scan agree bit for bit.  packed: pack_step_kernel mode 0 with N = 4H (gate g's column block = g H/16 + cb).
// SimpleRNN forward step: same tile, one product.
// ---------------------------------------------------------------------------------------------
template <int J, int ACT, bool RD>
__device__ __forceinline__ void lstm_step_fwd_body(const StepArgs& a_in) {
    const StepArgs& a = a_in;
    int r0, cb;
    if (!tile_of(a, r0, cb)) return;
    constexpr int H = 64 * J, GH = 4 * H, NB = H / 64, CBN = H / 16;
    __shared__ __attribute__((aligned(16))) float red[4 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nact = min(16, a.bt - r0);
    const int row = tid >> 4, col = 16 * cb + (tid & 15);
    const bool ok = row < nact;
    const rsrc_t rXW = mk_rsrc(a.XW), rH = mk_rsrc(a.Hout), rG = mk_rsrc(a.gates), rC = mk_rsrc(a.aux);
    const int soG = (a.p0 + r0) * GH * 4, soH = (a.p0 + r0) * H * 4, soP = (a.pprev0 + r0) * H * 4;
    const int vg = ok ? (row * GH + col) * 4 : INVALID_OFF;
    const int vh = ok ? (row * H + col) * 4 : INVALID_OFF;
    float4 b[4][NB];
    float av[H / 16];
    const int arow = min(lane & 15, nact - 1), koff = a_koff<H>(lane, w);
    if (!a.first) {          // the U slices first, pinned, then the A rows: every load in flight before the first wait
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4* pk = reinterpret_cast<const float4*>(a.pk) + (size_t)((g * CBN + cb) * 4 + w) * NB * 64 + lane;
#pragma unroll
            for (int i = 0; i < NB; ++i) b[g][i] = pk[i * 64];
        }
        gload_vec(av, a.Hout + (long)(a.pprev0 + r0 + arow) * H + koff);
    }
    float xw[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xw[g] = bload(rXW, vg + g * H * 4, soG);
    const float cp = bload(rC, a.first ? INVALID_OFF : vh, soP);
    __builtin_amdgcn_sched_barrier(0);
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, unused[4];
    if (!a.first) {
        if constexpr (RD) {      // every gate reads h_prev through its own recurrent-dropout mask
            float4 b1[1][NB];
            float am[H / 16], o1[1], u1[1];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int j = 0; j < H / 16; ++j) am[j] = av[j];
                mask_vec(am, a.rmask, a.B, H, g, r0 + arow, koff, true);
#pragma unroll
                for (int i = 0; i < NB; ++i) b1[0][i] = b[g][i];
                if (g) __syncthreads();
                seqrec_cluster::cl_tiles_n<H, 1>(am, b1, red, tid, o1, u1);
                acc[g] = o1[0];
            }
        } else {
            seqrec_cluster::cl_tiles_n<H, 4>(av, b, red, tid, acc, unused);
        }
    }
    float gi, gf, gg, go, c, h;
    lstm_cell_fwd<ACT>(acc[0] + xw[0], acc[1] + xw[1], acc[2] + xw[2], acc[3] + xw[3], cp, col < a.H_real, gi, gf, gg, go, c, h, a.act_rt);
    bstore(rH, vh, soH, h);
    bstore(rC, vh, soH, c);
    bstore(rG, vg, soG, gi);
    bstore(rG, vg + H * 4, soG, gf);
    bstore(rG, vg + 2 * H * 4, soG, gg);
    bstore(rG, vg + 3 * H * 4, soG, go);
}

template <int J, int ACT> __global__ __launch_bounds__(256) void lstm_step_fwd_nd(StepArgs a) { lstm_step_fwd_body<J, ACT, false>(a); }
template <int J, int ACT> __global__ __launch_bounds__(256) void lstm_step_fwd_rd(StepArgs a) { lstm_step_fwd_body<J, ACT, true>(a); }
template <int J, int ACT>
__global__ __launch_bounds__(256) void srnn_step_fwd(StepArgs a_in) {
    const StepArgs& a = a_in;
    int r0, cb;
    if (!tile_of(a, r0, cb)) return;
    constexpr int H = 64 * J;
    __shared__ float red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nact = min(16, a.bt - r0);
    const int row = tid >> 4, col = 16 * cb + (tid & 15);
    const bool ok = row < nact;
    const rsrc_t rXW = mk_rsrc(a.XW), rH = mk_rsrc(a.Hout);
    const int soH = (a.p0 + r0) * H * 4, soP = (a.pprev0 + r0) * H * 4;
    const int vh = ok ? (row * H + col) * 4 : INVALID_OFF;
    float acc = 0.f;
    float4 b[H / 64];
    float av[H / 16];
    const int arow = min(lane & 15, nact - 1), koff = a_koff<H>(lane, w);
    if (!a.first) {
        const float4* pk = reinterpret_cast<const float4*>(a.pk) + (size_t)(cb * 4 + w) * (H / 64) * 64 + lane;
#pragma unroll
        for (int i = 0; i < H / 64; ++i) b[i] = pk[i * 64];
        gload_vec(av, a.Hout + (long)(a.pprev0 + r0 + arow) * H + koff);
    }
    const float xw = bload(rXW, vh, soH);
    __builtin_amdgcn_sched_barrier(0);       // every load issued before the first wait
    if (!a.first) {
        if (a.rmask) mask_vec(av, a.rmask, a.B, H, 0, r0 + arow, koff, true);
        acc = tile_16x16_reg<H>(av, b, red, tid);
    }
    float y = act_fwd<ACT>(acc + xw, a.act_rt);
    if (col >= a.H_real) y = 0.f;
    bstore(rH, vh, soH, y);
}

// pointwise part of one backward step: dPre[p] from dh (= dHout + carried dh), the stash and, for the
// LSTM, the carried dc (dCc).  One thread per (row, hidden col).  grid = ceil(bt*H / 256)
template <int CELL, int ACT>
__global__ void pointwise_bwd_step(StepArgs a_in) {
    const StepArgs& a = a_in;
    const int H = a.H;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)a.bt * H) return;
    const int row = (int)(e / H), col = (int)(e % H);
    const long q = (long)a.p0 + row;
    float dh = a.dHout[q * H + col];
    if (row < a.bnext) dh += a.dHc[q * H + col];
    if (CELL == SEQREC_CELL_SIMPLERNN) {
        a.dPre[q * H + col] = dh * act_grad<ACT>(a.Hout[q * H + col], a.act_rt);
    } else {
        const long GH = 4L * H;
        const float* gp = a.gates + q * GH + col;
        const float gi = gp[0], gf = gp[H], gg = gp[2 * H], go = gp[3 * H];
        const float cn = a.aux[q * H + col];
        const float cp = a.first ? 0.f : a.aux[((long)a.pprev0 + row) * H + col];
        const float dcin = row < a.bnext ? a.tmpc[q * H + col] : 0.f;      // dc carried from step t+1
        float di, df, dg, dout, dcp;
        lstm_cell_bwd<ACT>(dh, dcin, gi, gf, gg, go, cn, cp, di, df, dg, dout, dcp, a.act_rt);
        float* o = a.dPre + q * GH + col;
        o[0] = di;
        o[H] = df;
        o[2 * H] = dg;
        o[3 * H] = dout;
        if (!a.first) a.tmpc[((long)a.pprev0 + row) * H + col] = dcp;   // dc for the previous token
    }
}

// dHc[prev token][own 16 cols] = dPre[p][0:K] . packed(U^T)      (K = G*H), skipped at t = 0
template <int K>
__global__ __launch_bounds__(256) void gemm_bwd_step(StepArgs a_in, int H, int ldp) {
    const StepArgs& a = a_in;
    int r0, cb;
    if (!tile_of(a, r0, cb)) return;
    __shared__ float red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nact = min(16, a.bt - r0);
    const int row = tid >> 4, col = 16 * cb + (tid & 15);
    const bool ok = row < nact;
    float4 b[K / 64];
    const float4* pk = reinterpret_cast<const float4*>(a.pk) + (size_t)(cb * 4 + w) * (K / 64) * 64 + lane;
#pragma unroll
    for (int i = 0; i < K / 64; ++i) b[i] = pk[i * 64];
    __builtin_amdgcn_sched_barrier(0);       // U slice requested first (see gru_step_bwd)
    const long pt = (long)a.p0 + r0;
    float acc;
    {
        const int arow = min(lane & 15, nact - 1), koff = a_koff<K>(lane, w);
        float av[K / 16];
        gload_vec(av, a.dPre + (pt + arow) * ldp + koff);
        acc = tile_16x16_reg<K>(av, b, red, tid);
    }
    if (a.rmask) {
        const int srow = min(r0 + row, a.B - 1);
        if (K == H) {
            acc *= a.rmask[(long)srow * H + col];
        } else {      // LSTM: K = 4H split over the waves = gate blocks i,f,c,o, each with its own mask
            acc = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc += red[g * 256 + tid] * a.rmask[((long)g * a.B + srow) * H + col];
        }
    }
    const rsrc_t rC = mk_rsrc(a.dHc);
    bstore(rC, ok ? (row * H + col) * 4 : INVALID_OFF, (a.pprev0 + r0) * H * 4, acc);
}

bool xcd_placement() {
    static const bool on = seqrec_env("SEQREC_SCAN_XCD", 1) != 0;   // A/B switch
    return on;
}
// grid = (row blocks, column blocks) -> the 1-D placement grid of tile_of()
unsigned place_grid(dim3 grid, StepArgs& a) {
    a.cbn = (int)grid.y;
    a.xcd = xcd_placement() ? 1 : 0;
    return a.xcd ? 8u * grid.y * ((grid.x + 7) / 8) : grid.x * grid.y;
}

// One scan call = a PLAN: the ordered list of its dependent launches (kernel, grid, arguments).  The plan is either
// issued eagerly (one hipLaunchKernel per entry) or replayed through a captured hipGraph whose kernel nodes are
// REWRITTEN with this batch's exact geometry before every replay (hipGraphExecKernelNodeSetParams): the kernels
// keep taking everything from their arguments -- no device-side step table, no oversized grids -- while the host
// pays ~0.6 us per node plus one graph launch instead of ~3 us per launch (tools/graph_update_probe.hip).
struct Launch { const void* fn; unsigned grid; StepArgs a; int i0, i1; int nargs; };
typedef std::vector<Launch> Plan;

template <typename KF> int plan_step(Plan& pl, KF kern, dim3 grid, const StepArgs& a_in) {
    Launch L;
    L.fn = reinterpret_cast<const void*>(kern);
    L.a = a_in;
    L.grid = place_grid(grid, L.a);
    L.a.tag = (int)pl.size();
    L.i0 = L.i1 = 0;
    L.nargs = 1;
    pl.push_back(L);
    return 0;
}

#define STEP_DISPATCH(KERN, PHASE, GRID)                                                        \
    do {                                                                                        \
        int rc__ = SEQREC_E_SHAPE;                                                              \
        switch (J * 4 + act) {                                                                  \
            case 4 * 1 + 0: rc__ = plan_step(pl, KERN<1, 0, PHASE>, GRID, a); break;            \
            case 4 * 1 + 1: rc__ = plan_step(pl, KERN<1, 1, PHASE>, GRID, a); break;            \
            case 4 * 1 + 2: rc__ = plan_step(pl, KERN<1, 2, PHASE>, GRID, a); break;            \
            case 4 * 1 + 3: rc__ = plan_step(pl, KERN<1, 3, PHASE>, GRID, a); break;            \
            case 4 * 2 + 0: rc__ = plan_step(pl, KERN<2, 0, PHASE>, GRID, a); break;            \
            case 4 * 2 + 1: rc__ = plan_step(pl, KERN<2, 1, PHASE>, GRID, a); break;            \
            case 4 * 2 + 2: rc__ = plan_step(pl, KERN<2, 2, PHASE>, GRID, a); break;            \
            case 4 * 2 + 3: rc__ = plan_step(pl, KERN<2, 3, PHASE>, GRID, a); break;            \
            case 4 * 4 + 0: rc__ = plan_step(pl, KERN<4, 0, PHASE>, GRID, a); break;            \
            case 4 * 4 + 1: rc__ = plan_step(pl, KERN<4, 1, PHASE>, GRID, a); break;            \
            case 4 * 4 + 2: rc__ = plan_step(pl, KERN<4, 2, PHASE>, GRID, a); break;            \
            case 4 * 4 + 3: rc__ = plan_step(pl, KERN<4, 3, PHASE>, GRID, a); break;            \
            case 4 * 8 + 0: rc__ = plan_step(pl, KERN<8, 0, PHASE>, GRID, a); break;            \
            case 4 * 8 + 1: rc__ = plan_step(pl, KERN<8, 1, PHASE>, GRID, a); break;            \
            case 4 * 8 + 2: rc__ = plan_step(pl, KERN<8, 2, PHASE>, GRID, a); break;            \
            case 4 * 8 + 3: rc__ = plan_step(pl, KERN<8, 3, PHASE>, GRID, a); break;            \
        }                                                                                       \
        if (rc__) return rc__;                                                                  \
    } while (0)

#define CELL_DISPATCH(KERN, GRID)                                                              \
    do {                                                                                        \
        int rc__ = SEQREC_E_SHAPE;                                                              \
        switch (J * 4 + act) {                                                                  \
            case 4 * 1 + 0: rc__ = plan_step(pl, KERN<1, 0>, GRID, a); break;                   \
            case 4 * 1 + 1: rc__ = plan_step(pl, KERN<1, 1>, GRID, a); break;                   \
            case 4 * 1 + 2: rc__ = plan_step(pl, KERN<1, 2>, GRID, a); break;                   \
            case 4 * 1 + 3: rc__ = plan_step(pl, KERN<1, 3>, GRID, a); break;                   \
            case 4 * 2 + 0: rc__ = plan_step(pl, KERN<2, 0>, GRID, a); break;                   \
            case 4 * 2 + 1: rc__ = plan_step(pl, KERN<2, 1>, GRID, a); break;                   \
            case 4 * 2 + 2: rc__ = plan_step(pl, KERN<2, 2>, GRID, a); break;                   \
            case 4 * 2 + 3: rc__ = plan_step(pl, KERN<2, 3>, GRID, a); break;                   \
            case 4 * 4 + 0: rc__ = plan_step(pl, KERN<4, 0>, GRID, a); break;                   \
            case 4 * 4 + 1: rc__ = plan_step(pl, KERN<4, 1>, GRID, a); break;                   \
            case 4 * 4 + 2: rc__ = plan_step(pl, KERN<4, 2>, GRID, a); break;                   \
            case 4 * 4 + 3: rc__ = plan_step(pl, KERN<4, 3>, GRID, a); break;                   \
            case 4 * 8 + 0: rc__ = plan_step(pl, KERN<8, 0>, GRID, a); break;                   \
            case 4 * 8 + 1: rc__ = plan_step(pl, KERN<8, 1>, GRID, a); break;                   \
            case 4 * 8 + 2: rc__ = plan_step(pl, KERN<8, 2>, GRID, a); break;                   \
            case 4 * 8 + 3: rc__ = plan_step(pl, KERN<8, 3>, GRID, a); break;                   \
        }                                                                                       \
        if (rc__) return rc__;                                                                  \
    } while (0)

template <int CELL> int plan_pointwise(Plan& pl, int act, const StepArgs& a) {
    Launch L;
    L.fn = act == 0 ? reinterpret_cast<const void*>(pointwise_bwd_step<CELL, 0>)
         : act == 1 ? reinterpret_cast<const void*>(pointwise_bwd_step<CELL, 1>)
         : act == 2 ? reinterpret_cast<const void*>(pointwise_bwd_step<CELL, 2>)
                    : reinterpret_cast<const void*>(pointwise_bwd_step<CELL, 3>);
    L.a = a;
    L.a.tag = (int)pl.size();
    L.grid = (unsigned)(((long)a.bt * a.H + 255) / 256);
    L.i0 = L.i1 = 0;
    L.nargs = 1;
    pl.push_back(L);
    return 0;
}

int plan_gru_first(Plan& pl, int act, bool bwd, const StepArgs& a) {
    Launch L;
    if (bwd) L.fn = act == 0 ? reinterpret_cast<const void*>(gru_first_step_bwd<0>)
                  : act == 1 ? reinterpret_cast<const void*>(gru_first_step_bwd<1>)
                  : act == 2 ? reinterpret_cast<const void*>(gru_first_step_bwd<2>) : reinterpret_cast<const void*>(gru_first_step_bwd<3>);
    else L.fn = act == 0 ? reinterpret_cast<const void*>(gru_first_step_fwd<0>)
              : act == 1 ? reinterpret_cast<const void*>(gru_first_step_fwd<1>)
              : act == 2 ? reinterpret_cast<const void*>(gru_first_step_fwd<2>) : reinterpret_cast<const void*>(gru_first_step_fwd<3>);
    L.a = a;
    L.a.tag = (int)pl.size();
    L.grid = (unsigned)(((long)a.bt * a.H + 255) / 256);
    L.i0 = L.i1 = 0;
    L.nargs = 1;
    pl.push_back(L);
    return 0;
}

int plan_gemm_bwd(Plan& pl, int K, dim3 grid2, const StepArgs& a_in, int H, int ldp) {
    Launch L;
    switch (K) {
        case 64: L.fn = reinterpret_cast<const void*>(gemm_bwd_step<64>); break;
        case 128: L.fn = reinterpret_cast<const void*>(gemm_bwd_step<128>); break;
        case 256: L.fn = reinterpret_cast<const void*>(gemm_bwd_step<256>); break;
        case 512: L.fn = reinterpret_cast<const void*>(gemm_bwd_step<512>); break;
        case 1024: L.fn = reinterpret_cast<const void*>(gemm_bwd_step<1024>); break;
        case 2048: L.fn = reinterpret_cast<const void*>(gemm_bwd_step<2048>); break;
        default: return SEQREC_E_SHAPE;
    }
    L.a = a_in;
    L.grid = place_grid(grid2, L.a);
    L.a.tag = (int)pl.size();
    L.i0 = H; L.i1 = ldp;
    L.nargs = 3;
    pl.push_back(L);
    return 0;
}

bool ok_shape(int cell, int act, int H, int H_real, int T, int B) {
    if (cell < 0 || cell > 2 || act < 0 || act > SEQREC_ACT_ELU) return false;
    if (!(H == 64 || H == 128 || H == 256 || H == 512)) return false;
    if (H_real < 1 || H_real > H || T < 0 || B < 0) return false;
    if ((long)B * T * 4 * H * 4 >= 0x7FFFFFF0L) return false;
    return true;
}

int issue_eager(Plan& pl, hipStream_t st) {
    for (Launch& L : pl) {
        void* argv[3] = {&L.a, &L.i0, &L.i1};
        const hipError_t e = hipLaunchKernel(L.fn, dim3(L.grid), dim3(256), argv, 0, st);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // namespace

// layouts, in this order inside upack: fwd [z|r] (H x 2H), fwd h (H x H), bwd U_h^T (H x H), bwd [U_z U_r]^T (2H x H)
static int pack_u_impl(int cell, int H, const float* U, float* upack, const SampleJob* sample, hipStream_t st, BatchJob* batch = nullptr) {
    if (cell < 0 || cell > 2) return SEQREC_E_UNSUPPORTED;
    if (!(H == 64 || H == 128 || H == 256 || H == 512)) return SEQREC_E_SHAPE;
    if (!U || !upack) return SEQREC_E_ARG;
    const long HH = (long)H * H;
    SampleJob sj = {};
    sj.njobs = -1;
    if (sample) sj = *sample;
    const unsigned gx = sample ? (unsigned)std::max(128, std::min(1024, (sample->K + 3) / 4)) : 128u;
    const unsigned extra = sample ? 1u : 0u;
    PackStepArgs pa = {};
    pa.U = U; pa.out = upack;
    int nj;
    if (cell == SEQREC_CELL_LSTM) {
        pa.ldu = 4 * H;
        pa.job[0] = PackStepJob{0, H, 4 * H, 0, 0};               // fwd: [U_i U_f U_c U_o], N = 4H column blocks
        pa.job[1] = PackStepJob{0, 4 * H, H, 1, 4 * HH};          // bwd: U^T, K = 4H
        nj = 2;
    } else if (cell == SEQREC_CELL_SIMPLERNN) {
        pa.ldu = H;
        pa.job[0] = PackStepJob{0, H, H, 0, 0};
        pa.job[1] = PackStepJob{0, H, H, 1, HH};
        nj = 2;
    } else {
        pa.ldu = 3 * H;
        pa.job[0] = PackStepJob{0, H, 2 * H, 0, 0};
        pa.job[1] = PackStepJob{2 * H, H, H, 0, 2 * HH};
        pa.job[2] = PackStepJob{2 * H, H, H, 1, 3 * HH};
        pa.job[3] = PackStepJob{0, 2 * H, H, 1, 4 * HH};
        pa.job[4] = PackStepJob{2 * H, H, H, 1, 6 * HH, 1};        // U_h^T for the wide BPTT tile
        nj = 5;
    }
    if (sample) sj.njobs = nj;
    if (batch) {
        batch->row = nj + (int)extra;
        batch->gbx = (std::max(batch->B, batch->T + 1) + 255) / 256;
        const unsigned gxb = std::max(gx, (unsigned)(batch->T * batch->gbx));
        hipLaunchKernelGGL(pack_step_batch_kernel, dim3(gxb, nj + extra + 1), dim3(256), 0, st, pa, sj, *batch);
    } else {
        hipLaunchKernelGGL(pack_step_kernel, dim3(gx, nj + extra), dim3(256), 0, st, pa, sj);
    }
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_rnn_pack_u_stepwise(int cell, int H, const float* U, float* upack, void* stream) {
    return pack_u_impl(cell, H, U, upack, nullptr, as_stream(stream));
}
// seqrec_rnn_pack_u_stepwise + seqrec_sample_gather in ONE launch (both open a sampled-softmax training step and depend
// only on the weights): same outputs as the two calls
extern "C" int seqrec_rnn_pack_u_sample(int cell, int H, const float* U, float* upack, uint64_t seed, uint64_t step, int K,
                                        const uint32_t* thresh, const int32_t* alias, int V, const float* table, int width,
                                        const float* logq, int32_t* neg_out, float* rows_out, float* logq_out, void* stream) {
    if (K <= 0 || V <= 0 || width <= 0) return SEQREC_E_ARG;
    if (!thresh || !alias || !table || !neg_out || !rows_out || (logq_out && !logq)) return SEQREC_E_ARG;
    if ((width & 3) == 0 && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(rows_out)) & 15)) return SEQREC_E_ARG;
    SampleJob sj = {};
    sj.key = key64(seed, 1); sj.step = step; sj.K = K; sj.V = V; sj.width = width;
    sj.thresh = thresh; sj.alias = alias; sj.table = table; sj.logq = logq; sj.neg = neg_out; sj.rows = rows_out; sj.lq = logq_out;
    return pack_u_impl(cell, H, U, upack, &sj, as_stream(stream));
}

// ... + seqrec_pack_batch_host in the same launch: the three openers of a training step on a batch drawn from an HBM-resident data set
extern "C" int seqrec_rnn_pack_u_sample_batch(int cell, int H, const float* U, float* upack, uint64_t seed, uint64_t step, int K,
                                              const uint32_t* thresh, const int32_t* alias, int V, const float* table, int width,
                                              const float* logq, int32_t* neg_out, float* rows_out, float* logq_out,
                                              const int32_t* flat, const int64_t* starts, const int32_t* sess_host,
                                              const int32_t* step_off_host, int B, int T, int32_t* sess_out, int32_t* step_off_out,
                                              int32_t* ids, int32_t* tgt, int32_t* prev, void* stream) {
    if (K <= 0 || V <= 0 || width <= 0 || B <= 0 || T <= 0) return SEQREC_E_ARG;
    if (!thresh || !alias || !table || !neg_out || !rows_out || (logq_out && !logq)) return SEQREC_E_ARG;
    if ((width & 3) == 0 && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(rows_out)) & 15)) return SEQREC_E_ARG;
    if ((long)B + T + 1 > PACK_MERGED_MAX) return SEQREC_E_SHAPE;
    if (!flat || !starts || !sess_host || !step_off_host || !sess_out || !step_off_out || !ids || !tgt || !prev) return SEQREC_E_ARG;
    if (step_off_host[T] < 0 || step_off_host[0] != 0) return SEQREC_E_ARG;
    SampleJob sj = {};
    sj.key = key64(seed, 1); sj.step = step; sj.K = K; sj.V = V; sj.width = width;
    sj.thresh = thresh; sj.alias = alias; sj.table = table; sj.logq = logq; sj.neg = neg_out; sj.rows = rows_out; sj.lq = logq_out;
    BatchJob bj = {};
    bj.flat = flat; bj.starts = reinterpret_cast<const long*>(starts); bj.B = B; bj.T = T;
    bj.sess_out = sess_out; bj.step_off_out = step_off_out; bj.ids = ids; bj.tgt = tgt; bj.prev = prev;
    for (int i = 0; i <= T; ++i) bj.v[i] = step_off_host[i];
    for (int i = 0; i < B; ++i) bj.v[T + 1 + i] = sess_host[i];
    return pack_u_impl(cell, H, U, upack, &sj, as_stream(stream), &bj);
}

// ---- launch-graph cache (hidden state of the library, part 2 of 2): per (stream, launch SEQUENCE -- the ordered kernel
// functions of a plan: cell, activation, H, direction and T decide it) one hipGraph and a small RING of executables
// instantiated from it.  A replay rewrites every node of an executable with the batch's arguments
// (hipGraphExecKernelNodeSetParams) and launches it.  Nothing says that an executable whose previous launch is still
// queued keeps the arguments it was launched with when its nodes are rewritten, so that never happens: every launch
// records an event behind itself, an executable is only rewritten once its event has completed, a free one is taken
// from the ring (grown to GRAPH_RING executables, then the host waits for the oldest).  A training loop that enqueues
// steps without host syncs (bench.py, fit_generator) is the case this is for; tests/test_gpu_ops.py replays two
// DIFFERENT batches back to back behind a long kernel and compares with the eager issue.
namespace {
constexpr size_t GRAPH_RING = 4;
struct GraphExec { hipGraphExec_t exec; hipEvent_t done; bool launched; };
struct GraphEntry { hipGraph_t graph; std::vector<hipGraphNode_t> nodes; std::vector<GraphExec> ring; size_t next; };   // the node handles live in `graph`
typedef std::pair<hipStream_t, std::vector<const void*>> GraphKey;
std::map<GraphKey, GraphEntry> g_graphs;
std::mutex g_graph_mu;

void destroy_entry(GraphEntry& ent) {
    for (GraphExec& x : ent.ring) {
        if (x.launched) (void)hipEventSynchronize(x.done);       // never destroy an executable that may still be queued
        (void)hipEventDestroy(x.done);
        (void)hipGraphExecDestroy(x.exec);
    }
    (void)hipGraphDestroy(ent.graph);
}

void node_params(Launch& L, void** argv, hipKernelNodeParams& p) {
    argv[0] = &L.a; argv[1] = &L.i0; argv[2] = &L.i1;
    p = hipKernelNodeParams{};
    p.func = const_cast<void*>(L.fn);
    p.gridDim = dim3(L.grid); p.blockDim = dim3(256); p.sharedMemBytes = 0;
    p.kernelParams = argv; p.extra = nullptr;
}

int add_exec(GraphEntry& ent) {
    GraphExec x{};
    hipError_t e = hipGraphInstantiate(&x.exec, ent.graph, nullptr, nullptr, 0);
    if (e != hipSuccess) return (int)e;
    e = hipEventCreateWithFlags(&x.done, hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipGraphExecDestroy(x.exec); return (int)e; }
    x.launched = false;
    ent.ring.push_back(x);
    return 0;
}

int issue_graph(Plan& pl, hipStream_t st) {
    GraphKey key;
    key.first = st;
    key.second.resize(pl.size());
    for (size_t i = 0; i < pl.size(); ++i) key.second[i] = pl[i].fn;
    std::lock_guard<std::mutex> lk(g_graph_mu);
    auto it = g_graphs.find(key);
    hipError_t e;
    if (it == g_graphs.end()) {
        // build the chain explicitly (node i depends on node i - 1): the node handles are then known in launch order
        GraphEntry ent;
        e = hipGraphCreate(&ent.graph, 0);
        if (e != hipSuccess) return (int)e;
        ent.nodes.reserve(pl.size());
        ent.next = 0;
        for (size_t i = 0; i < pl.size(); ++i) {
            void* argv[3];
            hipKernelNodeParams p;
            node_params(pl[i], argv, p);
            hipGraphNode_t node;
            e = hipGraphAddKernelNode(&node, ent.graph, i ? &ent.nodes[i - 1] : nullptr, i ? 1 : 0, &p);
            if (e != hipSuccess) { (void)hipGraphDestroy(ent.graph); return (int)e; }
            ent.nodes.push_back(node);
        }
        if (g_graphs.size() >= 512) {                      // bounded: drop everything, rebuild on demand
            for (auto& kv : g_graphs) destroy_entry(kv.second);
            g_graphs.clear();
        }
        it = g_graphs.emplace(key, ent).first;
    }
    GraphEntry& ent = it->second;
    // an executable that is not in flight: the next one of the ring if its last launch has completed, else a new one, else wait
    GraphExec* x = nullptr;
    for (size_t k = 0; k < ent.ring.size() && !x; ++k) {
        GraphExec& c = ent.ring[(ent.next + k) % ent.ring.size()];
        if (!c.launched || hipEventQuery(c.done) == hipSuccess) { x = &c; ent.next = (ent.next + k + 1) % ent.ring.size(); }
    }
    if (!x && ent.ring.size() < GRAPH_RING) {
        const int rc = add_exec(ent);
        if (rc) return rc;
        x = &ent.ring.back();
        ent.next = 0;
    }
    if (!x) {
        x = &ent.ring[ent.next];
        ent.next = (ent.next + 1) % ent.ring.size();
        e = hipEventSynchronize(x->done);
        if (e != hipSuccess) return (int)e;
    }
    for (size_t i = 0; i < pl.size(); ++i) {
        void* argv[3];
        hipKernelNodeParams p;
        node_params(pl[i], argv, p);
        e = hipGraphExecKernelNodeSetParams(x->exec, ent.nodes[i], &p);
        if (e != hipSuccess) return (int)e;
    }
    e = hipGraphLaunch(x->exec, st);
    if (e != hipSuccess) return (int)e;
    x->launched = true;
    e = hipEventRecord(x->done, st);
    return e == hipSuccess ? 0 : (int)e;
}
}  // namespace

#ifdef SEQREC_STAMP
extern "C" int seqrec_debug_stamps(unsigned long long* host_out, int clear) {
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 256 * 8);
    if (e != hipSuccess) return (int)e;
    if (clear) {
        static unsigned long long zero[256 * 8] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), zero, sizeof(zero));
    }
    return e == hipSuccess ? 0 : (int)e;
}
#endif

extern "C" int seqrec_graph_cache_clear(void) {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    for (auto& kv : g_graphs) destroy_entry(kv.second);
    g_graphs.clear();
    return 0;
}
// Frees what the library keeps for `stream`: the cluster scans' exchange flags and the launch graphs captured for it.
// Synchronises the stream first.  (A stream that is destroyed without this call leaks 16 KB of device memory and its graphs.)
extern "C" int seqrec_release_stream(void* stream) {
    hipStream_t st = as_stream(stream);
    const hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    seqrec_cluster_release_stream(st);
    std::lock_guard<std::mutex> lk(g_graph_mu);
    for (auto it = g_graphs.begin(); it != g_graphs.end();) {
        if (it->first.first == st) { destroy_entry(it->second); it = g_graphs.erase(it); }
        else ++it;
    }
    return 0;
}

extern "C" int seqrec_rnn_fwd_stepwise(int cell, int act, int H, int H_real, int T, int B,
                                       const int32_t* step_off, const int32_t* step_off_host, const float* XW,
                                       float* Hout, float* gates, float* aux, const float* upack,
                                       const float* rmask, int use_graph, void* stream) {
    (void)step_off;                                             // kept in the signature; the kernels take their geometry as arguments
    if (!ok_shape(cell, act, H, H_real, T, B)) return SEQREC_E_SHAPE;
    if (T == 0 || B == 0) return 0;
    if (!step_off_host || !XW || !Hout || !upack) return SEQREC_E_ARG;
    if (cell != SEQREC_CELL_SIMPLERNN && (!gates || !aux)) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    const int J = H / 64;
    const long HH = (long)H * H;
    const int32_t* soh = step_off_host;
    const int act_rt = act;
    if (act <= SEQREC_ACT_LINEAR) {                   // cluster form: one launch, in-kernel exchange (rnn_cluster.hip, rnn_cluster2.hip)
        int rc = 0;
        if (seqrec_cluster_fwd(cell, act, H, H_real, T, B, soh, XW, Hout, gates, aux, upack, rmask, st, &rc)) return rc;
    } else {
        act = SEQREC_ACT_OTHER;                       // sigmoid ... elu: the shared instance of the step-wise kernels, kind in StepArgs
    }
    Plan pl;
    pl.reserve(2 * (size_t)T);
    StepArgs a = {};
    a.H = H; a.H_real = H_real; a.XW = XW; a.Hout = Hout; a.gates = gates; a.aux = aux;
    a.rmask = rmask; a.B = B; a.act_rt = act_rt;
    for (int t = 0; t < T; ++t) {
        const int bt = soh[t + 1] - soh[t];
        if (bt <= 0) break;
        a.p0 = soh[t]; a.bt = bt; a.pprev0 = t > 0 ? soh[t - 1] : 0; a.first = t == 0;
        a.bnext = t + 1 < T ? soh[t + 2] - soh[t + 1] : 0;
        const unsigned rb = (unsigned)((bt + 15) / 16);
        a.pk = upack;
        if (cell == SEQREC_CELL_GRU && t == 0) {
            plan_gru_first(pl, act, false, a);                     // h_prev = 0: the whole step is pointwise
        } else if (cell == SEQREC_CELL_GRU) {
            STEP_DISPATCH(gru_step_fwd, 0, dim3(rb, 2 * H / 16));
            a.pk = upack + 2 * HH;
            STEP_DISPATCH(gru_step_fwd, 1, dim3(rb, H / 16));
        } else if (cell == SEQREC_CELL_LSTM) {
            if (rmask) { CELL_DISPATCH(lstm_step_fwd_rd, dim3(rb, H / 16)); }
            else { CELL_DISPATCH(lstm_step_fwd_nd, dim3(rb, H / 16)); }
        } else {
            CELL_DISPATCH(srnn_step_fwd, dim3(rb, H / 16));
        }
    }
    return use_graph ? issue_graph(pl, st) : issue_eager(pl, st);
}

// workspace: 2 * N_tok * H floats (carried dh per token, dcar / carried dc per token)
namespace {
// dHout = sum of the parts (the forms of the scan that read dHout from ONE array): same order of additions as the reduce
// launch of the producing GEMM (slabs in slab order, then the row term)
__global__ void dh_parts_sum_kernel(seqrec_dh_parts p, int H, long total, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.n_slabs; ++s) v += p.slabs[(long)s * p.slab_stride + i];
        if (p.add_table) {
            const long q = i / H;
            const int ix = p.add_index[q];
            v += ix < 0 ? 0.f : (p.add_scale ? p.add_scale[q] : 1.f) * p.add_table[(long)ix * p.add_ld + (i - q * H)];
        }
        out[i] = v;
    }
}
}  // namespace

static int rnn_bwd_stepwise_impl(int cell, int act, int H, int H_real, int T, int B, const int32_t* step_off_host, int64_t n_tok,
                                 const float* dHout, const seqrec_dh_parts* parts, float* dh_scratch, const float* Hout,
                                 const float* gates, const float* aux, float* dPre, const float* upack, float* workspace,
                                 const float* rmask, int use_graph, void* stream);

extern "C" int seqrec_rnn_bwd_stepwise(int cell, int act, int H, int H_real, int T, int B,
                                       const int32_t* step_off, const int32_t* step_off_host, int64_t n_tok,
                                       const float* dHout, const float* Hout, const float* gates, const float* aux,
                                       float* dPre, const float* upack, float* workspace, const float* rmask,
                                       int use_graph, void* stream) {
    (void)step_off;
    return rnn_bwd_stepwise_impl(cell, act, H, H_real, T, B, step_off_host, n_tok, dHout, nullptr, nullptr, Hout, gates, aux, dPre,
                                 upack, workspace, rmask, use_graph, stream);
}
extern "C" int seqrec_rnn_bwd_stepwise_parts(int cell, int act, int H, int H_real, int T, int B,
                                             const int32_t* step_off, const int32_t* step_off_host, int64_t n_tok,
                                             const seqrec_dh_parts* parts, float* dHout_scratch, const float* Hout,
                                             const float* gates, const float* aux, float* dPre, const float* upack,
                                             float* workspace, const float* rmask, int use_graph, void* stream) {
    (void)step_off;
    if (!parts || !parts->slabs || parts->n_slabs < 1 || !dHout_scratch) return SEQREC_E_ARG;
    if (parts->n_slabs > 1 && parts->slab_stride < n_tok * H) return SEQREC_E_ARG;
    if (parts->add_table && (!parts->add_index || parts->add_ld < H)) return SEQREC_E_ARG;
    return rnn_bwd_stepwise_impl(cell, act, H, H_real, T, B, step_off_host, n_tok, nullptr, parts, dHout_scratch, Hout, gates, aux,
                                 dPre, upack, workspace, rmask, use_graph, stream);
}

static int rnn_bwd_stepwise_impl(int cell, int act, int H, int H_real, int T, int B, const int32_t* step_off_host, int64_t n_tok,
                                 const float* dHout, const seqrec_dh_parts* parts, float* dh_scratch, const float* Hout,
                                 const float* gates, const float* aux, float* dPre, const float* upack, float* workspace,
                                 const float* rmask, int use_graph, void* stream) {
    if (!ok_shape(cell, act, H, H_real, T, B)) return SEQREC_E_SHAPE;
    if (T == 0 || B == 0) return 0;
    if (!step_off_host || (!dHout && !parts) || !Hout || !dPre || !upack || !workspace || n_tok <= 0) return SEQREC_E_ARG;
    if (cell != SEQREC_CELL_SIMPLERNN && (!gates || !aux)) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    const int J = H / 64;
    const long HH = (long)H * H;
    const int32_t* soh = step_off_host;
    // cluster form: one launch, in-kernel exchange (rnn_cluster.hip, rnn_cluster2.hip).  Only the GRU kernel adds the parts
    // of dHout in its own loads; every other form reads dHout from one array
    const int act_rt = act;
    const bool core_act = act <= SEQREC_ACT_LINEAR;    // sigmoid ... elu exist in the step-wise form only (one shared kernel instance)
    if (core_act && (!parts || cell == SEQREC_CELL_GRU)) {
        int rc = 0;
        if (seqrec_cluster_bwd(cell, act, H, H_real, T, B, soh, dHout, Hout, gates, aux, dPre, upack, rmask, st, &rc, parts)) return rc;
    }
    if (parts) {
        const long total = (long)soh[T] * H;
        if (total > 0) {
            hipLaunchKernelGGL(dh_parts_sum_kernel, dim3((unsigned)std::min<long>(2048, (total + 255) / 256)), dim3(256), 0, st, *parts, H,
                               total, dh_scratch);
            SEQREC_LAUNCH_CHECK();
        }
        dHout = dh_scratch;
        if (core_act && cell != SEQREC_CELL_GRU) {
            int rc = 0;
            if (seqrec_cluster_bwd(cell, act, H, H_real, T, B, soh, dHout, Hout, gates, aux, dPre, upack, rmask, st, &rc, nullptr)) return rc;
        }
    }
    if (!core_act) act = SEQREC_ACT_OTHER;
    Plan pl;
    pl.reserve(2 * (size_t)T);
    StepArgs a = {};
    a.H = H; a.H_real = H_real; a.Hout = const_cast<float*>(Hout); a.gates = const_cast<float*>(gates);
    a.aux = const_cast<float*>(aux); a.dHout = dHout; a.dPre = dPre;
    a.dHc = workspace; a.tmpc = workspace + n_tok * H;
    a.rmask = rmask; a.B = B; a.act_rt = act_rt;
    for (int t = T - 1; t >= 0; --t) {
        const int bt = soh[t + 1] - soh[t];
        if (bt <= 0) continue;
        a.p0 = soh[t]; a.bt = bt; a.pprev0 = t > 0 ? soh[t - 1] : 0; a.first = t == 0;
        a.bnext = t + 1 < T ? soh[t + 2] - soh[t + 1] : 0;
        const unsigned rb = (unsigned)((bt + 15) / 16);
        if (cell == SEQREC_CELL_GRU && t == 0) {
            plan_gru_first(pl, act, true, a);                      // no recurrent product at step 0
        } else if (cell == SEQREC_CELL_GRU) {
            static const int wide_rb = (int)seqrec_env("SEQREC_SCAN_WIDE_RB", 9);   // A/B switch (0 = never)
            if (wide_rb > 0 && (int)rb >= wide_rb && H >= 128) {
                a.pk = upack + 6 * HH;
                CELL_DISPATCH(gru_step_bwd0_wide, dim3(rb, H / 64));
            } else {
                a.pk = upack + 3 * HH;
                STEP_DISPATCH(gru_step_bwd, 0, dim3(rb, H / 16));
            }
            if (t > 0) {
                a.pk = upack + 4 * HH;
                STEP_DISPATCH(gru_step_bwd, 1, dim3(rb, H / 16));
            }
        } else {
            const int G = cell == SEQREC_CELL_LSTM ? 4 : 1;
            int rc = cell == SEQREC_CELL_LSTM ? plan_pointwise<SEQREC_CELL_LSTM>(pl, act, a)
                                              : plan_pointwise<SEQREC_CELL_SIMPLERNN>(pl, act, a);
            if (rc) return rc;
            if (t > 0) {
                a.pk = upack + (long)G * HH;
                if ((rc = plan_gemm_bwd(pl, G * H, dim3(rb, H / 16), a, H, G * H))) return rc;
            }
        }
    }
    return use_graph ? issue_graph(pl, st) : issue_eager(pl, st);
}

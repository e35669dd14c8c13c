// fp32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32): exact fp32
// (bitwise a k-ordered fmaf chain), so the 1e-3 parity budget of the hot path is untouched.
//
// Block = 256 threads = 4 waves in a 2x2 grid; block tile BM x BN (64 or 128 each), BK = 16.
// Both operand tiles live k-major in LDS ([k][m] / [k][n], row stride R+2 floats) so that the
// MFMA operand read  A[i = lane&31][k = lane>>5]  is one conflict-free ds_read_b32 per lane.
// Global->LDS staging goes through registers (issue next tile's loads, compute, then write),
// one barrier per K tile, two LDS buffers.
//
// Roofline: MFMA-bound, 157.3 TFLOP/s fp32; algorithmic flops = 2*M*N*K.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    long M, N, K, lda, ldb, ldc;
    long k_per_split;      // multiple of 16
    int accumulate;        // C += result (only when splits == 1)
    int a_vec, b_vec;      // 16-byte vector loads legal for this operand
    int tiles_n;
    // epilogue 1 (Recall@K): rank[row] += #{col != tgt[row] : acc + bias[col] > thr[row]}
    int epi;
    const int* tgt; const float* thr; int* rank;
    // fused gather of the A operand (AIDX kernels): A_KC  -> row m of A is A[a_idx[m] * lda + k]  (x.W with x = E[ids])
    //                                               !A_KC -> slice k of A is A[a_idx[k] * lda + m] (Hout[prev]^T . dPre)
    // a negative index is an all-zero row
    const int* a_idx;
    // fused row add where the final C is written: C[m, :] += add_scale[m] * add_table[add_idx[m] * add_ld + :]
    // (dH += dlt * Eout[tgt]); add_idx < 0 adds nothing
    const float* add_table; const int* add_idx; const float* add_scale; long add_ld;
};
__device__ __forceinline__ float row_add(const GemmArgs& g, long row, long col) {
    if (!g.add_table) return 0.f;
    const int id = g.add_idx[row];
    return id < 0 ? 0.f : (g.add_scale ? g.add_scale[row] : 1.f) * g.add_table[(long)id * g.add_ld + col];
}

constexpr int BK = 16;          // K granule of split-K bookkeeping; kernels use BKT = 16 or 32

// Load the 4 consecutive-k (KCONTIG) or consecutive-r (!KCONTIG) elements that thread `idx` owns.
template <int BR, int BKT, bool KCONTIG>
__device__ __forceinline__ float4 load_tile4(const float* __restrict__ X, long ld, long r0, long k0,
                                             long R, long Kend, int idx, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KCONTIG) {
        const int r = idx / (BKT / 4), kq = idx % (BKT / 4);
        const long gr = r0 + r, gk = k0 + 4 * kq;
        if (gr < R) {
            const float* p = X + gr * ld + gk;
            if (vec_ok && gk + 3 < Kend) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (gk + 0 < Kend) v.x = p[0];
                if (gk + 1 < Kend) v.y = p[1];
                if (gk + 2 < Kend) v.z = p[2];
                if (gk + 3 < Kend) v.w = p[3];
            }
        }
    } else {
        const int k = idx / (BR / 4), rq = idx % (BR / 4);
        const long gk = k0 + k, gr = r0 + 4 * rq;
        if (gk < Kend) {
            const float* p = X + gk * ld + gr;
            if (vec_ok && gr + 3 < R) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (gr + 0 < R) v.x = p[0];
                if (gr + 1 < R) v.y = p[1];
                if (gr + 2 < R) v.z = p[2];
                if (gr + 3 < R) v.w = p[3];
            }
        }
    }
    return v;
}

// interior tiles (whole tile inside the matrix, 16-byte loads legal): no predicates, one dwordx4 per thread
template <int BR, int BKT, bool KCONTIG>
__device__ __forceinline__ float4 load_tile4_fast(const float* __restrict__ X, long ld, long r0, long k0, int idx) {
    if (KCONTIG) {
        const int r = idx / (BKT / 4), kq = idx % (BKT / 4);
        return *reinterpret_cast<const float4*>(X + (r0 + r) * ld + k0 + 4 * kq);
    } else {
        const int k = idx / (BR / 4), rq = idx % (BR / 4);
        return *reinterpret_cast<const float4*>(X + (k0 + k) * ld + r0 + 4 * rq);
    }
}

template <int BR, int BKT, bool KCONTIG>
__device__ __forceinline__ void store_tile4(float* __restrict__ S, int idx, float4 v) {
    constexpr int LD = BR + 2;
    if (KCONTIG) {
        const int r = idx / (BKT / 4), kq = idx % (BKT / 4);
        S[(4 * kq + 0) * LD + r] = v.x;
        S[(4 * kq + 1) * LD + r] = v.y;
        S[(4 * kq + 2) * LD + r] = v.z;
        S[(4 * kq + 3) * LD + r] = v.w;
    } else {
        const int k = idx / (BR / 4), rq = idx % (BR / 4);
        float* q = S + k * LD + 4 * rq;
        q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
    }
}

// 4 floats of one (possibly gathered) A row / slice starting at p; zeros for a null row
__device__ __forceinline__ float4 load_row4(const float* __restrict__ p, long c0, long cend, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!p) return v;
    if (vec_ok && c0 + 3 < cend) return *reinterpret_cast<const float4*>(p + c0);
    if (c0 + 0 < cend) v.x = p[c0 + 0];
    if (c0 + 1 < cend) v.y = p[c0 + 1];
    if (c0 + 2 < cend) v.z = p[c0 + 2];
    if (c0 + 3 < cend) v.w = p[c0 + 3];
    return v;
}

template <int BM, int BN, int BKT, bool A_KC, bool B_KC, bool AIDX = false>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& g, const int tile, const int zsplit, const int nsplits) {
    constexpr int LDA = BM + 2, LDB = BN + 2;
    constexpr int TM = BM / 64, TN = BN / 64;     // 32x32 MFMA tiles per wave
    constexpr int NA = BM * BKT / 1024, NB = BN * BKT / 1024;     // float4 loads per thread per operand
    __shared__ float As[2][BKT * LDA];
    __shared__ float Bs[2][BKT * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)(tile / g.tiles_n) * BM, n0 = (long)(tile % g.tiles_n) * BN;
    const long kbeg = (long)zsplit * g.k_per_split;
    const long kend = min(g.K, kbeg + g.k_per_split);
    const int nk = (int)((kend - kbeg + BKT - 1) / BKT);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // workgroup-uniform: the whole tile is inside A, B and this split's K range and 16-byte loads are
    // legal -> a main loop without any predicate (one dwordx4 per operand piece); else the guarded loop
    const bool interior = g.a_vec && g.b_vec && m0 + BM <= g.M && n0 + BN <= g.N && (kend - kbeg) % BKT == 0;
    auto main_loop = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        float4 ra[NA], rb[NB];
        // AIDX: gathered A.  A_KC: the thread's rows are fixed over K -> resolve their base pointers once.
        // !A_KC: the index runs along K -> the ids of tile kt+1 are fetched while tile kt loads (one tile ahead
        // of the data they address, so the id -> data chain never sits in front of the MFMAs).
        const float* arow[NA];
        int kid[NA];
        if (AIDX) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = tid + 256 * i;
                if (A_KC) {
                    const long gr = m0 + idx / (BKT / 4);
                    const int id = gr < g.M ? g.a_idx[gr] : -1;
                    arow[i] = id >= 0 ? g.A + (long)id * g.lda : nullptr;
                } else {
                    const long gk = kbeg + idx / (BM / 4);
                    kid[i] = gk < kend ? g.a_idx[gk] : -1;
                }
            }
        }
        auto gload = [&](int kt) {
            const long k0 = kbeg + (long)kt * BKT;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (AIDX) {
                    const int idx = tid + 256 * i;
                    if (A_KC) {
                        ra[i] = load_row4(arow[i], k0 + 4 * (idx % (BKT / 4)), kend, g.a_vec);
                    } else {
                        const float* p = kid[i] >= 0 ? g.A + (long)kid[i] * g.lda : nullptr;
                        ra[i] = load_row4(p, m0 + 4 * (idx % (BM / 4)), g.M, g.a_vec);
                        const long gk = k0 + BKT + idx / (BM / 4);
                        kid[i] = gk < kend ? g.a_idx[gk] : -1;
                    }
                } else {
                    ra[i] = FAST ? load_tile4_fast<BM, BKT, A_KC>(g.A, g.lda, m0, k0, tid + 256 * i)
                                 : load_tile4<BM, BKT, A_KC>(g.A, g.lda, m0, k0, g.M, kend, tid + 256 * i, g.a_vec);
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                rb[i] = FAST ? load_tile4_fast<BN, BKT, B_KC>(g.B, g.ldb, n0, k0, tid + 256 * i)
                             : load_tile4<BN, BKT, B_KC>(g.B, g.ldb, n0, k0, g.N, kend, tid + 256 * i, g.b_vec);
        };
        auto sstore = [&](int buf) {
#pragma unroll
            for (int i = 0; i < NA; ++i) store_tile4<BM, BKT, A_KC>(As[buf], tid + 256 * i, ra[i]);
#pragma unroll
            for (int i = 0; i < NB; ++i) store_tile4<BN, BKT, B_KC>(Bs[buf], tid + 256 * i, rb[i]);
        };
        if (nk > 0) {
            gload(0);
            sstore(0);
        }
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) gload(kt + 1);
            const float* as = As[cur] + wm * (BM / 2) + (lane & 31);
            const float* bs = Bs[cur] + wn * (BN / 2) + (lane & 31);
            // all MFMA operand fragments of this K tile first (one ds_read_b32 each), then the MFMAs
            // back to back: hipcc otherwise pairs every read with its use and exposes the LDS latency
            // in front of each MFMA (lgkmcnt(0) per pair).
            float a[BKT / 2][TM], b[BKT / 2][TN];
#pragma unroll
            for (int ks = 0; ks < BKT / 2; ++ks) {
                const int kk = 2 * ks + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TM; ++i) a[ks][i] = as[kk * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[ks][j] = bs[kk * LDB + 32 * j];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < BKT / 2; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) sstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    };
    if (interior) main_loop(std::true_type{});
    else main_loop(std::false_type{});

    if (g.epi == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const long col = n0 + wn * (BN / 2) + 32 * j + (lane & 31);
                const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = m0 + wm * (BM / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    bool pred = false;
                    if (row < g.M && col < g.N) pred = (col != g.tgt[row]) && (acc[i][j][r] + bv > g.thr[row]);
                    const unsigned long long bal = __ballot(pred);
                    const int cnt = __popcll(lane < 32 ? (bal & 0xFFFFFFFFull) : (bal >> 32));
                    if ((lane & 31) == 0 && cnt > 0 && row < g.M) atomicAdd(g.rank + row, cnt);
                }
            }
        return;
    }
    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cb = g.C + (long)zsplit * g.M * g.ldc;   // split-K slabs use ldc = N
    const bool splits = nsplits > 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long col = n0 + wn * (BN / 2) + 32 * j + (lane & 31);
            if (col >= g.N) continue;
            const float bv = (!splits && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + wm * (BM / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    float* c = Cb + row * g.ldc + col;
                    float v = acc[i][j][r] + bv;
                    if (!splits) {
                        v += row_add(g, row, col);
                        if (g.accumulate) v += *c;
                    }
                    *c = v;
                }
            }
        }
}


// ===================================================================================================
// v2 tile body: both operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging,
// no ds_write pass) into a 3-stage ring, TWO K tiles in flight across each barrier (counted vmcnt, raw
// s_barrier).  The LDS images are lane-linear per DMA (1 KB per wave-instruction), so every swizzle is
// applied on the per-lane SOURCE address and undone by the reading lane:
//   K-contiguous operand  [r][16 k], 16-B chunk q of row r stored at chunk q ^ ((r >> 2) & 3): the MFMA operand of
//       lane (row, kh) = 8 consecutive k = two conflict-free ds_read_b128;
//   r-contiguous operand  [16 k][R r]: R = 128 -> rows of the wave's two 32-row MFMA tiles interleaved (row 2l + i),
//       one ds_read_b64 per k feeds both tiles; R = 64 -> chunk ^ 8 on k rows 8..15, one ds_read_b32 per k.
// K order inside a 16-deep tile: MFMA step j multiplies k = j (lanes 0-31) and k = 8 + j (lanes 32-63) -- the same for A
// and B, so the sum is the exact fp32 sum in that fixed order.  Rows / K slices outside the matrix, null (negative)
// gather indices and the K tail read 16 zero bytes (g_zero16) instead of being predicated.
// LDS reads are inline asm: hipcc would otherwise put s_waitcnt vmcnt(0) in front of every ds_read that follows an
// LDS-DMA (it cannot see that the stages differ) and drain the ring every K tile.
__device__ __attribute__((aligned(16))) float g_zero16[4];

typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(uintptr_t)(const lds_float*)p; }
template <int OFF> __device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
    f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ f32x2 lds_read_b64(unsigned addr) {
    f32x2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}
template <int OFF> __device__ __forceinline__ float lds_read_b32(unsigned addr) {
    float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ void glds16(const float* src, float* lds_dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

#ifdef SEQREC_GEMM_ABLATE      // timing-only diagnostic build (tools/bench_gemm2.py): 1 no MFMA, 2 no in-loop DMA, 4 no C stores, 8 no LDS reads
__device__ int g_ablate;
__device__ unsigned long long g_gemm_stamps[4096 * 8];     // per workgroup: s_memrealtime (100 MHz) at 6 points + HW_ID
#define ABL(bit) (g_ablate_s & (bit))
#define GSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_gemm_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned long long g_gemm_phase[4096 * 8];      // per workgroup (wave 0): shader cycles summed per loop section + iterations
#define PH(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define PH(i) do {} while (0)
#define ABL(bit) false
#define GSTAMP(i) do {} while (0)
#endif
constexpr int V2_STAGES = 3;
constexpr int V2_IDX_CAP = 1024;      // K range per workgroup of a K-slice-gathered A operand (ids parked in LDS)
template <int BM, int BN, bool AIDX_K> constexpr int v2_lds_floats() { return V2_STAGES * (BM + BN) * 16 + (AIDX_K ? V2_IDX_CAP : 0); }

// one operand's per-lane DMA source state: NI wave-instructions per K tile
template <int R> struct V2Src {
    static constexpr int NI = R / 64;
    const float* p[NI];     // source of the NEXT K tile (or g_zero16)
    long step[NI];          // floats per K tile (0: stays on the zero chunk)
    int koff[NI];           // first k (within the tile) this lane's chunk covers: tail validity
    int roff[NI];           // r-contiguous gathered operand: column offset of the lane's chunk (else unused)
};

template <int R, bool KC>
__device__ __forceinline__ void v2_src_setup(V2Src<R>& s, const float* X, long ld, long r0, long Rtot, long kbeg,
                                             const int* row_idx, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < R / 64; ++i) {
        const int p = wave + 4 * i;
        if (KC) {
            const int r = 16 * p + (lane >> 2);
            const int sc = (lane & 3) ^ ((r >> 2) & 3);
            const long gr = r0 + r;
            long rowid = gr;
            bool ok = gr < Rtot;
            if (row_idx) { const int id = ok ? row_idx[gr] : -1; ok = id >= 0; rowid = id; }
            s.koff[i] = 4 * sc; s.roff[i] = 0;
            s.p[i] = ok ? X + rowid * ld + kbeg + 4 * sc : g_zero16;
            s.step[i] = ok ? 16 : 0;
        } else {
            const int k = R == 64 ? 4 * p + (lane >> 4) : 2 * p + (lane >> 5);
            const int c = lane & (R / 4 - 1);
            const int sc = R == 64 ? c ^ ((k & 8) ? 8 : 0) : c;
            const bool ok = r0 + 4 * sc < Rtot;
            s.koff[i] = k; s.roff[i] = ok ? 4 * sc : -1;
            s.p[i] = ok ? X + (kbeg + k) * ld + r0 + 4 * sc : g_zero16;
            s.step[i] = ok ? 16 * ld : 0;
        }
    }
}

// bijective XCD remap: workgroups are dealt round-robin to the 8 XCDs, so give XCD x a CONTIGUOUS run of work items
// (neighbouring tiles share their A rows / B columns through one L2)
__device__ __forceinline__ int xcd_tile(int b, int n) {
    const int x = b & 7, q = n >> 3, r = n & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// work item (= workgroup) -> (problem, tile, K split)
constexpr int GROUP_MAX = 6;
struct GemmGroup { GemmArgs g[GROUP_MAX]; int ntiles[GROUP_MAX]; };
struct PlainMap {
    const GemmArgs* g; int ntiles, nsplits;
    __device__ int count() const { return ntiles * nsplits; }
    __device__ void get(int item, int& p, int& tile, int& z, int& ns) const {
        const int l = xcd_tile(item, ntiles * nsplits);
        z = l / ntiles; tile = l - z * ntiles; p = 0; ns = nsplits;
    }
    __device__ GemmArgs load(int) const { return *g; }
};
struct GroupMap {          // the descriptors stay in the kernel-argument segment: problem p is picked by scalar selects
    const GemmGroup* gg; int nprob, nsplits, total;
    __device__ int count() const { return total; }
    __device__ void get(int item, int& p, int& tile, int& z, int& ns) const {
        int l = xcd_tile(item, total);
        int nt = gg->ntiles[0];
        p = 0;
#pragma unroll
        for (int i = 1; i < GROUP_MAX; ++i) {                // scalar selects: the problem sizes sit in the kernel arguments
            const int c = gg->ntiles[i - 1] * nsplits;
            if (p == i - 1 && i < nprob && l >= c) { l -= c; p = i; nt = gg->ntiles[i]; }
        }
        z = l / nt; tile = l - z * nt; ns = nsplits;
    }
    __device__ GemmArgs load(int p) const {
        GemmArgs r = gg->g[0];
#pragma unroll
        for (int i = 1; i < GROUP_MAX; ++i) if (p == i) r = gg->g[i];
        return r;
    }
};

__device__ __forceinline__ void glds16_to(const float* src, unsigned lds_byte_addr_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte_addr_wave_uniform, 16, 0, 0);
}

// One work item (tile x K split of one problem) per workgroup.  K loop: a 3-stage LDS ring filled by LDS-DMA three K tiles
// ahead of the MFMAs; the operand fragments of tile kt+1 are read into the second register set while the MFMAs of tile kt
// run from the first, and every non-MFMA instruction of a step (wait, barrier, DMA issue, fragment reads, bookkeeping) sits
// in the shadow of one of that step's MFMAs -- the order is pinned with sched_barriers.  The loop is split by what a step
// still has to do (FULL: wait + barrier + DMA + reads; the last three steps drop the DMA, then the partial wait, then
// everything), so the steady state carries no per-step conditions.
enum { STEP_FULL = 0, STEP_NOISSUE = 1, STEP_NOISSUE0 = 2, STEP_LAST = 3 };

template <int BM, int BN, bool A_KC, bool B_KC, bool AIDX, class Map>
__device__ __forceinline__ void gemm2_body(const Map& map, float* smem) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int STAGE = (BM + BN) * 16;           // floats per ring stage: A image then B image
    constexpr unsigned SB = STAGE * 4;              // bytes per stage
    constexpr bool AIDX_K = AIDX && !A_KC;           // the index (if the item has one) runs along K
    constexpr int NLD = BM / 64 + BN / 64;           // LDS-DMA instructions per wave per K tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
#ifdef SEQREC_GEMM_ABLATE
    const int g_ablate_s = __builtin_amdgcn_readfirstlane(g_ablate);
#endif
    GSTAMP(0);
#ifdef SEQREC_GEMM_ABLATE
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_gemm_stamps[blockIdx.x * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);   // HW_ID, XCC_ID
#endif
    int p, tile, zsplit, nsplits;
    map.get(blockIdx.x, p, tile, zsplit, nsplits);
    const GemmArgs g = map.load(p);
    const long m0 = (long)(tile / g.tiles_n) * BM, n0 = (long)(tile % g.tiles_n) * BN;
    const long kbeg = (long)zsplit * g.k_per_split;
    const long kend = min(g.K, kbeg + g.k_per_split);
    const int nk = (int)((kend - kbeg + 15) / 16);
    const bool ktail = ((kend - kbeg) & 15) != 0;
    const bool gather = AIDX_K && g.a_idx != nullptr;

    // per-lane LDS read addresses (bytes) inside a stage
    unsigned ra0, ra1 = 0, rb0, rb1 = 0;
    const unsigned sm = lds_addr(smem);
    if (A_KC) {
        const int r = wm * (BM / 2) + lr, f = (lr >> 2) & 3;
        ra0 = sm + (unsigned)(r * 64 + (((2 * kh) ^ f) * 16));
        ra1 = sm + (unsigned)(r * 64 + (((2 * kh + 1) ^ f) * 16));
    } else if (TM == 2) {
        ra0 = sm + (unsigned)((8 * kh) * (BM * 4) + (wm * 64 + 2 * lr) * 4);
    } else {
        ra0 = sm + (unsigned)((8 * kh) * (BM * 4) + (((wm * 32 + lr) ^ (kh ? 32 : 0)) * 4));
    }
    if (B_KC) {
        const int r = wn * (BN / 2) + lr, f = (lr >> 2) & 3;
        rb0 = sm + (unsigned)(BM * 64 + r * 64 + (((2 * kh) ^ f) * 16));
        rb1 = sm + (unsigned)(BM * 64 + r * 64 + (((2 * kh + 1) ^ f) * 16));
    } else if (TN == 2) {
        rb0 = sm + (unsigned)(BM * 64 + (8 * kh) * (BN * 4) + (wn * 64 + 2 * lr) * 4);
    } else {
        rb0 = sm + (unsigned)(BM * 64 + (8 * kh) * (BN * 4) + (((wn * 32 + lr) ^ (kh ? 32 : 0)) * 4));
    }

    // ---------------- DMA side
    V2Src<BM> sa; V2Src<BN> sb;
    v2_src_setup<BM, A_KC>(sa, g.A, g.lda, m0, g.M, kbeg, (AIDX && A_KC) ? g.a_idx : nullptr, wave, lane);
    v2_src_setup<BN, B_KC>(sb, g.B, g.ldb, n0, g.N, kbeg, nullptr, wave, lane);
    // K-slice gather: the item's K range of the index is parked in LDS; the ids of the NEXT tile to issue are read one
    // issue ahead (nid), behind the step's closing lgkmcnt(0)
    int* ids_s = reinterpret_cast<int*>(smem + V2_STAGES * STAGE);
    const unsigned ids_a = lds_addr(smem + V2_STAGES * STAGE);
    int nid[BM / 64];
    auto read_ids = [&](int kt) {
        const int t = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int i = 0; i < BM / 64; ++i) nid[i] = __float_as_int(lds_read_b32<0>(ids_a + (unsigned)((t * 16 + sa.koff[i]) * 4)));
    };
    // Every inline-asm LDS read below is invisible to hipcc's bookkeeping in TWO ways: it inserts no wait for the result, and
    // it considers the destination register free again as soon as its (last) reader is scheduled -- or at once when nothing
    // reads it.  The LDS unit writes the register when the data returns, whatever the compiler has put there meanwhile.  So
    // every wait is TIED to the registers it covers ("+v": the value is redefined at the wait, hence live from the read to
    // the wait, and every reader depends on the wait).
    auto tie_ids = [&]() {
        if constexpr (AIDX_K) {
#pragma unroll
            for (int i = 0; i < BM / 64; ++i) asm volatile("" : "+v"(nid[i]));
        }
    };
    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tie_ids(); __builtin_amdgcn_sched_barrier(0); };
    if (gather) {
        for (int i = tid; i < nk * 16; i += 256) ids_s[i] = kbeg + i < kend ? g.a_idx[kbeg + i] : -1;
        __syncthreads();
        read_ids(0);
        lgkm0();
    }
    const unsigned dmaA = sm + (unsigned)wave * 1024u, dmaB = sm + (unsigned)(BM * 64) + (unsigned)wave * 1024u;   // + stage + 4096 i
    int kt_i = 0;                 // next K tile to issue
    unsigned sw = 0;              // byte offset of the ring stage it goes to
    const float* zero16 = g_zero16;
    asm volatile("" : "+v"(zero16));                   // keep the address in registers (else: a GOT load + wait in every step)
    auto issue = [&]() {
        if (__builtin_expect(ktail && kt_i == nk - 1, 0)) {   // workgroup-uniform, last tile only: k past the end reads zeros
            asm volatile("" ::: "memory");             // a real branch, not 8 selects per step
            const long k0 = kbeg + (long)kt_i * 16;
#pragma unroll
            for (int i = 0; i < BM / 64; ++i)
                if (k0 + sa.koff[i] >= kend) sa.p[i] = zero16;
#pragma unroll
            for (int i = 0; i < BN / 64; ++i)
                if (k0 + sb.koff[i] >= kend) sb.p[i] = zero16;
        }
        const bool dma = !(ABL(2) && kt_i >= 3);
        if (gather) {
            // the ids of this tile were requested by an inline-asm ds_read one issue ago: the wait is tied to their registers
#pragma unroll
            for (int i = 0; i < BM / 64; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nid[i]) :: "memory");
        }
#pragma unroll
        for (int i = 0; i < BM / 64; ++i) {
            const float* src = sa.p[i];
            sa.p[i] += sa.step[i];
            if (gather) src = (nid[i] >= 0 && sa.roff[i] >= 0) ? g.A + (long)nid[i] * g.lda + m0 + sa.roff[i] : zero16;
            if (dma) glds16_to(src, dmaA + sw + 4096u * i);
        }
#pragma unroll
        for (int i = 0; i < BN / 64; ++i) {
            if (dma) glds16_to(sb.p[i], dmaB + sw + 4096u * i);
            sb.p[i] += sb.step[i];
        }
        // (no read behind the last tile: round 3 -- in the K-tail variant of this code that read's result had no reader, hipcc
        // handed its register to the address arithmetic that follows, and the LDS data arriving later overwrote it: one wave's
        // operands garbage whenever other processes' traffic stretched the LDS latency; tools/mp_stress.py)
        if (gather && kt_i + 1 < nk) read_ids(kt_i + 1);
        ++kt_i;
        sw = sw == 2 * SB ? 0u : sw + SB;
    };

    // ---------------- operand fragments: two register sets (static indices: steps are instantiated per parity)
    float a[2][TM][8], b[2][TN][8];
    // the same wait with the fragment registers of set P redefined behind it: the MFMAs that read them carry a data
    // dependency on the wait (an asm load is not in hipcc's wait bookkeeping; position alone orders nothing for the optimiser)
    auto lgkm0_frags = [&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(a[P][i][j]));
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(b[P][i][j]));
        tie_ids();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_a = [&](auto Pc, unsigned so) {
        constexpr int P = decltype(Pc)::value;
        const unsigned ra0s = ra0 + so, ra1s = ra1 + so;      // one address add per operand half, immediates for the rest
#ifdef SEQREC_GEMM_ABLATE
        if (ABL(8)) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) a[P][i][j] = 1.0f;
            return;
        }
#endif
        if constexpr (A_KC) {
            static_for<0, TM>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const f32x4 v0 = lds_read_b128<i * 2048>(ra0s), v1 = lds_read_b128<i * 2048>(ra1s);
                a[P][i][0] = v0[0]; a[P][i][1] = v0[1]; a[P][i][2] = v0[2]; a[P][i][3] = v0[3];
                a[P][i][4] = v1[0]; a[P][i][5] = v1[1]; a[P][i][6] = v1[2]; a[P][i][7] = v1[3];
            });
        } else {
            static_for<0, 8>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (TM == 2) { const f32x2 v = lds_read_b64<j * BM * 4>(ra0s); a[P][0][j] = v[0]; a[P][TM - 1][j] = v[1]; }
                else a[P][0][j] = lds_read_b32<j * BM * 4>(ra0s);
            });
        }
    };
    auto read_b = [&](auto Pc, unsigned so) {
        constexpr int P = decltype(Pc)::value;
        const unsigned rb0s = rb0 + so, rb1s = rb1 + so;
#ifdef SEQREC_GEMM_ABLATE
        if (ABL(8)) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i) b[P][i][j] = 0.5f;
            return;
        }
#endif
        if constexpr (B_KC) {
            static_for<0, TN>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const f32x4 v0 = lds_read_b128<i * 2048>(rb0s), v1 = lds_read_b128<i * 2048>(rb1s);
                b[P][i][0] = v0[0]; b[P][i][1] = v0[1]; b[P][i][2] = v0[2]; b[P][i][3] = v0[3];
                b[P][i][4] = v1[0]; b[P][i][5] = v1[1]; b[P][i][6] = v1[2]; b[P][i][7] = v1[3];
            });
        } else {
            static_for<0, 8>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (TN == 2) { const f32x2 v = lds_read_b64<j * BN * 4>(rb0s); b[P][0][j] = v[0]; b[P][TN - 1][j] = v[1]; }
                else b[P][0][j] = lds_read_b32<j * BN * 4>(rb0s);
            });
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mfma_j = [&](auto Pc, auto Jc) {
        constexpr int P = decltype(Pc)::value, j = decltype(Jc)::value;
#ifdef SEQREC_GEMM_ABLATE
        if (ABL(1)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(a[P][i][j]));
#pragma unroll
            for (int i = 0; i < TN; ++i) asm volatile("" :: "v"(b[P][i][j]));
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[P][i][j], b[P][jj][j], acc[i][jj], 0, 0, 0);
    };

    // ---------------- prologue: three tiles in flight, tile 0 into fragment set 0
    issue();
    if (AIDX_K) lgkm0();
    if (nk > 1) issue();
    if (AIDX_K) lgkm0();
    if (nk > 2) issue();
    if (AIDX_K) lgkm0();
    GSTAMP(1);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NLD) : "memory");
    else if (nk == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    GSTAMP(2);
    read_a(std::integral_constant<int, 0>{}, 0u);
    read_b(std::integral_constant<int, 0>{}, 0u);
    lgkm0_frags(std::integral_constant<int, 0>{});
    unsigned sr = SB;             // byte offset of the ring stage the next fragment reads come from

    // step kt: MFMAs of tile kt from set P.  Under them: wait for tile kt+1 (this wave's DMAs), barrier (every wave's
    // DMAs of tile kt+1 have landed; everyone finished reading tile kt in the previous step, so its stage may be refilled),
    // DMA of tile kt+3 into that stage, fragment reads of tile kt+1 into set P^1.
    auto step = [&](auto Pc, auto Kc) {
        constexpr int KIND = decltype(Kc)::value;
        using Q = std::integral_constant<int, decltype(Pc)::value ^ 1>;
        using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
        using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;
        using J4 = std::integral_constant<int, 4>; using J5 = std::integral_constant<int, 5>;
        using J6 = std::integral_constant<int, 6>; using J7 = std::integral_constant<int, 7>;
        __builtin_amdgcn_sched_barrier(0);
        mfma_j(Pc, J0{});
        __builtin_amdgcn_sched_barrier(0);
        if (KIND == STEP_FULL || KIND == STEP_NOISSUE) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");
        if (KIND == STEP_NOISSUE0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_j(Pc, J1{});
        __builtin_amdgcn_sched_barrier(0);
        if (KIND != STEP_LAST) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfma_j(Pc, J2{});
        __builtin_amdgcn_sched_barrier(0);
        if (KIND == STEP_FULL) issue();
        __builtin_amdgcn_sched_barrier(0);
        mfma_j(Pc, J3{});
        __builtin_amdgcn_sched_barrier(0);
        if (KIND != STEP_LAST) read_a(Q{}, sr);
        __builtin_amdgcn_sched_barrier(0);
        mfma_j(Pc, J4{});
        __builtin_amdgcn_sched_barrier(0);
        if (KIND != STEP_LAST) read_b(Q{}, sr);
        __builtin_amdgcn_sched_barrier(0);
        mfma_j(Pc, J5{});
        __builtin_amdgcn_sched_barrier(0);
        sr = sr == 2 * SB ? 0u : sr + SB;
        __builtin_amdgcn_sched_barrier(0);
        mfma_j(Pc, J6{});
        mfma_j(Pc, J7{});
        __builtin_amdgcn_sched_barrier(0);
        if (KIND != STEP_LAST) lgkm0_frags(Q{});
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using KF = std::integral_constant<int, STEP_FULL>;
    using KN = std::integral_constant<int, STEP_NOISSUE>;
    using KZ = std::integral_constant<int, STEP_NOISSUE0>;
    using KL = std::integral_constant<int, STEP_LAST>;
    auto tail = [&](auto Pc, int rem) {          // the last rem <= 3 steps, starting on set P
        using Q = std::integral_constant<int, decltype(Pc)::value ^ 1>;
        if (rem == 3) { step(Pc, KN{}); step(Q{}, KZ{}); step(Pc, KL{}); }
        else if (rem == 2) { step(Pc, KZ{}); step(Q{}, KL{}); }
        else step(Pc, KL{});
    };
    int kt = 0;
    for (; kt + 4 < nk; kt += 2) { step(P0{}, KF{}); step(P1{}, KF{}); }
    if (kt + 3 < nk) { step(P0{}, KF{}); ++kt; tail(P1{}, nk - kt); }
    else tail(P0{}, nk - kt);
    GSTAMP(3);

    // ---------------- epilogue.  C/D map of the 32x32 MFMA: column lane = lane&31, row lane = (r&3) + 8*(r>>2) + 4*(lane>>5);
    // tile (i, jj) covers rows 32 i + row lane (K-contiguous A, or one tile) or 2 row lane + i (interleaved), same for columns
    float* Cb = g.C + (long)zsplit * g.M * g.ldc;   // split-K slabs use ldc = N
    const bool splits = nsplits > 1;
    constexpr bool ROW_IL = !A_KC && TM == 2, COL_IL = !B_KC && TN == 2;
    const bool pair_ok = COL_IL && ((g.ldc & 1) == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 7) == 0);
    // fast path (interior tile, nothing to add): one per-lane base pointer + workgroup-uniform byte offsets
    const bool plain = (splits || (!g.bias && !g.add_table && !g.accumulate)) && m0 + BM <= g.M && n0 + BN <= g.N &&
                       g.ldc < (1l << 22) && (!COL_IL || pair_ok) && !ABL(4);
    if (plain) {
        const unsigned ld4 = (unsigned)g.ldc * 4u;
        char* base = reinterpret_cast<char*>(Cb + (m0 + wm * (BM / 2) + (ROW_IL ? 8 * kh : 4 * kh)) * g.ldc + n0 + wn * (BN / 2) + (COL_IL ? 2 * lr : lr));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2);
                const unsigned ro = (unsigned)(ROW_IL ? 2 * rl + i : 32 * i + rl) * ld4;
                if (COL_IL) {
                    *reinterpret_cast<f32x2*>(base + ro) = f32x2{acc[i][0][r], acc[i][TN - 1][r]};
                } else {
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) *reinterpret_cast<float*>(base + ro + 128u * jj) = acc[i][jj][r];
                }
            }
        GSTAMP(4);
        GSTAMP(5);
#ifdef SEQREC_GEMM_ABLATE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GSTAMP(6);
#endif
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * kh;
            const long row = m0 + wm * (BM / 2) + (ROW_IL ? 2 * rl + i : 32 * i + rl);
            if (row >= g.M) continue;
            if (ABL(4) && acc[i][0][r] != 12345.678f) continue;
            float v[TN];
            long col[TN];
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                col[jj] = n0 + wn * (BN / 2) + (COL_IL ? 2 * lr + jj : 32 * jj + lr);
                v[jj] = acc[i][jj][r];
                if (col[jj] < g.N && !splits) {
                    if (g.bias) v[jj] += g.bias[col[jj]];
                    v[jj] += row_add(g, row, col[jj]);
                    if (g.accumulate) v[jj] += Cb[row * g.ldc + col[jj]];
                }
            }
            if (COL_IL && pair_ok && col[TN - 1] < g.N) {
                *reinterpret_cast<f32x2*>(Cb + row * g.ldc + col[0]) = f32x2{v[0], v[TN - 1]};
            } else {
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
                    if (col[jj] < g.N) Cb[row * g.ldc + col[jj]] = v[jj];
            }
        }
    GSTAMP(4);
    GSTAMP(5);
#ifdef SEQREC_GEMM_ABLATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GSTAMP(6);
#endif
}

// register budget: the 64x64 tile must keep 5 workgroups on a CU (c3: 1 280 tiles = 5 per CU; at 4 a fifth of them
// runs as a second round), the larger tiles 3 / 2
#define GEMM2_WAVES(BM, BN) __attribute__((amdgpu_waves_per_eu((BM) * (BN) <= 4096 ? 5 : (BM) * (BN) <= 8192 ? 3 : 2)))
template <int BM, int BN, bool A_KC, bool B_KC, bool AIDX>
__global__ __launch_bounds__(256) GEMM2_WAVES(BM, BN) void gemm2_f32_kernel(GemmArgs g, int ntiles, int nsplits) {
    __shared__ __attribute__((aligned(16))) float smem[v2_lds_floats<BM, BN, AIDX && !A_KC>()];
    const PlainMap map{&g, ntiles, nsplits};
    gemm2_body<BM, BN, A_KC, B_KC, AIDX>(map, smem);
}

// TWO independent products of different layouts in one launch (round 4): work items [0, n0) belong to the first, the rest to the
// second; a workgroup runs ONE of the two layout bodies.  For dH = dlogits . Eneg (k-contiguous A) beside dEneg = dlogits^T . H
// (row-contiguous operands): both read dlogits, neither fills the chip alone at c3 (480 + 512 workgroups on 1 280 slots), and run
// side by side they take 46.7 us against 54.2 back to back (tools/pair_probe.py).
struct PairMap {
    const GemmArgs* g; int ntiles, nsplits, off;
    __device__ int count() const { return ntiles * nsplits; }
    __device__ void get(int item, int& p, int& tile, int& z, int& ns) const {
        const int l = xcd_tile(item - off, ntiles * nsplits);
        z = l / ntiles; tile = l - z * ntiles; p = 0; ns = nsplits;
    }
    __device__ GemmArgs load(int) const { return *g; }
};
template <int BM, int BN, bool A0, bool B0, bool A1, bool B1>
__global__ __launch_bounds__(256) GEMM2_WAVES(BM, BN) void gemm2_f32_pair_kernel(GemmArgs g0, int nt0, int ns0, GemmArgs g1, int nt1, int ns1) {
    __shared__ __attribute__((aligned(16))) float smem[v2_lds_floats<BM, BN, false>()];
    const int n0 = nt0 * ns0;
    if ((int)blockIdx.x < n0) {
        const PairMap map{&g0, nt0, ns0, 0};
        gemm2_body<BM, BN, A0, B0, false>(map, smem);
    } else {
        const PairMap map{&g1, nt1, ns1, n0};
        gemm2_body<BM, BN, A1, B1, false>(map, smem);
    }
}

template <int BM, int BN, int BKT, bool A_KC, bool B_KC, bool AIDX = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
#ifdef SEQREC_PROBE_XCD_SKIP       // timing probe only (tools/overlap_probe.py): workgroups dealt to the first N XCDs do nothing
    if ((int)(blockIdx.x & 7) < SEQREC_PROBE_XCD_SKIP) return;
#endif
    gemm_tile_body<BM, BN, BKT, A_KC, B_KC, AIDX>(g, blockIdx.x, blockIdx.z, gridDim.z);
}

// grouped launch: blockIdx.y picks one of up to 4 independent problems of the same layout
template <int BM, int BN, int BKT, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_grouped_kernel(GemmGroup gg) {
    const int p = blockIdx.y;
    if ((int)blockIdx.x >= gg.ntiles[p]) return;
    if (gg.g[p].a_idx) gemm_tile_body<BM, BN, BKT, A_KC, B_KC, true>(gg.g[p], blockIdx.x, blockIdx.z, gridDim.z);
    else gemm_tile_body<BM, BN, BKT, A_KC, B_KC, false>(gg.g[p], blockIdx.x, blockIdx.z, gridDim.z);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) GEMM2_WAVES(BM, BN) void gemm2_f32_grouped_kernel(GemmGroup gg, int nprob, int nsplits, int total) {   // layout (A, B both r-contiguous) only
    __shared__ __attribute__((aligned(16))) float smem[v2_lds_floats<BM, BN, true>()];
    const GroupMap map{&gg, nprob, nsplits, total};
    gemm2_body<BM, BN, false, false, true>(map, smem);
}

struct ReduceGroup { const float* ws[GROUP_MAX]; float* C[GROUP_MAX]; const float* bias[GROUP_MAX]; long M[GROUP_MAX], N[GROUP_MAX], ldc[GROUP_MAX];
                     int accumulate[GROUP_MAX]; };
__global__ void splitk_reduce_grouped_kernel(ReduceGroup r, int splits) {
    const int p = blockIdx.y;
    const long total = r.M[p] * r.N[p];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += r.ws[p][(long)z * total + i];
        const long row = i / r.N[p], col = i % r.N[p];
        if (r.bias[p]) s += r.bias[p][col];
        float* o = r.C[p] + row * r.ldc[p] + col;
        if (r.accumulate[p]) s += *o;
        *o = s;
    }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long M, long N,
                                     float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                     int accumulate, GemmArgs g) {
    const long total = M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += ws[(long)z * total + i];   // fixed order: deterministic
        const long r = i / N, c = i % N;
        if (bias) s += bias[c];
        s += row_add(g, r, c);
        float* o = C + r * ldc + c;
        if (accumulate) s += *o;
        *o = s;
    }
}

template <int BM, int BN, int BKT>
int launch_gemm_bk(int a_kc, int b_kc, GemmArgs& g, int splits, hipStream_t st) {
    const long tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    g.tiles_n = (int)tiles_n;
    dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)splits), block(256);
    if (g.a_idx) {                // gathered A operand: 64x64 tiles only (the caller picks them)
        if constexpr (BM == 64 && BN == 64 && BKT == 16) {
            if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 16, true, true, true>), grid, block, 0, st, g);
            else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 16, true, false, true>), grid, block, 0, st, g);
            else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 16, false, true, true>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 16, false, false, true>), grid, block, 0, st, g);
            SEQREC_LAUNCH_CHECK();
            return 0;
        } else {
            return SEQREC_E_UNSUPPORTED;
        }
    }
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BKT, true, true>), grid, block, 0, st, g);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BKT, true, false>), grid, block, 0, st, g);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BKT, false, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BKT, false, false>), grid, block, 0, st, g);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
int g_v2_tile = -1, g_v2_gtile = -1;      // seqrec_debug_gemm_tile(): tests force every tile shape through every layout
// v2 (LDS-DMA ring) launcher.  Eligible: 16-byte loads legal on both operands, K % 4 == 0 for a K-contiguous operand
// (its 16-byte chunks run along K), a K-slice-gathered A with at most V2_IDX_CAP k per workgroup, plain epilogue.
inline bool gemm2_eligible(int a_kc, int b_kc, const GemmArgs& g) {
    static const bool on = seqrec_env("SEQREC_GEMM_V2", 1) != 0;      // tuning switch
    if (!on || !g.a_vec || !g.b_vec || g.epi != 0 || g.K <= 0) return false;
    if ((a_kc || b_kc) && (g.K % 4 != 0)) return false;
    if (g.a_idx && !a_kc && g.k_per_split > V2_IDX_CAP) return false;
    if (g.a_idx && a_kc && b_kc) return false;               // not instantiated (no caller)
    if (g.a_idx && !a_kc && b_kc) return false;
    return true;
}
inline int gemm2_grid(long nitems) { return (int)nitems; }      // one work item per workgroup
template <int BM, int BN>
int launch_gemm2(int a_kc, int b_kc, GemmArgs& g, int splits, hipStream_t st) {
    const long tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    g.tiles_n = (int)tiles_n;
    const int ntiles = (int)(tiles_m * tiles_n);
    dim3 grid((unsigned)gemm2_grid((long)ntiles * splits)), block(256);
    if (g.a_idx) {
        if (a_kc) hipLaunchKernelGGL((gemm2_f32_kernel<BM, BN, true, false, true>), grid, block, 0, st, g, ntiles, splits);
        else hipLaunchKernelGGL((gemm2_f32_kernel<BM, BN, false, false, true>), grid, block, 0, st, g, ntiles, splits);
    } else if (a_kc && b_kc) hipLaunchKernelGGL((gemm2_f32_kernel<BM, BN, true, true, false>), grid, block, 0, st, g, ntiles, splits);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm2_f32_kernel<BM, BN, true, false, false>), grid, block, 0, st, g, ntiles, splits);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm2_f32_kernel<BM, BN, false, true, false>), grid, block, 0, st, g, ntiles, splits);
    else hipLaunchKernelGGL((gemm2_f32_kernel<BM, BN, false, false, false>), grid, block, 0, st, g, ntiles, splits);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
// tile choice of v2 (measured at the c3 shapes, a 25 088-row batch and 4096^3, tools/bench_gemm2.py): the per-workgroup
// prologue + epilogue favour MANY small workgroups until there are enough tiles for several rounds per CU
inline int gemm2_tile(long M, long N, int splits) {
    static const int forced = (int)seqrec_env("SEQREC_GEMM_V2_TILE", 0);      // tuning switch
    if (g_v2_tile > 0) return g_v2_tile;
    if (forced) return forced;
    const long t128 = ((M + 127) / 128) * ((N + 127) / 128) * splits;
    const long t12864 = ((M + 127) / 128) * ((N + 63) / 64) * splits;
    if (t128 >= 1024) return 3;
    if (t12864 >= 2048) return 2;
    return 1;
}
int launch_gemm2_auto(int a_kc, int b_kc, GemmArgs& g, int splits, hipStream_t st) {
    switch (gemm2_tile(g.M, g.N, splits)) {
        case 3: return launch_gemm2<128, 128>(a_kc, b_kc, g, splits, st);
        case 2: return launch_gemm2<128, 64>(a_kc, b_kc, g, splits, st);
        default: return launch_gemm2<64, 64>(a_kc, b_kc, g, splits, st);
    }
}

// one barrier per 32-deep K tile when the per-split K is long enough (halves the barrier count);
// 128x128 stays at 16 (LDS: 2 x 2 x 32 x 130 x 4 B would cost occupancy)
template <int BM, int BN>
int launch_gemm(int a_kc, int b_kc, GemmArgs& g, int splits, hipStream_t st) {
    static const bool bk32 = seqrec_env("SEQREC_GEMM_BK32", 0) != 0;   // tuning switch
    if (bk32 && !g.a_idx && BM * BN < 128 * 128 && g.k_per_split >= 64 && g.k_per_split % 32 == 0)
        return launch_gemm_bk<BM, BN, (BM * BN < 128 * 128 ? 32 : 16)>(a_kc, b_kc, g, splits, st);
    return launch_gemm_bk<BM, BN, 16>(a_kc, b_kc, g, splits, st);
}

}  // namespace

#ifdef SEQREC_GEMM_ABLATE
extern "C" void seqrec_debug_gemm_stamps(unsigned long long* out, int clear) {      // diagnostic build only
    if (clear) { (void)hipMemset((void*)nullptr, 0, 0); static unsigned long long z[4096 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), z, sizeof(z)); }
    else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 4096 * 8);
}
extern "C" void seqrec_debug_gemm_phases(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_phase), sizeof(unsigned long long) * 4096 * 8);
}
#endif
extern "C" void seqrec_debug_gemm_tile(int tile, int grouped_tile) {
    g_v2_tile = tile; g_v2_gtile = grouped_tile;
#ifdef SEQREC_GEMM_ABLATE
    const int m = getenv("SEQREC_GEMM_ABLATE") ? atoi(getenv("SEQREC_GEMM_ABLATE")) : 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &m, sizeof(int));
#endif
}

extern "C" int64_t seqrec_gemm_workspace_floats(int64_t M, int64_t N, int splitk) {
    return splitk > 1 ? (int64_t)splitk * M * N : 0;
}

extern "C" int seqrec_gemm_f32(int a_kcontig, int b_kcontig, int64_t M, int64_t N, int64_t K,
                               const float* A, int64_t lda, const float* B, int64_t ldb,
                               float* C, int64_t ldc, const float* bias, int accumulate,
                               int splitk, float* workspace, void* stream) {
    return seqrec_gemm_f32_fused(a_kcontig, b_kcontig, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, splitk, workspace,
                                 nullptr, stream);
}

static int gemm_f32_impl(int a_kcontig, int b_kcontig, int64_t M, int64_t N, int64_t K,
                         const float* A, int64_t lda, const float* B, int64_t ldb,
                         float* C, int64_t ldc, const float* bias, int accumulate,
                         int splitk, float* workspace, const seqrec_gemm_fuse* fuse, void* stream, int* slabs_out);

extern "C" int seqrec_gemm_f32_fused(int a_kcontig, int b_kcontig, int64_t M, int64_t N, int64_t K,
                                     const float* A, int64_t lda, const float* B, int64_t ldb,
                                     float* C, int64_t ldc, const float* bias, int accumulate,
                                     int splitk, float* workspace, const seqrec_gemm_fuse* fuse, void* stream) {
    if (!C) return SEQREC_E_ARG;
    return gemm_f32_impl(a_kcontig, b_kcontig, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, splitk, workspace, fuse, stream,
                         nullptr);
}

// split-K partial products LEFT in the workspace (slab s = workspace + s * M * N, row stride N): no reduce launch, the
// consumer adds the slabs where it reads them (seqrec_rows_job.n_slabs: the row scatter of dX / dEneg)
extern "C" int seqrec_gemm_f32_slabs(int a_kcontig, int b_kcontig, int64_t M, int64_t N, int64_t K,
                                     const float* A, int64_t lda, const float* B, int64_t ldb,
                                     int splitk, float* workspace, int* n_slabs, void* stream) {
    if (!workspace || !n_slabs) return SEQREC_E_ARG;
    *n_slabs = 0;
    return gemm_f32_impl(a_kcontig, b_kcontig, M, N, K, A, lda, B, ldb, nullptr, N, nullptr, 0, splitk, workspace, nullptr, stream,
                         n_slabs);
}

// the pair launch: product 0 = C0 (+ its fused row add, split-K with the reduce launch behind the pair), product 1 left as slabs.
// Taken when both products are LDS-DMA eligible at 64 x 64 tiles, un-gathered, with the layouts (1,0) and (0,0), and together at most
// ~1 500 work items (one round of the chip: larger products fill it alone -- c4); otherwise the two products go one after the other,
// exactly as seqrec_gemm_f32_fused and seqrec_gemm_f32_slabs would issue them.
extern "C" int seqrec_gemm_f32_pair(seqrec_gemm_pair* p, void* stream) {
    if (!p || !p->C0 || !p->ws1 || p->splitk0 < 1 || p->splitk1 < 1) return SEQREC_E_ARG;
    if (p->M0 <= 0 || p->N0 <= 0 || p->K0 <= 0 || p->M1 <= 0 || p->N1 <= 0 || p->K1 <= 0) return SEQREC_E_ARG;
    if (!p->A0 || !p->B0 || !p->A1 || !p->B1 || (p->splitk0 > 1 && !p->ws0)) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    auto fill = [](GemmArgs& g, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, int splitk, int& splits) {
        g.A = A; g.B = B; g.bias = nullptr; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.accumulate = 0;
        g.epi = 0; g.tgt = nullptr; g.thr = nullptr; g.rank = nullptr; g.a_idx = nullptr;
        g.a_vec = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 4 == 0);
        g.b_vec = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 4 == 0);
        long kps = (K + splitk - 1) / splitk;
        kps = (kps + 31) / 32 * 32;
        if (kps == 0) kps = BK;
        splits = (int)((K + kps - 1) / kps);
        if (splits < 1) splits = 1;
        g.k_per_split = kps;
    };
    GemmArgs g0{}, g1{};
    int s0 = 1, s1 = 1;
    fill(g0, p->M0, p->N0, p->K0, p->A0, p->lda0, p->B0, p->ldb0, p->splitk0, s0);
    fill(g1, p->M1, p->N1, p->K1, p->A1, p->lda1, p->B1, p->ldb1, p->splitk1, s1);
    if (p->add_table && (!p->add_index || p->add_ld < p->N0)) return SEQREC_E_ARG;
    g0.add_table = p->add_table; g0.add_idx = p->add_index; g0.add_scale = p->add_scale; g0.add_ld = p->add_ld;
    const long nt0 = ((p->M0 + 63) / 64) * ((p->N0 + 63) / 64), nt1 = ((p->M1 + 63) / 64) * ((p->N1 + 63) / 64);
    const bool together = p->a_kc0 == 1 && p->b_kc0 == 0 && p->a_kc1 == 0 && p->b_kc1 == 0 && gemm2_eligible(1, 0, g0) && gemm2_eligible(0, 0, g1)
                          && gemm2_tile(p->M0, p->N0, s0) == 1 && gemm2_tile(p->M1, p->N1, s1) == 1 && nt0 * s0 + nt1 * s1 <= 1536;
    if (!together) {
        seqrec_gemm_fuse f = {};
        f.add_table = p->add_table; f.add_index = p->add_index; f.add_scale = p->add_scale; f.add_ld = p->add_ld;
        int rc = gemm_f32_impl(p->a_kc0, p->b_kc0, p->M0, p->N0, p->K0, p->A0, p->lda0, p->B0, p->ldb0, p->C0, p->ldc0, nullptr, 0, p->splitk0,
                               p->ws0, &f, stream, nullptr);
        if (rc) return rc;
        int ns = 0;
        rc = gemm_f32_impl(p->a_kc1, p->b_kc1, p->M1, p->N1, p->K1, p->A1, p->lda1, p->B1, p->ldb1, nullptr, p->N1, nullptr, 0, p->splitk1,
                           p->ws1, nullptr, stream, &ns);
        p->n_slabs1 = ns;
        p->together = 0;
        return rc;
    }
    if (s0 > 1) { g0.C = p->ws0; g0.ldc = p->N0; } else { g0.C = p->C0; g0.ldc = p->ldc0; }
    g1.C = p->ws1; g1.ldc = p->N1;
    g0.tiles_n = (int)((p->N0 + 63) / 64); g1.tiles_n = (int)((p->N1 + 63) / 64);
    hipLaunchKernelGGL((gemm2_f32_pair_kernel<64, 64, true, false, false, false>), dim3((unsigned)(nt0 * s0 + nt1 * s1)), dim3(256), 0, st,
                       g0, (int)nt0, s0, g1, (int)nt1, s1);
    SEQREC_LAUNCH_CHECK();
    if (s0 > 1) {
        const long total = p->M0 * p->N0;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)min((long)2048, (total + 255) / 256)), dim3(256), 0, st, p->ws0, s0, (long)p->M0,
                           (long)p->N0, p->C0, (long)p->ldc0, (const float*)nullptr, 0, g0);
        SEQREC_LAUNCH_CHECK();
    }
    p->n_slabs1 = s1;
    p->together = 1;
    return 0;
}

static int gemm_f32_impl(int a_kcontig, int b_kcontig, int64_t M, int64_t N, int64_t K,
                         const float* A, int64_t lda, const float* B, int64_t ldb,
                         float* C, int64_t ldc, const float* bias, int accumulate,
                         int splitk, float* workspace, const seqrec_gemm_fuse* fuse, void* stream, int* slabs_out) {
    if (M < 0 || N < 0 || K < 0 || (!C && !slabs_out)) return SEQREC_E_ARG;
    if (M == 0 || N == 0) return 0;
    if ((K > 0 && (!A || !B)) || splitk < 1) return SEQREC_E_ARG;
    if (splitk > 1 && !workspace) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    GemmArgs g{};
    if (fuse) {
        if (fuse->add_table && (!fuse->add_index || fuse->add_ld < N)) return SEQREC_E_ARG;
        g.a_idx = fuse->a_index;
        g.add_table = fuse->add_table; g.add_idx = fuse->add_index; g.add_scale = fuse->add_scale; g.add_ld = fuse->add_ld;
    }
    g.A = A; g.B = B; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.accumulate = accumulate;
    g.epi = 0; g.tgt = nullptr; g.thr = nullptr; g.rank = nullptr;
    g.a_vec = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 4 == 0);
    g.b_vec = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 4 == 0);
    long kps = (K + splitk - 1) / splitk;
    kps = (kps + 31) / 32 * 32;                       // multiple of both K-tile depths
    if (kps == 0) kps = BK;
    int splits = (int)((K + kps - 1) / kps);
    if (splits < 1) splits = 1;
    g.k_per_split = kps;
    if (splits > 1 || slabs_out) { g.C = workspace; g.ldc = N; } else { g.C = C; g.ldc = ldc; }
    if (slabs_out) *slabs_out = splits;
    // tile choice: the biggest tile that still puts >= 4 workgroups on every CU (256 CUs); measured on
    // the c3 shapes, 64x64 beats 128x64 below that (K is short: prologue/epilogue dominate)
    const long t128 = ((M + 127) / 128) * ((N + 127) / 128) * splits;
    const long t12864 = ((M + 127) / 128) * ((N + 63) / 64) * splits;
    static const long thr = seqrec_env("SEQREC_GEMM_TILE_THR", 1024);   // tuning switch
    int rc;
    if (gemm2_eligible(a_kcontig, b_kcontig, g)) rc = launch_gemm2_auto(a_kcontig, b_kcontig, g, splits, st);
    else if (g.a_idx) rc = launch_gemm<64, 64>(a_kcontig, b_kcontig, g, splits, st);
    else if (t128 >= thr) rc = launch_gemm<128, 128>(a_kcontig, b_kcontig, g, splits, st);
    else if (t12864 >= thr) rc = launch_gemm<128, 64>(a_kcontig, b_kcontig, g, splits, st);
    else rc = launch_gemm<64, 64>(a_kcontig, b_kcontig, g, splits, st);
    if (rc) return rc;
    if (splits > 1 && !slabs_out) {
        const long total = M * N;
        int blocks = (int)min((long)2048, (total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, splits, (long)M, (long)N,
                           C, (long)ldc, bias, accumulate, g);
        SEQREC_LAUNCH_CHECK();
    }
    return 0;
}

namespace {
// thr[i] = hd[i,:] . Eout[tgt[i],:] + bout[tgt[i]]   (one wave per row)
__global__ void target_score_kernel(const float* __restrict__ hd, int H, const float* __restrict__ Eout,
                                    const float* __restrict__ bout, const int* __restrict__ tgt, long n,
                                    float* __restrict__ thr) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int t = tgt[row];
    const float* h = hd + row * H;
    const float* e = Eout + (long)t * H;
    float d = 0.f;
    for (int j = lane; j < H; j += 64) d += h[j] * e[j];
    d = wave_sum(d);
    if (lane == 0) thr[row] = d + (bout ? bout[t] : 0.f);
}
}  // namespace

namespace {
int rank_count_launch(const float* hd, int H, const float* Eout, const float* bout, const int32_t* tgt, const float* thr,
                      int64_t n, int V, int32_t* rank, hipStream_t st) {
    GemmArgs g{};
    g.A = hd; g.B = Eout; g.C = nullptr; g.bias = bout;
    g.M = n; g.N = V; g.K = H; g.lda = H; g.ldb = H; g.ldc = 0;
    g.k_per_split = (H + 31) / 32 * 32;
    g.accumulate = 0;
    g.a_vec = ((reinterpret_cast<uintptr_t>(hd) & 15) == 0) && (H % 4 == 0);
    g.b_vec = ((reinterpret_cast<uintptr_t>(Eout) & 15) == 0) && (H % 4 == 0);
    g.epi = 1; g.tgt = tgt; g.thr = thr; g.rank = rank;
    return launch_gemm<128, 128>(1, 1, g, 1, st);
}
}  // namespace

extern "C" int seqrec_rank_count(const float* hd, int H, const float* Eout, const float* bout,
                                 const int32_t* tgt, int64_t n, int V, int32_t* rank, float* thr_workspace,
                                 void* stream) {
    if (n < 0 || V <= 0 || H <= 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!hd || !Eout || !tgt || !rank || !thr_workspace) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(target_score_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, hd, H, Eout, bout, tgt, (long)n,
                       thr_workspace);
    SEQREC_LAUNCH_CHECK();
    return rank_count_launch(hd, H, Eout, bout, tgt, thr_workspace, n, V, rank, st);
}

extern "C" int seqrec_target_score(const float* hd, int H, const float* Eout, const float* bout, const int32_t* tgt,
                                   int64_t n, float* thr, void* stream) {
    if (n < 0 || H <= 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!hd || !Eout || !tgt || !thr) return SEQREC_E_ARG;
    hipLaunchKernelGGL(target_score_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), hd, H, Eout, bout,
                       tgt, (long)n, thr);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

// row-sharded form: thresholds are given (the target rows live on other ranks), `tgt_local` is the
// target's row in THIS shard or -1; rank accumulates the shard's count.
extern "C" int seqrec_rank_count_thr(const float* hd, int H, const float* Eout, const float* bout,
                                     const int32_t* tgt_local, const float* thr, int64_t n, int V, int32_t* rank,
                                     void* stream) {
    if (n < 0 || V < 0 || H <= 0) return SEQREC_E_ARG;
    if (n == 0 || V == 0) return 0;
    if (!hd || !Eout || !tgt_local || !thr || !rank) return SEQREC_E_ARG;
    return rank_count_launch(hd, H, Eout, bout, tgt_local, thr, n, V, rank, as_stream(stream));
}

// ---- grouped form: up to 4 problems with the same layout flags and split count in ONE launch
// (used for the weight gradients dW, dU_zr, dU_h that share K = N_tok)
static int gemm_grouped_impl(int count, int a_kcontig, int b_kcontig, const seqrec_gemm_desc* d, int splitk, float* workspace,
                             void* stream, int* slabs_out);
extern "C" int seqrec_gemm_f32_grouped(int count, int a_kcontig, int b_kcontig, const seqrec_gemm_desc* d,
                                       int splitk, float* workspace, void* stream) {
    return gemm_grouped_impl(count, a_kcontig, b_kcontig, d, splitk, workspace, stream, nullptr);
}
// the grouped products with their split-K partial sums LEFT in the workspace (problem i at the offset the reducing form
// uses: sum_{j<i} n_slabs M_j N_j; its slab s at + s M_i N_i, row stride N_i): descs' C / ldc / bias / accumulate are not
// touched here -- seqrec_opt_sqnorm_slabs finishes the products where the gradient norm reads them anyway
extern "C" int seqrec_gemm_f32_grouped_slabs(int count, int a_kcontig, int b_kcontig, const seqrec_gemm_desc* d,
                                             int splitk, float* workspace, int* n_slabs, void* stream) {
    if (!workspace || !n_slabs) return SEQREC_E_ARG;
    *n_slabs = 0;
    return gemm_grouped_impl(count, a_kcontig, b_kcontig, d, splitk, workspace, stream, n_slabs);
}
static int gemm_grouped_impl(int count, int a_kcontig, int b_kcontig, const seqrec_gemm_desc* d, int splitk, float* workspace,
                             void* stream, int* slabs_out) {
    if (count < 1 || count > GROUP_MAX || !d || splitk < 1) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    GemmGroup gg = {};
    ReduceGroup rg = {};
    long K = d[0].K, maxtiles = 0, wsoff = 0;
    long kps = (K + splitk - 1) / splitk;
    kps = (kps + 31) / 32 * 32;
    if (kps == 0) kps = 32;
    const int splits = (int)((K + kps - 1) / kps) < 1 ? 1 : (int)((K + kps - 1) / kps);
    if (splits > 1 && !workspace) return SEQREC_E_ARG;
    for (int i = 0; i < count; ++i) {
        if (d[i].K != K || d[i].M <= 0 || d[i].N <= 0 || !d[i].A || !d[i].B || (!d[i].C && !slabs_out)) return SEQREC_E_ARG;
        GemmArgs& g = gg.g[i];
        g.A = d[i].A; g.B = d[i].B; g.bias = (splits > 1 || slabs_out) ? nullptr : d[i].bias;
        g.M = d[i].M; g.N = d[i].N; g.K = K; g.lda = d[i].lda; g.ldb = d[i].ldb;
        g.accumulate = slabs_out ? 0 : d[i].accumulate;
        g.epi = 0; g.tgt = nullptr; g.thr = nullptr; g.rank = nullptr;
        g.a_idx = d[i].a_index;
        g.add_table = nullptr; g.add_idx = nullptr; g.add_scale = nullptr; g.add_ld = 0;
        g.a_vec = ((reinterpret_cast<uintptr_t>(d[i].A) & 15) == 0) && (d[i].lda % 4 == 0);
        g.b_vec = ((reinterpret_cast<uintptr_t>(d[i].B) & 15) == 0) && (d[i].ldb % 4 == 0);
        g.k_per_split = kps;
        const long tm = (g.M + 63) / 64, tn = (g.N + 63) / 64;
        g.tiles_n = (int)tn;
        gg.ntiles[i] = (int)(tm * tn);
        if (tm * tn > maxtiles) maxtiles = tm * tn;
        if (splits > 1 || slabs_out) { g.C = workspace + wsoff; g.ldc = g.N; } else { g.C = d[i].C; g.ldc = d[i].ldc; }
        rg.ws[i] = workspace + wsoff; rg.C[i] = d[i].C; rg.bias[i] = d[i].bias; rg.M[i] = g.M; rg.N[i] = g.N;
        rg.ldc[i] = d[i].ldc; rg.accumulate[i] = d[i].accumulate;
        wsoff += (long)splits * g.M * g.N;
    }
    bool v2 = !a_kcontig && !b_kcontig;
    for (int i = 0; i < count && v2; ++i) v2 = gemm2_eligible(0, 0, gg.g[i]);
    static const int gtile_env = (int)seqrec_env("SEQREC_GEMM_V2_GTILE", 1);      // tuning switch
    const int gtile = g_v2_gtile > 0 ? g_v2_gtile : gtile_env;
    if (v2 && gtile == 2) {
        maxtiles = 0;
        for (int i = 0; i < count; ++i) {
            const long tm = (gg.g[i].M + 127) / 128, tn = (gg.g[i].N + 63) / 64;
            gg.g[i].tiles_n = (int)tn; gg.ntiles[i] = (int)(tm * tn);
            if (tm * tn > maxtiles) maxtiles = tm * tn;
        }
    }
    dim3 grid((unsigned)maxtiles, (unsigned)count, (unsigned)splits), block(256);
    if (v2) {
        long total = 0;
        for (int i = 0; i < count; ++i) total += (long)gg.ntiles[i] * splits;
        dim3 pgrid((unsigned)gemm2_grid(total));
        if (gtile == 2) hipLaunchKernelGGL((gemm2_f32_grouped_kernel<128, 64>), pgrid, block, 0, st, gg, count, splits, (int)total);
        else hipLaunchKernelGGL((gemm2_f32_grouped_kernel<64, 64>), pgrid, block, 0, st, gg, count, splits, (int)total);
    }
    else if (a_kcontig && b_kcontig) hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 16, true, true>), grid, block, 0, st, gg);
    else if (a_kcontig && !b_kcontig) hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 16, true, false>), grid, block, 0, st, gg);
    else if (!a_kcontig && b_kcontig) hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 16, false, true>), grid, block, 0, st, gg);
    else hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 16, false, false>), grid, block, 0, st, gg);
    SEQREC_LAUNCH_CHECK();
    if (slabs_out) { *slabs_out = splits; return 0; }
    if (splits > 1) {
        hipLaunchKernelGGL(splitk_reduce_grouped_kernel, dim3(256, count), dim3(256), 0, st, rg, splits);
        SEQREC_LAUNCH_CHECK();
    }
    return 0;
}

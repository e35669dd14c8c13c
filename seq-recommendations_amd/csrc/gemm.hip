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

template <int BM, int BN, int BKT, bool A_KC, bool B_KC, bool AIDX = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
#ifdef SEQREC_PROBE_XCD_SKIP       // timing probe only (tools/overlap_probe.py): workgroups dealt to the first N XCDs do nothing
    if ((int)(blockIdx.x & 7) < SEQREC_PROBE_XCD_SKIP) return;
#endif
    gemm_tile_body<BM, BN, BKT, A_KC, B_KC, AIDX>(g, blockIdx.x, blockIdx.z, gridDim.z);
}

// grouped launch: blockIdx.y picks one of up to 4 independent problems of the same layout
struct GemmGroup { GemmArgs g[4]; int ntiles[4]; };
template <int BM, int BN, int BKT, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_grouped_kernel(GemmGroup gg) {
    const int p = blockIdx.y;
    if ((int)blockIdx.x >= gg.ntiles[p]) return;
    if (gg.g[p].a_idx) gemm_tile_body<BM, BN, BKT, A_KC, B_KC, true>(gg.g[p], blockIdx.x, blockIdx.z, gridDim.z);
    else gemm_tile_body<BM, BN, BKT, A_KC, B_KC, false>(gg.g[p], blockIdx.x, blockIdx.z, gridDim.z);
}

struct ReduceGroup { const float* ws[4]; float* C[4]; const float* bias[4]; long M[4], N[4], ldc[4]; int accumulate[4]; };
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
// one barrier per 32-deep K tile when the per-split K is long enough (halves the barrier count);
// 128x128 stays at 16 (LDS: 2 x 2 x 32 x 130 x 4 B would cost occupancy)
template <int BM, int BN>
int launch_gemm(int a_kc, int b_kc, GemmArgs& g, int splits, hipStream_t st) {
    static const bool bk32 = getenv("SEQREC_GEMM_BK32") && atoi(getenv("SEQREC_GEMM_BK32")) != 0;   // tuning switch
    if (bk32 && !g.a_idx && BM * BN < 128 * 128 && g.k_per_split >= 64 && g.k_per_split % 32 == 0)
        return launch_gemm_bk<BM, BN, (BM * BN < 128 * 128 ? 32 : 16)>(a_kc, b_kc, g, splits, st);
    return launch_gemm_bk<BM, BN, 16>(a_kc, b_kc, g, splits, st);
}

}  // namespace

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

extern "C" int seqrec_gemm_f32_fused(int a_kcontig, int b_kcontig, int64_t M, int64_t N, int64_t K,
                                     const float* A, int64_t lda, const float* B, int64_t ldb,
                                     float* C, int64_t ldc, const float* bias, int accumulate,
                                     int splitk, float* workspace, const seqrec_gemm_fuse* fuse, void* stream) {
    if (M < 0 || N < 0 || K < 0 || !C) return SEQREC_E_ARG;
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
    if (splits > 1) { g.C = workspace; g.ldc = N; } else { g.C = C; g.ldc = ldc; }
    // tile choice: the biggest tile that still puts >= 4 workgroups on every CU (256 CUs); measured on
    // the c3 shapes, 64x64 beats 128x64 below that (K is short: prologue/epilogue dominate)
    const long t128 = ((M + 127) / 128) * ((N + 127) / 128) * splits;
    const long t12864 = ((M + 127) / 128) * ((N + 63) / 64) * splits;
    static const long thr = getenv("SEQREC_GEMM_TILE_THR") ? atol(getenv("SEQREC_GEMM_TILE_THR")) : 1024;   // tuning switch
    int rc;
    if (g.a_idx) rc = launch_gemm<64, 64>(a_kcontig, b_kcontig, g, splits, st);
    else if (t128 >= thr) rc = launch_gemm<128, 128>(a_kcontig, b_kcontig, g, splits, st);
    else if (t12864 >= thr) rc = launch_gemm<128, 64>(a_kcontig, b_kcontig, g, splits, st);
    else rc = launch_gemm<64, 64>(a_kcontig, b_kcontig, g, splits, st);
    if (rc) return rc;
    if (splits > 1) {
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
extern "C" int seqrec_gemm_f32_grouped(int count, int a_kcontig, int b_kcontig, const seqrec_gemm_desc* d,
                                       int splitk, float* workspace, void* stream) {
    if (count < 1 || count > 4 || !d || splitk < 1) return SEQREC_E_ARG;
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
        if (d[i].K != K || d[i].M <= 0 || d[i].N <= 0 || !d[i].A || !d[i].B || !d[i].C) return SEQREC_E_ARG;
        GemmArgs& g = gg.g[i];
        g.A = d[i].A; g.B = d[i].B; g.bias = splits > 1 ? nullptr : d[i].bias;
        g.M = d[i].M; g.N = d[i].N; g.K = K; g.lda = d[i].lda; g.ldb = d[i].ldb;
        g.accumulate = d[i].accumulate;
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
        if (splits > 1) { g.C = workspace + wsoff; g.ldc = g.N; } else { g.C = d[i].C; g.ldc = d[i].ldc; }
        rg.ws[i] = workspace + wsoff; rg.C[i] = d[i].C; rg.bias[i] = d[i].bias; rg.M[i] = g.M; rg.N[i] = g.N;
        rg.ldc[i] = d[i].ldc; rg.accumulate[i] = d[i].accumulate;
        wsoff += (long)splits * g.M * g.N;
    }
    dim3 grid((unsigned)maxtiles, (unsigned)count, (unsigned)splits), block(256);
    if (a_kcontig && b_kcontig) hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 16, true, true>), grid, block, 0, st, gg);
    else if (a_kcontig && !b_kcontig) hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 16, true, false>), grid, block, 0, st, gg);
    else if (!a_kcontig && b_kcontig) hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 16, false, true>), grid, block, 0, st, gg);
    else hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 16, false, false>), grid, block, 0, st, gg);
    SEQREC_LAUNCH_CHECK();
    if (splits > 1) {
        hipLaunchKernelGGL(splitk_reduce_grouped_kernel, dim3(256, count), dim3(256), 0, st, rg, splits);
        SEQREC_LAUNCH_CHECK();
    }
    return 0;
}

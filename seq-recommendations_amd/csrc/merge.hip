// Deterministic merge of the row-sparse gradients ("sort by row + segment sum, no float
// atomics on the hot rows"): the selectable alternative to the float-atomic scatter of ops.hip.
//
// The reference's dense Adagrad (experiments_methods.py:41) adds the contributions of a batch to a table
// row in ONE fixed order; float atomics add them in arrival order, so two runs with identical seeds agree
// to rounding only.  Here every scatter list of one table is keyed (row, contribution index), sorted by
// row with a STABLE radix sort (rocPRIM device radix sort over the 32-bit row id, the index rides along
// as the value: equal rows stay in increasing contribution order), and the sorted array is summed per
// row in that order with plain loads and ONE plain store per row:
//   * merge_tile_kernel: one wave per tile of 64 sorted positions walks the runs of equal rows inside its
//     tile; a run that lies inside the tile is summed and stored; a run that crosses a tile border leaves
//     its partial sum in the workspace (head = continues from the previous tile, tail = continues into
//     the next);
//   * merge_chain_kernel: one wave per run that crosses borders adds its partials in tile order.
// The grouping of the additions depends only on the sorted order -- bitwise reproducible.  It writes what
// the atomic path writes (gtab[row] += sum, slot[row] = smallest contribution index), so the norm /
// Adagrad kernels that follow are shared.  HBM/L2-bound: 4*width B read per contribution, 4*width B
// written per unique row; no float atomics.
#include "common.h"
#include <limits.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int TILE = 64;                     // sorted positions per wave
constexpr unsigned FILLER = 0x7FFFFFFFu;     // key of rows[i] < 0 (sorts last, never summed)

struct MergeJobs { seqrec_rows_job j[4]; int count; long total; long off[5]; };

__global__ void merge_fill_kernel(MergeJobs m, unsigned* __restrict__ keys, int* __restrict__ vals) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= m.total) return;
    int q = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) q += (k < m.count && g >= m.off[k]) ? 1 : 0;
    const seqrec_rows_job& J = m.j[q];
    const long i = g - m.off[q];
    const int r = J.rows[i];
    keys[g] = r < 0 ? FILLER : (unsigned)r;
    vals[g] = J.base + (int)i;
}

// contribution v (= base_j + i) -> its value row and scale (wave-uniform)
struct Src { const float* p; float s; };
__device__ __forceinline__ Src locate(const MergeJobs& m, int v) {
    int q = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) q += (k < m.count && v >= m.j[k].base) ? 1 : 0;
    const seqrec_rows_job& J = m.j[q];
    const long i = v - J.base;
    Src s;
    s.p = J.vals + i * J.ldv;
    s.s = J.row_scale ? J.row_scale[i] : 1.f;
    return s;
}

// workspace per tile: flags (bit0 head partial present, bit1 the head run fills the whole tile and goes on,
// bit2 tail partial present), tail_first = smallest contribution index of the tail run, then the partial rows
struct TileMeta { int flags; int tail_first; unsigned tail_key; int pad; };

// NC = ceil(width / 64) accumulators per lane (columns lane, lane + 64, ...)
template <int NC>
__global__ __launch_bounds__(256) void merge_tile_kernel(MergeJobs m, const unsigned* __restrict__ keys,
                                                         const int* __restrict__ vals, TileMeta* __restrict__ meta,
                                                         float* __restrict__ part, int ntiles) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const int width = m.j[0].width;
    float* __restrict__ gtab = m.j[0].gtab;
    int* __restrict__ slot = m.j[0].slot;
    const long p0 = (long)t * TILE;
    const int cnt = (int)min((long)TILE, m.total - p0);
    const unsigned myk = lane < cnt ? keys[p0 + lane] : FILLER;
    const int myv = lane < cnt ? vals[p0 + lane] : 0;
    const unsigned prevk = p0 > 0 ? keys[p0 - 1] : FILLER;
    const unsigned nextk = p0 + cnt < m.total ? keys[p0 + cnt] : FILLER;
    float* head = part + (long)t * 2 * width;
    float* tail = head + width;
    int flags = 0, tail_first = 0;
    unsigned tail_key = FILLER;
    float acc[NC];
    int q = 0;
    while (q < cnt) {
        const unsigned k = (unsigned)__shfl((int)myk, q, 64);
        if (k == FILLER) break;
        const int first = __shfl(myv, q, 64);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.f;
        const int q0 = q;
        // run [q0, q1): same key; contributions added in sorted (= increasing index) order
        while (q < cnt && (unsigned)__shfl((int)myk, q, 64) == k) {
            const Src s = locate(m, __shfl(myv, q, 64));
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = lane + 64 * c;
                if (col < width) acc[c] += s.p[col] * s.s;
            }
            ++q;
        }
        const bool from_prev = (q0 == 0) && (prevk == k);
        const bool to_next = (q == cnt) && (nextk == k);
        if (!from_prev && !to_next) {                      // the whole run lives in this tile: finished
            float* g = gtab + (long)k * width;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = lane + 64 * c;
                if (col < width) g[col] += acc[c];          // gtab is all-zero between steps (kept as += for callers that pre-load it)
            }
            if (lane == 0) slot[k] = min(slot[k], first);
        } else if (from_prev) {                            // continues a run that started in an earlier tile
            flags |= 1 | (to_next ? 2 : 0);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = lane + 64 * c;
                if (col < width) head[col] = acc[c];
            }
        } else {                                           // starts here, continues into the next tile
            flags |= 4;
            tail_first = first;
            tail_key = k;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = lane + 64 * c;
                if (col < width) tail[col] = acc[c];
            }
        }
    }
    if (lane == 0) {
        TileMeta tm;
        tm.flags = flags; tm.tail_first = tail_first; tm.tail_key = tail_key; tm.pad = 0;
        meta[t] = tm;
    }
}

template <int NC>
__global__ __launch_bounds__(256) void merge_chain_kernel(MergeJobs m, const TileMeta* __restrict__ meta,
                                                          const float* __restrict__ part, int ntiles) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const TileMeta tm = meta[t];
    if (!(tm.flags & 4)) return;                           // no run starts here and leaves the tile
    const int width = m.j[0].width;
    float acc[NC];
    const float* tail = part + ((long)t * 2 + 1) * width;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = lane + 64 * c;
        acc[c] = col < width ? tail[col] : 0.f;
    }
    for (int u = t + 1; u < ntiles; ++u) {                 // partials of the following tiles, in tile order
        const int f = meta[u].flags;
        if (!(f & 1)) break;
        const float* head = part + (long)u * 2 * width;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = lane + 64 * c;
            if (col < width) acc[c] += head[col];
        }
        if (!(f & 2)) break;
    }
    float* g = m.j[0].gtab + (long)tm.tail_key * width;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = lane + 64 * c;
        if (col < width) g[col] += acc[c];
    }
    if (lane == 0) m.j[0].slot[tm.tail_key] = min(m.j[0].slot[tm.tail_key], tm.tail_first);
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Layout { size_t keys_in, keys_out, vals_in, vals_out, meta, part, tmp, total; };
Layout layout(long n_total, int width, size_t tmp_bytes) {
    Layout L;
    const size_t n = (size_t)n_total;
    const size_t ntiles = (n + TILE - 1) / TILE;
    size_t o = 0;
    L.keys_in = o; o = align_up(o + 4 * n, 256);
    L.keys_out = o; o = align_up(o + 4 * n, 256);
    L.vals_in = o; o = align_up(o + 4 * n, 256);
    L.vals_out = o; o = align_up(o + 4 * n, 256);
    L.meta = o; o = align_up(o + sizeof(TileMeta) * ntiles, 256);
    L.part = o; o = align_up(o + 4 * (size_t)width * 2 * ntiles, 256);
    L.tmp = o; o = align_up(o + tmp_bytes, 256);
    L.total = o;
    return L;
}

hipError_t sort_pairs(void* tmp, size_t& tmp_bytes, unsigned* kin, unsigned* kout, int* vin, int* vout, long n, hipStream_t st) {
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, 31u, st);
}

}  // namespace

extern "C" int64_t seqrec_rows_merge_workspace_bytes(int64_t n_total, int width) {
    if (n_total <= 0 || width <= 0) return 0;
    size_t tmp_bytes = 0;
    if (sort_pairs(nullptr, tmp_bytes, nullptr, nullptr, nullptr, nullptr, (long)n_total, nullptr) != hipSuccess) return -1;
    return (int64_t)layout((long)n_total, width, tmp_bytes).total;
}

extern "C" int seqrec_rows_merge_sorted(const seqrec_rows_job* jobs, int count, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
    if (count < 1 || count > 4 || !jobs) return SEQREC_E_ARG;
    MergeJobs m = {};
    // the lists of ONE table, ordered by base; bases must be disjoint index ranges (they identify the contribution)
    int order[4] = {0, 1, 2, 3};
    for (int a = 0; a < count; ++a)
        for (int b = a + 1; b < count; ++b)
            if (jobs[order[b]].base < jobs[order[a]].base) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
    long total = 0;
    int used = 0;
    for (int a = 0; a < count; ++a) {
        const seqrec_rows_job& j = jobs[order[a]];
        if (j.n < 0 || j.width <= 0) return SEQREC_E_ARG;
        if (j.n == 0) continue;
        if (!j.gtab || !j.slot || !j.rows || !j.vals) return SEQREC_E_ARG;
        if (j.n_slabs > 1) return SEQREC_E_UNSUPPORTED;          // slab sums are read by the atomic scatter only
        if (used && (j.gtab != m.j[0].gtab || j.slot != m.j[0].slot || j.width != m.j[0].width)) return SEQREC_E_ARG;
        if (used && (long)j.base < (long)m.j[used - 1].base + m.j[used - 1].n) return SEQREC_E_ARG;     // overlapping index ranges
        if ((long)j.base + j.n > (long)INT_MAX) return SEQREC_E_ARG;
        m.j[used] = j;
        m.off[used] = total;
        total += j.n;
        ++used;
    }
    if (total == 0) return 0;
    m.count = used;
    m.total = total;
    m.off[used] = total;
    const int width = m.j[0].width;
    if (width > 64 * 32) return SEQREC_E_SHAPE;
    if (!workspace) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    size_t tmp_bytes = 0;
    if (sort_pairs(nullptr, tmp_bytes, nullptr, nullptr, nullptr, nullptr, total, st) != hipSuccess) return SEQREC_E_UNSUPPORTED;
    const Layout L = layout(total, width, tmp_bytes);
    if ((int64_t)L.total > workspace_bytes) return SEQREC_E_ARG;
    char* ws = static_cast<char*>(workspace);
    unsigned* kin = reinterpret_cast<unsigned*>(ws + L.keys_in);
    unsigned* kout = reinterpret_cast<unsigned*>(ws + L.keys_out);
    int* vin = reinterpret_cast<int*>(ws + L.vals_in);
    int* vout = reinterpret_cast<int*>(ws + L.vals_out);
    TileMeta* meta = reinterpret_cast<TileMeta*>(ws + L.meta);
    float* part = reinterpret_cast<float*>(ws + L.part);
    hipLaunchKernelGGL(merge_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, m, kin, vin);
    SEQREC_LAUNCH_CHECK();
    const hipError_t e = sort_pairs(ws + L.tmp, tmp_bytes, kin, kout, vin, vout, total, st);
    if (e != hipSuccess) return (int)e;
    const int ntiles = (int)((total + TILE - 1) / TILE);
    const dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
    const int nc = (width + 63) / 64;
#define MERGE_LAUNCH(NC)                                                                          \
    do {                                                                                          \
        hipLaunchKernelGGL(merge_tile_kernel<NC>, grid, block, 0, st, m, kout, vout, meta, part, ntiles); \
        SEQREC_LAUNCH_CHECK();                                                                    \
        hipLaunchKernelGGL(merge_chain_kernel<NC>, grid, block, 0, st, m, meta, part, ntiles);    \
        SEQREC_LAUNCH_CHECK();                                                                    \
    } while (0)
    if (nc <= 1) MERGE_LAUNCH(1);
    else if (nc <= 2) MERGE_LAUNCH(2);
    else if (nc <= 4) MERGE_LAUNCH(4);
    else if (nc <= 8) MERGE_LAUNCH(8);
    else if (nc <= 16) MERGE_LAUNCH(16);
    else MERGE_LAUNCH(32);
#undef MERGE_LAUNCH
    return 0;
}

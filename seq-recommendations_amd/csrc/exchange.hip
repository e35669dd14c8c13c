// Row-exchange plumbing of the multi-GPU step (distributed.py; SURVEY 8e: item tables row-sharded, rows moved by
// all-to-all).  No reference counterpart (the reference is single-process).  Three launches bracket the two collectives of
// a step -- each replaces a chain of gathers / index rewrites / split-K reduce launches of the round-2 step:
//   exchange_pack       owner side, before all-to-all #1: requested rows + this rank's stratified negative draws + their ids
//   exchange_unpack     requester side, after it: the K negative rows into one contiguous matrix, their global ids, log-Q
//   exchange_grad_pack  requester side, before all-to-all #2: the row gradients in routing order, taken straight from the
//                       split-K slabs of dX / dEneg and from dlt * Hd (no reduce launch, no staging copy)
// All are HBM-bound row copies: one wave per row, 16 bytes per lane.
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ int draw_alias(uint64_t key, uint64_t step, int K, int k, const uint32_t* __restrict__ thresh,
                                          const int* __restrict__ alias, int V) {      // == sample_negatives_kernel (oracle/rng.py)
    const uint64_t r = rand64(key, step * (uint64_t)K + (uint64_t)k);
    const uint64_t hi = r >> 32;
    const uint32_t lo = (uint32_t)(r & 0xFFFFFFFFu);
    const int j = (int)((hi * (uint64_t)V) >> 32);
    return lo < thresh[j] ? j : alias[j];
}
__device__ __forceinline__ void copy_row(float* __restrict__ dst, const float* __restrict__ src, int width, int lane) {
    if ((width & 3) == 0) {
        for (int c = lane; c < width / 4; c += 64) reinterpret_cast<float4*>(dst)[c] = src ? reinterpret_cast<const float4*>(src)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int c = lane; c < width; c += 64) dst[c] = src ? src[c] : 0.f;
    }
}

struct PackArgs {
    const float* table; long table_rows; int width;
    const int* kinds; long m_tot;                  // owner-side rows: >= 0 local table row, -1 id row, -2 negative row
    const int* got; long got_len;                  // non-null: kinds[j] >= 0 is an INDEX into the received request list `got`
    uint64_t key, step; int n_neg, V_local, row_offset;
    const uint32_t* thresh; const int* alias;
    const int* neg_slots;                          // [n_neg] owner-side row of negative i
    const int* id_rows; int n_id_rows, per_peer, id_rows_per_peer, id_mul, id_add;   // id row (p, r): ids of negatives p*per_peer + r*width + c
    float* sendbuf; int* rows_eff; unsigned* status;
    // per-item output bias (nullable, all three together): bias_out[j] = the bias of the item owner-side row j stands for (rows of
    // the output table and my draws; 0 elsewhere), bias_rows[j] = its row in the bias table or -1 -- the scatter list of the
    // bias gradients that come back in the same positions
    const float* bias_table; float* bias_out; int* bias_rows;
};
__global__ void exchange_pack_kernel(PackArgs a) {
    const int lane = threadIdx.x & 63;
    const long wv = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wv < a.m_tot) {                            // requested rows
        int k = a.kinds[wv];
        if (k == -2) return;
        if (k >= 0 && a.got) {                     // the row a peer asked for: its local row number arrived by all-to-all
            if (k < a.got_len) k = a.got[k];
            else { if (lane == 0 && a.status) atomicOr(a.status, (unsigned)SEQREC_STATUS_BAD_INDEX); k = (int)a.table_rows; }
            if (k < 0) k = (int)a.table_rows;      // a negative row number from a peer is a bad index too (zero row + status)
        }
        int r = k;
        if (k >= a.table_rows) { if (lane == 0 && a.status) atomicOr(a.status, (unsigned)SEQREC_STATUS_BAD_INDEX); r = -1; }
        if (k == -1) { if (lane == 0) { a.rows_eff[wv] = -1; if (a.bias_out) { a.bias_out[wv] = 0.f; a.bias_rows[wv] = -1; } } return; }      // id rows are written whole by their own waves below
        copy_row(a.sendbuf + wv * a.width, r >= 0 ? a.table + (long)r * a.width : nullptr, a.width, lane);
        if (lane == 0) {
            a.rows_eff[wv] = r;
            if (a.bias_out) {
                const int br = r >= a.row_offset ? r - a.row_offset : -1;      // rows of the output table carry a bias (tied: every row)
                a.bias_out[wv] = br >= 0 ? a.bias_table[br] : 0.f;
                a.bias_rows[wv] = br;
            }
        }
        return;
    }
    long i = wv - a.m_tot;
    if (i < a.n_neg) {                             // this rank's draws for every requester
        const int id = draw_alias(a.key, a.step, a.n_neg, (int)i, a.thresh, a.alias, a.V_local);
        const int r = a.row_offset + id, pos = a.neg_slots[i];
        copy_row(a.sendbuf + (long)pos * a.width, a.table + (long)r * a.width, a.width, lane);
        if (lane == 0) {
            a.rows_eff[pos] = r;
            if (a.bias_out) { a.bias_out[pos] = a.bias_table[id]; a.bias_rows[pos] = id; }
        }
        return;
    }
    i -= a.n_neg;
    if (i < a.n_id_rows) {                         // id rows: global ids of the draws, bit-cast into the float buffer; rest zero
        const int p = (int)(i / a.id_rows_per_peer), rr = (int)(i % a.id_rows_per_peer);
        int* dst = reinterpret_cast<int*>(a.sendbuf + (long)a.id_rows[i] * a.width);
        for (int c = lane; c < a.width; c += 64) {
            const int q = rr * a.width + c;
            dst[c] = q < a.per_peer ? draw_alias(a.key, a.step, a.n_neg, p * a.per_peer + q, a.thresh, a.alias, a.V_local) * a.id_mul + a.id_add : 0;
        }
    }
}

__global__ void exchange_unpack_kernel(const float* __restrict__ recv, int width, const int* __restrict__ neg_rows,
                                       const int* __restrict__ negid_idx, int K, const float* __restrict__ logq, long logq_rows,
                                       float* __restrict__ Eneg, int* __restrict__ neg, float* __restrict__ lq_neg,
                                       unsigned* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= K) return;
    copy_row(Eneg + (long)k * width, recv + (long)neg_rows[k] * width, width, lane);
    if (lane == 0) {
        const int id = reinterpret_cast<const int*>(recv)[negid_idx[k]];
        neg[k] = id;
        if (lq_neg) {                              // the id was written by a PEER: never index with it unchecked
            const bool ok = id >= 0 && (long)id < logq_rows;
            lq_neg[k] = ok ? logq[id] : 0.f;
            if (!ok && status) atomicOr(status, SEQREC_STATUS_BAD_INDEX);
        }
    }
}

struct GradPackArgs {
    const int* back_idx; long n_tot; int n, K, width;
    const float* dX; int dx_slabs; long dx_stride;
    const float* Hd; const float* dlt;
    const float* dEneg; int dn_slabs; long dn_stride;
    float* out;
    const float* dbn; float* bias_grad;              // nullable: bias_grad[j] = dlt of a target row / dbn[k] of negative k / 0
};
__global__ void exchange_grad_pack_kernel(GradPackArgs a) {
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= a.n_tot) return;
    const int b = a.back_idx[j];
    if (a.bias_grad && lane == 0)
        a.bias_grad[j] = (b >= a.n && b < 2 * a.n) ? a.dlt[b - a.n] : ((b >= 2 * a.n && b < 2 * a.n + a.K) ? a.dbn[b - 2 * a.n] : 0.f);
    float* dst = a.out + j * a.width;
    const int w4 = a.width / 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < 0 || b >= 2 * a.n + a.K) {
        for (int c = lane; c < w4; c += 64) reinterpret_cast<float4*>(dst)[c] = z4;
    } else if (b >= a.n && b < 2 * a.n) {                                  // target row gradient: dlt * Hd
        const float s = a.dlt[b - a.n];
        const float4* src = reinterpret_cast<const float4*>(a.Hd + (long)(b - a.n) * a.width);
        for (int c = lane; c < w4; c += 64) { const float4 v = src[c]; reinterpret_cast<float4*>(dst)[c] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }
    } else {                                                               // dX / dEneg row: the slabs added in slab order
        const bool in = b < a.n;
        const float* base = in ? a.dX + (long)b * a.width : a.dEneg + (long)(b - 2 * a.n) * a.width;
        const int ns = in ? a.dx_slabs : a.dn_slabs;
        const long st = in ? a.dx_stride : a.dn_stride;
        for (int c = lane; c < w4; c += 64) {
            float4 v = reinterpret_cast<const float4*>(base)[c];
            for (int s = 1; s < ns; ++s) { const float4 u = reinterpret_cast<const float4*>(base + s * st)[c]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            reinterpret_cast<float4*>(dst)[c] = v;
        }
    }
}

}  // namespace

extern "C" int seqrec_exchange_pack(const float* table, int64_t table_rows, int width, const int32_t* kinds, const int32_t* got,
                                    int64_t got_len, int64_t m_tot, uint64_t seed, uint64_t step, int n_neg, const uint32_t* thresh, const int32_t* alias,
                                    int V_local, int32_t row_offset, const int32_t* neg_slots, const int32_t* id_rows,
                                    int n_id_rows, int per_peer, int32_t id_mul, int32_t id_add, float* sendbuf,
                                    int32_t* rows_eff, uint32_t* status, const float* bias_table, float* bias_out,
                                    int32_t* bias_rows, void* stream) {
    if ((bias_table || bias_out || bias_rows) && !(bias_table && bias_out && bias_rows)) return SEQREC_E_ARG;
    if (m_tot < 0 || width <= 0 || (width & 3) || n_neg < 0 || n_id_rows < 0 || table_rows <= 0) return SEQREC_E_ARG;
    if (!table || !kinds || !sendbuf || !rows_eff || (got && got_len < 0)) return SEQREC_E_ARG;
    if (n_neg > 0 && (!thresh || !alias || !neg_slots || V_local <= 0 || per_peer <= 0 || n_neg % per_peer)) return SEQREC_E_ARG;
    if (n_id_rows > 0 && (!id_rows || n_neg <= 0 || n_id_rows % (n_neg / per_peer))) return SEQREC_E_ARG;
    if ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(sendbuf)) & 15) return SEQREC_E_ARG;
    const long waves = (long)m_tot + n_neg + n_id_rows;
    if (waves == 0) return 0;
    PackArgs a = {};
    a.table = table; a.table_rows = (long)table_rows; a.width = width; a.kinds = kinds; a.m_tot = (long)m_tot; a.got = got; a.got_len = (long)got_len;
    a.key = key64(seed, 1); a.step = step; a.n_neg = n_neg; a.V_local = V_local; a.row_offset = row_offset;
    a.thresh = thresh; a.alias = alias; a.neg_slots = neg_slots; a.id_rows = id_rows; a.n_id_rows = n_id_rows;
    a.per_peer = per_peer; a.id_rows_per_peer = n_neg > 0 ? n_id_rows / (n_neg / per_peer) : 1; a.id_mul = id_mul; a.id_add = id_add;
    a.sendbuf = sendbuf; a.rows_eff = rows_eff; a.status = status;
    a.bias_table = bias_table; a.bias_out = bias_out; a.bias_rows = bias_rows;
    hipLaunchKernelGGL(exchange_pack_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), a);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int seqrec_exchange_unpack(const float* recv, int width, const int32_t* neg_rows, const int32_t* negid_idx, int K,
                                      const float* logq, int64_t logq_rows, float* Eneg, int32_t* neg, float* lq_neg,
                                      uint32_t* status, void* stream) {
    if (K < 0 || width <= 0 || (width & 3) || (lq_neg && logq_rows <= 0)) return SEQREC_E_ARG;
    if (K == 0) return 0;
    if (!recv || !neg_rows || !negid_idx || !Eneg || !neg || (lq_neg && !logq)) return SEQREC_E_ARG;
    if ((reinterpret_cast<uintptr_t>(recv) | reinterpret_cast<uintptr_t>(Eneg)) & 15) return SEQREC_E_ARG;
    hipLaunchKernelGGL(exchange_unpack_kernel, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, as_stream(stream), recv, width, neg_rows,
                       negid_idx, K, logq, (long)logq_rows, Eneg, neg, lq_neg, status);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int seqrec_exchange_grad_pack(const int32_t* back_idx, int64_t n_tot, int n, int K, int width, const float* dX,
                                         int dx_slabs, int64_t dx_stride, const float* Hd, const float* dlt, const float* dEneg,
                                         int dn_slabs, int64_t dn_stride, float* out, const float* dbn, float* bias_grad,
                                         void* stream) {
    if (bias_grad && K > 0 && !dbn) return SEQREC_E_ARG;
    if (n_tot < 0 || n < 0 || K < 0 || width <= 0 || (width & 3) || dx_slabs < 1 || dn_slabs < 1) return SEQREC_E_ARG;
    if (n_tot == 0) return 0;
    if (!back_idx || !out || (n > 0 && (!dX || !Hd || !dlt)) || (K > 0 && !dEneg)) return SEQREC_E_ARG;
    if ((dx_slabs > 1 && dx_stride < (int64_t)n * width) || (dn_slabs > 1 && dn_stride < (int64_t)K * width)) return SEQREC_E_ARG;
    if ((reinterpret_cast<uintptr_t>(dX) | reinterpret_cast<uintptr_t>(Hd) | reinterpret_cast<uintptr_t>(dEneg) | reinterpret_cast<uintptr_t>(out)) & 15) return SEQREC_E_ARG;
    if (((dx_stride | dn_stride) & 3) != 0) return SEQREC_E_ARG;
    GradPackArgs a = {};
    a.back_idx = back_idx; a.n_tot = (long)n_tot; a.n = n; a.K = K; a.width = width;
    a.dX = dX; a.dx_slabs = dx_slabs; a.dx_stride = (long)dx_stride; a.Hd = Hd; a.dlt = dlt;
    a.dEneg = dEneg; a.dn_slabs = dn_slabs; a.dn_stride = (long)dn_stride; a.out = out;
    a.dbn = dbn; a.bias_grad = bias_grad;
    hipLaunchKernelGGL(exchange_grad_pack_kernel, dim3((unsigned)((n_tot + 3) / 4)), dim3(256), 0, as_stream(stream), a);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

// Bandwidth-side kernels of the hot path: row gather, softmax/CE rows, bias column sums, the
// row-sparse gradient path (scatter-add / norm / Adagrad) and the dense optimizer, counter RNG.
// All are HBM/L2-bound integer-index + fp32 streaming work: 16 B per lane where rows allow it,
// one wave per table row so that a 256-float row is exactly one 1-KiB coalesced wave access.
#include "common.h"
#include <limits.h>
#include <cstdlib>
#include <algorithm>

namespace {

// ---------------------------------------------------------------------------------------------
// gather: out[i,:] = table[ids[i],:] * scale[i] + bias   (algorithmic bytes: 8*width per row)
// ---------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ void gather_rows_kernel(const float* __restrict__ table, const int* __restrict__ ids,
                                   float* __restrict__ out, long n, int width, const float* __restrict__ row_scale,
                                   const float* __restrict__ bias, int accumulate, long table_rows, unsigned* __restrict__ status) {
    // table_rows > 0: an id >= table_rows reads NOTHING (zero row) and is recorded in *status -- a stale or corrupted index
    // then cannot pull arbitrary memory into a gradient (seqrec_gather_rows_bounded)
    const int per = VEC4 ? width / 4 : width;
    const long total = n * per;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
        const long i = c / per;
        const int q = (int)(c % per);
        int id = ids[i];
        if (table_rows > 0 && id >= table_rows) {
            if (q == 0 && status) atomicOr(status, (unsigned)SEQREC_STATUS_BAD_INDEX);
            id = -1;
        }
        const float s = (row_scale && id >= 0) ? row_scale[i] : 1.f;
        if (VEC4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (id >= 0) v = reinterpret_cast<const float4*>(table + (long)id * width)[q];
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            if (bias) {
                const float4 b = reinterpret_cast<const float4*>(bias)[q];
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            float4* op = reinterpret_cast<float4*>(out + i * width) + q;
            if (accumulate) { const float4 o = *op; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *op = v;
        } else {
            float v = id >= 0 ? table[(long)id * width + q] : 0.f;
            v *= s;
            if (bias) v += bias[q];
            if (accumulate) v += out[i * width + q];
            out[i * width + q] = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// full softmax + Keras/Theano CE, one wave per token row
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void full_softmax_ce_row(float* __restrict__ logits, long ld, const int* __restrict__ tgt, long row,
                                                    int V, float inv_denom, float* __restrict__ loss_rows,
                                                    float* __restrict__ probs, int lane) {
    float* x = logits + row * ld;
    float m = -INFINITY;
    for (int j = lane; j < V; j += 64) m = fmaxf(m, x[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < V; j += 64) s += expf(x[j] - m);
    s = wave_sum(s);
    // Theano-backend categorical_crossentropy renormalises the softmax output once more
    float sp = 0.f;
    for (int j = lane; j < V; j += 64) sp += expf(x[j] - m) / s;
    sp = wave_sum(sp);
    float active = 0.f;
    int t = -1;
    if (tgt) {
        t = tgt[row];
        const float pt = (expf(x[t] - m) / s) / sp;
        const float lo = 1e-7f, hi = 1.0f - 1e-7f;
        active = (pt >= lo && pt <= hi) ? inv_denom : 0.f;
        if (lane == 0) loss_rows[row] = -logf(fminf(fmaxf(pt, lo), hi));
    }
    for (int j = lane; j < V; j += 64) {
        const float p = expf(x[j] - m) / s;
        if (probs) probs[row * V + j] = p;
        if (tgt) x[j] = (p - (j == t ? 1.f : 0.f)) * active;
    }
}

__global__ void full_softmax_ce_kernel(float* __restrict__ logits, long ld, const int* __restrict__ tgt, long n,
                                       int V, float inv_denom, float* __restrict__ loss_rows,
                                       float* __restrict__ probs) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row < n) full_softmax_ce_row(logits, ld, tgt, row, V, inv_denom, loss_rows, probs, lane);
}

// ---------------------------------------------------------------------------------------------
// sampled softmax + CE over {target} U K negatives, one wave per token row
// ---------------------------------------------------------------------------------------------
// ROWS = false: target row = Eout[tgt[i]], bout/logq indexed by item id (single-GPU tables)
// ROWS = true : target row = Eout[i] (rows already fetched from their owners), lq_t[i] / lq_n[k]
//               are the per-candidate log-Q values (multi-GPU, row-sharded tables)
template <bool ROWS>
__device__ __forceinline__ void sampled_softmax_ce_row(float* __restrict__ ln, long ld, const float* __restrict__ hd, int H,
                                                       const float* __restrict__ Eout, const float* __restrict__ bout,
                                                       const float* __restrict__ logq, const float* __restrict__ lq_n,
                                                       const int* __restrict__ tgt, const int* __restrict__ neg, long row, int K,
                                                       float inv_denom, float* __restrict__ loss_rows, float* __restrict__ dlt,
                                                       int lane, const int* __restrict__ trow = nullptr, long et_ld = 0) {
    const int t = tgt[row];
    const float* h = hd + row * H;
    const float* et = ROWS ? Eout + (trow ? (long)trow[row] : row) * (et_ld ? et_ld : (long)H) : Eout + (long)t * H;
    float d = 0.f;
    for (int j = lane; j < H; j += 64) d += h[j] * et[j];
    float lt = wave_sum(d);
    if (ROWS) {
        if (logq) lt -= logq[row];
    } else {
        if (bout) lt += bout[t];
        if (logq) lt -= logq[t];
    }
    float* x = ln + row * ld;
    float m = lt;
    for (int k = lane; k < K; k += 64) {
        const int v = neg[k];
        float l = x[k];
        if (ROWS) {
            if (lq_n) l -= lq_n[k];
        } else {
            if (bout) l += bout[v];
            if (lq_n) l -= lq_n[k];
            else if (logq) l -= logq[v];
        }
        if (v == t) l = -INFINITY;
        x[k] = l;
        m = fmaxf(m, l);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += expf(x[k] - m);
    const float etg = expf(lt - m);
    s = wave_sum(s) + etg;
    const float pt = etg / s;
    const float lo = 1e-7f, hi = 1.0f - 1e-7f;
    const float active = (pt >= lo && pt <= hi) ? inv_denom : 0.f;
    if (lane == 0) {
        loss_rows[row] = -logf(fminf(fmaxf(pt, lo), hi));
        dlt[row] = (pt - 1.f) * active;
    }
    for (int k = lane; k < K; k += 64) x[k] = (expf(x[k] - m) / s) * active;
}
template <bool ROWS>
__global__ void sampled_softmax_ce_kernel(float* __restrict__ ln, long ld, const float* __restrict__ hd, int H,
                                          const float* __restrict__ Eout, const float* __restrict__ bout,
                                          const float* __restrict__ logq, const float* __restrict__ lq_n,
                                          const int* __restrict__ tgt, const int* __restrict__ neg, long n, int K,
                                          float inv_denom, float* __restrict__ loss_rows, float* __restrict__ dlt,
                                          const int* __restrict__ trow, long et_ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row < n) sampled_softmax_ce_row<ROWS>(ln, ld, hd, H, Eout, bout, logq, lq_n, tgt, neg, row, K, inv_denom, loss_rows, dlt, lane, trow, et_ld);
}

__global__ void reduce_sum_kernel(const float* __restrict__ x, long n, float* __restrict__ out, int accumulate) {
    __shared__ float part[1024];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + part[0];
}

// column sums: block (64 columns x 4 row-lanes), rows strided over gridDim.y; deterministic two-stage
__global__ void colsum_partial_kernel(const float* __restrict__ X, long n, int width, long ld, float* __restrict__ part) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    __shared__ float sh[4][64];
    float s = 0.f;
    if (col < width)
        for (long i = (long)blockIdx.y * 4 + rl; i < n; i += (long)gridDim.y * 4) s += X[i * ld + col];
    sh[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < width) part[(long)blockIdx.y * width + col] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int nparts, int width, float* __restrict__ out, int accumulate) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= width) return;
    float s = 0.f;
#pragma unroll 8
    for (int p = 0; p < nparts; ++p) s += part[(long)p * width + col];   // fixed order: deterministic
    out[col] = (accumulate ? out[col] : 0.f) + s;
}

__global__ void mul_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = x[i] * m[i];
}
__global__ void fill_f32_kernel(float* x, float v, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void fill_i32_kernel(int* x, int v, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = v;
}

// ---------------------------------------------------------------------------------------------
// row-sparse gradient path; one wave per contribution row
// ---------------------------------------------------------------------------------------------
__global__ void rows_scatter_add_kernel(float* __restrict__ gtab, int* __restrict__ slot, const int* __restrict__ rows,
                                        const float* __restrict__ vals, long ldv, const float* __restrict__ row_scale,
                                        long n, int width, int base) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r = rows[i];
    if (r < 0) return;                              // filler contribution (wave-uniform)
    const float s = row_scale ? row_scale[i] : 1.f;
    float* g = gtab + (long)r * width;
    const float* v = vals + i * ldv;
    for (int c = lane; c < width; c += 64) atomicAdd(g + c, v[c] * s);
    if (lane == 0) atomicMin(slot + r, base + (int)i);
}

__global__ void rows_sqnorm_kernel(const float* __restrict__ gtab, const int* __restrict__ slot,
                                   const int* __restrict__ rows, long n, int width, int base, float* __restrict__ sq) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + wv;
    __shared__ float part[16];
    float s = 0.f;
    if (i < n) {
        const int r = rows[i];
        if (r >= 0 && slot[r] == base + (int)i) {
            const float* g = gtab + (long)r * width;
            for (int c = lane; c < width; c += 64) s += g[c] * g[c];
        }
    }
    s = wave_sum(s);
    if (lane == 0) part[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += part[k];
        if (t != 0.f) atomicAdd(sq, t);
    }
}

__global__ void rows_adagrad_kernel(float* __restrict__ table, float* __restrict__ accum, float* __restrict__ gtab,
                                    int* __restrict__ slot, const int* __restrict__ rows, long n, int width, int base,
                                    float lr, float eps, const float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r = rows[i];
    if (r < 0 || slot[r] != base + (int)i) return;  // wave-uniform: exactly one owner per touched row
    const float sc = scale[0];
    const long o = (long)r * width;
    for (int c = lane; c < width; c += 64) {
        const float g = gtab[o + c] * sc;
        const float a = accum[o + c] + g * g;
        accum[o + c] = a;
        table[o + c] -= lr * g / (sqrtf(a) + eps);
        gtab[o + c] = 0.f;
    }
    if (lane == 0) slot[r] = INT_MAX;
}

// dense optimizer
__global__ void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ sq) {
    __shared__ float part[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sq, part[0] + part[1] + part[2] + part[3]);
}
__global__ void clip_scale_kernel(const float* __restrict__ sq, float clipnorm, float* __restrict__ scale) {
    const float nrm = sqrtf(sq[0]);
    scale[0] = (clipnorm > 0.f && nrm >= clipnorm) ? clipnorm / nrm : 1.f;   // Keras clip_norm
}
__global__ void adagrad_dense_kernel(float* __restrict__ p, float* __restrict__ a, const float* __restrict__ g, long n,
                                     float lr, float eps, const float* __restrict__ scale) {
    const float sc = scale[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gv = g[i] * sc;
        const float av = a[i] + gv * gv;
        a[i] = av;
        p[i] -= lr * gv / (sqrtf(av) + eps);
    }
}

// multi-tensor forms: one launch for all dense tensors / all scatter lists of a step
struct DenseMulti { float* p[8]; float* a[8]; const float* g[8]; long n[8]; };
__global__ void sqnorm_multi_kernel(DenseMulti m, float* __restrict__ sq) {
    const float* g = m.g[blockIdx.y];
    const long n = m.n[blockIdx.y];
    __shared__ float part[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = part[0] + part[1] + part[2] + part[3];
        if (t != 0.f) atomicAdd(sq, t);
    }
}
__global__ void adagrad_multi_kernel(DenseMulti m, float lr, float eps, const float* __restrict__ scale) {
    float* p = m.p[blockIdx.y];
    float* a = m.a[blockIdx.y];
    const float* g = m.g[blockIdx.y];
    const long n = m.n[blockIdx.y];
    const float sc = scale[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gv = g[i] * sc;
        const float av = a[i] + gv * gv;
        a[i] = av;
        p[i] -= lr * gv / (sqrtf(av) + eps);
    }
}

struct RowsMulti { seqrec_rows_job j[4]; };
__global__ void rows_scatter_add_multi_kernel(RowsMulti m) {
    // every wave owns RPW consecutive contributions and issues their loads together (row ids, then all value pieces)
    // before the first atomic: one contribution per wave was a chain of dependent round trips (id -> 4 x (load -> atomic))
    // and four times as many workgroups to dispatch
    constexpr int RPW = 4;
    const seqrec_rows_job& J = m.j[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const long i0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW;
    if (i0 >= J.n) return;
    int r[RPW];
    float s[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) r[k] = (i0 + k < J.n) ? J.rows[i0 + k] : -1;
#pragma unroll
    for (int k = 0; k < RPW; ++k) s[k] = (J.row_scale && r[k] >= 0) ? J.row_scale[i0 + k] : 1.f;
    if (J.width == 256) {
        float x[RPW][4];
#pragma unroll
        for (int k = 0; k < RPW; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[k][j] = r[k] >= 0 ? J.vals[(i0 + k) * J.ldv + lane + 64 * j] : 0.f;
        for (int sl = 1; sl < J.n_slabs; ++sl) {
            const float* v = J.vals + (long)sl * J.slab_stride;
#pragma unroll
            for (int k = 0; k < RPW; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) x[k][j] += r[k] >= 0 ? v[(i0 + k) * J.ldv + lane + 64 * j] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            if (r[k] < 0) continue;
            float* g = J.gtab + (long)r[k] * 256;
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(g + lane + 64 * j, x[k][j] * s[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            if (r[k] < 0) continue;
            float* g = J.gtab + (long)r[k] * J.width;
            const float* v = J.vals + (i0 + k) * J.ldv;
            for (int c = lane; c < J.width; c += 64) {
                float xv = v[c];
                for (int sl = 1; sl < J.n_slabs; ++sl) xv += v[(long)sl * J.slab_stride + c];
                atomicAdd(g + c, xv * s[k]);
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < RPW; ++k)
            if (r[k] >= 0) atomicMin(J.slot + r[k], J.base + (int)(i0 + k));
    }
}
// The same with the contributions of a workgroup (12 waves x RPW = 48) COMBINED per row before the atomics.  Item ids are
// Zipf: in a c3 batch ~2 500 input contributions hit ~950 rows and the most popular row takes ~250 of them; float atomics
// on one address serialise (~20 ns each: tools/rows_probe.py, 18.8 us with the batch's rows against 13.8 us with as many
// distinct rows).  Every wave parks its scaled rows in LDS; the FIRST contribution of a row inside the workgroup (ballot
// over the 48 ids) adds the others' rows in index order and issues the only atomics for that row.
template <int NC>
__global__ void __launch_bounds__(768) rows_scatter_combine_kernel(RowsMulti m) {
    // rows wider than 256 (c4: 512) go in column blocks of W = 256: blockIdx.z picks the block, the LDS image stays 48 KB
    constexpr int RPW = 4, NW = 12, CPB = RPW * NW, W = 64 * NC;
    __shared__ float vals[CPB * W];
    __shared__ int rid[64];
    seqrec_rows_job J = m.j[blockIdx.y];
    const int width = J.width;                          // row stride of the gradient table
    J.vals += (long)blockIdx.z * W;
    J.gtab += (long)blockIdx.z * W;
    const long b0 = (long)blockIdx.x * CPB;
    if (b0 >= J.n) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long i0 = b0 + wv * RPW;
    int r[RPW];
    float s[RPW], x[RPW][NC];
#pragma unroll
    for (int k = 0; k < RPW; ++k) r[k] = (i0 + k < J.n) ? J.rows[i0 + k] : -1;
#pragma unroll
    for (int k = 0; k < RPW; ++k) s[k] = (J.row_scale && r[k] >= 0) ? J.row_scale[i0 + k] : 1.f;
#pragma unroll
    for (int k = 0; k < RPW; ++k)
#pragma unroll
        for (int j = 0; j < NC; ++j) x[k][j] = r[k] >= 0 ? J.vals[(i0 + k) * J.ldv + lane + 64 * j] : 0.f;
    for (int sl = 1; sl < J.n_slabs; ++sl) {            // split-K slabs of the producing GEMM, added in slab order
        const float* v = J.vals + (long)sl * J.slab_stride;
#pragma unroll
        for (int k = 0; k < RPW; ++k)
#pragma unroll
            for (int j = 0; j < NC; ++j) x[k][j] += r[k] >= 0 ? v[(i0 + k) * J.ldv + lane + 64 * j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k)
#pragma unroll
        for (int j = 0; j < NC; ++j) x[k][j] *= s[k];
    if (threadIdx.x >= CPB && threadIdx.x < 64) rid[threadIdx.x] = -1;
    if (lane < RPW) {
        int mine = r[0];
#pragma unroll
        for (int k = 1; k < RPW; ++k) if (lane == k) mine = r[k];
        rid[wv * RPW + lane] = mine;
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k)
#pragma unroll
        for (int j = 0; j < NC; ++j) vals[(wv * RPW + k) * W + lane + 64 * j] = x[k][j];
    __syncthreads();
    const int other = rid[lane];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        if (r[k] < 0) continue;                                          // wave-uniform
        const int c = wv * RPW + k;
        unsigned long long same = __ballot(other == r[k]);
        if (__ffsll((long long)same) - 1 != c) continue;                 // an earlier contribution of the workgroup leads this row
        same &= same - 1;                                                // the others, in index order
        float acc[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[j] = x[k][j];
        while (same) {
            const int o = __ffsll((long long)same) - 1;
            same &= same - 1;
#pragma unroll
            for (int j = 0; j < NC; ++j) acc[j] += vals[o * W + lane + 64 * j];
        }
        float* g = J.gtab + (long)r[k] * width;
#pragma unroll
        for (int j = 0; j < NC; ++j) atomicAdd(g + lane + 64 * j, acc[j]);
        if (lane == 0 && blockIdx.z == 0) atomicMin(J.slot + r[k], J.base + (int)(i0 + k));   // the leader is the smallest index of its row here
    }
}
__global__ void __launch_bounds__(1024) rows_sqnorm_multi_kernel(RowsMulti m, float* __restrict__ sq) {
    // every wave owns RPW consecutive contributions and issues their (random, HBM-latency-bound)
    // accesses together: RPW slot reads, then RPW predicated row reads -- instead of RPW dependent
    // round trips; one same-address atomic per 16*RPW contributions (1024-thread workgroups: see opt_sqnorm_kernel).
    constexpr int RPW = 4;
    const seqrec_rows_job& J = m.j[blockIdx.y];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long i0 = ((long)blockIdx.x * (blockDim.x >> 6) + wv) * RPW;
    __shared__ float part[16];
    int r[RPW];
    bool own[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) r[k] = (i0 + k < J.n) ? J.rows[i0 + k] : -1;
#pragma unroll
    for (int k = 0; k < RPW; ++k) own[k] = r[k] >= 0 && J.slot[r[k]] == J.base + (int)(i0 + k);
    float s = 0.f;
    for (int c = lane; c < J.width; c += 64) {
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const float g = own[k] ? J.gtab[(long)r[k] * J.width + c] : 0.f;
            s += g * g;
        }
    }
    s = wave_sum(s);
    if (lane == 0) part[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
        if (t != 0.f) atomicAdd(sq, t);
    }
}
__global__ void rows_adagrad_multi_kernel(RowsMulti m, float lr, float eps, const float* __restrict__ scale) {
    const seqrec_rows_job& J = m.j[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= J.n) return;
    const int r = J.rows[i];
    if (r < 0 || J.slot[r] != J.base + (int)i) return;
    const float sc = scale[0];
    const long o = (long)r * J.width;
    for (int c = lane; c < J.width; c += 64) {
        const float g = J.gtab[o + c] * sc;
        const float a = J.accum[o + c] + g * g;
        J.accum[o + c] = a;
        J.table[o + c] -= lr * g / (sqrtf(a) + eps);
        J.gtab[o + c] = 0.f;
    }
    if (lane == 0) J.slot[r] = INT_MAX;
}

// fused optimizer launches: the norm of EVERYTHING (dense tensors + owned rows) in one launch, then clip
// scale + dense Adagrad + row-sparse Adagrad in one launch (the step otherwise spends six ~5 us
// launches here).  blockIdx.y < nd: dense tensor, else scatter list blockIdx.y - nd.
struct OptPlan { DenseMulti d; RowsMulti r; int nd, nr; const float* loss_rows; long n_loss; float* loss_out; };
// batch loss in the same launch as the gradient norm (one spare workgroup): loss_out[0] = sum of the per-token CE in a
// fixed order, loss_out[1] = its token mean -- no separate reduction launch, no host-side division
__device__ __forceinline__ void block_loss_reduce(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float red[256];             // the first 256 threads of the workgroup add, whatever its size: one summation order
    if (threadIdx.x < 256) {
        float s = 0.f;
        for (long i = threadIdx.x; i < n; i += 256) s += x[i];
        red[threadIdx.x] = s;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = red[0]; out[1] = red[0] / (float)n; }
}
__global__ void loss_reduce_kernel(const float* __restrict__ x, long n, float* __restrict__ out) { block_loss_reduce(x, n, out); }
// 1024-thread workgroups: every workgroup ends in ONE atomic on the same address, and those serialise at ~10 ns each
// (tools/rows_probe.py: the launch took 15 us whatever the rows were, with ~940 workgroups of 256)
// products that arrive as split-K slabs (seqrec_gemm_f32_grouped_slabs): grid rows nd .. nd + np - 1 add the slabs in slab
// order (== the reduce launch of the reducing form), WRITE the product to its place (C, row stride ldc) for the update
// launch, and add its squares to the norm -- the reduce launch of the weight gradients rides in the norm launch
struct SlabPieces { const float* ws[4]; float* C[4]; long M[4], N[4], ldc[4]; int np, n_slabs; };
__global__ void __launch_bounds__(1024) opt_sqnorm_kernel(OptPlan pl, SlabPieces sp, float* __restrict__ sq) {
    if ((int)blockIdx.y == pl.nd + sp.np + pl.nr) {         // spare row: the batch loss
        if (blockIdx.x == 0) block_loss_reduce(pl.loss_rows, pl.n_loss, pl.loss_out);
        return;
    }
    __shared__ float part[16];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if ((int)blockIdx.y < pl.nd) {
        const float* g = pl.d.g[blockIdx.y];
        const long n = pl.d.n[blockIdx.y];
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
    } else if ((int)blockIdx.y < pl.nd + sp.np) {
        const int p = blockIdx.y - pl.nd;
        const float* __restrict__ ws = sp.ws[p];
        const long N = sp.N[p], total = sp.M[p] * N, ldc = sp.ldc[p];
        // four elements per thread and trip, their loads issued together slab by slab: a trip was (n_slabs dependent-looking loads ->
        // store), five trips for a 256 x 768 gradient on this launch's 40 workgroups -- a chain of memory latencies, not bandwidth
        // (each element still sums its slabs in slab order: the same bits as the grouped reduce launch)
        const long stride = (long)gridDim.x * blockDim.x;
        for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < sp.n_slabs; ++z) {
                const float* __restrict__ w = ws + (long)z * total;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long i = i0 + u * stride;
                    v[u] += i < total ? w[i] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long i = i0 + u * stride;
                if (i < total) {
                    sp.C[p][ldc == N ? i : (i / N) * ldc + i % N] = v[u];
                    s += v[u] * v[u];
                }
            }
        }
    } else {
        constexpr int RPW = 4;
        const seqrec_rows_job& J = pl.r.j[blockIdx.y - pl.nd - sp.np];
        const long i0 = ((long)blockIdx.x * (blockDim.x >> 6) + wv) * RPW;
        int r[RPW];
        bool own[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) r[k] = (i0 + k < J.n) ? J.rows[i0 + k] : -1;
#pragma unroll
        for (int k = 0; k < RPW; ++k) own[k] = r[k] >= 0 && J.slot[r[k]] == J.base + (int)(i0 + k);
        if ((J.width & 3) == 0 && ((uintptr_t)J.gtab & 15) == 0) {       // 16 bytes per lane: a 256-wide row is one load
            for (int c = lane; c < J.width / 4; c += 64) {
#pragma unroll
                for (int k = 0; k < RPW; ++k) {
                    const float4 g = own[k] ? reinterpret_cast<const float4*>(J.gtab + (long)r[k] * J.width)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                    s += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
                }
            }
        } else {
            for (int c = lane; c < J.width; c += 64) {
#pragma unroll
                for (int k = 0; k < RPW; ++k) {
                    const float g = own[k] ? J.gtab[(long)r[k] * J.width + c] : 0.f;
                    s += g * g;
                }
            }
        }
    }
    s = wave_sum(s);
    if (lane == 0) part[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
        if (t != 0.f) atomicAdd(sq, t);
    }
}
// deterministic form of the norm (merge = "sorted"): every block stores its partial sum, a single block adds the
// partials in index order (reduce_sum_kernel) -- no float atomics, bitwise reproducible
__global__ void opt_sqnorm_partial_kernel(OptPlan pl, float* __restrict__ partials) {
    if ((int)blockIdx.y == pl.nd + pl.nr) {
        if (blockIdx.x == 0) block_loss_reduce(pl.loss_rows, pl.n_loss, pl.loss_out);
        return;
    }
    __shared__ float part[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if ((int)blockIdx.y < pl.nd) {
        const float* g = pl.d.g[blockIdx.y];
        const long n = pl.d.n[blockIdx.y];
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
    } else {
        constexpr int RPW = 4;
        const seqrec_rows_job& J = pl.r.j[blockIdx.y - pl.nd];
        const long i0 = ((long)blockIdx.x * (blockDim.x >> 6) + wv) * RPW;
        int r[RPW];
        bool own[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) r[k] = (i0 + k < J.n) ? J.rows[i0 + k] : -1;
#pragma unroll
        for (int k = 0; k < RPW; ++k) own[k] = r[k] >= 0 && J.slot[r[k]] == J.base + (int)(i0 + k);
        for (int c = lane; c < J.width; c += 64) {
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                const float g = own[k] ? J.gtab[(long)r[k] * J.width + c] : 0.f;
                s += g * g;
            }
        }
    }
    s = wave_sum(s);
    if (lane == 0) part[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ void opt_apply_kernel(OptPlan pl, const float* __restrict__ sq, float clipnorm, float lr, float eps,
                                 float* __restrict__ scale_out, float* __restrict__ zero_next, const float* __restrict__ grad_div,
                                 unsigned* __restrict__ status, const float* __restrict__ sq_extra) {
    // grad_div (nullable device scalar): the gradients in memory are SUMS still to be divided by it (the global token
    // count of a multi-GPU step, known only after the all-reduce): norm and update use g / grad_div
    const float sqv = sq_extra ? sq[0] + sq_extra[0] : sq[0], gd = grad_div ? grad_div[0] : 1.f;
    const float inv_div = 1.f / gd;
    const float nrm = sqrtf(sqv) * inv_div;
    const float sc = ((clipnorm > 0.f && nrm >= clipnorm) ? clipnorm / nrm : 1.f) * inv_div;   // Keras clip_norm (== clip_scale_kernel)
    // A squared norm that is not a finite number >= 0, a divisor that is not a finite number > 0 or a scale that is not a
    // finite number > 0 can only come from a wrong input (an overflowing or uninitialised gradient value, a wild token
    // count).  Applied, the first two poison every weight and the last is a SILENT no-op of the whole step (scale 0:
    // gradients, accumulators and weights all unchanged).  Neither happens: the condition is recorded in *status
    // (SEQREC_STATUS_*; the engine raises on it at its next host sync) and the launch leaves everything as it is.
    unsigned bad = 0;
    if (!(sqv >= 0.f) || !isfinite(sqv)) bad |= SEQREC_STATUS_BAD_NORM;
    if (!(gd > 0.f) || !isfinite(gd)) bad |= SEQREC_STATUS_BAD_DIVISOR;
    if (!bad && (!(sc > 0.f) || !isfinite(sc))) bad |= SEQREC_STATUS_BAD_SCALE;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        scale_out[0] = sc;
        if (zero_next) zero_next[0] = 0.f;            // the OTHER norm slot: nobody reads or adds to it during this step
        if (bad && status) atomicOr(status, bad);
    }
    // `bad` is uniform over the launch: every thread read the same three scalars.  A refused step still CLEARS what the row
    // jobs own -- the gradient rows the scatter filled and the owner slots it claimed -- without touching table or
    // accumulator: the next step's scatter must find all-zero gradient rows and free slots (a stale atomicMin claim would
    // keep slot[r] == base + i from ever matching again: the row frozen, its gradient row growing with every step)
    if ((int)blockIdx.y < pl.nd) {
        if (bad) return;
        float* p = pl.d.p[blockIdx.y];
        float* a = pl.d.a[blockIdx.y];
        const float* g = pl.d.g[blockIdx.y];
        const long n = pl.d.n[blockIdx.y];
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
            const float gv = g[i] * sc;
            const float av = a[i] + gv * gv;
            a[i] = av;
            p[i] -= lr * gv / (sqrtf(av) + eps);
        }
        return;
    }
    const seqrec_rows_job& J = pl.r.j[blockIdx.y - pl.nd];
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= J.n) return;
    const int r = J.rows[i];
    if (r < 0 || J.slot[r] != J.base + (int)i) return;
    const long o = (long)r * J.width;
    if (bad) {                                         // discard mode (see above)
        for (int c = lane; c < J.width; c += 64) J.gtab[o + c] = 0.f;
        if (lane == 0) J.slot[r] = INT_MAX;
        return;
    }
    if ((J.width & 3) == 0 && (((uintptr_t)J.gtab | (uintptr_t)J.accum | (uintptr_t)J.table) & 15) == 0) {
        // 16 bytes per lane: a 256-wide row is ONE load per array and one store per array -- the element-wise loop below is
        // four dependent rounds of (3 loads -> 3 stores), the stores of a round fencing the loads of the next (may alias)
        for (int c = lane; c < J.width / 4; c += 64) {
            float4 g = reinterpret_cast<const float4*>(J.gtab + o)[c];
            float4 a = reinterpret_cast<const float4*>(J.accum + o)[c];
            float4 p = reinterpret_cast<const float4*>(J.table + o)[c];
            g.x *= sc; g.y *= sc; g.z *= sc; g.w *= sc;
            a.x += g.x * g.x; a.y += g.y * g.y; a.z += g.z * g.z; a.w += g.w * g.w;
            p.x -= lr * g.x / (sqrtf(a.x) + eps); p.y -= lr * g.y / (sqrtf(a.y) + eps);
            p.z -= lr * g.z / (sqrtf(a.z) + eps); p.w -= lr * g.w / (sqrtf(a.w) + eps);
            reinterpret_cast<float4*>(J.accum + o)[c] = a;
            reinterpret_cast<float4*>(J.table + o)[c] = p;
            reinterpret_cast<float4*>(J.gtab + o)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        for (int c = lane; c < J.width; c += 64) {
            const float g = J.gtab[o + c] * sc;
            const float a = J.accum[o + c] + g * g;
            J.accum[o + c] = a;
            J.table[o + c] -= lr * g / (sqrtf(a) + eps);
            J.gtab[o + c] = 0.f;
        }
    }
    if (lane == 0) J.slot[r] = INT_MAX;
}

// counter RNG (oracle/rng.py)
__global__ void sample_negatives_kernel(uint64_t key, uint64_t step, int K, const uint32_t* __restrict__ thresh,
                                        const int* __restrict__ alias, int V, int* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const uint64_t r = rand64(key, step * (uint64_t)K + (uint64_t)k);
    const uint64_t hi = r >> 32;
    const uint32_t lo = (uint32_t)(r & 0xFFFFFFFFu);
    const int j = (int)((hi * (uint64_t)V) >> 32);
    out[k] = lo < thresh[j] ? j : alias[j];
}
// draw + row gather (+ log-Q gather) of the K shared negatives in one launch: wave k draws negative k
// (every lane computes the same draw) and copies its table row
__global__ void sample_gather_kernel(uint64_t key, uint64_t step, int K, const uint32_t* __restrict__ thresh,
                                     const int* __restrict__ alias, int V, const float* __restrict__ table, int width,
                                     const float* __restrict__ logq, int* __restrict__ neg, float* __restrict__ rows,
                                     float* __restrict__ lq) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= K) return;
    const uint64_t r = rand64(key, step * (uint64_t)K + (uint64_t)k);
    const uint64_t hi = r >> 32;
    const uint32_t lo = (uint32_t)(r & 0xFFFFFFFFu);
    const int j = (int)((hi * (uint64_t)V) >> 32);
    const int id = lo < thresh[j] ? j : alias[j];
    const float* src = table + (long)id * width;
    float* dst = rows + (long)k * width;
    if ((width & 3) == 0) {
        for (int c = lane; c < width / 4; c += 64) reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(src)[c];
    } else {
        for (int c = lane; c < width; c += 64) dst[c] = src[c];
    }
    if (lane == 0) {
        neg[k] = id;
        if (lq) lq[k] = logq[id];
    }
}
__global__ void dropout_mask_kernel(uint64_t key, const long* __restrict__ rowkey, long n_rows, int width, long ld,
                                    uint32_t thr, float inv_keep, float* __restrict__ out) {
    const long total = n_rows * width;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / width;
        const int j = (int)(i % width);
        const uint64_t c = (uint64_t)(rowkey ? rowkey[r] : r) * (uint64_t)width + (uint64_t)j;
        const uint64_t v = rand64(key, c) >> 40;
        out[r * ld + j] = v < (uint64_t)thr ? inv_keep : 0.f;
    }
}

inline int grid_for(long work, int per_block, int cap = 4096) {
    long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace

extern "C" int seqrec_abi_version(void) { return SEQREC_ABI_VERSION; }
extern "C" const char* seqrec_build_arch(void) { return "gfx950"; }

static int gather_rows_impl(const float* table, int64_t table_rows, const int32_t* ids, float* out, int64_t n, int width,
                            const float* row_scale, const float* bias, int accumulate, uint32_t* status, void* stream) {
    if (n < 0 || width <= 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!table || !ids || !out) return SEQREC_E_ARG;
    const bool vec = (width % 4 == 0) && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out) |
                                           reinterpret_cast<uintptr_t>(bias)) & 15) == 0;
    const long total = (long)n * (vec ? width / 4 : width);
    const int blocks = grid_for(total, 256, 8192);
    if (vec) hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(stream), table, ids, out, (long)n, width, row_scale, bias, accumulate, (long)table_rows, status);
    else hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(stream), table, ids, out, (long)n, width, row_scale, bias, accumulate, (long)table_rows, status);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_gather_rows(const float* table, const int32_t* ids, float* out, int64_t n, int width,
                                  const float* row_scale, const float* bias, int accumulate, void* stream) {
    return gather_rows_impl(table, 0, ids, out, n, width, row_scale, bias, accumulate, nullptr, stream);
}
extern "C" int seqrec_gather_rows_bounded(const float* table, int64_t table_rows, const int32_t* ids, float* out, int64_t n,
                                          int width, const float* row_scale, const float* bias, int accumulate,
                                          uint32_t* status, void* stream) {
    if (table_rows <= 0) return SEQREC_E_ARG;
    return gather_rows_impl(table, table_rows, ids, out, n, width, row_scale, bias, accumulate, status, stream);
}

extern "C" int seqrec_full_softmax_ce(float* logits, int64_t ld, const int32_t* tgt, int64_t n, int V,
                                      float inv_denom, float* loss_rows, float* probs, void* stream) {
    if (n < 0 || V <= 0 || ld < V) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!logits || (tgt && !loss_rows) || (!tgt && !probs)) return SEQREC_E_ARG;
    hipLaunchKernelGGL(full_softmax_ce_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), logits,
                       (long)ld, tgt, (long)n, V, inv_denom, loss_rows, probs);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

namespace {
// single-pass variant: the row (K <= 64*KR logits) is held in registers -- one read, one write
template <bool ROWS, int KR>
__device__ __forceinline__ void sampled_softmax_ce_reg_row(float* __restrict__ ln, long ld, const float* __restrict__ hd, int H,
                                                           const float* __restrict__ Eout, const float* __restrict__ bout,
                                                           const float* __restrict__ logq, const float* __restrict__ lq_n,
                                                           const int* __restrict__ tgt, const int* __restrict__ neg, long row, int K,
                                                           float inv_denom, float* __restrict__ loss_rows, float* __restrict__ dlt,
                                                           int vec, int lane, const int* __restrict__ trow = nullptr, long et_ld = 0) {
    // lane owns logits k = 256*i + 4*lane + e (e < 4): one 16-byte access per lane per chunk when the
    // row is 16-byte aligned (vec), else four dword accesses with the same mapping
    constexpr int NC = KR / 4;
    const int t = tgt[row];
    float* x = ln + row * ld;
    float v[KR];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int k0 = 256 * i + 4 * lane;
        float4 q = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if ((vec & 1) && k0 + 3 < K) {
            q = *reinterpret_cast<const float4*>(x + k0);
        } else {
            if (k0 + 0 < K) q.x = x[k0 + 0];
            if (k0 + 1 < K) q.y = x[k0 + 1];
            if (k0 + 2 < K) q.z = x[k0 + 2];
            if (k0 + 3 < K) q.w = x[k0 + 3];
        }
        const float qq[4] = {q.x, q.y, q.z, q.w};
        if ((vec & 2) && k0 + 3 < K) {
            // candidate ids (and their log-Q) as ONE 16-byte load each: the element-wise form costs 64 dword loads with a
            // predicated branch and a wait apiece per row -- the kernel spent its time there, not in the 32 exps
            const int4 ng = *reinterpret_cast<const int4*>(neg + k0);
            float4 lq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lq_n) lq = *reinterpret_cast<const float4*>(lq_n + k0);
            const int ids[4] = {ng.x, ng.y, ng.z, ng.w};
            const float lqs[4] = {lq.x, lq.y, lq.z, lq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float l = qq[e];
                if (ROWS) {
                    l -= lqs[e];
                } else {
                    if (bout) l += bout[ids[e]];
                    if (lq_n) l -= lqs[e];
                    else if (logq) l -= logq[ids[e]];
                }
                if (ids[e] == t) l = -INFINITY;
                v[4 * i + e] = l;
            }
            continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            float l = qq[e];
            if (k < K) {
                const int id = neg[k];
                if (ROWS) {
                    if (lq_n) l -= lq_n[k];
                } else {
                    if (bout) l += bout[id];
                    if (lq_n) l -= lq_n[k];              // per-candidate log-Q gathered once per step by the caller
                    else if (logq) l -= logq[id];
                }
                if (id == t) l = -INFINITY;
            }
            v[4 * i + e] = l;
        }
    }
    const float* h = hd + row * H;
    const float* et = ROWS ? Eout + (trow ? (long)trow[row] : row) * (et_ld ? et_ld : (long)H) : Eout + (long)t * H;
    float d = 0.f;
    for (int j = lane; j < H; j += 64) d += h[j] * et[j];
    float lt = wave_sum(d);
    if (ROWS) {
        if (logq) lt -= logq[row];
    } else {
        if (bout) lt += bout[t];
        if (logq) lt -= logq[t];
    }
    float m = lt;
#pragma unroll
    for (int i = 0; i < KR; ++i) m = fmaxf(m, v[i]);
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < KR; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
    const float etg = expf(lt - m);
    s = wave_sum(s) + etg;
    const float pt = etg / s;
    const float lo = 1e-7f, hi = 1.0f - 1e-7f;
    const float active = (pt >= lo && pt <= hi) ? inv_denom : 0.f;
    if (lane == 0) {
        loss_rows[row] = -logf(fminf(fmaxf(pt, lo), hi));
        dlt[row] = (pt - 1.f) * active;
    }
    const float sc = active / s;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int k0 = 256 * i + 4 * lane;
        if ((vec & 1) && k0 + 3 < K) {
            *reinterpret_cast<float4*>(x + k0) = make_float4(v[4 * i] * sc, v[4 * i + 1] * sc, v[4 * i + 2] * sc, v[4 * i + 3] * sc);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + e < K) x[k0 + e] = v[4 * i + e] * sc;
        }
    }
}
template <bool ROWS, int KR>
__global__ void sampled_softmax_ce_reg_kernel(float* __restrict__ ln, long ld, const float* __restrict__ hd, int H,
                                              const float* __restrict__ Eout, const float* __restrict__ bout,
                                              const float* __restrict__ logq, const float* __restrict__ lq_n,
                                              const int* __restrict__ tgt, const int* __restrict__ neg, long n, int K,
                                              float inv_denom, float* __restrict__ loss_rows, float* __restrict__ dlt,
                                              const int* __restrict__ trow, long et_ld, int vec) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row < n)
        sampled_softmax_ce_reg_row<ROWS, KR>(ln, ld, hd, H, Eout, bout, logq, lq_n, tgt, neg, row, K, inv_denom, loss_rows, dlt,
                                             vec, lane, trow, et_ld);
}

template <bool ROWS>
int launch_sampled(float* ln, long ld, const float* hd, int H, const float* Eout, const float* bout, const float* logq,
                   const float* lq_n, const int* tgt, const int* neg, long n, int K, float inv_denom, float* loss_rows,
                   float* dlt, hipStream_t st, const int* trow = nullptr, long et_ld = 0) {
    // one wave per row; with few rows (an MSNBC-shaped batch has ~2.5 k) single-wave workgroups spread evenly over the
    // 256 CUs (10 per CU) where 4-wave workgroups leave some CUs with 3 and some with 2 (tuning switch: SEQREC_CE_BLOCK)
    static const int ce_block = (int)seqrec_env("SEQREC_CE_BLOCK", 64);
    const int wpb = (ce_block == 256 || n > 16384) ? 4 : 1;
    const dim3 grid((unsigned)((n + wpb - 1) / wpb)), block(64 * wpb);
    // bit 0: 16-byte accesses to the logit rows are legal; bit 1: to the candidate id / log-Q vectors
    const int vec = ((((reinterpret_cast<uintptr_t>(ln) & 15) == 0) && (ld % 4 == 0)) ? 1 : 0) |
                    ((((reinterpret_cast<uintptr_t>(neg) & 15) == 0) && (!lq_n || (reinterpret_cast<uintptr_t>(lq_n) & 15) == 0)) ? 2 : 0);
#define SS_ARGS ln, ld, hd, H, Eout, bout, logq, lq_n, tgt, neg, n, K, inv_denom, loss_rows, dlt, trow, et_ld
#define SS_ARGSV SS_ARGS, vec
    if (K <= 64 * 8) hipLaunchKernelGGL((sampled_softmax_ce_reg_kernel<ROWS, 8>), grid, block, 0, st, SS_ARGSV);
    else if (K <= 64 * 16) hipLaunchKernelGGL((sampled_softmax_ce_reg_kernel<ROWS, 16>), grid, block, 0, st, SS_ARGSV);
    else if (K <= 64 * 32) hipLaunchKernelGGL((sampled_softmax_ce_reg_kernel<ROWS, 32>), grid, block, 0, st, SS_ARGSV);
    else if (K <= 64 * 64) hipLaunchKernelGGL((sampled_softmax_ce_reg_kernel<ROWS, 64>), grid, block, 0, st, SS_ARGSV);
    else hipLaunchKernelGGL(sampled_softmax_ce_kernel<ROWS>, grid, block, 0, st, SS_ARGS);
#undef SS_ARGS
#undef SS_ARGSV
    SEQREC_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int seqrec_sampled_softmax_ce(float* ln, int64_t ld, const float* hd, int H, const float* Eout,
                                         const float* bout, const float* logq, const float* cand_logq,
                                         const int32_t* tgt, const int32_t* neg, int64_t n, int K, float inv_denom,
                                         float* loss_rows, float* dlt, void* stream) {
    if (n < 0 || K < 0 || H <= 0 || ld < K) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!ln || !hd || !Eout || !tgt || (K > 0 && !neg) || !loss_rows || !dlt) return SEQREC_E_ARG;
    return launch_sampled<false>(ln, (long)ld, hd, H, Eout, bout, logq, cand_logq, tgt, neg, (long)n, K, inv_denom, loss_rows,
                                 dlt, as_stream(stream));
}

extern "C" int seqrec_sampled_softmax_ce_rows(float* ln, int64_t ld, const float* hd, int H, const float* Etgt,
                                              const float* lq_tgt, const float* lq_neg, const int32_t* tgt,
                                              const int32_t* neg, int64_t n, int K, float inv_denom,
                                              float* loss_rows, float* dlt, void* stream) {
    if (n < 0 || K < 0 || H <= 0 || ld < K) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!ln || !hd || !Etgt || !tgt || (K > 0 && !neg) || !loss_rows || !dlt) return SEQREC_E_ARG;
    return launch_sampled<true>(ln, (long)ld, hd, H, Etgt, nullptr, lq_tgt, lq_neg, tgt, neg, (long)n, K, inv_denom,
                                loss_rows, dlt, as_stream(stream));
}

// the same with the target rows read THROUGH an index: row i of the target table is table[tgt_row[i] * table_ld + :] (the
// received rows of the exchange buffer, no staging copy)
extern "C" int seqrec_sampled_softmax_ce_rows_idx(float* ln, int64_t ld, const float* hd, int H, const float* table, int64_t table_ld,
                                                  const int32_t* tgt_row, const float* lq_tgt, const float* lq_neg,
                                                  const int32_t* tgt, const int32_t* neg, int64_t n, int K, float inv_denom,
                                                  float* loss_rows, float* dlt, void* stream) {
    if (n < 0 || K < 0 || H <= 0 || ld < K || table_ld < H) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!ln || !hd || !table || !tgt_row || !tgt || (K > 0 && !neg) || !loss_rows || !dlt) return SEQREC_E_ARG;
    return launch_sampled<true>(ln, (long)ld, hd, H, table, nullptr, lq_tgt, lq_neg, tgt, neg, (long)n, K, inv_denom,
                                loss_rows, dlt, as_stream(stream), tgt_row, (long)table_ld);
}

extern "C" int seqrec_reduce_sum(const float* x, int64_t n, float* out, int accumulate, void* stream) {
    if (n < 0 || !out || (n > 0 && !x)) return SEQREC_E_ARG;
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, as_stream(stream), x, (long)n, out, accumulate);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int seqrec_colsum(const float* X, int64_t n, int width, int64_t ld, float* out, int accumulate,
                             float* workspace, void* stream) {
    if (n < 0 || width <= 0 || !out || !workspace) return SEQREC_E_ARG;
    if (n > 0 && !X) return SEQREC_E_ARG;
    constexpr int NP = 32;                       // partial rows (workspace holds up to 64)
    const int np = (int)(n < NP * 4 ? (n + 3) / 4 : NP);
    hipStream_t st = as_stream(stream);
    if (np > 0) {
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((width + 63) / 64, np), dim3(256), 0, st, X, (long)n, width, (long)ld, workspace);
        SEQREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(colsum_final_kernel, dim3((width + 63) / 64), dim3(64), 0, st, workspace, np, width, out, accumulate);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int seqrec_mul(const float* x, const float* m, float* y, int64_t n, void* stream) {
    if (n < 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!x || !m || !y) return SEQREC_E_ARG;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), x, m, y, (long)n);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_fill_f32(float* x, float v, int64_t n, void* stream) {
    if (n < 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!x) return SEQREC_E_ARG;
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), x, v, (long)n);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
// ---------------------------------------------------------------------------------------------
// top-K prediction at catalogue scale ("top-K form at large V"): a running top-64 per row,
// one candidate per lane, merged with one chunk of scores at a time -- no n x V matrix is ever kept
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
// state_val/state_idx [n,64]: the 64 best (score, item) pairs seen so far, unordered (init -inf / -1)
__global__ void topk_merge_kernel(const float* __restrict__ scores, long ld, long n, int width, int col0,
                                  const float* __restrict__ bias, float* __restrict__ state_val, int* __restrict__ state_idx) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float sv = state_val[row * 64 + lane];
    int si = state_idx[row * 64 + lane];
    float m = wave_min(sv);
    const float* x = scores + row * ld;
    for (int c0 = 0; c0 < width; c0 += 64) {
        const int c = c0 + lane;
        float v = c < width ? x[c] + (bias ? bias[col0 + c] : 0.f) : -INFINITY;
        bool pending = v > m;
        while (__any(pending)) {
            const float vmax = wave_max(pending ? v : -INFINITY);
            const int src = __ffsll((unsigned long long)__ballot(pending && v == vmax)) - 1;     // best pending candidate
            const int dst = __ffsll((unsigned long long)__ballot(sv == m)) - 1;                  // slot holding the minimum
            if (lane == dst) { sv = vmax; si = col0 + c0 + src; }
            if (lane == src) pending = false;
            m = wave_min(sv);
            pending = pending && v > m;
        }
    }
    state_val[row * 64 + lane] = sv;
    state_idx[row * 64 + lane] = si;
}
// sorted output: out[row, r] = r-th best of the 64 (ties broken towards the lower item id)
__global__ void topk_finish_kernel(const float* __restrict__ state_val, const int* __restrict__ state_idx, long n, int k,
                                   float* __restrict__ out_val, int* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float v = state_val[row * 64 + lane];
    const int id = state_idx[row * 64 + lane];
    int rank = 0;
    for (int j = 0; j < 64; ++j) {
        const float vj = __shfl(v, j, 64);
        const int ij = __shfl(id, j, 64);
        rank += (vj > v || (vj == v && (unsigned)ij < (unsigned)id)) ? 1 : 0;
    }
    if (rank < k) {
        out_val[row * k + rank] = v;
        out_idx[row * k + rank] = id;
    }
}
}  // namespace
extern "C" int seqrec_topk_merge(const float* scores, int64_t ld, int64_t n, int width, int col0, const float* bias,
                                 float* state_val, int32_t* state_idx, void* stream) {
    if (n < 0 || width < 0 || ld < width || col0 < 0) return SEQREC_E_ARG;
    if (n == 0 || width == 0) return 0;
    if (!scores || !state_val || !state_idx) return SEQREC_E_ARG;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), scores, (long)ld, (long)n,
                       width, col0, bias, state_val, state_idx);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_topk_finish(const float* state_val, const int32_t* state_idx, int64_t n, int k, float* out_val,
                                  int32_t* out_idx, void* stream) {
    if (n < 0 || k < 1 || k > 64) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!state_val || !state_idx || !out_val || !out_idx) return SEQREC_E_ARG;
    hipLaunchKernelGGL(topk_finish_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), state_val, state_idx,
                       (long)n, k, out_val, out_idx);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Gaussian-prior / L2 kernel regularizer (model.py:71-91): R = strength * sum (w - mean)^2
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void prior_grad_kernel(const float* __restrict__ w, const float* __restrict__ means, long n, float strength,
                                  float* __restrict__ grad, float* __restrict__ loss) {
    __shared__ float part[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = w[i] - (means ? means[i] : 0.f);
        s += d * d;
        if (grad) grad[i] += 2.f * strength * d;
    }
    if (!loss) return;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, strength * (part[0] + part[1] + part[2] + part[3]));
}
}  // namespace
extern "C" int seqrec_prior_grad(const float* w, const float* means, int64_t n, float strength, float* grad,
                                 float* loss_accum, void* stream) {
    if (n < 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!w || (!grad && !loss_accum)) return SEQREC_E_ARG;
    hipLaunchKernelGGL(prior_grad_kernel, dim3(grid_for(n, 1024, 1024)), dim3(256), 0, as_stream(stream), w, means, (long)n,
                       strength, grad, loss_accum);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// device-side ragged batcher: the dataset stays in HBM as one flat id array
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void pack_batch_kernel(const int* __restrict__ flat, const long* __restrict__ starts, const int* __restrict__ sess,
                                  const int* __restrict__ step_off, int* __restrict__ ids, int* __restrict__ tgt,
                                  int* __restrict__ prev) {
    const int t = blockIdx.y;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int p0 = step_off[t];
    if (r >= step_off[t + 1] - p0) return;
    const long base = starts[sess[r]] + t;
    const int p = p0 + r;
    ids[p] = flat[base];
    tgt[p] = flat[base + 1];
    prev[p] = t > 0 ? step_off[t - 1] + r : -1;
}
// the same with the batch's step offsets and session indices handed over IN THE KERNEL ARGUMENTS (<= 3840 bytes): no
// host -> device copy in front of the launch (a ~5 us blit on the stream).  The launch also leaves both arrays in HBM
// for later readers (history features, evaluation).
struct PackHost { int v[SEQREC_PACK_HOST_MAX]; };
__global__ void pack_batch_host_kernel(const int* __restrict__ flat, const long* __restrict__ starts, const PackHost h, int B,
                                       int T, int* __restrict__ sess_out, int* __restrict__ step_off_out,
                                       int* __restrict__ ids, int* __restrict__ tgt, int* __restrict__ prev) {
    const int t = blockIdx.y;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        if (r < B) sess_out[r] = h.v[T + 1 + r];
        if (r <= T) step_off_out[r] = h.v[r];
    }
    const int p0 = h.v[t];
    if (r >= h.v[t + 1] - p0) return;
    const long base = starts[h.v[T + 1 + r]] + t;
    const int p = p0 + r;
    ids[p] = flat[base];
    tgt[p] = flat[base + 1];
    prev[p] = t > 0 ? h.v[t - 1] + r : -1;
}
// xs[p, v] = 1 (or the count, freq != 0) for every item v among the session's items 0..t  (datasets.py:97-113)
__global__ void history_features_kernel(const int* __restrict__ flat, const long* __restrict__ starts,
                                        const int* __restrict__ sess, const int* __restrict__ step_off, int x_dim, long ld,
                                        int freq, float* __restrict__ xs) {
    const int t = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int p0 = step_off[t];
    if (r >= step_off[t + 1] - p0) return;
    float* row = xs + (long)(p0 + r) * ld;
    for (int c = lane; c < (int)ld; c += 64) row[c] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const long base = starts[sess[r]];
    for (int j = lane; j <= t; j += 64) {
        const int v = flat[base + j];
        if (v >= 0 && v < x_dim) {
            if (freq) atomicAdd(row + v, 1.f);
            else row[v] = 1.f;
        }
    }
}
}  // namespace
extern "C" int seqrec_pack_batch(const int32_t* flat, const int64_t* starts, const int32_t* sess, const int32_t* step_off,
                                 int B, int T, int32_t* ids, int32_t* tgt, int32_t* prev, void* stream) {
    if (B < 0 || T < 0) return SEQREC_E_ARG;
    if (B == 0 || T == 0) return 0;
    if (!flat || !starts || !sess || !step_off || !ids || !tgt || !prev) return SEQREC_E_ARG;
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)((B + 255) / 256), (unsigned)T), dim3(256), 0, as_stream(stream), flat,
                       reinterpret_cast<const long*>(starts), sess, step_off, ids, tgt, prev);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_pack_batch_host(const int32_t* flat, const int64_t* starts, const int32_t* sess_host,
                                      const int32_t* step_off_host, int B, int T, int32_t* sess_out, int32_t* step_off_out,
                                      int32_t* ids, int32_t* tgt, int32_t* prev, void* stream) {
    if (B < 0 || T < 0) return SEQREC_E_ARG;
    if ((long)B + T + 1 > SEQREC_PACK_HOST_MAX) return SEQREC_E_SHAPE;
    if (!sess_host || !step_off_host || !sess_out || !step_off_out) return SEQREC_E_ARG;
    if (B == 0 || T == 0) return 0;
    if (!flat || !starts || !ids || !tgt || !prev) return SEQREC_E_ARG;
    if (step_off_host[T] < 0 || step_off_host[0] != 0) return SEQREC_E_ARG;
    PackHost h;
    for (int i = 0; i <= T; ++i) h.v[i] = step_off_host[i];
    for (int i = 0; i < B; ++i) h.v[T + 1 + i] = sess_host[i];
    const unsigned gx = (unsigned)((std::max(B, T + 1) + 255) / 256);
    hipLaunchKernelGGL(pack_batch_host_kernel, dim3(gx, (unsigned)T), dim3(256), 0, as_stream(stream), flat,
                       reinterpret_cast<const long*>(starts), h, B, T, sess_out, step_off_out, ids, tgt, prev);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_history_features(const int32_t* flat, const int64_t* starts, const int32_t* sess,
                                       const int32_t* step_off, int B, int T, int x_dim, int64_t ld, int freq, float* xs,
                                       void* stream) {
    if (B < 0 || T < 0 || x_dim <= 0 || ld < x_dim) return SEQREC_E_ARG;
    if (B == 0 || T == 0) return 0;
    if (!flat || !starts || !sess || !step_off || !xs) return SEQREC_E_ARG;
    hipLaunchKernelGGL(history_features_kernel, dim3((unsigned)((B + 3) / 4), (unsigned)T), dim3(256), 0, as_stream(stream), flat,
                       reinterpret_cast<const long*>(starts), sess, step_off, x_dim, (long)ld, freq, xs);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

namespace {
__global__ void index_affine_i32_kernel(int* __restrict__ dst, const int* __restrict__ dpos, const int* __restrict__ src,
                                        const int* __restrict__ spos, long n, int mul, int add) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long d = dpos ? dpos[i] : i, q = spos ? spos[i] : i;
        if (d >= 0 && q >= 0) dst[d] = src[q] * mul + add;
    }
}
}  // namespace
extern "C" int seqrec_index_affine_i32(int32_t* dst, const int32_t* dst_pos, const int32_t* src, const int32_t* src_pos,
                                       int64_t n, int32_t mul, int32_t add, void* stream) {
    if (n < 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!dst || !src) return SEQREC_E_ARG;
    hipLaunchKernelGGL(index_affine_i32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), dst, dst_pos, src,
                       src_pos, (long)n, mul, add);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_fill_i32(int32_t* x, int32_t v, int64_t n, void* stream) {
    if (n < 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!x) return SEQREC_E_ARG;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), x, v, (long)n);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int seqrec_rows_scatter_add(float* gtab, int32_t* slot, const int32_t* rows, const float* vals,
                                       int64_t ldv, const float* row_scale, int64_t n, int width, int32_t base,
                                       void* stream) {
    if (n < 0 || width <= 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!gtab || !slot || !rows || !vals) return SEQREC_E_ARG;
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), gtab, slot,
                       rows, vals, (long)ldv, row_scale, (long)n, width, base);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_rows_sqnorm(const float* gtab, const int32_t* slot, const int32_t* rows, int64_t n,
                                  int width, int32_t base, float* sq_accum, void* stream) {
    if (n < 0 || width <= 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!gtab || !slot || !rows || !sq_accum) return SEQREC_E_ARG;
    hipLaunchKernelGGL(rows_sqnorm_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), gtab, slot, rows,
                       (long)n, width, base, sq_accum);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_rows_adagrad(float* table, float* accum, float* gtab, int32_t* slot, const int32_t* rows,
                                   int64_t n, int width, int32_t base, float lr, float eps, const float* scale,
                                   void* stream) {
    if (n < 0 || width <= 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!table || !accum || !gtab || !slot || !rows || !scale) return SEQREC_E_ARG;
    hipLaunchKernelGGL(rows_adagrad_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), table, accum,
                       gtab, slot, rows, (long)n, width, base, lr, eps, scale);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int seqrec_sqnorm(const float* g, int64_t n, float* sq_accum, void* stream) {
    if (n < 0 || !sq_accum) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!g) return SEQREC_E_ARG;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid_for(n, 1024, 1024)), dim3(256), 0, as_stream(stream), g, (long)n, sq_accum);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_clip_scale(const float* sq_accum, float clipnorm, float* scale, void* stream) {
    if (!sq_accum || !scale) return SEQREC_E_ARG;
    hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, as_stream(stream), sq_accum, clipnorm, scale);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_adagrad_dense(float* p, float* a, const float* g, int64_t n, float lr, float eps,
                                    const float* scale, void* stream) {
    if (n < 0) return SEQREC_E_ARG;
    if (n == 0) return 0;
    if (!p || !a || !g || !scale) return SEQREC_E_ARG;
    hipLaunchKernelGGL(adagrad_dense_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), p, a, g, (long)n, lr, eps, scale);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int seqrec_sample_negatives(uint64_t seed, uint64_t step, int K, const uint32_t* thresh,
                                       const int32_t* alias, int V, int32_t* out, void* stream) {
    if (K < 0 || V <= 0) return SEQREC_E_ARG;
    if (K == 0) return 0;
    if (!thresh || !alias || !out) return SEQREC_E_ARG;
    hipLaunchKernelGGL(sample_negatives_kernel, dim3((K + 255) / 256), dim3(256), 0, as_stream(stream), key64(seed, 1), step, K,
                       thresh, alias, V, out);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_sample_gather(uint64_t seed, uint64_t step, int K, const uint32_t* thresh, const int32_t* alias,
                                    int V, const float* table, int width, const float* logq, int32_t* neg_out,
                                    float* rows_out, float* logq_out, void* stream) {
    if (K < 0 || V <= 0 || width <= 0) return SEQREC_E_ARG;
    if (K == 0) return 0;
    if (!thresh || !alias || !table || !neg_out || !rows_out || (logq_out && !logq)) return SEQREC_E_ARG;
    if ((width & 3) == 0 && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(rows_out)) & 15)) return SEQREC_E_ARG;
    hipLaunchKernelGGL(sample_gather_kernel, dim3((K + 3) / 4), dim3(256), 0, as_stream(stream), key64(seed, 1), step, K, thresh,
                       alias, V, table, width, logq, neg_out, rows_out, logq_out);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_dropout_mask(uint64_t seed, uint64_t stream_id, const int64_t* rowkey, int64_t n_rows,
                                   int width, int64_t ld, double rate, float* out, void* stream) {
    if (n_rows < 0 || width <= 0 || ld < width || rate < 0.0 || rate >= 1.0) return SEQREC_E_ARG;
    if (n_rows == 0) return 0;
    if (!out) return SEQREC_E_ARG;
    const double keep = 1.0 - rate;
    const uint32_t thr = (uint32_t)llrint(keep * 16777216.0);
    const float inv_keep = 1.0f / (float)keep;      // oracle: float32(1) / float32(keep)
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n_rows * width, 256)), dim3(256), 0, as_stream(stream),
                       key64(seed, stream_id), reinterpret_cast<const long*>(rowkey), (long)n_rows, width, (long)ld, thr,
                       inv_keep, out);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int seqrec_sqnorm_multi(int count, const float* const* g, const int64_t* n, float* sq_accum, void* stream) {
    if (count < 0 || count > 8 || !sq_accum || (count > 0 && (!g || !n))) return SEQREC_E_ARG;
    if (count == 0) return 0;
    DenseMulti m = {};
    for (int i = 0; i < count; ++i) { if (n[i] < 0 || (n[i] > 0 && !g[i])) return SEQREC_E_ARG; m.g[i] = g[i]; m.n[i] = n[i]; }
    hipLaunchKernelGGL(sqnorm_multi_kernel, dim3(32, count), dim3(256), 0, as_stream(stream), m, sq_accum);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_adagrad_dense_multi(int count, float* const* p, float* const* a, const float* const* g,
                                          const int64_t* n, float lr, float eps, const float* scale, void* stream) {
    if (count < 0 || count > 8 || !scale || (count > 0 && (!p || !a || !g || !n))) return SEQREC_E_ARG;
    if (count == 0) return 0;
    DenseMulti m = {};
    for (int i = 0; i < count; ++i) {
        if (n[i] < 0 || (n[i] > 0 && (!p[i] || !a[i] || !g[i]))) return SEQREC_E_ARG;
        m.p[i] = p[i]; m.a[i] = a[i]; m.g[i] = g[i]; m.n[i] = n[i];
    }
    hipLaunchKernelGGL(adagrad_multi_kernel, dim3(256, count), dim3(256), 0, as_stream(stream), m, lr, eps, scale);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
static int fill_rows_multi(const seqrec_rows_job* jobs, int count, RowsMulti& m, long& maxn) {
    if (count < 0 || count > 4 || (count > 0 && !jobs)) return SEQREC_E_ARG;
    maxn = 0;
    for (int i = 0; i < count; ++i) {
        const seqrec_rows_job& j = jobs[i];
        if (j.n < 0 || j.width <= 0) return SEQREC_E_ARG;
        if (j.n > 0 && (!j.gtab || !j.slot || !j.rows)) return SEQREC_E_ARG;
        m.j[i] = j;
        if (j.n > maxn) maxn = j.n;
    }
    return 0;
}
extern "C" int seqrec_rows_scatter_add_multi(const seqrec_rows_job* jobs, int count, void* stream) {
    RowsMulti m = {};
    long maxn;
    int rc = fill_rows_multi(jobs, count, m, maxn);
    if (rc || maxn == 0) return rc;
    for (int i = 0; i < count; ++i) if (jobs[i].n > 0 && !jobs[i].vals) return SEQREC_E_ARG;
    int w0 = jobs[0].width;
    for (int i = 0; i < count; ++i) if (jobs[i].n > 0 && (jobs[i].width != w0 || jobs[i].ldv < w0)) w0 = 0;
    static const bool combine = seqrec_env("SEQREC_SCATTER_COMBINE", 1) != 0;
    if (combine && (w0 == 64 || w0 == 128 || (w0 > 0 && w0 % 256 == 0 && w0 <= 2048))) {
        const dim3 grid((unsigned)((maxn + 47) / 48), count, w0 > 256 ? w0 / 256 : 1);
        if (w0 == 64) hipLaunchKernelGGL(rows_scatter_combine_kernel<1>, grid, dim3(768), 0, as_stream(stream), m);
        else if (w0 == 128) hipLaunchKernelGGL(rows_scatter_combine_kernel<2>, grid, dim3(768), 0, as_stream(stream), m);
        else hipLaunchKernelGGL(rows_scatter_combine_kernel<4>, grid, dim3(768), 0, as_stream(stream), m);
        SEQREC_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(rows_scatter_add_multi_kernel, dim3((unsigned)((maxn + 15) / 16), count), dim3(256), 0, as_stream(stream), m);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_rows_sqnorm_multi(const seqrec_rows_job* jobs, int count, float* sq_accum, void* stream) {
    RowsMulti m = {};
    long maxn;
    int rc = fill_rows_multi(jobs, count, m, maxn);
    if (rc || maxn == 0) return rc;
    if (!sq_accum) return SEQREC_E_ARG;
    hipLaunchKernelGGL(rows_sqnorm_multi_kernel, dim3((unsigned)((maxn + 63) / 64), count), dim3(1024), 0, as_stream(stream), m, sq_accum);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_rows_adagrad_multi(const seqrec_rows_job* jobs, int count, float lr, float eps, const float* scale,
                                         void* stream) {
    RowsMulti m = {};
    long maxn;
    int rc = fill_rows_multi(jobs, count, m, maxn);
    if (rc || maxn == 0) return rc;
    if (!scale) return SEQREC_E_ARG;
    for (int i = 0; i < count; ++i) if (jobs[i].n > 0 && (!jobs[i].table || !jobs[i].accum)) return SEQREC_E_ARG;
    hipLaunchKernelGGL(rows_adagrad_multi_kernel, dim3((unsigned)((maxn + 3) / 4), count), dim3(256), 0, as_stream(stream), m, lr, eps, scale);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

namespace {
int fill_opt_plan(int n_dense, float* const* params, float* const* accums, const float* const* grads, const int64_t* numel,
                  const seqrec_rows_job* jobs, int n_jobs, bool need_state, OptPlan& pl, long& maxn) {
    if (n_dense < 0 || n_dense > 8 || n_jobs < 0 || n_jobs > 4 || n_dense + n_jobs == 0) return SEQREC_E_ARG;
    if (n_dense && (!grads || !numel || (need_state && (!params || !accums)))) return SEQREC_E_ARG;
    pl = OptPlan{};
    pl.nd = n_dense; pl.nr = n_jobs;
    for (int i = 0; i < n_dense; ++i) {
        if (numel[i] < 0 || (numel[i] > 0 && (!grads[i] || (need_state && (!params[i] || !accums[i]))))) return SEQREC_E_ARG;
        pl.d.g[i] = grads[i]; pl.d.n[i] = (long)numel[i];
        pl.d.p[i] = need_state ? params[i] : nullptr; pl.d.a[i] = need_state ? accums[i] : nullptr;
    }
    maxn = 0;
    if (n_jobs) {
        if (!jobs) return SEQREC_E_ARG;
        RowsMulti m = {};
        const int rc = fill_rows_multi(jobs, n_jobs, m, maxn);
        if (rc) return rc;
        pl.r = m;
    }
    return 0;
}
}  // namespace

extern "C" int seqrec_loss_reduce(const float* loss_rows, int64_t n, float* loss_out, void* stream) {
    if (n <= 0 || !loss_rows || !loss_out) return SEQREC_E_ARG;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, as_stream(stream), loss_rows, (long)n, loss_out);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_opt_sqnorm_slabs(int n_dense, const float* const* grads, const int64_t* numel,
                                       int n_products, const seqrec_gemm_desc* products, int n_slabs, const float* workspace,
                                       const seqrec_rows_job* jobs_host, int n_jobs, float* sq_accum,
                                       const float* loss_rows, int64_t n_loss, float* loss_out, void* stream) {
    if (n_products < 0 || n_products > 4 || n_dense < 0 || n_dense + n_products > 8) return SEQREC_E_ARG;
    if (n_products && (!products || !workspace || n_slabs < 1)) return SEQREC_E_ARG;
    OptPlan pl;
    long maxn = 0;
    if (n_dense + n_jobs > 0) {
        const int rc = fill_opt_plan(n_dense, nullptr, nullptr, grads, numel, jobs_host, n_jobs, false, pl, maxn);
        if (rc) return rc;
    } else {
        if (!n_products) return SEQREC_E_ARG;
        pl = OptPlan{};
    }
    if (!sq_accum || (loss_out && (!loss_rows || n_loss <= 0))) return SEQREC_E_ARG;
    SlabPieces sp = {};
    sp.np = n_products; sp.n_slabs = n_slabs;
    long off = 0;
    for (int i = 0; i < n_products; ++i) {
        const seqrec_gemm_desc& d = products[i];
        if (d.M <= 0 || d.N <= 0 || !d.C || d.ldc < d.N) return SEQREC_E_ARG;
        if (d.bias || d.accumulate) return SEQREC_E_UNSUPPORTED;
        sp.ws[i] = workspace + off; sp.C[i] = d.C; sp.M[i] = d.M; sp.N[i] = d.N; sp.ldc[i] = d.ldc;
        off += (long)n_slabs * d.M * d.N;
    }
    pl.loss_rows = loss_rows; pl.n_loss = (long)n_loss; pl.loss_out = loss_out;
    const unsigned gx = (unsigned)std::max<long>(8, (maxn + 63) / 64);
    hipLaunchKernelGGL(opt_sqnorm_kernel, dim3(gx, n_dense + n_products + n_jobs + (loss_out ? 1 : 0)), dim3(1024), 0,
                       as_stream(stream), pl, sp, sq_accum);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_opt_sqnorm(int n_dense, const float* const* grads, const int64_t* numel,
                                 const seqrec_rows_job* jobs_host, int n_jobs, float* sq_accum,
                                 const float* loss_rows, int64_t n_loss, float* loss_out, void* stream) {
    return seqrec_opt_sqnorm_slabs(n_dense, grads, numel, 0, nullptr, 0, nullptr, jobs_host, n_jobs, sq_accum, loss_rows, n_loss,
                                   loss_out, stream);
}
extern "C" int64_t seqrec_opt_sqnorm_ordered_floats(int n_dense, int n_jobs, int64_t max_job_rows) {
    const long gx = std::max<long>(32, (max_job_rows + 15) / 16);
    return (int64_t)gx * (n_dense + n_jobs);
}
extern "C" int seqrec_opt_sqnorm_ordered(int n_dense, const float* const* grads, const int64_t* numel,
                                         const seqrec_rows_job* jobs_host, int n_jobs, float* partials,
                                         int64_t partials_floats, float* sq_out, int accumulate,
                                         const float* loss_rows, int64_t n_loss, float* loss_out, void* stream) {
    OptPlan pl;
    long maxn;
    const int rc = fill_opt_plan(n_dense, nullptr, nullptr, grads, numel, jobs_host, n_jobs, false, pl, maxn);
    if (rc) return rc;
    if (!sq_out || !partials || (loss_out && (!loss_rows || n_loss <= 0))) return SEQREC_E_ARG;
    pl.loss_rows = loss_rows; pl.n_loss = (long)n_loss; pl.loss_out = loss_out;
    const unsigned gx = (unsigned)std::max<long>(32, (maxn + 15) / 16);
    const long np = (long)gx * (n_dense + n_jobs);
    if (np > partials_floats) return SEQREC_E_ARG;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(opt_sqnorm_partial_kernel, dim3(gx, n_dense + n_jobs + (loss_out ? 1 : 0)), dim3(256), 0, st, pl, partials);
    SEQREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, st, partials, np, sq_out, accumulate);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
extern "C" int seqrec_opt_apply(int n_dense, float* const* params, float* const* accums, const float* const* grads,
                                const int64_t* numel, const seqrec_rows_job* jobs_host, int n_jobs, const float* sq,
                                float clipnorm, float lr, float eps, float* scale_out, float* zero_next,
                                const float* grad_div, uint32_t* status, const float* sq_extra, void* stream) {
    OptPlan pl;
    long maxn;
    const int rc = fill_opt_plan(n_dense, params, accums, grads, numel, jobs_host, n_jobs, true, pl, maxn);
    if (rc) return rc;
    if (!sq || !scale_out) return SEQREC_E_ARG;
    const unsigned gx = (unsigned)std::max<long>(n_dense ? 256 : 1, (maxn + 3) / 4);
    hipLaunchKernelGGL(opt_apply_kernel, dim3(gx, n_dense + n_jobs), dim3(256), 0, as_stream(stream), pl, sq, clipnorm, lr, eps,
                       scale_out, zero_next, grad_div, status, sq_extra);
    SEQREC_LAUNCH_CHECK();
    return 0;
}

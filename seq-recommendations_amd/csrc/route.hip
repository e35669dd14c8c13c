// Host-side routing arithmetic of the row exchange (distributed.py).  No device code: these entry points run on
// the CPU and exist because the same arithmetic in numpy (RowExchange.plan_seg_many + ShardedEngine._prepare_unified:
// ~60 small array operations per batch) cost 0.7 ms of host time per batch -- more than the 0.55 ms the GPU needs for the
// step it plans.  One pass over the batch's 2n requests each.
//
// Requests of a batch: q < n asks for input row ids[q], q >= n for target row tgt[q - n]; item v lives on rank v % R at
// local row v / R of E, or -- untied tables -- at local row (rows of E on that rank) + v / R of the unified table.
// Requester-side buffer: peer j's segment = [my requests to j in request order | `extra` rows chosen by j];
// owner-side buffer:     peer i's segment = [i's requests to me             | `extra` rows I choose for i].
#include "common.h"
#include <vector>

extern "C" int seqrec_route_count_host(const int32_t* ids, const int32_t* tgt, int64_t n, int R, int64_t* counts) {
    if (n < 0 || R < 1 || !counts || (n > 0 && (!ids || !tgt))) return SEQREC_E_ARG;
    for (int j = 0; j < R; ++j) counts[j] = 0;
    for (int64_t q = 0; q < n; ++q) {
        if (ids[q] < 0 || tgt[q] < 0) return SEQREC_E_ARG;
        ++counts[ids[q] % R];
        ++counts[tgt[q] % R];
    }
    return 0;
}

// send[base[j] + t] = the local row of my t-th request to peer j (request order); req_rank[q] = position of request q in the
// owner-sorted order (stable).  base: absolute start of this batch's segment for peer j inside the send list.
extern "C" int seqrec_route_fill_host(const int32_t* ids, const int32_t* tgt, int64_t n, int R, int64_t V_in, int tied,
                                      const int64_t* counts, const int64_t* base, int32_t* send, int32_t* req_rank) {
    if (n < 0 || R < 1 || !counts || !base || (n > 0 && (!ids || !tgt || !send || !req_rank))) return SEQREC_E_ARG;
    std::vector<int64_t> next(R), start(R);
    int64_t acc = 0;
    for (int j = 0; j < R; ++j) { start[j] = acc; acc += counts[j]; next[j] = 0; }
    for (int64_t q = 0; q < 2 * n; ++q) {
        const int64_t v = q < n ? ids[q] : tgt[q - n];
        const int j = (int)(v % R);
        int64_t want = v / R;
        if (q >= n && !tied) want += (V_in - j + R - 1) / R;           // E rows held by the owner come first in its unified table
        const int64_t t = next[j]++;
        send[base[j] + t] = (int32_t)want;
        req_rank[q] = (int32_t)(start[j] + t);
    }
    return 0;
}

// The batch's int32 blob (ShardedEngine.prepare: ONE host -> device copy per batch), in this order:
//   step_off[T+1] prev[n] ids[n] tgt[n] neg_slots[R Kr] id_rows[R nid] take_in[n] take_tgt[n] neg_rows[R Kr] negid_idx[R Kr]
//   back_idx[n_tot] own_src[m_tot] ntok[1] (lq_tgt[n])
// sc / rc: my requests per peer / the peers' requests to me; extra = Kr + nid rows per peer pair; got_off[i]: where peer i's
// requests of THIS batch start in the received list.
extern "C" int64_t seqrec_route_blob_host(const int32_t* step_off, int T, const int32_t* prev, const int32_t* ids, const int32_t* tgt,
                                          int64_t n, int R, int Kr, int nid, int w, const int64_t* sc, const int64_t* rc,
                                          const int32_t* req_rank, const int64_t* got_off, float ntok,
                                          const float* lq_tgt, int32_t* blob, int64_t blob_len) {
    if (n < 0 || T < 0 || R < 1 || Kr < 0 || nid < 0 || w < 1 || !sc || !rc || !got_off || !blob) return SEQREC_E_ARG;
    const int64_t extra = (int64_t)Kr + nid;
    int64_t m = 0;
    for (int i = 0; i < R; ++i) m += rc[i];
    const int64_t n_tot = 2 * n + R * extra, m_tot = m + R * extra;
    const int64_t need = (T + 1) + 5 * n + 3 * (int64_t)R * Kr + (int64_t)R * nid + n_tot + m_tot + 1 + (lq_tgt ? n : 0);
    if (blob_len < need) return SEQREC_E_ARG;
    int32_t* p = blob;
    for (int t = 0; t <= T; ++t) *p++ = step_off[t];
    for (int64_t q = 0; q < n; ++q) *p++ = prev[q];
    for (int64_t q = 0; q < n; ++q) *p++ = ids[q];
    for (int64_t q = 0; q < n; ++q) *p++ = tgt[q];
    // owner side: rows of my draws / of their id rows -- own_extra[i][e] = rc_end[i] + extra * i + e
    std::vector<int64_t> rc_end(R), sc_end(R);
    { int64_t a = 0, b = 0; for (int i = 0; i < R; ++i) { a += rc[i]; rc_end[i] = a; b += sc[i]; sc_end[i] = b; } }
    for (int i = 0; i < R; ++i) for (int e = 0; e < Kr; ++e) *p++ = (int32_t)(rc_end[i] + extra * i + e);            // neg_slots
    for (int i = 0; i < R; ++i) for (int e = 0; e < nid; ++e) *p++ = (int32_t)(rc_end[i] + extra * i + Kr + e);      // id_rows
    // requester side: position of request q = its sorted rank + extra * (owner of q)
    int32_t* take = p;
    for (int64_t q = 0; q < 2 * n; ++q) {
        const int64_t v = q < n ? ids[q] : tgt[q - n];
        *p++ = (int32_t)(req_rank[q] + extra * (v % R));
    }
    for (int j = 0; j < R; ++j) for (int e = 0; e < Kr; ++e) *p++ = (int32_t)(sc_end[j] + extra * j + e);            // neg_rows
    for (int j = 0; j < R; ++j) for (int q = 0; q < Kr; ++q) *p++ = (int32_t)((sc_end[j] + extra * j + Kr + q / w) * w + q % w);   // negid_idx
    int32_t* back = p;                                                                                               // back_idx
    for (int64_t i = 0; i < n_tot; ++i) back[i] = -1;
    for (int64_t q = 0; q < 2 * n; ++q) back[take[q]] = (int32_t)q;
    for (int j = 0; j < R; ++j) for (int e = 0; e < Kr; ++e) back[sc_end[j] + extra * j + e] = (int32_t)(2 * n + (int64_t)j * Kr + e);
    p += n_tot;
    // owner-side rows: requests of peer i as an index got_off[i] + t into the received request list, the extras as their kind
    // (seqrec_exchange_pack's `kinds` with the received list as `got`: -2 at the rows of my draws, -1 at their id rows)
    int32_t* own = p;
    { int64_t pos = 0;
      for (int i = 0; i < R; ++i) {
          for (int64_t t = 0; t < rc[i]; ++t) own[pos + t] = (int32_t)(got_off[i] + t);
          pos += rc[i];
          for (int e = 0; e < Kr; ++e) own[pos + e] = -2;
          for (int e = 0; e < nid; ++e) own[pos + Kr + e] = -1;
          pos += extra;
      } }
    p += m_tot;
    { int32_t bits; __builtin_memcpy(&bits, &ntok, 4); *p++ = bits; }
    if (lq_tgt) { __builtin_memcpy(p, lq_tgt, (size_t)n * 4); p += n; }
    return (int64_t)(p - blob);
}

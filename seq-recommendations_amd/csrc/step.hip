// seqrec_train_cell: the launch sequence of a sampled-softmax training step's cell issued from ONE host call
// (include/seqrec_hip.h, "the cell of a sampled-softmax training step").  No kernels of its own: every launch is an entry
// point of this library called with the arguments engine.Engine.train_step / distributed.ShardedEngine._cell_unified pass
// call by call -- the Python sequence is the specification, this is its host-side fast path (bit-identical results).
#include "common.h"

#define STEP_TRY(expr)            \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != 0) return rc__; \
    } while (0)

extern "C" int64_t seqrec_cell_plan_bytes(void) { return (int64_t)sizeof(seqrec_cell_plan); }

extern "C" int seqrec_train_cell(seqrec_cell_plan* p, void* st) {
    if (!p || p->n <= 0 || p->Hp <= 0 || p->G <= 0 || p->K <= 0 || p->Dp <= 0 || !p->step_off_host) return SEQREC_E_ARG;
    if (p->batch && !(p->pack_u && p->sample && (p->stages & 1))) return SEQREC_E_ARG;
    const int64_t n = p->n;
    const int Hp = p->Hp, GHp = p->G * p->Hp, K = p->K, Dp = p->Dp;
    if (p->stages & 1) {
        if (p->pack_u) {
            if (p->sample && p->batch)
                STEP_TRY(seqrec_rnn_pack_u_sample_batch(p->cell, Hp, p->U, p->upack, p->seed, p->step, K, p->thresh, p->alias, p->V,
                                                        p->sample_table, Hp, p->sample_logq, p->neg_out, p->Eneg_out, p->lq_neg_out,
                                                        p->flat, p->starts, p->sess_host, p->step_off_host, p->B, p->T, p->sess_out,
                                                        p->step_off_out, p->ids_out, p->tgt_out, p->prev_out, st));
            else if (p->sample)
                STEP_TRY(seqrec_rnn_pack_u_sample(p->cell, Hp, p->U, p->upack, p->seed, p->step, K, p->thresh, p->alias, p->V,
                                                  p->sample_table, Hp, p->sample_logq, p->neg_out, p->Eneg_out, p->lq_neg_out, st));
            else
                STEP_TRY(seqrec_rnn_pack_u_stepwise(p->cell, Hp, p->U, p->upack, st));
        }
        seqrec_gemm_fuse fx = {};
        fx.a_index = p->x_index;
        STEP_TRY(seqrec_gemm_f32_fused(1, 0, n, GHp, Dp, p->x_table, p->x_ld, p->W, GHp, p->XW, GHp, p->bias, 0, 1, nullptr, &fx, st));
        STEP_TRY(seqrec_rnn_fwd_stepwise(p->cell, p->act, Hp, p->H_real, p->T, p->B, nullptr, p->step_off_host, p->XW, p->Hout, p->gates,
                                         p->aux, p->upack, nullptr, p->use_graph, st));
        STEP_TRY(seqrec_gemm_f32(1, 1, n, K, Hp, p->Hout, Hp, p->Eneg, Hp, p->ln, K, nullptr, 0, 1, nullptr, st));
        if (p->lq_tgt || p->tgt_index != p->tgt_ids)
            STEP_TRY(seqrec_sampled_softmax_ce_rows_idx(p->ln, K, p->Hout, Hp, p->tgt_table, p->tgt_ld, p->tgt_index, p->lq_tgt, p->lq_neg,
                                                        p->tgt_ids, p->neg, n, K, p->inv_denom, p->loss_rows, p->dlt, st));
        else
            STEP_TRY(seqrec_sampled_softmax_ce(p->ln, K, p->Hout, Hp, p->tgt_table, nullptr, p->logq_table, p->lq_neg, p->tgt_ids, p->neg, n, K,
                                               p->inv_denom, p->loss_rows, p->dlt, st));
    }
    if (p->stages & 2) {
        p->ns_deneg = 0;
        if (p->deneg_mode == 3) {          // dH and dEneg -- both products of dlogits -- in ONE launch (seqrec_gemm_f32_pair)
            seqrec_gemm_pair q = {};
            q.a_kc0 = 1; q.b_kc0 = 0; q.M0 = n; q.N0 = Hp; q.K0 = K; q.A0 = p->ln; q.lda0 = K; q.B0 = p->Eneg; q.ldb0 = Hp;
            q.C0 = p->dHd; q.ldc0 = Hp; q.splitk0 = p->sk_dh; q.ws0 = p->gemm_ws;
            q.add_table = p->tgt_table; q.add_index = p->tgt_index; q.add_scale = p->dlt; q.add_ld = p->tgt_ld;
            q.a_kc1 = 0; q.b_kc1 = 0; q.M1 = K; q.N1 = Hp; q.K1 = n; q.A1 = p->ln; q.lda1 = K; q.B1 = p->Hout; q.ldb1 = Hp;
            q.splitk1 = p->sk_deneg < 1 ? 1 : p->sk_deneg; q.ws1 = p->dEneg_slabs;
            STEP_TRY(seqrec_gemm_f32_pair(&q, st));
            p->ns_deneg = q.n_slabs1;
        } else {
            seqrec_gemm_fuse fh = {};
            fh.add_table = p->tgt_table; fh.add_index = p->tgt_index; fh.add_scale = p->dlt; fh.add_ld = p->tgt_ld;
            STEP_TRY(seqrec_gemm_f32_fused(1, 0, n, Hp, K, p->ln, K, p->Eneg, Hp, p->dHd, Hp, nullptr, 0, p->sk_dh, p->sk_dh > 1 ? p->gemm_ws : nullptr,
                                           &fh, st));
        }
        if (p->deneg_mode == 1) {
            int ns = 0;
            STEP_TRY(seqrec_gemm_f32_slabs(0, 0, K, Hp, n, p->ln, K, p->Hout, Hp, p->sk_deneg < 1 ? 1 : p->sk_deneg, p->dEneg_slabs, &ns, st));
            p->ns_deneg = ns;
        }
        STEP_TRY(seqrec_rnn_bwd_stepwise(p->cell, p->act, Hp, p->H_real, p->T, p->B, nullptr, p->step_off_host, n, p->dHd, p->Hout, p->gates,
                                         p->aux, p->dPre, p->upack, p->scan_ws, nullptr, p->use_graph, st));
        seqrec_gemm_desc* d = p->descs_out;
        int c = 0;
        auto put = [&](int64_t M, int64_t N, const float* A, int64_t lda, const float* B, float* C, const int32_t* idx) {
            d[c].M = M; d[c].N = N; d[c].K = n; d[c].A = A; d[c].lda = lda; d[c].B = B; d[c].ldb = GHp; d[c].C = C; d[c].ldc = GHp;
            d[c].bias = nullptr; d[c].accumulate = 0; d[c].a_index = idx;
            ++c;
        };
        if (p->cell == SEQREC_CELL_GRU) {
            put(Hp, 2 * Hp, p->Hout, Hp, p->dPre, p->dU, p->prev);
            put(Hp, Hp, p->aux, Hp, p->dPre + 2 * Hp, p->dU + 2 * Hp, nullptr);
        } else {
            put(Hp, GHp, p->Hout, Hp, p->dPre, p->dU, p->prev);
        }
        put(Dp, GHp, p->x_table, p->x_ld, p->dPre, p->dW, p->x_index);
        if (p->db) put(1, GHp, p->ones, 4, p->dPre, p->db, nullptr);
        const int n_w = c;
        if (p->deneg_mode == 2) {          // dEneg = dln^T . Hout rides as the LAST problem (its own B operand and row stride)
            d[c].M = K; d[c].N = Hp; d[c].K = n; d[c].A = p->ln; d[c].lda = K; d[c].B = p->Hout; d[c].ldb = Hp; d[c].C = p->Hout; d[c].ldc = Hp;
            d[c].bias = nullptr; d[c].accumulate = 0; d[c].a_index = nullptr;
            ++c;
        }
        p->n_descs = n_w;
        p->ns_wgrad = 1;
        p->deneg_off = 0;
        if (p->wgrad_slabs) {
            int ns = 0;
            STEP_TRY(seqrec_gemm_f32_grouped_slabs(c, 0, 0, d, p->sk_wgrad, p->wgrad_ws, &ns, st));
            p->ns_wgrad = ns;
            if (p->deneg_mode == 2) {
                int64_t off = 0;
                for (int i = 0; i < n_w; ++i) off += (int64_t)ns * d[i].M * d[i].N;
                p->deneg_off = off;
                p->ns_deneg = ns;
            }
        } else {
            if (p->deneg_mode == 2) return SEQREC_E_ARG;
            STEP_TRY(seqrec_gemm_f32_grouped(c, 0, 0, d, p->sk_wgrad, p->sk_wgrad > 1 ? p->wgrad_ws : nullptr, st));
        }
    }
    if (p->stages & 4) {
        int ns = 0;
        STEP_TRY(seqrec_gemm_f32_slabs(1, 1, n, Dp, GHp, p->dPre, GHp, p->W, GHp, p->sk_dx < 1 ? 1 : p->sk_dx, p->dX_slabs, &ns, st));
        p->ns_dx = ns;
    }
    return 0;
}

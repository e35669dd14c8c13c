"""Worker for the gloo (CPU) tests of distributed.RowExchange; run under torch.distributed.run."""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def seg_plan_checks(D, ex, E, rank, R):
    """plan_seg / fetch_seg / push_seg: requests against TWO stacked tables plus `extra` owner-chosen
    rows per peer, forward rows and reverse gradients against a dense single-process reference."""
    V, w = E.shape
    g2 = torch.Generator().manual_seed(2)
    F = torch.randn(V, w, generator=g2, dtype=torch.float64)               # second table (Eout), same width
    nE = D.shard_size(V, rank, R)
    uni = torch.cat([D.shard_rows(E, rank, R), D.shard_rows(F, rank, R)])  # my unified shard: E rows then F rows
    gr = torch.Generator().manual_seed(300 + rank)
    for case, extra in enumerate((0, 3, 5)):
        n1 = 0 if (case == 2 and rank == 0) else 40 + 9 * rank
        n2 = 25 + 4 * rank
        a = torch.randint(0, V, (n1,), generator=gr)
        b = torch.randint(0, V, (n2,), generator=gr)
        if n1 > 5:
            a[:5] = 11
        owner = torch.cat([a % R, b % R])
        off = (V - (b % R) + R - 1) // R                                     # E rows held by the owner of b
        plan = ex.plan_seg(owner, torch.cat([a // R, b // R + off]), extra=extra)
        n = n1 + n2
        assert plan.n_tot == n + R * extra and sum(plan.req_split) == plan.n_tot and sum(plan.own_split) == plan.m_tot
        # owner side: requested rows + `extra` rows I pick per peer (here: F rows of my shard, ids known to me)
        idx = plan.own_rows.clone().long()
        pick = torch.randint(0, uni.shape[0] - nE, (R, extra), generator=gr) + nE
        if extra:
            idx[plan.own_extra.long().reshape(-1)] = pick.reshape(-1)
        assert (idx >= 0).all()
        recv = ex.fetch_seg(plan, uni[idx].contiguous())
        got = recv[plan.req_pos.long()]
        assert torch.equal(got[:n1], E[a]) and torch.equal(got[n1:], F[b]), "fetch_seg mismatch (case %d)" % case
        picks = [None] * R
        dist.all_gather_object(picks, pick)
        for j in range(R):                                                   # extras from owner j: its picks for me
            want = D.shard_rows(F, j, R)[picks[j][rank] - D.shard_size(V, j, R)]
            assert torch.equal(recv[plan.req_extra.long()[j]], want), "extras mismatch (case %d)" % case
        # reverse: gradients for requests and extras
        gq = torch.randn(n, w, generator=gr, dtype=torch.float64)
        gx = torch.randn(R * extra, w, generator=gr, dtype=torch.float64)
        src = torch.cat([gq, gx])
        back = plan.back_src.clone().long()
        if extra:
            back[plan.req_extra.long().reshape(-1)] = n + torch.arange(R * extra)
        assert sorted(back.tolist()) == list(range(n + R * extra))
        gown = ex.push_seg(plan, src[back].contiguous())
        acc = torch.zeros_like(uni)
        acc.index_add_(0, idx, gown)
        everything = [None] * R
        dist.all_gather_object(everything, (a, b, gq, gx, pick))
        refE = torch.zeros(V, w, dtype=torch.float64)
        for aj, bj, gqj, gxj, pj in everything:
            refE.index_add_(0, aj, gqj[: aj.numel()])
        assert torch.allclose(acc[:nE], D.shard_rows(refE, rank, R), atol=1e-12), "push_seg E mismatch (case %d)" % case
        # second table: target-row gradients of every rank + the extras every owner picked for every requester
        refF2 = torch.zeros(V, w, dtype=torch.float64)
        for j, (aj, bj, gqj, gxj, pj) in enumerate(everything):
            refF2.index_add_(0, bj, gqj[aj.numel():])
        for o in range(R):                                                   # owner o, requester j
            for j in range(R):
                if extra:
                    rows_global = (everything[o][4][j] - D.shard_size(V, o, R)) * R + o
                    refF2.index_add_(0, rows_global, everything[j][3].view(R, extra, w)[o])
        assert torch.allclose(acc[nE:], D.shard_rows(refF2, rank, R), atol=1e-12), "push_seg F mismatch (case %d)" % case


def many_plan_checks(D, ex, rank, R):
    """plan_seg_many (two collectives + one host sync for M batches, host index arithmetic) must build exactly the
    plans plan_seg builds batch by batch -- including an empty request list and an owner nobody asks."""
    import numpy as np
    gr = np.random.default_rng(500 + rank)
    V = 997
    reqs = []
    for b in range(5):
        n = 0 if (b == 3 and rank == R - 1) else int(gr.integers(20, 90))
        ids = gr.integers(0, V, size=n)
        if b == 1:
            ids = ids // R * R                     # everything goes to owner 0
        reqs.append((ids % R, (ids // R).astype(np.int32)))
    for extra in (0, 4):
        many = ex.plan_seg_many(reqs, extra=extra)
        for (o, w), pm in zip(reqs, many):
            ps = ex.plan_seg(torch.from_numpy(o), torch.from_numpy(w), extra=extra)
            assert (pm.n, pm.extra, pm.n_tot, pm.m_tot) == (ps.n, ps.extra, ps.n_tot, ps.m_tot)
            assert pm.req_split == ps.req_split and pm.own_split == ps.own_split
            for f in ("req_pos", "req_extra", "own_rows", "own_extra", "back_src"):
                assert torch.equal(getattr(pm, f).long(), getattr(ps, f).long()), (f, extra)


def unified_plan_checks(D, ex, rank, R):
    """RowExchange.plan_unified (native host planner, csrc/route.hip) must produce exactly the index blob the numpy
    arithmetic of plan_seg_many + the unified step's derived arrays gives -- untied and tied tables, with and without the
    log-Q vector, including a batch whose requests all go to one owner."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_sessions
    Bt = importlib.import_module("seq-recommendations_amd.batching")
    gr = np.random.default_rng(900 + rank)
    V, w, Kr = 1201, 8, 11
    nid = -(-Kr // w)
    lqh = gr.normal(size=V).astype(np.float32)
    for tied in (False, True):
        rbs = []
        for b in range(4):
            sess = make_sessions(gr, 9 + 3 * rank + b, V, 2, 9)
            if b == 2:
                sess = [[(x // R) * R for x in s_] for s_ in sess]           # every row lives on owner 0
            rbs.append(Bt.pack_sessions(sess))
        for lq in (None, lqh):
            got = ex.plan_unified(rbs, V, tied, Kr, nid, w, lq_host=lq)
            # reference: the numpy planner + the derived arrays of the unified step
            reqs = []
            for rb in rbs:
                ids, tgt = rb.ids.astype(np.int64), rb.tgt.astype(np.int64)
                off = np.zeros_like(tgt) if tied else (V - tgt % R + R - 1) // R
                reqs.append((np.concatenate([ids % R, tgt % R]), np.concatenate([ids // R, tgt // R + off])))
            ref = ex.plan_seg_many(reqs, extra=Kr + nid, device_fields=False, tokens=[rb.n_tok for rb in rbs])
            q = np.arange(Kr)
            for rb, (blob, parts, plan), pr in zip(rbs, got, ref):
                n, h = rb.n_tok, pr.host
                oe, re_ = h["own_extra"], h["req_extra"]
                back = h["back"].copy()
                back[re_[:, :Kr].reshape(-1)] = 2 * n + np.arange(R * Kr)
                # owner-side row kinds (seqrec_exchange_pack): the request's index into the received list, -2 at my draws, -1 at id rows
                own_kinds = h["own_src"].copy()
                own_kinds[oe[:, :Kr].reshape(-1)] = -2
                own_kinds[oe[:, Kr:Kr + nid].reshape(-1)] = -1
                want = {"step_off": rb.step_off, "prev": rb.prev, "ids": rb.ids, "tgt": rb.tgt, "neg_slots": oe[:, :Kr].reshape(-1),
                        "id_rows": oe[:, Kr:Kr + nid].reshape(-1), "take_in": h["req_pos"][:n], "take_tgt": h["req_pos"][n:],
                        "neg_rows": re_[:, :Kr].reshape(-1), "negid_idx": (re_[:, Kr + q // w] * w + (q % w)[None, :]).reshape(-1),
                        "back_idx": back, "own_src": own_kinds,
                        "ntok": np.array([pr.n_global], np.float32).view(np.int32)}
                if lq is not None:
                    want["lq_tgt"] = lq[rb.tgt].view(np.int32)
                assert [nm for nm, _, _ in parts] == list(want), [nm for nm, _, _ in parts]
                bl = blob.numpy()
                for nm, o, cnt in parts:
                    assert cnt == len(want[nm]), (nm, cnt, len(want[nm]))
                    assert np.array_equal(bl[o:o + cnt], np.asarray(want[nm], dtype=np.int64).astype(np.int32)), (nm, tied, lq is not None)
                assert (plan.n_tot, plan.m_tot, plan.req_split, plan.own_split, plan.n_global) == \
                    (pr.n_tot, pr.m_tot, pr.req_split, pr.own_split, pr.n_global)
                assert torch.equal(plan.got_pad, pr.got_pad)
            # round 4: planning in two halves (ShardedEngine.prepare_begin / prepare_end) -- the count exchange queued, OTHER
            # collectives of the training loop issued in between, the rest later: the same blobs and plans as the one-shot call
            hd = ex.plan_unified_begin(rbs, V, tied, Kr, nid, w, lq_host=lq)
            t = torch.ones(3)
            dist.all_reduce(t)                                     # (a training step's collective between the two halves)
            assert float(t[0]) == R
            got2 = ex.plan_unified_end(hd)
            for (b1, p1, q1), (b2, p2, q2) in zip(got, got2):
                assert p1 == p2 and torch.equal(b1, b2) and torch.equal(q1.got_pad, q2.got_pad)
                assert (q1.n_tot, q1.m_tot, q1.req_split, q1.own_split, q1.n_global) == (q2.n_tot, q2.m_tot, q2.req_split, q2.own_split, q2.n_global)


def main():
    dist.init_process_group("gloo")
    rank, R = dist.get_rank(), dist.get_world_size()
    D = importlib.import_module("seq-recommendations_amd.distributed")
    ex = D.RowExchange(dist, None, "cpu")
    V, w = 1003, 8
    g = torch.Generator().manual_seed(1)
    E = torch.randn(V, w, generator=g, dtype=torch.float64)            # identical on every rank
    shard = D.shard_rows(E, rank, R)
    assert shard.shape[0] == D.shard_size(V, rank, R)
    take = lambda src, idx: src[idx.long()]
    gl = lambda idx: shard[idx.long()]
    gr = torch.Generator().manual_seed(100 + rank)
    for case in range(4):
        if case == 0:
            n = 200 + 17 * rank
            ids = torch.randint(0, V, (n,), generator=gr)
            ids[:20] = 7                                                  # duplicates (Zipf head)
        elif case == 1:
            ids = torch.randint(0, V, (50,), generator=gr) // R * R       # every request goes to owner 0
        elif case == 2:
            ids = torch.zeros(0, dtype=torch.long) if rank == 0 else torch.randint(0, V, (31,), generator=gr)
        else:
            ids = torch.arange(rank, V, R)                                # only my own rows
        plan = ex.plan(ids)
        assert plan.n == ids.numel() and sum(plan.send_counts) == ids.numel() and plan.m == sum(plan.recv_counts)
        rows = ex.fetch(plan, gl, w, take)
        assert torch.equal(rows, E[ids]), "fetch mismatch (case %d)" % case
        # backward: contributions reach the owners; summed shards == dense scatter-add of everyone
        grads = torch.randn(ids.numel(), w, generator=gr, dtype=torch.float64)
        contrib, local = ex.push(plan, grads, take)
        gshard = torch.zeros_like(shard)
        gshard.index_add_(0, local.long(), contrib)
        all_ids = [None] * R
        all_g = [None] * R
        dist.all_gather_object(all_ids, ids)
        dist.all_gather_object(all_g, grads)
        ref = torch.zeros(V, w, dtype=torch.float64)
        for i, gg in zip(all_ids, all_g):
            ref.index_add_(0, i, gg)
        assert torch.allclose(gshard, D.shard_rows(ref, rank, R), atol=1e-12), "push mismatch (case %d)" % case
    seg_plan_checks(D, ex, E, rank, R)
    many_plan_checks(D, ex, rank, R)
    unified_plan_checks(D, ex, rank, R)
    x = torch.arange(R * 3, dtype=torch.float32).view(R, 3) + 100 * rank
    y = ex.swap_fixed(x)
    for i in range(R):
        assert torch.equal(y[i], torch.arange(rank * 3, rank * 3 + 3, dtype=torch.float32) + 100 * i)
    # the flat dense-gradient bucket: sum over ranks
    flat = torch.full((10,), float(rank + 1))
    dist.all_reduce(flat)
    assert torch.all(flat == R * (R + 1) / 2)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()

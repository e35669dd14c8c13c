#!/usr/bin/env python3
"""Row-sparse tail of the step in isolation (developer tool, GPU box): scatter-add, norm and update launches of c3's
shape, timed with events, on the real duplicate structure of a batch (Zipf items: a few rows take hundreds of
contributions) against the same number of all-distinct rows.  The difference is what same-row atomics cost.

    python tools/rows_probe.py [reps]
"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
Sm = importlib.import_module("seq-recommendations_amd.sampling")
ptr, call = L.ptr, L.call
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
V, W, K = 1_000_000, 256, 2000
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
gen = Sy.SyntheticSessions(V, seed=1234)
flat, starts = gen.generate(512 * 4)
probs = Sm.log_uniform_probs(V, gen.proposal_rank())
rng = np.random.default_rng(5)
E = torch.randn(V, W, device=dev) * 0.1; Eo = torch.randn(V, W, device=dev) * 0.1
AE = torch.zeros(V, W, device=dev); AEo = torch.zeros(V, W, device=dev)
GE = torch.zeros(V, W, device=dev); GEo = torch.zeros(V, W, device=dev)
SE = torch.full((V,), 2**31 - 1, dtype=torch.int32, device=dev); SEo = SE.clone()
dense = [torch.randn(256, 768, device=dev) * 0.01, torch.randn(256, 768, device=dev) * 0.01, torch.randn(768, device=dev) * 0.01]
dp = [torch.randn_like(t) for t in dense]; da = [torch.zeros_like(t) for t in dense]
sq = torch.zeros(2, device=dev); scale = torch.zeros(1, device=dev); loss_out = torch.zeros(2, device=dev)
flush = torch.empty(96 << 20, device=dev)


def run(tag, ids, tgt, neg, cold):
    n = len(ids)
    ti = torch.from_numpy(ids.astype(np.int32)).to(dev); tt = torch.from_numpy(tgt.astype(np.int32)).to(dev)
    tn = torch.from_numpy(neg.astype(np.int32)).to(dev)
    dX = torch.randn(n, W, device=dev) * 0.01; Hd = torch.randn(n, W, device=dev) * 0.01; dlt = torch.randn(n, device=dev)
    dEn = torch.randn(K, W, device=dev) * 0.01; lrows = torch.rand(n, device=dev)
    jobs = [dict(table=Eo, accum=AEo, gtab=GEo, slot=SEo, rows=tt, vals=Hd, ldv=W, row_scale=dlt, n=n, width=W, base=0),
            dict(table=Eo, accum=AEo, gtab=GEo, slot=SEo, rows=tn, vals=dEn, ldv=W, row_scale=None, n=K, width=W, base=n),
            dict(table=E, accum=AE, gtab=GE, slot=SE, rows=ti, vals=dX, ldv=W, row_scale=None, n=n, width=W, base=0)]
    arr, cnt = L.rows_jobs(jobs)
    gp, pp, ap_ = L.ptr_array(dense), L.ptr_array(dp), L.ptr_array(da)
    nn = L.i64_array([t.numel() for t in dense])
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    for r in range(reps):
        if cold:
            flush.normal_()
        sq.zero_()
        ev[r][0].record()
        call("seqrec_rows_scatter_add_multi", arr, cnt, st)
        ev[r][1].record()
        call("seqrec_opt_sqnorm", len(dense), gp, nn, arr, cnt, ptr(sq), ptr(lrows), n, ptr(loss_out), st)
        ev[r][2].record()
        call("seqrec_opt_apply", len(dense), pp, ap_, gp, nn, arr, cnt, ptr(sq), 1.0, 0.01, 1e-8, ptr(scale), None, None, None, None, st)
        ev[r][3].record()
    torch.cuda.synchronize()
    t = np.array([[ev[r][i].elapsed_time(ev[r][i + 1]) * 1e3 for i in range(3)] for r in range(5, reps)])
    u = len(np.unique(np.concatenate([tt.cpu().numpy(), tn.cpu().numpy()]))) + len(np.unique(ids))
    print("%-22s %s contributions %5d distinct rows %5d | scatter %6.1f  sqnorm %6.1f  apply %6.1f us (median)" %
          (tag, "cold" if cold else "warm", 2 * n + K, u, *np.median(t, axis=0)))


for b in range(2):
    rb = Bt.pack_flat(flat, starts, np.arange(b * 512, (b + 1) * 512))
    ids, tgt = np.asarray(rb.ids), np.asarray(rb.tgt)
    n = len(ids)
    neg = rng.choice(V, size=K, p=probs)
    for cold in (False, True):
        run("batch %d real" % b, ids, tgt, neg, cold)
        perm = rng.permutation(V)
        run("batch %d distinct" % b, perm[:n], perm[n:2 * n], perm[2 * n:2 * n + K], cold)

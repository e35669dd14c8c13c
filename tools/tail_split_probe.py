#!/usr/bin/env python3
"""Does it pay to SPLIT the cluster scan at the step t* where <= 2 row blocks are left alive, and run the tail as its own 16-32
workgroup launch BESIDE the output-side work (logits GEMM + CE + dH GEMM) of the finished tokens?  (developer tool, GPU box;
VERDICT r3 item 3 -- round 3's probe measured an UNSPLIT 512-workgroup scan beside GEMMs started at the scan's first step.)

The split is emulated without touching the kernels: the bulk scan is a batch whose session lengths are clipped at t*, the tail scan
a batch of the <= 32 sessions that are longer, with their remaining steps -- same launches, same per-step work as the split would
issue (the tail only lacks the one-off load of h at t* - 1).  Forward shapes of c3 (GRU 256, K 2000) and c4 (LSTM 512, K 4000).

Measured (HIP events on the main stream, mean of `reps`):
  A  serial, unsplit      scan(full) -> logits(n) -> CE(n) -> dH(n)
  B  serial, split        scan(bulk) -> scan(tail) -> logits(n) -> CE(n) -> dH(n)
  C  overlapped           scan(bulk) -> fork -> { scan(tail) on main || logits/CE/dH(bulk tokens) on side } -> join -> logits/CE/dH(tail tokens)
  C0 fork/join alone      C without the side-stream work and without the tail-token launches (cost of the two events)
  D  like C, the tail scan on a HIGH-priority side stream and the GEMMs on main
  and for the BPTT:  E serial  bwd(full) + dEneg     F  { bwd(tail) || dEneg } -> join -> bwd(bulk)
"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
ptr, call = L.ptr, L.call
lib = L.load()
dev = "cuda"


def run(cell, H, K, seeds=(0, 1, 2), reps=30):
    G = {"gru": 3, "lstm": 4}[cell]
    ci = L.CELL[cell]
    gen = Sy.SyntheticSessions(100000, seed=1234)
    flat, starts = gen.generate(512 * (max(seeds) + 1))
    main = torch.cuda.current_stream()
    st = main.cuda_stream
    side = torch.cuda.Stream()
    hi = torch.cuda.Stream(priority=-1)
    U = (torch.randn(H, G * H, device=dev) * (0.5 / np.sqrt(H))).contiguous()
    up = torch.empty(int(lib.seqrec_rnn_upack_floats(ci, H)), device=dev)
    call("seqrec_rnn_pack_u_stepwise", ci, H, ptr(U), ptr(up), st)
    V = 100000
    Et = (torch.randn(V, H, device=dev) * 0.01).contiguous()
    neg = torch.randint(0, V, (K,), dtype=torch.int32, device=dev)
    Eneg = Et[neg.long()].contiguous()
    res = []
    for sd in seeds:
        sel = np.arange(512 * sd, 512 * (sd + 1))
        sess = [flat[starts[i]:starts[i + 1]].tolist() for i in sel]
        rb = Bt.pack_sessions(sess)
        n, T = rb.n_tok, rb.T
        so = rb.step_off
        Bt_ = np.diff(so)                                 # live rows per step
        tstar = int(np.argmax(Bt_ <= 32)) if (Bt_ <= 32).any() else T
        bulk = Bt.pack_sessions([s[: tstar + 1] for s in sess])
        tail = Bt.pack_sessions([s[tstar:] for s in sess if len(s) - 1 > tstar])
        n1, n2 = bulk.n_tok, tail.n_tok
        assert n1 + n2 == n, (n1, n2, n)

        def bufs(m):
            return dict(XW=torch.randn(m, G * H, device=dev) * 0.3, Hout=torch.zeros(m, H, device=dev), gates=torch.zeros(m, G * H, device=dev),
                        aux=torch.zeros(m, H, device=dev), dH=torch.randn(m, H, device=dev) * 0.1, dPre=torch.zeros(m, G * H, device=dev),
                        ws=torch.zeros(2 * m * H, device=dev))
        bf, bb, bt = bufs(n), bufs(max(n1, 1)), bufs(max(n2, 1))
        ln = torch.empty(n, K, device=dev); dHd = torch.empty(n, H, device=dev); gws = torch.empty(4 * n * H, device=dev); gws2 = torch.empty(4 * n * H, device=dev)
        tgt = torch.randint(0, V, (n,), dtype=torch.int32, device=dev)
        loss = torch.empty(n, device=dev); dlt = torch.empty(n, device=dev)
        dEn = torch.empty(K, H, device=dev); ews = torch.empty(4 * K * H, device=dev)

        def fwd(r, b, s):
            call("seqrec_rnn_fwd_stepwise", ci, 0, H, H, r.T, r.B, None, r.step_off.ctypes.data, ptr(b["XW"]), ptr(b["Hout"]), ptr(b["gates"]), ptr(b["aux"]), ptr(up), None, 0, s)

        def bwd(r, b, s):
            call("seqrec_rnn_bwd_stepwise", ci, 0, H, H, r.T, r.B, None, r.step_off.ctypes.data, r.n_tok, ptr(b["dH"]), ptr(b["Hout"]), ptr(b["gates"]), ptr(b["aux"]),
                 ptr(b["dPre"]), ptr(up), ptr(b["ws"]), None, 0, s)

        def outside(lo, m, s, ws):                         # logits + CE + dH of tokens [lo, lo + m)
            if m <= 0:
                return
            Hd = bf["Hout"][lo:lo + m]
            call("seqrec_gemm_f32", 1, 1, m, K, H, ptr(Hd), H, ptr(Eneg), H, ptr(ln[lo:lo + m]), K, None, 0, 1, None, s)
            call("seqrec_sampled_softmax_ce", ptr(ln[lo:lo + m]), K, ptr(Hd), H, ptr(Et), None, None, None, ptr(tgt[lo:lo + m]), ptr(neg), m, K, 1.0 / n,
                 ptr(loss[lo:lo + m]), ptr(dlt[lo:lo + m]), s)
            sk = 3 if m >= 1024 else 1
            call("seqrec_gemm_f32", 1, 0, m, H, K, ptr(ln[lo:lo + m]), K, ptr(Eneg), H, ptr(dHd[lo:lo + m]), H, None, 0, sk, ptr(ws) if sk > 1 else None, s)

        def deneg(s):
            call("seqrec_gemm_f32", 0, 0, K, H, n, ptr(ln), K, ptr(bf["Hout"]), H, ptr(dEn), H, None, 0, 4, ptr(ews), s)

        def A():
            fwd(rb, bf, st); outside(0, n, st, gws)

        def B():
            fwd(bulk, bb, st); fwd(tail, bt, st); outside(0, n, st, gws)

        def C(work=True):
            fwd(bulk, bb, st)
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            fwd(tail, bt, st)
            if work:
                outside(0, n1, side.cuda_stream, gws2)
            ev2 = torch.cuda.Event(); ev2.record(side); main.wait_event(ev2)
            if work:
                outside(n1, n2, st, gws)

        def D():
            fwd(bulk, bb, st)
            ev = torch.cuda.Event(); ev.record(main); hi.wait_event(ev)
            fwd(tail, bt, hi.cuda_stream)
            outside(0, n1, st, gws)
            ev2 = torch.cuda.Event(); ev2.record(hi); main.wait_event(ev2)
            outside(n1, n2, st, gws)

        def E():
            bwd(rb, bf, st); deneg(st)

        def F():
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            bwd(tail, bt, st)
            deneg(side.cuda_stream)
            ev2 = torch.cuda.Event(); ev2.record(side); main.wait_event(ev2)
            bwd(bulk, bb, st)

        def Fs():                                           # split BPTT, serial
            bwd(tail, bt, st); deneg(st); bwd(bulk, bb, st)

        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        parts = dict(scan_full=timed(lambda: fwd(rb, bf, st)), scan_bulk=timed(lambda: fwd(bulk, bb, st)), scan_tail=timed(lambda: fwd(tail, bt, st)),
                     out_all=timed(lambda: outside(0, n, st, gws)), out_bulk=timed(lambda: outside(0, n1, st, gws)), out_tail=timed(lambda: outside(n1, n2, st, gws)),
                     bwd_full=timed(lambda: bwd(rb, bf, st)), bwd_bulk=timed(lambda: bwd(bulk, bb, st)), bwd_tail=timed(lambda: bwd(tail, bt, st)), deneg=timed(lambda: deneg(st)))
        r = dict(seed=sd, T=T, tstar=tstar, n=n, n_tail=n2, rows_tail=tail.B, A=timed(A), B=timed(B), C=timed(C), C0=timed(lambda: C(False)), D=timed(D), E=timed(E), Fs=timed(Fs), F=timed(F))
        r.update(parts)
        res.append(r)
        print("%s H=%d K=%d seed %d: T %d, t* %d, tokens %d (tail %d in %d rows)" % (cell, H, K, sd, T, tstar, n, n2, tail.B))
        print("   parts  scan full %.1f = bulk %.1f + tail %.1f | out-side all %.1f, bulk %.1f, tail %.1f | bwd full %.1f = bulk %.1f + tail %.1f | dEneg %.1f"
              % tuple(parts[k] for k in ("scan_full", "scan_bulk", "scan_tail", "out_all", "out_bulk", "out_tail", "bwd_full", "bwd_bulk", "bwd_tail", "deneg")))
        print("   fwd   A serial unsplit %.1f | B serial split %.1f | C overlapped %.1f (fork/join alone, with both scans: %.1f) | D tail on high-priority stream %.1f"
              % (r["A"], r["B"], r["C"], r["C0"], r["D"]))
        print("   bwd   E serial unsplit %.1f | split serial %.1f | F tail || dEneg %.1f" % (r["E"], r["Fs"], r["F"]))
    errs = int(lib.seqrec_cluster_scan_errors(st))
    print("   exchange timeouts: %d" % errs)
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c4"]
    if "c3" in which:
        run("gru", 256, 2000)
    if "c4" in which:
        run("lstm", 512, 4000, seeds=(0, 1))

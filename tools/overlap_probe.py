#!/usr/bin/env python3
"""Does a GEMM stream running beside the scan slow the scan's launch chain?  (developer tool, GPU box)
Background: a second stream is pre-loaded with logits-shaped GEMMs; the scan is timed on the main
stream meanwhile.  With SEQREC_LIB pointing at a -DSEQREC_PROBE_XCD_SKIP=2 build of the library (same hipcc
line as seq-recommendations_amd/build.py plus that define) the background GEMM leaves XCDs 0-1, where the
late scan steps run, alone.  Measured (r01): scan alone 171 us; beside full-chip GEMMs 373 us; beside
GEMMs that avoid XCDs 0-1 359 us -- the slowdown is queue arbitration between the streams, not CUs."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
ptr, call = L.ptr, L.call
H, G, cell = 256, 3, "gru"
gen = Sy.SyntheticSessions(100000, seed=1234)
flat, starts = gen.generate(512 * 4)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
st = main.cuda_stream
U = (torch.randn(H, G * H, device="cuda") * (0.5 / np.sqrt(H))).contiguous()
up2 = torch.empty(int(L.load().seqrec_rnn_upack_floats(L.CELL[cell], H)), device="cuda")
call("seqrec_rnn_pack_u_stepwise", L.CELL[cell], H, ptr(U), ptr(up2), st)
M, N, K = 2546, 2000, 256
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
def bg(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            call("seqrec_gemm_f32", 1, 1, M, N, K, ptr(A), K, ptr(B), K, ptr(C), N, None, 0, 1, None, side.cuda_stream)
for mode in ("alone", "with background GEMMs"):
    res = []
    for i in range(4):
        rb = Bt.pack_flat(flat, starts, np.arange(i * 512, (i + 1) * 512))
        n = rb.n_tok
        so = torch.from_numpy(rb.step_off).cuda()
        XW = torch.randn(n, G * H, device="cuda") * 0.5
        Hout = torch.empty(n, H, device="cuda"); gates = torch.empty(n, G * H, device="cuda"); aux = torch.empty(n, H, device="cuda")
        soh = rb.step_off
        def f2(): call("seqrec_rnn_fwd_stepwise", L.CELL[cell], 0, H, H, rb.T, rb.B, ptr(so), soh.ctypes.data, ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up2), None, 0, st)
        f2(); torch.cuda.synchronize()
        if mode != "alone":
            bg(400)                      # ~15 ms of queued GEMMs
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f2()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 100.0)
    print("%-24s fwd scan %s us  (mean %.1f)" % (mode, ["%.1f" % x for x in res], np.mean(res)))

#!/usr/bin/env python3
"""Time the scan kernels alone on c3-shaped batches (developer tool, GPU box only).

Launch-chain floor: build a second library with empty step kernels and point SEQREC_LIB at it --
    cd seq-recommendations_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics \
        -I../../include -DSEQREC_SCAN_EMPTY gemm.hip ops.hip rnn.hip rnn_step.hip -o ../../tools/bin/libseqrec_scan_empty.so
    SEQREC_LIB=$PWD/tools/bin/libseqrec_scan_empty.so python tools/bench_scan.py gru 256
(the step-wise columns then show the cost of the dependent launches alone, eager and as a replayed graph)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
ptr, call = L.ptr, L.call

def main():
    cell = sys.argv[1] if len(sys.argv) > 1 else "gru"
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    G = L.N_GATES[cell]
    gen = Sy.SyntheticSessions(100000, seed=1234)
    flat, starts = gen.generate(512 * 8)
    st = torch.cuda.current_stream().cuda_stream
    U = (torch.randn(H, G * H, device="cuda") * (0.5 / np.sqrt(H))).contiguous()
    up = torch.empty(int(L.load().seqrec_rnn_upack_floats(L.CELL[cell], H)), device="cuda")
    call("seqrec_rnn_pack_u", L.CELL[cell], H, ptr(U), ptr(up), st)
    up2 = torch.empty_like(up)
    call("seqrec_rnn_pack_u_stepwise", L.CELL[cell], H, ptr(U), ptr(up2), st)
    res = []
    for i in range(8):
        rb = Bt.pack_flat(flat, starts, np.arange(i * 512, (i + 1) * 512))
        n = rb.n_tok
        so = torch.from_numpy(rb.step_off).cuda()
        XW = torch.randn(n, G * H, device="cuda") * 0.5
        Hout = torch.empty(n, H, device="cuda"); gates = torch.empty(n, G * H, device="cuda"); aux = torch.empty(n, H, device="cuda")
        dH = torch.randn(n, H, device="cuda") * 0.1; dPre = torch.empty(n, G * H, device="cuda")
        def f(): call("seqrec_rnn_fwd", L.CELL[cell], 0, H, H, rb.T, rb.B, ptr(so), ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up), st)
        def b(): call("seqrec_rnn_bwd", L.CELL[cell], 0, H, H, rb.T, rb.B, ptr(so), ptr(dH), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(up), st)
        ws = torch.empty(2 * n * H, device="cuda")
        soh = rb.step_off
        def f2(): call("seqrec_rnn_fwd_stepwise", L.CELL[cell], 0, H, H, rb.T, rb.B, ptr(so), soh.ctypes.data, ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up2), None, 0, st)
        def b2(): call("seqrec_rnn_bwd_stepwise", L.CELL[cell], 0, H, H, rb.T, rb.B, ptr(so), soh.ctypes.data, n, ptr(dH), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(up2), ptr(ws), None, 0, st)
        def f3(): call("seqrec_rnn_fwd_stepwise", L.CELL[cell], 0, H, H, rb.T, rb.B, ptr(so), soh.ctypes.data, ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up2), None, 1, st)
        def b3(): call("seqrec_rnn_bwd_stepwise", L.CELL[cell], 0, H, H, rb.T, rb.B, ptr(so), soh.ctypes.data, n, ptr(dH), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(up2), ptr(ws), None, 1, st)
        fns = (f, b, f2, b2, f3, b3)
        out = []
        for fn in fns:
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 100.0)     # us per launch
        res.append((rb.T, n) + tuple(out))
    for r in res:
        T, n = r[0], r[1]
        print("T=%2d n_tok=%5d  " % (T, n) + "  ".join("%7.1f us (%.2f/step)" % (x, x / T) for x in r[2:]))
    print("mean [persistent fwd, bwd, stepwise fwd, bwd, graph fwd, bwd]:", ["%.1f" % np.mean([r[i] for r in res]) for i in range(2, len(res[0]))])

main()

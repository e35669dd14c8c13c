#!/usr/bin/env python3
"""Where does a step of the cluster LSTM scans spend its time?  (developer tool, GPU box; DIAGNOSTIC build only)

    python tools/build_diag.py stamp        # -DSEQREC_CLUSTER_STAMP build -> tools/diag/libseqrec_clstamp.so
    SEQREC_LIB=$PWD/tools/diag/libseqrec_clstamp.so python tools/cluster_stamps2.py [H]

Workgroup (row block 0, column block 1) sums s_memrealtime (100 MHz) between marked points of every step; MSNBC-shaped
batches of 512 sessions (c4: H = 512)."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
ptr, call = L.ptr, L.call
H, G = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 4
lib = L.load()
lib.seqrec_debug_cluster_stamps2.argtypes = [ctypes.c_void_p]
lib.seqrec_debug_cluster_stamps2.restype = None
gen = Sy.SyntheticSessions(100000, seed=1234)
flat, starts = gen.generate(512 * 4)
st = torch.cuda.current_stream().cuda_stream
U = (torch.randn(H, G * H, device="cuda") * (0.5 / np.sqrt(H))).contiguous()
up = torch.empty(int(lib.seqrec_rnn_upack_floats(1, H)), device="cuda")
call("seqrec_rnn_pack_u_stepwise", 1, H, ptr(U), ptr(up), st)
lab_f = ["loop top (xw in registers)", "wait exchange (h of t-1)", "load h rows (DMA + LDS read)", "4 gate tiles + reduce", "cell, stores, drain, flag"]
lab_b = ["loop top", "pointwise + 4 dPre stores", "drain, barrier, flag, prefetch issue", "wait exchange (dPre of t)", "pieces: DMA ring + 128 MFMA", "reduce"]
tf, tb, sf, sb = np.zeros(12), np.zeros(12), 0, 0
buf = (ctypes.c_ulonglong * 32)()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tms = []
for b in range(4):
    rb = Bt.pack_flat(flat, starts, np.arange(b * 512, (b + 1) * 512))
    n = rb.n_tok
    XW = torch.randn(n, G * H, device="cuda") * 0.3
    dH = torch.randn(n, H, device="cuda") * 0.1
    Hout = torch.zeros(n, H, device="cuda"); gates = torch.zeros(n, G * H, device="cuda"); aux = torch.zeros(n, H, device="cuda")
    dPre = torch.zeros(n, G * H, device="cuda"); ws = torch.zeros(2 * n * H, device="cuda")
    for rep in range(3):
        ev[0].record()
        call("seqrec_rnn_fwd_stepwise", 1, 0, H, H, rb.T, rb.B, None, rb.step_off.ctypes.data, ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up), None, 0, st)
        ev[1].record()
        call("seqrec_rnn_bwd_stepwise", 1, 0, H, H, rb.T, rb.B, None, rb.step_off.ctypes.data, n, ptr(dH), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(up), ptr(ws), None, 0, st)
        ev[2].record()
        torch.cuda.synchronize()
    tms.append((rb.T, n, ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3))
    lib.seqrec_debug_cluster_stamps2(buf)
    s = np.array(buf[:], dtype=np.float64)
    tf += s[:12]; sf += s[12]; tb += s[16:28]; sb += s[28]
print("batches (T, tokens, forward us, BPTT us):", [(t, n, round(a), round(b)) for t, n, a, b in tms])
for name, lab, tot, steps in (("forward", lab_f, tf, sf), ("BPTT", lab_b, tb, sb)):
    print("cluster LSTM %s H=%d, row block 0 / column block 1, %d steps: mean ns per step and section" % (name, H, steps))
    for l, v in zip(lab, tot / max(steps, 1) * 10.0):
        print("   %-40s %7.0f ns" % (l, v))
    print("   %-40s %7.0f ns" % ("sum", tot.sum() / max(steps, 1) * 10.0))

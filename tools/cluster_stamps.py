#!/usr/bin/env python3
"""Where does a step of the cluster GRU forward scan spend its time?  (developer tool, GPU box; DIAGNOSTIC build only)

    cd seq-recommendations_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -I../../include \
        -DSEQREC_CLUSTER_STAMP gemm.hip ops.hip rnn.hip rnn_step.hip rnn_cluster.hip merge.hip -o ../../tools/bin/libseqrec_clstamp.so
    SEQREC_LIB=$PWD/tools/bin/libseqrec_clstamp.so python tools/cluster_stamps.py

Workgroup (row block 0, column block 1) sums s_memrealtime (100 MHz) between marked points of every step t >= 1."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
ptr, call = L.ptr, L.call
H, G = 256, 3
lib = L.load()
lib.seqrec_debug_cluster_stamps.argtypes = [ctypes.c_void_p]
lib.seqrec_debug_cluster_stamps.restype = None
gen = Sy.SyntheticSessions(100000, seed=1234)
flat, starts = gen.generate(512 * 4)
st = torch.cuda.current_stream().cuda_stream
U = (torch.randn(H, G * H, device="cuda") * (0.5 / np.sqrt(H))).contiguous()
up = torch.empty(int(lib.seqrec_rnn_upack_floats(2, H)), device="cuda")
call("seqrec_rnn_pack_u_stepwise", 2, H, ptr(U), ptr(up), st)
labels = ["loop top + xw loads issued", "wait exchange 2 (h of t-1)", "load h rows", "r tile (16 MFMA) + reduce", "epilogue r, store, drain, flag",
          "z tile under the exchange", "wait exchange 1 (r*h)", "load r*h rows", "h tile (16 MFMA) + reduce", "epilogue h, store, drain, flag"]
labels_b = ["loop top, element-wise d", "store d", "drain, barrier, flag (A)", "prefetch of step t-1 issued", "wait exchange A (d)", "load d rows",
            "drh tile (16 MFMA) + reduce", "dpre_z / dpre_r, stores, drain, barrier, flag (B)", "wait exchange B", "load [dz|dr] rows (K = 2H)",
            "[dz|dr] tile (32 MFMA) + reduce, carry"]
tot = np.zeros(10); steps = 0; totb = np.zeros(11); stepsb = 0
buf = (ctypes.c_ulonglong * 32)()
for b in range(4):
    rb = Bt.pack_flat(flat, starts, np.arange(b * 512, (b + 1) * 512))
    n = rb.n_tok
    XW = torch.randn(n, G * H, device="cuda") * 0.3
    Hout = torch.zeros(n, H, device="cuda"); gates = torch.zeros(n, G * H, device="cuda"); aux = torch.zeros(n, H, device="cuda")
    dH = torch.randn(n, H, device="cuda") * 0.1; dPre = torch.zeros(n, G * H, device="cuda"); ws = torch.empty(2 * n * H, device="cuda")
    for rep in range(3):
        call("seqrec_rnn_fwd_stepwise", 2, 0, H, H, rb.T, rb.B, None, rb.step_off.ctypes.data, ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up), None, 0, st)
        torch.cuda.synchronize()
    for rep in range(3):
        call("seqrec_rnn_bwd_stepwise", 2, 0, H, H, rb.T, rb.B, None, rb.step_off.ctypes.data, n, ptr(dH), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(up), ptr(ws), None, 0, st)
        torch.cuda.synchronize()
    lib.seqrec_debug_cluster_stamps(buf)
    s = np.array(buf[:], dtype=np.float64)
    tot += s[:10]; steps += s[12]; totb += s[13:24]; stepsb += s[25]
print("cluster forward, row block 0 / column block 1, %d steps: mean ns per step and section" % steps)
for l, v in zip(labels, tot / steps * 10.0):
    print("   %-36s %7.0f ns" % (l, v))
print("   %-36s %7.0f ns" % ("sum", tot.sum() / steps * 10.0))
print("cluster BPTT, row block 0 / column block 1, %d steps: mean ns per step and section" % stepsb)
for l, v in zip(labels_b, totb / max(stepsb, 1) * 10.0):
    print("   %-52s %7.0f ns" % (l, v))
print("   %-52s %7.0f ns" % ("sum", totb.sum() / max(stepsb, 1) * 10.0))

#!/usr/bin/env python3
"""How many polls does a wait of the cluster GRU scans take?  (developer tool, GPU box; DIAGNOSTIC build only)

    cd seq-recommendations_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -I../../include \
        -DSEQREC_CLUSTER_SPINS gemm.hip ops.hip rnn.hip rnn_step.hip rnn_cluster.hip merge.hip -o ../../tools/bin/libseqrec_clspins.so
    SEQREC_LIB=$PWD/tools/bin/libseqrec_clspins.so python tools/cluster_spins.py

Every wave counts the polls of each of its waits (a poll = one device-scope load of the group's flags + its round trip)."""
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
lib.seqrec_debug_cluster_spins.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.seqrec_debug_cluster_spins.restype = None
gen = Sy.SyntheticSessions(100000, seed=1234)
flat, starts = gen.generate(512 * 4)
st = torch.cuda.current_stream().cuda_stream
U = (torch.randn(H, G * H, device="cuda") * (0.5 / np.sqrt(H))).contiguous()
up = torch.empty(int(lib.seqrec_rnn_upack_floats(2, H)), device="cuda")
call("seqrec_rnn_pack_u_stepwise", 2, H, ptr(U), ptr(up), st)
buf = (ctypes.c_ulonglong * 8)()
for direction in ("fwd", "bwd"):
    lib.seqrec_debug_cluster_spins(buf, 1)
    for b in range(4):
        rb = Bt.pack_flat(flat, starts, np.arange(b * 512, (b + 1) * 512))
        n = rb.n_tok
        XW = torch.randn(n, G * H, device="cuda") * 0.3
        Hout = torch.zeros(n, H, device="cuda"); gates = torch.zeros(n, G * H, device="cuda"); aux = torch.zeros(n, H, device="cuda")
        dH = torch.randn(n, H, device="cuda") * 0.1; dPre = torch.zeros(n, G * H, device="cuda"); ws = torch.zeros(2 * n * H, device="cuda")
        so = rb.step_off
        call("seqrec_rnn_fwd_stepwise", 2, 0, H, H, rb.T, rb.B, None, so.ctypes.data, ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up), None, 0, st)
        torch.cuda.synchronize()
        if direction == "fwd":
            lib.seqrec_debug_cluster_spins(buf, 1)
        for rep in range(3):
            if direction == "fwd":
                call("seqrec_rnn_fwd_stepwise", 2, 0, H, H, rb.T, rb.B, None, so.ctypes.data, ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up), None, 0, st)
            else:
                call("seqrec_rnn_bwd_stepwise", 2, 0, H, H, rb.T, rb.B, None, so.ctypes.data, n, ptr(dH), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(up), ptr(ws), None, 0, st)
            torch.cuda.synchronize()
        if direction == "fwd":
            lib.seqrec_debug_cluster_spins(buf, 0)
            s = np.array(buf[:], dtype=np.float64)
            print("fwd batch %d: %d waits, polls per wait 1..7+: %s  mean %.2f" % (b, s[0], np.round(s[1:] / s[0], 3), (s[1:] * np.arange(1, 8)).sum() / s[0]))
    if direction == "bwd":
        lib.seqrec_debug_cluster_spins(buf, 0)
        s = np.array(buf[:], dtype=np.float64)
        print("fwd x4 + bwd x12: %d waits, polls per wait 1..7+: %s  mean %.2f" % (s[0], np.round(s[1:] / s[0], 3), (s[1:] * np.arange(1, 8)).sum() / s[0]))

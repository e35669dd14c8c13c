#!/usr/bin/env python3
"""Where does a scan step launch spend its time?  (developer tool, GPU box; DIAGNOSTIC build only)

    cd seq-recommendations_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics \
        -I../../include -DSEQREC_STAMP gemm.hip ops.hip rnn.hip rnn_step.hip merge.hip -o ../../tools/bin/libseqrec_stamp.so
    SEQREC_LIB=$PWD/tools/bin/libseqrec_stamp.so SEQREC_SCAN_WIDE_RB=0 python tools/stamp_probe.py

One workgroup per launch (row block 0, column block 1, wave 0) stamps s_memtime at: 0 entry, 1 all loads issued,
2 every load returned (the diagnostic build waits vmcnt(0) there; the product build does not), 5 operand ready (BPTT:
d computed), 3 tile product + LDS reduce done, 4 stores drained; 6/7 = s_memrealtime (100 MHz) at entry/exit.
Shares, not lengths: the stamps' fences forbid overlaps the real kernel has."""
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
lib.seqrec_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
gen = Sy.SyntheticSessions(100000, seed=1234)
flat, starts = gen.generate(512 * 4)
st = torch.cuda.current_stream().cuda_stream
U = (torch.randn(H, G * H, device="cuda") * (0.5 / np.sqrt(H))).contiguous()
up = torch.empty(int(lib.seqrec_rnn_upack_floats(2, H)), device="cuda")
call("seqrec_rnn_pack_u_stepwise", 2, H, ptr(U), ptr(up), st)
big = torch.empty(64 << 20, device="cuda")      # 256 MB written between forward and backward: cold stash, like a real step
acc = {}
for b in range(4):
    rb = Bt.pack_flat(flat, starts, np.arange(b * 512, (b + 1) * 512))
    n = rb.n_tok
    XW = torch.randn(n, G * H, device="cuda") * 0.3
    Hout = torch.zeros(n, H, device="cuda"); gates = torch.zeros(n, G * H, device="cuda"); aux = torch.zeros(n, H, device="cuda")
    dH = torch.randn(n, H, device="cuda") * 0.1; dPre = torch.zeros(n, G * H, device="cuda"); ws = torch.zeros(2 * n * H, device="cuda")
    so = rb.step_off
    buf = (ctypes.c_ulonglong * (256 * 8))()
    for direction in ("fwd", "bwd"):
        big.normal_()                                 # evict L2 / Infinity Cache
        torch.cuda.synchronize()
        lib.seqrec_debug_stamps(buf, 1)
        if direction == "fwd":
            call("seqrec_rnn_fwd_stepwise", 2, 0, H, H, rb.T, rb.B, None, so.ctypes.data, ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up), None, 0, st)
        else:
            call("seqrec_rnn_bwd_stepwise", 2, 0, H, H, rb.T, rb.B, None, so.ctypes.data, n, ptr(dH), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(up), ptr(ws), None, 0, st)
        torch.cuda.synchronize()
        lib.seqrec_debug_stamps(buf, 0)
        s = np.array(buf[:], dtype=np.uint64).reshape(256, 8).astype(np.int64)
        nl = 2 * rb.T - 1
        for tag in range(1, min(nl, 256)):
            row = s[tag]
            if row[0] == 0 or row[4] == 0:
                continue
            t = (tag + 1) // 2 if direction == "fwd" else rb.T - 1 - (tag // 2)
            phase = (tag - 1) % 2 if direction == "fwd" else tag % 2
            bt = int(so[t + 1] - so[t])
            width = "narrow" if bt <= 128 else "wide"
            key = (direction, phase, width)
            p5 = row[5] if row[5] else row[2]
            seg = [row[1] - row[0], row[2] - row[1], p5 - row[2], row[3] - p5, row[4] - row[3], row[4] - row[0], (row[7] - row[6]) * 10.0]
            acc.setdefault(key, []).append(seg)
print("segment means in shader cycles: issue loads | wait all loads | operand (d) | product+reduce | epilogue+drain | total | total ns (100 MHz clock)")
for key in sorted(acc):
    a = np.array(acc[key], dtype=np.float64)
    print(key, "n=%d" % len(a), " ".join("%8.0f" % x for x in a.mean(0)))

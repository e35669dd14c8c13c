#!/usr/bin/env python3
"""How do CU-mask bits map to XCDs, and does a GEMM on a masked stream disturb a cluster scan?  (developer tool, GPU box)"""
import ctypes, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
ptr, call = L.ptr, L.call
lib = L.load()
import subprocess
so = os.path.join(ROOT, "gpurun_out", "libcumask_probe.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "cumask_probe.hip"), "-o", so])
hl = ctypes.CDLL(so)
hl.seqrec_stream_create_masked.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
hl.seqrec_debug_xcc_count.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    out = ctypes.c_void_p()
    L.check(hl.seqrec_stream_create_masked(arr, len(words), ctypes.byref(out)), "create_masked")
    return torch.cuda.ExternalStream(out.value), out.value

def xcc_hist(stream_ptr, n=4096):
    c = torch.zeros(16, dtype=torch.int32, device="cuda")
    L.check(hl.seqrec_debug_xcc_count(n, ptr(c), stream_ptr), "xcc_count")
    torch.cuda.synchronize()
    return c[:8].tolist()

print("CUs:", torch.cuda.get_device_properties(0).multi_processor_count)
print("unmasked:", xcc_hist(torch.cuda.current_stream().cuda_stream))
for name, words in [("word0 only", [0xFFFFFFFF] + [0] * 7), ("word1 only", [0, 0xFFFFFFFF] + [0] * 6), ("even bits", [0x55555555] * 8),
                    ("bits = 0 mod 8", [0x01010101] * 8), ("bits 2..7 mod 8", [0xFCFCFCFC] * 8), ("words 2..7", [0, 0] + [0xFFFFFFFF] * 6)]:
    s, p = masked_stream(words)
    print("%-18s" % name, xcc_hist(p))

# ---- scan tail beside a GEMM on a masked stream
H, G = 256, 3
gen = Sy.SyntheticSessions(100000, seed=1234)
flat, starts = gen.generate(512)
rb = Bt.pack_flat(flat, starts, np.arange(512))
n = rb.n_tok
st = torch.cuda.current_stream().cuda_stream
U = (torch.randn(H, G * H, device="cuda") * (0.5 / np.sqrt(H))).contiguous()
up = torch.empty(int(lib.seqrec_rnn_upack_floats(2, H)), device="cuda")
call("seqrec_rnn_pack_u_stepwise", 2, H, ptr(U), ptr(up), st)
XW = torch.randn(n, G * H, device="cuda") * 0.3
Hout = torch.zeros(n, H, device="cuda"); gates = torch.zeros(n, G * H, device="cuda"); aux = torch.zeros(n, H, device="cuda")
A = torch.randn(2560, 256, device="cuda"); Bm = torch.randn(2000, 256, device="cuda"); Cm = torch.empty(2560, 2000, device="cuda")
def scan():
    call("seqrec_rnn_fwd_stepwise", 2, 0, H, H, rb.T, rb.B, None, rb.step_off.ctypes.data, ptr(XW), ptr(Hout), ptr(gates), ptr(aux), ptr(up), None, 0, st)
def gemm(sp):
    call("seqrec_gemm_f32", 1, 1, 2560, 2000, 256, ptr(A), 256, ptr(Bm), 256, ptr(Cm), 2000, None, 0, 1, None, sp)
def timed(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("T", rb.T, "tokens", n)
print("scan alone           %.1f us" % timed(scan))
print("gemm alone (main)    %.1f us" % timed(lambda: gemm(st)))
for name, words in [("all CUs", [0xFFFFFFFF] * 8), ("bits 2..7 mod 8", [0xFCFCFCFC] * 8), ("words 2..7", [0, 0] + [0xFFFFFFFF] * 6), ("bits 4..7 mod 8", [0xF0F0F0F0] * 8)]:
    s, p = masked_stream(words)
    def g_alone():
        gemm(p)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    gemm(p); torch.cuda.synchronize()
    t0.record(s)
    for _ in range(20): gemm(p)
    t1.record(s); torch.cuda.synchronize()
    ga = t0.elapsed_time(t1) / 20 * 1e3
    # both: scan on main, 3 GEMMs on the side stream at the same time
    def both():
        ev = torch.cuda.Event(); ev.record()
        s.wait_event(ev)
        scan()
        for _ in range(3): gemm(p)
        ev2 = torch.cuda.Event(); ev2.record(s)
        torch.cuda.current_stream().wait_event(ev2)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    both(); torch.cuda.synchronize()
    tot = sc = 0.0
    for _ in range(10):
        ev = torch.cuda.Event(); ev.record(); s.wait_event(ev)
        e0.record(); scan(); e1.record()
        for _ in range(3): gemm(p)
        evj = torch.cuda.Event(); evj.record(s); torch.cuda.current_stream().wait_event(evj)
        e2.record(); torch.cuda.synchronize()
        sc += e0.elapsed_time(e1); tot += e0.elapsed_time(e2)
    print("%-18s gemm alone on it %.1f us | scan beside 3 gemms %.1f us, scan + 3 gemms together %.1f us" % (name, ga, sc / 10 * 1e3, tot / 10 * 1e3))

#!/usr/bin/env python3
"""Upper bound of the mixed-layout grouped launch: how much do two INDEPENDENT GEMM launches of the c3 backward gain
when they may overlap freely?  (developer tool, GPU box)  Pair B = dX (layout k-contiguous / k-contiguous, 3 split-K slabs) beside the grouped
weight-gradient launch (dU_zr, dU_h, dW, db, dEneg; row-contiguous layouts, gathered A); pair A = dH beside dEneg on its own.
  serial     both on one stream, N times
  free       each on its own stream, N launches enqueued back to back on both, NO event between them (a merged launch cannot do better than
             this: same workgroups, same dispatcher, no barrier between the two products)
The training step needs a fork and a join around such a pair (measured in round 3: loses 1.5 %); a merged kernel would not."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
ptr, call = L.ptr, L.call
lib = L.load()
dev = "cuda"
n, H, G, K, D = 2560, 256, 3, 2000, 256
GH = G * H
rng = torch.Generator(device=dev); rng.manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=rng) * 0.1
dPre, W, Hout, aux, E, ln = R(n, GH), R(D, GH), R(n, H), R(n, H), R(50000, D), R(n, K)
Eneg = R(K, H)
ids = torch.randint(0, 50000, (n,), dtype=torch.int32, device=dev)
prev = torch.arange(n, dtype=torch.int32, device=dev) - 97
ones = torch.ones(n * 4, device=dev)
dU, dW, db = torch.empty(H, GH, device=dev), torch.empty(D, GH, device=dev), torch.empty(GH, device=dev)
ws_w = torch.empty(8 * (H * GH + D * GH + GH + K * H), device=dev)
ws_x = torch.empty(4 * n * D, device=dev)
ws_h = torch.empty(4 * n * H, device=dev); dH = torch.empty(n, H, device=dev)
ws_e = torch.empty(4 * K * H, device=dev)
descs = L.gemm_descs([(H, 2 * H, n, Hout, H, dPre, GH, dU, GH, prev), (H, H, n, aux, H, dPre[:, 2 * H:], GH, dU[:, 2 * H:], GH),
                      (D, GH, n, E, D, dPre, GH, dW, GH, ids), (1, GH, n, ones, 4, dPre, GH, db, GH), (K, H, n, ln, K, Hout, H, Hout, H)])
ns = ctypes.c_int(0)
def wgrad(s):
    call("seqrec_gemm_f32_grouped_slabs", 5, 0, 0, descs, 4, ptr(ws_w), ctypes.addressof(ns), s)
def dx(s):
    call("seqrec_gemm_f32_slabs", 1, 1, n, D, GH, ptr(dPre), GH, ptr(W), GH, 3, ptr(ws_x), ctypes.addressof(ns), s)
def dh(s):
    call("seqrec_gemm_f32", 1, 0, n, H, K, ptr(ln), K, ptr(Eneg), H, ptr(dH), H, None, 0, 3, ptr(ws_h), s)
def deneg(s):
    call("seqrec_gemm_f32_slabs", 0, 0, K, H, n, ptr(ln), K, ptr(Hout), H, 4, ptr(ws_e), ctypes.addressof(ns), s)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
def timed(fa, fb, free, N=200):
    for _ in range(5):
        fa(main.cuda_stream); fb(main.cuda_stream)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    sb = side if free else main
    e0.record(main)
    if free:
        side.wait_event(e0)
    for _ in range(N):
        fa(main.cuda_stream); fb(sb.cuda_stream)
    e1.record(main)
    if free:
        e2.record(side); torch.cuda.synchronize()
        return max(e0.elapsed_time(e1), e0.elapsed_time(e2)) / N * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3
def alone(f, N=200):
    for _ in range(5): f(main.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): f(main.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3
print("c3 shapes, n = %d tokens; us per launch / pair" % n)
print("alone: wgrad+dEneg %.1f  dX %.1f  dH %.1f  dEneg %.1f" % (alone(wgrad), alone(dx), alone(dh), alone(deneg)))
print("pair B (dX | wgrad+dEneg): serial %.1f  free-running on two streams %.1f" % (timed(wgrad, dx, False), timed(wgrad, dx, True)))
print("pair A (dH | dEneg):       serial %.1f  free-running on two streams %.1f" % (timed(dh, deneg, False), timed(dh, deneg, True)))

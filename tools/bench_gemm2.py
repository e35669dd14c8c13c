#!/usr/bin/env python3
"""v1 vs v2 (LDS-DMA) GEMM at the c3 shapes, every v2 tile, split-K sweep (developer tool, GPU box).

    SEQREC_GEMM_V2=0 python tools/bench_gemm2.py   # v1 only (the env switch is read once per process)
    python tools/bench_gemm2.py                    # v2: tiles 1 (64x64), 2 (128x64), 3 (128x128)
"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
ptr, call = L.ptr, L.call
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
n = int(os.environ.get("NTOK", 2560))
# name, M, N, K, akc, bkc, split-K candidates
shapes = [("logits", n, 2000, 256, 1, 1, (1,)), ("dH", n, 256, 2000, 1, 0, (2, 3, 4, 5, 6, 8)), ("dEneg", 2000, 256, n, 0, 0, (2, 3, 4, 5, 8, 10)),
          ("xw", n, 768, 256, 1, 0, (1,)), ("dX", n, 256, 768, 1, 1, (1, 2, 3, 4)), ("dW", 256, 768, n, 0, 0, (4, 5, 8, 10, 16)),
          ("sat-logits", 25088, 2000, 256, 1, 1, (1,)), ("4096^3", 4096, 4096, 4096, 1, 0, (1,))]
v2 = not (os.environ.get("SEQREC_GEMM_V2") == "0")
only = os.environ.get("ONLY")
if only:
    shapes = [x for x in shapes if x[0] in only.split(",")]
tiles = tuple(int(t) for t in os.environ.get("TILES", "1,2,3").split(","))
check = not os.environ.get("SEQREC_GEMM_ABLATE")
flush = torch.empty(96 << 20, device="cuda")
for (name, M, N, K, akc, bkc, sks) in shapes:
    A = torch.randn((M, K) if akc else (K, M), device="cuda")
    B = torch.randn((N, K) if bkc else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    ref = (A if akc else A.T).double() @ (B.T if bkc else B).double()
    for tile in (tiles if v2 else (0,)):
        lib.seqrec_debug_gemm_tile(tile, 0)
        for sk in sks:
            ws = torch.empty(max(1, sk * M * N), device="cuda") if sk > 1 else None
            def f(): call("seqrec_gemm_f32", akc, bkc, M, N, K, ptr(A), K if akc else M, ptr(B), K if bkc else N, ptr(C), N, None, 0, sk, ptr(ws), st)
            f(); torch.cuda.synchronize()
            err = float((C.double() - ref).abs().max()) if check else -1.0
            reps = 20
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for e0, e1 in ev:
                e0.record(); f(); e1.record()
            torch.cuda.synchronize()
            us = sorted(e0.elapsed_time(e1) * 1000 for e0, e1 in ev)
            med = us[len(us) // 2]
            print("%-10s M=%5d N=%5d K=%5d %s tile=%d splitk=%2d : med %7.1f us  min %7.1f  %6.1f TFLOP/s  maxerr %.2e" % (
                name, M, N, K, "v2" if v2 else "v1", tile, sk, med, us[0], 2.0 * M * N * K / med / 1e6, err), flush=True)
lib.seqrec_debug_gemm_tile(0, 0)

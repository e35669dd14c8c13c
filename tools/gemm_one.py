#!/usr/bin/env python3
"""Run ONE GEMM configuration a few times (target of rocprofv3 passes; GPU box):  gemm_one.py <shape> <tile> <splitk> [reps]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
ptr, call = L.ptr, L.call
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
n = 2560
shapes = {"logits": (n, 2000, 256, 1, 1), "dH": (n, 256, 2000, 1, 0), "dEneg": (2000, 256, n, 0, 0), "sat": (25088, 2000, 256, 1, 1),
          "4096": (4096, 4096, 4096, 1, 0), "xw": (n, 768, 256, 1, 0), "dW": (256, 768, n, 0, 0)}
name, tile, sk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
M, N, K, akc, bkc = shapes[name]
A = torch.randn((M, K) if akc else (K, M), device="cuda")
B = torch.randn((N, K) if bkc else (K, N), device="cuda")
C = torch.empty(M, N, device="cuda")
ws = torch.empty(max(1, sk * M * N), device="cuda") if sk > 1 else None
lib.seqrec_debug_gemm_tile(tile, 0)
for _ in range(reps):
    call("seqrec_gemm_f32", akc, bkc, M, N, K, ptr(A), K if akc else M, ptr(B), K if bkc else N, ptr(C), N, None, 0, sk, ptr(ws), st)
    torch.cuda.synchronize()
print("ok", name, tile, sk)

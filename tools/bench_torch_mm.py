#!/usr/bin/env python3
"""Reference point only: what the vendor library (torch.mm -> hipBLASLt/rocBLAS, fp32) reaches on the
step's GEMM shapes, beside seqrec_gemm_f32 (developer tool, GPU box)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
ptr, call = L.ptr, L.call
torch.backends.cuda.matmul.allow_tf32 = False
st = torch.cuda.current_stream().cuda_stream
shapes = [("logits", 2546, 2000, 256, "nt"), ("xw", 2546, 768, 256, "nn"), ("dH", 2546, 256, 2000, "nn"), ("dEneg", 2000, 256, 2546, "tn"),
          ("dX", 2546, 256, 768, "nt"), ("dW", 256, 768, 2546, "tn"), ("big", 4096, 4096, 4096, "nn")]
def timeit(f, n=50):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for name, M, N, K, lay in shapes:
    A = torch.randn((M, K) if lay[0] == "n" else (K, M), device="cuda")
    B = torch.randn((K, N) if lay[1] == "n" else (N, K), device="cuda")
    C = torch.empty(M, N, device="cuda")
    a = A if lay[0] == "n" else A.t()
    b = B if lay[1] == "n" else B.t()
    t_lib = timeit(lambda: torch.mm(a, b, out=C))
    akc, bkc = int(lay[0] == "n"), int(lay[1] == "t")
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    sk = int(max(1, min(32, -(-1024 // tiles), K // 128)))
    ws = torch.empty(max(1, sk * M * N), device="cuda")
    C2 = torch.empty(M, N, device="cuda")
    t_own = timeit(lambda: call("seqrec_gemm_f32", akc, bkc, M, N, K, ptr(A), K if akc else M, ptr(B), K if bkc else N, ptr(C2), N, None, 0, sk, ptr(ws), st))
    err = (C - C2).abs().max().item() / max(C.abs().max().item(), 1e-9)
    print("%-6s M=%5d N=%5d K=%5d  torch.mm %7.1f us (%5.1f TF)   seqrec_gemm_f32 %7.1f us (%5.1f TF, splitk %d)   max rel diff %.1e" % (
        name, M, N, K, t_lib, 2e-6 * M * N * K / t_lib, t_own, 2e-6 * M * N * K / t_own, sk, err))

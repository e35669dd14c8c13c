#!/usr/bin/env python3
"""Per-workgroup timeline of one LDS-DMA GEMM launch (diagnostic build, -DSEQREC_GEMM_ABLATE; GPU box).

    SEQREC_LIB=$PWD/tools/bin/libseqrec_ablate.so [SEQREC_GEMM_V2_GRID=..] python tools/gemm_stamps.py logits 1

Every workgroup stamps s_memrealtime (100 MHz) at: 0 entry, 1 after setup + the two prologue DMA issues, 2 first tile
landed + barrier, 3 end of its first item's K loop, 4 after that item's C stores were issued, 5 after its last item,
6 stores drained.  Printed: kernel span and the distribution (min / median / max over workgroups) of each point relative
to the first workgroup's entry, in microseconds."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("seq-recommendations_amd._lib")
ptr, call = L.ptr, L.call
lib = L.load()
lib.seqrec_debug_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.seqrec_debug_gemm_stamps.restype = None
lib.seqrec_debug_gemm_phases.argtypes = [ctypes.c_void_p]
lib.seqrec_debug_gemm_phases.restype = None
st = torch.cuda.current_stream().cuda_stream
n = 2560
shapes = {"logits": (n, 2000, 256, 1, 1, 1), "dH": (n, 256, 2000, 1, 0, 4), "dEneg": (2000, 256, n, 0, 0, 8), "sat": (25088, 2000, 256, 1, 1, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "logits"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 1
M, N, K, akc, bkc, sk = shapes[name]
A = torch.randn((M, K) if akc else (K, M), device="cuda")
B = torch.randn((N, K) if bkc else (K, N), device="cuda")
C = torch.empty(M, N, device="cuda")
ws = torch.empty(max(1, sk * M * N), device="cuda") if sk > 1 else None
lib.seqrec_debug_gemm_tile(tile, 0)
def f(): call("seqrec_gemm_f32", akc, bkc, M, N, K, ptr(A), K if akc else M, ptr(B), K if bkc else N, ptr(C), N, None, 0, sk, ptr(ws), st)
for _ in range(3): f()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4096 * 8))()
for rep in range(2):
    lib.seqrec_debug_gemm_stamps(None, 1)
    torch.cuda.synchronize()
    f()
    torch.cuda.synchronize()
    lib.seqrec_debug_gemm_stamps(buf, 0)
    s = np.array(buf[:], dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    print("%s tile %d mask %s grid %s: %d workgroups, span %.2f us" % (name, tile, os.environ.get("SEQREC_GEMM_ABLATE", "0"),
          os.environ.get("SEQREC_GEMM_V2_GRID", "512"), len(s), (s[:, 1:7].max() - t0) / 100.0))
    for i, lab in enumerate(["entry", "setup+issue", "first tile landed", "K loop of item 0 done", "stores of item 0 issued", "last item done", "drained"]):
        v = (s[:, i] - t0) / 100.0
        v = v[s[:, i] > 0]
        if len(v): print("   %-26s min %7.2f  p10 %7.2f  med %7.2f  p90 %7.2f  max %7.2f" % (lab, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
    d = (s[:, 6] - s[:, 0]) / 100.0
    print("   workgroup lifetime         min %7.2f  med %7.2f  max %7.2f" % (d.min(), np.median(d), d.max()))
    hw = s[:, 7] & 0xFFFFFFFF
    xcc = (s[:, 7] >> 32) & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 50 + cu
    # workgroups co-resident on a CU at the moment the median workgroup is half-way through its life
    mid = np.median((s[:, 0] + s[:, 6]) // 2)
    live = (s[:, 0] <= mid) & (s[:, 6] >= mid)
    u, c = np.unique(key[live], return_counts=True)
    print("   at t = %.2f us: %d workgroups live on %d distinct CUs; workgroups per CU histogram %s; XCCs %s" % (
        (mid - t0) / 100.0, live.sum(), len(u), dict(zip(*np.unique(c, return_counts=True))), sorted(set(xcc.tolist()))))
    u2, c2 = np.unique(key, return_counts=True)
    print("   whole launch: %d distinct CUs, workgroups per CU over the launch %s" % (len(u2), dict(zip(*np.unique(c2, return_counts=True)))))
lib.seqrec_debug_gemm_tile(0, 0)

"""Keeps the GPU busy from a separate process for argv[1] seconds (tests/test_gpu_multiprocess.py)."""
import sys
import time

import torch

x = torch.randn(4096, 4096, device="cuda")
t_end = time.time() + float(sys.argv[1])
while time.time() < t_end:
    for _ in range(20):
        y = x @ x
    torch.cuda.synchronize()

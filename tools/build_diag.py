#!/usr/bin/env python3
"""Diagnostic builds of the library (developer tool): python tools/build_diag.py stamp|spins|tunables -> tools/diag/libseqrec_cl<kind>.so
(tools/diag/ is git-ignored but travels to the GPU box; select with SEQREC_LIB=...)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kind = sys.argv[1] if len(sys.argv) > 1 else "stamp"
flag = {"stamp": "-DSEQREC_CLUSTER_STAMP", "spins": "-DSEQREC_CLUSTER_SPINS", "tunables": "-DSEQREC_TUNABLES"}[kind]      # tunables: the env switches of the A/B scripts
src = ["gemm.hip", "ops.hip", "rnn.hip", "rnn_step.hip", "rnn_cluster.hip", "rnn_cluster2.hip", "merge.hip", "exchange.hip", "route.hip", "step.hip"]
os.makedirs(os.path.join(ROOT, "tools", "diag"), exist_ok=True)
out = os.path.join(ROOT, "tools", "diag", "libseqrec_cl%s.so" % kind)
cs = os.path.join(ROOT, "seq-recommendations_amd", "csrc")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", flag] + [os.path.join(cs, s) for s in src] + ["-o", out]
subprocess.check_call(cmd)
print(out)

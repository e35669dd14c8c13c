"""Build libseqrec_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.

Every source is compiled to an object under csrc/_obj/ (kept out of git; only sources whose text or
headers changed are recompiled, in parallel), then linked into the one shared object the ctypes binding
loads.  __graft_entry__.build() calls build(): an up-to-date library is reused, not rebuilt."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libseqrec_hip.so")
SOURCES = ["gemm.hip", "ops.hip", "rnn.hip", "rnn_step.hip", "rnn_cluster.hip", "rnn_cluster2.hip", "merge.hip", "exchange.hip", "route.hip", "step.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(HERE, "..", "include", "seqrec_hip.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def _stale():
    return _newer(LIB, [os.path.join(CSRC, s) for s in SOURCES] + _headers())


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared object.  Returns the path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libseqrec_hip.so")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()

    def compile_one(src):
        s, o = os.path.join(CSRC, src), os.path.join(OBJ, src + ".o")
        if not force and not _newer(o, [s] + hdrs):
            return None
        cmd = [hipcc] + CFLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, r.stdout, r.stderr))
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(OBJ, s + ".o") for s in SOURCES] + ["-o", LIB + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

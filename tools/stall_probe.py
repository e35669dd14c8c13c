#!/usr/bin/env python3
"""Where do the slow early steps of a fresh process come from?  (developer tool, GPU box)

Runs the c3 fresh-batch training loop of bench.py from engine creation and prints, per block of 10 steps:
synced wall ms/step, host enqueue ms/step, torch allocator reserved MB / segment count.  Under
`rocprofv3 --hip-trace --kernel-trace` (program directly after `--`) the same run gives the per-call
HIP timeline that VERDICT r1 item 8 asked for.  argv[1] = number of blocks (default 16); argv[2] = 'noreserve'
skips Engine.reserve() to show the allocator growth it removes."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
E = importlib.import_module("seq-recommendations_amd.engine")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
Sm = importlib.import_module("seq-recommendations_amd.sampling")
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cd = bench.CONFIGS["c3"]; V = cd["V"]
cfg = E.NetConfig(cell="gru", act="relu", H=256, V_in=V, V_out=V, input="embed", D=256, output="sampled", K=2000, logq=True, seed=1)
eng = E.Engine(cfg)
bench.init_params_device(eng, cd, 1)
gen = Sy.SyntheticSessions(V, seed=1234)
p = Sm.log_uniform_probs(V, gen.proposal_rank()); th, al = Sm.build_alias_table(p); eng.set_sampler(th, al, np.log(p).astype(np.float32))
flat, starts = gen.generate(40_000)
ds = eng.put_dataset(flat, starts)
if not (len(sys.argv) > 2 and sys.argv[2] == "noreserve"):
    eng.reserve(512 * 49)
stream = bench.BatchStream(0, 40_000, 512, 1)
torch.cuda.synchronize()
import gc
_gc = {"t0": 0.0, "log": []}
def _cb(phase, info):
    if phase == "start":
        _gc["t0"] = time.perf_counter()
    else:
        _gc["log"].append((info["generation"], (time.perf_counter() - _gc["t0"]) * 1e3, info["collected"]))
gc.callbacks.append(_cb)
if "freeze" in sys.argv[2:]:
    gc.collect(); gc.freeze()
step = 0
print("block  wall_ms/step  host_ms/step  reserved_MB  segments  max_step_host_ms")
for blk in range(blocks):
    t0 = time.perf_counter(); host = 0.0; worst = 0.0
    for i in range(10):
        h0 = time.perf_counter()
        d = eng.upload_device(ds, stream.sel(step))
        eng.train_step(d, step=step); step += 1
        h = time.perf_counter() - h0
        host += h; worst = max(worst, h)
    torch.cuda.synchronize()
    ms = torch.cuda.memory_stats()
    print("%5d  %12.3f  %12.3f  %11.1f  %8d  %16.3f" % (blk, (time.perf_counter() - t0) / 10 * 1e3, host / 10 * 1e3,
          ms["reserved_bytes.all.current"] / 2**20, ms["segment.all.current"], worst * 1e3),
          " gc(gen, ms):", [(g, round(t, 2)) for g, t, _ in _gc["log"] if t > 0.5], flush=True)
    _gc["log"].clear()

#!/usr/bin/env python3
"""How long does the step time take to settle after the engine is created? (developer tool, GPU box)"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
E = importlib.import_module("seq-recommendations_amd.engine")
Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
Sm = importlib.import_module("seq-recommendations_amd.sampling")
cd = bench.CONFIGS["c3"]; V = cd["V"]
cfg = E.NetConfig(cell="gru", act="relu", H=256, V_in=V, V_out=V, input="embed", D=256, output="sampled", K=2000, logq=True, seed=1)
eng = E.Engine(cfg)
bench.init_params_device(eng, cd, 1)
gen = Sy.SyntheticSessions(V, seed=1234)
p = Sm.log_uniform_probs(V, gen.proposal_rank()); th, al = Sm.build_alias_table(p); eng.set_sampler(th, al, np.log(p).astype(np.float32))
flat, starts = gen.generate(512 * 64)
bs = [eng.upload(Bt.pack_flat(flat, starts, np.arange(i * 512, (i + 1) * 512))) for i in range(64)]
if len(sys.argv) > 1 and sys.argv[1] == "spin":      # clock spin-up with neutral work first
    a = torch.randn(4096, 4096, device="cuda"); t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5: b = a @ a
    torch.cuda.synchronize()
torch.cuda.synchronize()
step = 0
out = []
for blk in range(25):
    t0 = time.perf_counter()
    for i in range(20):
        eng.train_step(bs[step % 64], step=step); step += 1
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 20 * 1e3)
print(" ".join("%.3f" % x for x in out))

#!/usr/bin/env python3
"""Where does ShardedEngine.prepare spend its host time?  (developer tool, GPU box) -- line-level timers around the phases of
RowExchange.plan_unified / ShardedEngine._prepare_unified for windows of 32 c3 batches on a 1-rank RCCL group."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29593")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
E = importlib.import_module("seq-recommendations_amd.engine")
Dm = importlib.import_module("seq-recommendations_amd.distributed")
Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
Sm = importlib.import_module("seq-recommendations_amd.sampling")
cd = bench.CONFIGS["c3"]; V = cd["V"]
cfg = E.NetConfig(cell="gru", act="relu", H=256, V_in=V, V_out=V, input="embed", D=256, output="sampled", K=2000, logq=True, seed=1)
eng = Dm.ShardedEngine(cfg, "cuda:0", dist)
bench.init_params_device(eng, cd, 1)
gen = Sy.SyntheticSessions(V, seed=1234)
p = Sm.log_uniform_probs(V, gen.proposal_rank()); th, al = Sm.build_alias_table(p); eng.set_sampler(th, al, np.log(p).astype(np.float32))
flat, starts = gen.generate(60_000)
stream = bench.BatchStream(0, 60_000, 512, 1)
import cProfile, pstats
step = 0
def window(w=32):
    global step
    t0 = time.perf_counter()
    rbs = [Bt.pack_flat(flat, starts, stream.sel(step + j)) for j in range(w)]
    t1 = time.perf_counter()
    ds = eng.prepare(rbs)
    t2 = time.perf_counter()
    for d in ds:
        eng.train_step(d, step=step); step += 1
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for _ in range(3):
    window()
torch.cuda.synchronize()
acc = np.zeros(3)
for _ in range(6):
    acc += np.array(window())
torch.cuda.synchronize()
print("per window of 32 (ms): pack %.2f  prepare %.2f  32 train_step enqueues %.2f" % tuple(acc / 6 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    rbs = [Bt.pack_flat(flat, starts, stream.sel(step + j)) for j in range(32)]
    ds = eng.prepare(rbs)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)

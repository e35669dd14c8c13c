#!/usr/bin/env python3
"""bench.py -- sessions/sec of the RNN next-item training step on MI355X.

Workload (BASELINE.json `metric`, configs[2] = "c3"): synthetic MSNBC-shaped sessions,
|items| = 1M, seq_len <= 50, batch 512 sessions per GPU, GRU hidden = 256 (Keras-2.0 GRU
equations, hard_sigmoid gates, relu), embedding width 256, sampled softmax with K = 2000 shared
log-uniform negatives, masked-token-mean CE, BPTT, global-norm clip 1.0 + Adagrad(lr 0.01,
eps 1e-8).  A "step" is one full training step on one batch whose index arrays are already
resident in HBM.  fp32 end to end (exact-fp32 MFMA).

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline     -- the dominant kernel of the step (by summed device time, HIP events on the launch
                  stream over a profiled pass of the same batches): algorithmic flops (or bytes)
                  per launch / mean launch duration against the gfx950 peak that bounds it;
  kernels      -- the same for every kernel class of the step;
  cpu_baseline -- the oracle (numpy fp32 restatement of the Keras/Theano path) timed on this box's
                  host cores on a bounded sample of the SAME workload (rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, Peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0             # HBM3E spec (6.29 TB/s measured-achievable)

CONFIGS = {
    # name: V, H, D, K, cell
    "c3": dict(V=1_000_000, H=256, D=256, K=2000, cell="gru",
               desc="c3: |items|=1M seq_len<=50 GRU hidden=256 embed=256 sampled-softmax K=2000 batch 512/GPU"),
    "c2": dict(V=100_000, H=128, D=128, K=1000, cell="gru",
               desc="c2: |items|=100k seq_len<=50 GRU hidden=128 embed=128 sampled-softmax K=1000 batch 512/GPU"),
    # c4's model (LSTM 512, K=4000) -- the reference's own cell type at catalogue scale
    "c4": dict(V=1_000_000, H=512, D=512, K=4000, cell="lstm",
               desc="c4: |items|=1M seq_len<=50 LSTM hidden=512 embed=512 sampled-softmax K=4000 batch 512/GPU"),
    # c5: tied input/output table at 5M items
    "c5": dict(V=5_000_000, H=256, D=256, K=2000, cell="gru", tied=True,
               desc="c5: |items|=5M seq_len<=50 GRU hidden=256 tied input/output table sampled-softmax K=2000 batch 512/GPU"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=512, help="sessions per GPU per step")
    ap.add_argument("--saturated", action="store_true", help="every session has 50 items (roofline runs)")
    ap.add_argument("--distinct-batches", type=int, default=64)
    ap.add_argument("--profile-steps", type=int, default=10)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--recall-steps", type=int, default=1500, help="extra training steps before Recall@20 (0 = skip)")
    ap.add_argument("--recall-sessions", type=int, default=2048)
    ap.add_argument("--force-sharded", action="store_true", help="use the row-sharded engine even on one GPU")
    ap.add_argument("--sharded-recall", action="store_true",
                    help="also compute Recall@20 through the sharded rank counting when N > 1 (default: only for N = 1)")
    return ap.parse_args()


def init_params_device(eng, cfg, seed):
    """E, Eout ~ U(-0.01, 0.01); W glorot-uniform; U orthogonal per gate; b = 0."""
    import torch
    g = torch.Generator(device=eng.dev)
    g.manual_seed(seed)
    c = eng.cfg
    H, D, G = c.H, c.D, eng.G
    with torch.no_grad():
        eng.P["E"].uniform_(-0.01, 0.01, generator=g)
        if "Eout" in eng.P:
            eng.P["Eout"].uniform_(-0.01, 0.01, generator=g)
        lim = float(np.sqrt(6.0 / (D + G * H)))
        eng.P["W"].uniform_(-lim, lim, generator=g)
        rs = np.random.default_rng(seed)
        U = np.concatenate([np.linalg.qr(rs.normal(size=(H, H)))[0] for _ in range(G)], axis=1)
        eng.set_param("U", U.astype(np.float32))
        if "b" in eng.P:
            eng.P["b"].zero_()


def kernel_model(cfgd, n_tok, K, B):
    """Algorithmic work per launch of every kernel class at n_tok tokens (SURVEY 8d formulas):
    name -> (bound, flops or bytes)."""
    H, D = cfgd["H"], cfgd["D"]
    G = 3 if cfgd["cell"] == "gru" else (4 if cfgd["cell"] == "lstm" else 1)
    m = {
        "seqrec_rnn_fwd": ("mfma", 2.0 * G * H * H * n_tok),
        "seqrec_rnn_bwd": ("mfma", 2.0 * G * H * H * n_tok),
        # step-wise scan: one C-ABI call = 2 launches per time step; work and time are per CALL
        "seqrec_rnn_fwd_stepwise": ("mfma", 2.0 * G * H * H * n_tok),
        "seqrec_rnn_bwd_stepwise": ("mfma", 2.0 * G * H * H * n_tok),
        "seqrec_gemm_f32[xw]": ("mfma", 2.0 * n_tok * D * G * H),
        "seqrec_gemm_f32[logits]": ("mfma", 2.0 * n_tok * K * H),
        "seqrec_gemm_f32[dH]": ("mfma", 2.0 * n_tok * K * H),
        "seqrec_gemm_f32[dEneg]": ("mfma", 2.0 * n_tok * K * H),
        "seqrec_gemm_f32[dW]": ("mfma", 2.0 * n_tok * D * G * H),
        "seqrec_gemm_f32[dX]": ("mfma", 2.0 * n_tok * D * G * H),
        "seqrec_gemm_f32[dU]": ("mfma", 2.0 * n_tok * H * G * H / (2 if G == 3 else 1)),   # GRU: two launches
        "seqrec_gemm_f32_grouped[dW+dU]": ("mfma", 2.0 * n_tok * G * H * (D + H)),          # dW and dU in one launch
        "seqrec_gather_rows[E]": ("hbm", 8.0 * D * n_tok),                                   # 4 B read + 4 B written / elt
    }
    return m


def pmc_traffic(kernel, t_mean, a):
    """HBM bytes per call of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_v3_c3_pmc_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs of
    this same command; FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads).
    Only valid for the workload those passes profiled (c3, MSNBC-shaped); None otherwise."""
    path = os.path.join(ROOT, "profiles", "r01_v3_c3_pmc_hbm_traffic.json")
    if a.config != "c3" or a.saturated or not os.path.exists(path):
        return None
    pm = json.load(open(path))
    pick = {"seqrec_rnn_fwd_stepwise": ("gru_step_fwd<4, 0, 0>", "gru_step_fwd<4, 0, 1>"),
            "seqrec_rnn_bwd_stepwise": ("gru_step_bwd<4, 0, 0>", "gru_step_bwd<4, 0, 1>")}.get(kernel)
    if not pick:
        return None
    tot = 0.0
    for i, frag in enumerate(pick):
        ent = next((v for k, v in pm.items() if frag in k), None)
        if ent is None:
            return None
        launches = t_mean - (1 if (i == 1 and "bwd" in kernel) else 0)     # bwd phase 1 is skipped at t = 0
        tot += launches * (ent["fetch_kib_x2"] + ent["write_kib"]) * 1024.0
    return {"bytes_per_call": round(tot), "source": "profiles/r01_v3_c3_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}


def pmc_mfma_util(kernel, a):
    """Counter-based matrix-pipe utilisation of the dominant kernel's launches from the committed rocprofv3 PMC
    pass (profiles/r01_v6_c3[_saturated]_pmc_mfma.json, tools/pmc_mfma.py); c3 only."""
    path = os.path.join(ROOT, "profiles", "r01_v6_c3%s_pmc_mfma.json" % ("_saturated" if a.saturated else ""))
    if a.config != "c3" or not os.path.exists(path):
        return None
    frag = {"seqrec_rnn_fwd_stepwise": "gru_step_fwd<4, 0, ", "seqrec_rnn_bwd_stepwise": "gru_step_bwd<4, 0, "}.get(kernel)
    if not frag:
        return None
    pm = json.load(open(path))
    out = {k[k.index("gru_step"):k.index(">") + 1]: v["mfma_util"] for k, v in pm.items() if frag in k}
    if not out:
        return None
    out["source"] = os.path.relpath(path, ROOT) + " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)"
    return out


def main():
    a = parse()
    # stdout carries exactly ONE line (the JSON): everything libraries print while we run (RCCL's
    # version banner, warnings) is sent to stderr by pointing fd 1 at fd 2 until the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    dist = None
    if world > 1 or a.force_sharded:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if os.environ.get("SEQREC_BENCH_BACKEND", "nccl") == "gloo-staged":
            # rehearsal of the N > 1 code path on a ONE-GPU box: every rank computes on cuda:0 and the
            # collectives are staged through the host (distributed.HostStagedDist); numbers are meaningless
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist = importlib.import_module("seq-recommendations_amd.distributed").HostStagedDist(dist)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    pkg = importlib.import_module("seq-recommendations_amd")
    pkg.require_hip()
    E = importlib.import_module("seq-recommendations_amd.engine")
    Bt = importlib.import_module("seq-recommendations_amd.batching")
    Sy = importlib.import_module("seq-recommendations_amd.synthetic")
    Sm = importlib.import_module("seq-recommendations_amd.sampling")

    cd = CONFIGS[a.config]
    V, H, D, K = cd["V"], cd["H"], cd["D"], cd["K"]
    dev = "cuda:%d" % local
    ncfg = E.NetConfig(cell=cd["cell"], act="relu", H=H, V_in=V, V_out=V, input="embed", D=D, output="sampled", K=K,
                       tied=bool(cd.get("tied", False)), use_bias=True, out_bias=False, logq=True, seed=1234)
    sharded = dist is not None
    if sharded:
        Dm = importlib.import_module("seq-recommendations_amd.distributed")
        eng = Dm.ShardedEngine(ncfg, dev, dist)
    else:
        eng = E.Engine(ncfg, dev)
    init_params_device(eng, cd, seed=1234 + rank)
    if sharded and world > 1:
        for k in ("W", "U", "b"):                       # replicated cell weights: rank 0's values everywhere
            dist.broadcast(eng.P[k], src=0)
        eng.upack_dirty = True
    gen = Sy.SyntheticSessions(V, seed=1234)
    probs = Sm.log_uniform_probs(V, gen.proposal_rank())
    th, al = Sm.build_alias_table(probs)
    logq = np.log(probs).astype(np.float32)
    if sharded:
        # shard-local proposal (rows rank, rank+R, ...) and the effective Q(v) = Q_shard(v) / R
        pl = probs[rank::world]
        pl = pl / pl.sum()
        thl, all_ = Sm.build_alias_table(pl)
        eng.set_sampler(thl, all_, (np.log(pl) - np.log(world)).astype(np.float32))
    else:
        eng.set_sampler(th, al, logq)

    # ---- batches: generated once, packed on the host, index arrays resident in HBM before timing
    nb = max(1, min(a.distinct_batches, a.steps + a.warmup))
    flat, starts = gen.generate(world * nb * a.batch + a.recall_sessions, saturated=a.saturated)
    batches = []
    for i in range(nb):
        sel = np.arange((rank * nb + i) * a.batch, (rank * nb + i + 1) * a.batch)
        batches.append(eng.upload(Bt.pack_flat(flat, starts, sel)))
    n_tok_mean = float(np.mean([b["n"] for b in batches]))
    t_max = int(max(b["T"] for b in batches))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    step = 0
    # settle pass (untimed, not part of --warmup): one step on every distinct batch so that the grow-only
    # workspaces reach their final size, every kernel variant is loaded and the clocks are up before the
    # warm-up starts -- a 0.7 ms step is otherwise measured through ~60 steps of allocator growth and
    # lazy code loading (tools/warmup_probe.py: 3.3 / 0.8 / 3.8 ms per step in the first three 20-step blocks)
    settle = max(128, nb)       # also covers a one-off runtime stall seen around steps 40-60 of a fresh process
    for i in range(settle):
        eng.train_step(batches[step % nb], lr=0.01, eps=1e-8, clipnorm=1.0, step=step)
        step += 1
    sync()
    for i in range(a.warmup):
        eng.train_step(batches[step % nb], lr=0.01, eps=1e-8, clipnorm=1.0, step=step)
        step += 1
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = eng.train_step(batches[step % nb], lr=0.01, eps=1e-8, clipnorm=1.0, step=step)
        step += 1
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / a.steps * 1e3
    sessions_per_s = a.steps * a.batch * world / dt
    tokens_per_s = a.steps * n_tok_mean * world / dt
    last_loss = float(loss.item())

    # ---- per-kernel device time (HIP events on the launch stream), same batches
    kern = {}
    roof = None
    notes = []          # a failing OPTIONAL leg (single process only) must not take the throughput line with it
    if a.profile_steps > 0:
        try:
            E.profile_start()
            for i in range(a.profile_steps):
                eng.train_step(batches[step % nb], lr=0.01, eps=1e-8, clipnorm=1.0, step=step)
                step += 1
            prof = E.profile_stop()
            model = kernel_model(cd, n_tok_mean, K, a.batch)
            tot = sum(ms for _, ms in prof.values())
            for name, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                per = ms / cnt
                ent = {"launches_per_step": cnt / a.profile_steps, "avg_us": round(per * 1e3, 2),
                       "share": round(ms / tot, 4)}
                if name in model:
                    bound, work = model[name]
                    if bound == "mfma":
                        ach = work / (per * 1e-3) / 1e12
                        ent.update(bound="mfma", achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                                   frac=round(ach / PEAK_F32_MFMA_TFLOPS, 5))
                    else:
                        ach = work / (per * 1e-3) / 1e9
                        ent.update(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                                   frac=round(ach / PEAK_HBM_GBS, 5))
                kern[name] = ent
            dom = next((k for k in kern if "bound" in kern[k]), None)
            if dom:
                e = kern[dom]
                t_mean = float(np.mean([b["T"] for b in batches]))
                roof = {"kernel": dom, "bound": e["bound"], "achieved": e["achieved"], "peak": e["peak"], "unit": e["unit"],
                        "frac": e["frac"], "traffic": pmc_traffic(dom, t_mean, a), "pmc_mfma_util": pmc_mfma_util(dom, a),
                        "avg_us": e["avg_us"],
                        "share_of_step": e["share"]}
        except Exception as e:                                   # noqa: BLE001
            if dist is not None:
                raise
            E._PROF = None
            notes.append("profile leg failed: %r" % (e,))

    # ---- Recall@20 on held-out sessions after some more training
    recall = None
    if a.recall_steps > 0 and sharded and (world == 1 or a.sharded_recall):
        for i in range(a.recall_steps):
            eng.train_step(batches[step % nb], lr=0.01, eps=1e-8, clipnorm=1.0, step=step)
            step += 1
        per = a.recall_sessions // world                       # every rank scores the same number of held-out sessions
        base = world * nb * a.batch + rank * per
        acc = torch.zeros(2, dtype=torch.float64, device=dev)
        for s in range(0, per, a.batch):
            d = eng.upload(Bt.pack_flat(flat, starts, np.arange(base + s, base + min(per, s + a.batch))))
            rk = eng.rank_counts(d)
            acc[0] += (rk < 20).sum()
            acc[1] += d["n"]
        dist.all_reduce(acc)
        recall = float(acc[0].item() / max(acc[1].item(), 1.0))
    if a.recall_steps > 0 and not sharded:
        try:
            for i in range(a.recall_steps):
                eng.train_step(batches[step % nb], lr=0.01, eps=1e-8, clipnorm=1.0, step=step)
                step += 1
            sel = np.arange(world * nb * a.batch, world * nb * a.batch + a.recall_sessions)
            hits = n = 0
            for s in range(0, len(sel), a.batch):
                d = eng.upload(Bt.pack_flat(flat, starts, sel[s:s + a.batch]))
                rk = eng.rank_counts(d)
                hits += int((rk < 20).sum().item())
                n += d["n"]
            recall = hits / max(n, 1)
        except Exception as e:                                   # noqa: BLE001
            notes.append("recall leg failed: %r" % (e,))

    # ---- CPU baseline: the oracle on a bounded sample of the same workload
    cpu = None
    if rank == 0 and world == 1 and not sharded and a.cpu_seconds > 0:
        try:
            cpu = cpu_baseline(a, cd, gen, flat, starts, th, al, logq)
        except Exception as e:                                   # noqa: BLE001
            notes.append("cpu_baseline leg failed: %r" % (e,))

    if rank == 0:
        out = {
            "metric": "sessions/sec", "value": round(sessions_per_s, 1), "unit": "sessions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cd["desc"] + (" (saturated: 50 items/session)" if a.saturated else " (MSNBC-shaped lengths)"),
                       "global_batch": a.batch * world, "seq_len": 50, "tokens_per_step_per_gpu": round(n_tok_mean, 1),
                       "t_max": t_max, "settle_steps": settle,
                       "parallelism": ("dp%d+row-sharded-tables" % world) if sharded else "single"},
            "tokens_per_s": round(tokens_per_s, 1), "final_loss": round(last_loss, 5), "recall_at_20": recall,
            "roofline": roof, "cpu_baseline": cpu, "kernels": kern,
        }
        if notes:
            out["notes"] = notes
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(a, cd, gen, flat, starts, th, al, logq):
    """Time the oracle (numpy fp32, all host cores through OpenBLAS) on the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import nn as onn
    from oracle import rng as orng
    V, H, D, K = cd["V"], cd["H"], cd["D"], cd["K"]
    G = onn.N_GATES[cd["cell"]]
    rs = np.random.default_rng(1234)
    p = {
        "E": rs.uniform(-0.01, 0.01, (V, D)).astype(np.float32),
        "Eout": rs.uniform(-0.01, 0.01, (V, H)).astype(np.float32),
        "W": rs.uniform(-1, 1, (D, G * H)).astype(np.float32) * np.float32(np.sqrt(6.0 / (D + G * H))),
        "U": np.concatenate([np.linalg.qr(rs.normal(size=(H, H)))[0] for _ in range(G)], axis=1).astype(np.float32),
        "b": np.zeros(G * H, np.float32),
    }
    acc = {k: np.zeros_like(v) for k, v in p.items()}
    net = onn.OracleNet(dict(cell=cd["cell"], act="relu", input="embed", output="sampled", tied=False,
                             use_bias=True, out_bias=False), p)

    def padded(i):
        sel = np.arange(i * a.batch, (i + 1) * a.batch)
        L = (starts[sel + 1] - starts[sel] - 1).astype(np.int64)
        T = int(L.max())
        ids = np.zeros((a.batch, T), np.int64); tgt = np.zeros((a.batch, T), np.int64); mask = np.zeros((a.batch, T), bool)
        for b, s in enumerate(sel):
            n = L[b]
            seq = flat[starts[s]:starts[s + 1]]
            ids[b, T - n:] = seq[:-1]; tgt[b, T - n:] = seq[1:]; mask[b, T - n:] = True
        return {"ids": ids, "tgt": tgt, "mask": mask}

    def one(i):
        neg = orng.sample_negatives(1234, i, K, th, al)
        net.forward(padded(i), negatives=neg, logq=logq)
        g = net.backward()
        onn.adagrad_step(p, acc, g, lr=0.01, eps=1e-8, clipnorm=1.0)

    one(0)                                   # warm-up (page in the tables)
    t0 = time.perf_counter()
    n = 0
    while True:
        one(n + 1)
        n += 1
        el = time.perf_counter() - t0
        if el >= a.cpu_seconds or n >= 200:
            break
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    return {"value": round(n * a.batch / el, 1), "unit": "sessions/s", "cores": cores, "kind": "port",
            "ms_per_step": round(el / n * 1e3, 2),
            "sample": "%d training steps of the same %s workload (batch %d, identical session generator, numpy fp32 "
                      "oracle, OpenBLAS threads = host cores), %.1f s" % (n, a.config, a.batch, el)}


if __name__ == "__main__":
    main()

"""bench.py's host-side pieces without a GPU: the CPU-baseline / parity leg under the DRIVER's flags
(--steps 20 --warmup 5: round 1 read past the generated sessions there), the batch stream (Keras fit's
per-epoch reshuffle, short last batch) and the token-order bridge between the packed GPU batch and the
oracle's padded view."""
import importlib

import numpy as np

import bench
from oracle import nn as onn
from oracle import rng as orng

Bt = importlib.import_module("seq-recommendations_amd.batching")
Sy = importlib.import_module("seq-recommendations_amd.synthetic")
Sm = importlib.import_module("seq-recommendations_amd.sampling")


def _world(n_train=300, n_test=40, batch=32):
    cd = bench.CONFIGS["tiny"]
    gen = Sy.SyntheticSessions(cd["V"], seed=1234)
    probs = Sm.log_uniform_probs(cd["V"], gen.proposal_rank())
    th, al = Sm.build_alias_table(probs)
    flat, starts = gen.generate(n_train + n_test)
    stream = bench.BatchStream(0, n_train, batch, 1234)
    return cd, flat, starts, stream, th, al, np.log(probs).astype(np.float32)


def test_cpu_leg_runs_with_driver_flags_and_few_generated_batches():
    a = bench.parse(["--gpus", "1", "--steps", "20", "--warmup", "5", "--config", "tiny", "--batch", "32"])
    assert (a.steps, a.warmup) == (20, 5)
    cd, flat, starts, stream, th, al, logq = _world(batch=a.batch)
    w = bench.host_weights(cd, 1234)
    sels = [stream.sel(i).copy() for i in range(3)]            # far fewer batches than the leg will run: it must wrap
    sample = np.arange(300, 340)
    out = bench.cpu_leg(cd, a.batch, flat, starts, sels, w, th, al, logq, 1234, seconds=0.5, parity_steps=6,
                        sample_sel=sample, max_steps=25)
    assert out["kind"] == "port" and out["cores"] >= 1 and out["value"] > 0 and out["unit"] == "sessions/s"
    assert len(out["losses"]) == 6 and all(np.isfinite(out["losses"]))
    assert out["steps_timed"] >= 5
    rank, bi, ti = out["ranks"]
    assert rank.shape == bi.shape and rank.min() >= 0 and rank.max() < cd["V"]
    # the leg must not touch the caller's weights (the GPU engine was loaded from the same arrays)
    w2 = bench.host_weights(cd, 1234)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    # losses come from a training trajectory: the first equals a fresh oracle forward on batch 0
    net = onn.OracleNet(dict(cell=cd["cell"], act="relu", input="embed", output="sampled", tied=False, use_bias=True,
                             out_bias=False), {k: v.copy() for k, v in w.items()})
    l0 = net.forward(bench.padded_batch(flat, starts, sels[0]), negatives=orng.sample_negatives(1234, 0, cd["K"], th, al),
                     logq=logq)["loss"]
    assert abs(l0 - out["losses"][0]) < 1e-6


def test_batch_stream_is_a_reshuffled_partition_per_epoch():
    s = bench.BatchStream(100, 70, 32, 7)
    assert s.per_epoch == 3
    e0 = np.concatenate([s.sel(i) for i in range(3)])
    e1 = np.concatenate([s.sel(i) for i in range(3, 6)])
    assert len(s.sel(2)) == 6                                   # short last batch, like Keras fit
    for e in (e0, e1):
        assert np.array_equal(np.sort(e), np.arange(100, 170))
    assert not np.array_equal(e0, e1)
    assert np.array_equal(s.sel(1), bench.BatchStream(100, 70, 32, 7).sel(1))     # deterministic replay (CPU leg)


def test_sample_order_maps_packed_tokens_to_the_padded_row_major_order():
    cd, flat, starts, stream, th, al, logq = _world()
    sel = np.arange(300, 340)
    rb = Bt.pack_flat(flat, starts, sel)
    pb = bench.padded_batch(flat, starts, sel)
    bi, ti = np.nonzero(pb["mask"])
    idx = bench.sample_order(rb)
    assert np.array_equal(rb.ids[idx], pb["ids"][bi, ti])
    assert np.array_equal(rb.tgt[idx], pb["tgt"][bi, ti])
    # with an empty and a one-item session in the selection
    sessions = [flat[starts[i]:starts[i + 1]].tolist() for i in sel[:6]]
    sessions[2] = []
    sessions[4] = sessions[4][:1]
    rb = Bt.pack_sessions(sessions)
    idx = bench.sample_order(rb)
    want = [v for s in sessions for v in s[:-1]]
    assert rb.ids[idx].tolist() == want


def test_cpu_rank_counts_exclude_the_target_and_match_a_direct_count():
    cd, flat, starts, stream, th, al, logq = _world()
    w = bench.host_weights(cd, 5)
    w["Eout"] = (w["Eout"] * 50).astype(np.float32)
    pb = bench.padded_batch(flat, starts, np.arange(300, 310))
    rank, bi, ti = bench.cpu_rank_counts(onn, cd, w, pb, chunk=128)
    xw = w["E"][np.where(pb["mask"], pb["ids"], 0)] @ w["W"] + w["b"]
    hs, _ = onn.rnn_forward("gru", "relu", xw * pb["mask"][:, :, None], pb["mask"], w["U"])
    sc = hs[bi, ti] @ w["Eout"].T
    tg = pb["tgt"][bi, ti]
    ts = sc[np.arange(len(tg)), tg]
    ref = (sc > ts[:, None]).sum(1)
    assert np.abs(rank - ref).max() <= 1                       # einsum vs matmul rounding of the target's own score
    assert (rank == ref).mean() > 0.9


def test_bench_gpus_n_launches_itself_and_relays_rank0s_line():
    """`python bench.py --gpus N` outside a launcher (the driver's command shape) must start torch.distributed.run as a child,
    print rank 0's ONE JSON line and exit with the child's code (VERDICT r3 missing 1).  Rehearsed without a GPU: the ranks
    rendezvous over gloo (SEQREC_BENCH_DRYRUN=1) and rank 0 reports what it saw."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SEQREC_BENCH_DRYRUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"dryrun": True, "n_gpus": 2, "ranks_seen": 2, "steps": 3}
    # a failing child fails the bench: no GPU here, so the real path raises in every rank
    env.pop("SEQREC_BENCH_DRYRUN")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--config", "tiny", "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


class _OracleAsDevice:
    """Stands in for bench.GpuSide without a GPU: an fp32 oracle that keeps its own state (optionally with a planted fault)."""

    def __init__(self, cd, flat, starts, sels, th, al, logq, seed, fault_step=None):
        self.cd, self.flat, self.starts, self.sels, self.th, self.al, self.logq, self.seed = cd, flat, starts, sels, th, al, logq, seed
        self.fault_step = fault_step

    def reset(self, weights):
        self.p = {k: v.astype(np.float32) for k, v in weights.items()}
        self.a = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.net = onn.OracleNet(bench._oracle_cfg(self.cd), self.p)

    def step(self, i):
        neg = orng.sample_negatives(self.seed, i, self.cd["K"], self.th, self.al)
        o = self.net.forward(bench.padded_batch(self.flat, self.starts, self.sels[i % len(self.sels)]), negatives=neg, logq=self.logq)
        g = self.net.backward()
        if i == self.fault_step:
            g["W"] = g["W"] * np.float32(1.05)                  # a wrong gradient in ONE tensor at ONE step
        self.sc = onn.adagrad_step(self.p, self.a, g, lr=0.01, eps=1e-8, clipnorm=1.0)
        return float(o["loss"])

    def read(self, name, rows):
        return self.p[name].copy() if rows is None else self.p[name][rows].copy()

    def write(self, name, rows, p, a):
        if rows is None:
            self.p[name][...] = p; self.a[name][...] = a
        else:
            self.p[name][rows] = p; self.a[name][rows] = a

    def scale(self):
        return self.sc


def test_resync_parity_passes_a_faithful_device_and_names_a_planted_fault():
    """bench.parity_resync: three paths from identical numbers every step.  A faithful fp32 path passes resync_verdict; the same
    path with ONE tensor's gradient 5 % off at ONE step fails, and the verdict names that step and tensor."""
    cd, flat, starts, stream, th, al, logq = _world()
    w = bench.host_weights(cd, 1234)
    sels = [stream.sel(i).copy() for i in range(5)]
    good = bench.parity_resync(cd, flat, starts, sels, w, th, al, logq, 1234, 5, _OracleAsDevice(cd, flat, starts, sels, th, al, logq, 1234))
    ok, why = bench.resync_verdict(good)
    assert ok, why
    assert max(good["loss_rel_gpu"]) <= 1e-6 and len(good["loss_f64"]) == 5
    assert set(good["update"]) == {"E", "Eout", "W", "U", "b"}
    for u in good["update"].values():                          # the stand-in IS the fp32 oracle: identical figures on both sides
        assert u["gpu_l2"] == u["cpu32_l2"] and u["gpu_flips"] == u["cpu32_flips"]
    bad = bench.parity_resync(cd, flat, starts, sels, w, th, al, logq, 1234, 5,
                              _OracleAsDevice(cd, flat, starts, sels, th, al, logq, 1234, fault_step=3))
    ok, why = bench.resync_verdict(bad)
    assert not ok and any("step 3" in x and " W " in x for x in why), why
    assert not any("step 4" in x for x in why)                 # re-synchronised: the fault does not leak into the next step
    # the fp64 arbiter of the free-running comparison
    l64 = bench.cpu_free_run(cd, flat, starts, sels, w, th, al, logq, 1234, 4, np.float64)
    l32 = bench.cpu_free_run(cd, flat, starts, sels, w, th, al, logq, 1234, 4, np.float32)
    assert max(abs(a - b) / abs(b) for a, b in zip(l32, l64)) < 1e-4

#!/usr/bin/env python3
"""bench.py -- sessions/sec of the RNN next-item training step on MI355X.

Workload (BASELINE.json `metric`, configs[2] = "c3"): synthetic MSNBC-shaped sessions,
|items| = 1M, seq_len <= 50, batch 512 sessions per GPU, GRU hidden = 256 (Keras-2.0 GRU
equations, hard_sigmoid gates, relu), embedding width 256, sampled softmax with K = 2000 shared
log-uniform negatives, masked-token-mean CE, BPTT, global-norm clip 1.0 + Adagrad(lr 0.01,
eps 1e-8).  fp32 end to end (exact-fp32 MFMA).

What a "step" is (the loop of the reference's fit, experiments_methods.py:41-45 -> model.py:179-182):
the SURVEY 8(d) data set -- 200k training and 20k held-out sessions per GPU -- is parked in HBM once; every
step takes the NEXT 512 sessions of a per-epoch shuffle, builds its ragged batch on the device
(Engine.upload_device: length sort on the host, id / target / link gather on the GPU) and runs one full
training step.  Batch construction is INSIDE the timed region and no batch is revisited within an epoch
(390 steps).  `--resident` restores the round-1 measurement (<= 64 pre-built batches cycled) for comparison.

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline     -- the dominant kernel of the step (by summed device time, HIP events on the launch
                  stream over a profiled pass): algorithmic flops (or bytes) per launch / mean launch
                  duration against the gfx950 peak that bounds it;
  kernels      -- the same for every kernel class of the step;
  cpu_baseline -- the oracle (numpy fp32 restatement of the Keras/Theano path) timed on this box's
                  host cores on a bounded sample of the SAME workload (rank 0, N = 1 only);
  parity       -- GPU vs that CPU path from IDENTICAL host-generated initial weights on the same first
                  batches and negatives: per-step loss of both, max relative difference (bound 1e-3,
                  BASELINE.json north_star) and Recall@20 of both on the same held-out token sample.
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
PARITY_BOUND = 1e-3               # north_star: loss / Recall@K within 1e-3 relative

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
    # tiny shape for the CPU-only test of the cpu_baseline / parity leg (tests/test_bench_cpu.py)
    "tiny": dict(V=600, H=24, D=16, K=40, cell="gru", desc="tiny: test shape"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=512, help="sessions per GPU per step")
    ap.add_argument("--saturated", action="store_true", help="every session has 50 items (roofline runs)")
    ap.add_argument("--train-sessions", type=int, default=200_000, help="training sessions per GPU (SURVEY 8d)")
    ap.add_argument("--test-sessions", type=int, default=20_000, help="held-out sessions (Recall@20)")
    ap.add_argument("--resident", action="store_true",
                    help="round-1 measurement: cycle <= --distinct-batches pre-built batches instead of fresh ones")
    ap.add_argument("--distinct-batches", type=int, default=64)
    ap.add_argument("--settle", type=int, default=8, help="untimed steps before --warmup (see DESIGN.md 5)")
    ap.add_argument("--merge", default="atomic", choices=["atomic", "sorted"],
                    help="row-gradient merge: float atomics, or the deterministic sort + ordered segment sum (csrc/merge.hip)")
    ap.add_argument("--tune-steps", type=int, default=96, help="untimed steps of the eager-vs-graph scan-issue calibration (0 = off)")
    ap.add_argument("--profile-steps", type=int, default=10)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--parity-steps", type=int, default=8, help="steps whose loss is compared GPU vs CPU (>= 5)")
    ap.add_argument("--parity-sessions", type=int, default=256, help="held-out sessions of the Recall@20 parity sample")
    ap.add_argument("--arbiter", default="auto", choices=["auto", "on", "off"],
                    help="fp64 arbiter + re-synchronised single-step comparison of the parity leg: 'auto' runs them when the free-running "
                         "trajectories part by more than the bound")
    ap.add_argument("--resync-steps", type=int, default=0, help="steps of the re-synchronised comparison (0 = --parity-steps)")
    ap.add_argument("--recall-steps", type=int, default=1500, help="extra training steps before Recall@20 (0 = skip)")
    ap.add_argument("--force-sharded", action="store_true", help="use the row-sharded engine even on one GPU")
    ap.add_argument("--sharded-recall", action="store_true",
                    help="also compute Recall@20 through the sharded rank counting when N > 1 (default: only for N = 1)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# initial weights: E, Eout ~ U(-0.01, 0.01); W glorot-uniform; U orthogonal per gate; b = 0
# ------------------------------------------------------------------------------------------------
def host_weights(cd, seed):
    """UNPADDED numpy fp32 initial weights, generated once on the host and loaded into BOTH the GPU
    engine (Engine.set_param) and the CPU oracle: the parity leg starts both from identical values."""
    V, H, D = cd["V"], cd["H"], cd["D"]
    G = 3 if cd["cell"] == "gru" else (4 if cd["cell"] == "lstm" else 1)
    rs = np.random.default_rng(seed)

    def uni(shape, lim):
        a = rs.random(shape, dtype=np.float32)
        a *= np.float32(2 * lim)
        a -= np.float32(lim)
        return a

    p = {"E": uni((V, D), 0.01)}
    if not cd.get("tied", False):
        p["Eout"] = uni((V, H), 0.01)
    p["W"] = uni((D, G * H), float(np.sqrt(6.0 / (D + G * H))))
    p["U"] = np.concatenate([np.linalg.qr(rs.normal(size=(H, H)))[0] for _ in range(G)], axis=1).astype(np.float32)
    b = np.zeros(G * H, np.float32)
    if cd["cell"] == "lstm":
        b[H:2 * H] = 1.0                                   # Keras unit_forget_bias
    p["b"] = b
    return p


def init_params_device(eng, cd, seed):
    """Same distributions drawn on the device (multi-GPU runs and the full-size tests, where no CPU leg
    needs the values).  Padded columns stay exactly zero."""
    import torch
    g = torch.Generator(device=eng.dev)
    g.manual_seed(seed)
    c = eng.cfg
    H, D, G = c.H, c.D, eng.G
    with torch.no_grad():
        eng.P["E"].zero_()
        eng.P["E"][:, : (H if c.tied else D)].uniform_(-0.01, 0.01, generator=g)
        if "Eout" in eng.P:
            eng.P["Eout"].zero_()
            eng.P["Eout"][:, :H].uniform_(-0.01, 0.01, generator=g)
    rs = np.random.default_rng(seed)
    lim = float(np.sqrt(6.0 / (D + G * H)))
    eng.set_param("W", rs.uniform(-lim, lim, (D, G * H)).astype(np.float32))
    eng.set_param("U", np.concatenate([np.linalg.qr(rs.normal(size=(H, H)))[0] for _ in range(G)], axis=1).astype(np.float32))
    if "b" in eng.P:
        b = np.zeros(G * H, np.float32)
        if c.cell == "lstm":
            b[H:2 * H] = 1.0
        eng.set_param("b", b)


# ------------------------------------------------------------------------------------------------
# the batch stream: Keras fit's order (model.py:179-182 -> Model.fit: reshuffle every epoch, last batch short)
# ------------------------------------------------------------------------------------------------
class BatchStream:
    """sel(i) = the session indices of global step i: epoch e = i // steps_per_epoch uses a permutation
    of [lo, lo + n_train) seeded by (seed, e).  Deterministic, so the CPU leg replays the same batches."""

    def __init__(self, lo, n_train, batch, seed):
        self.lo, self.n, self.batch, self.seed = lo, n_train, batch, seed
        self.per_epoch = -(-n_train // batch)
        self._e, self._perm = -1, None

    def sel(self, i):
        e, j = divmod(i, self.per_epoch)
        if e != self._e:
            self._perm = self.lo + np.random.default_rng([self.seed, e]).permutation(self.n)
            self._e = e
        return self._perm[j * self.batch:(j + 1) * self.batch]


def kernel_model(cfgd, n_tok, K, B, slabs=None, uniq=None, upack_floats=0):
    """Algorithmic work per launch of every kernel class at n_tok tokens (SURVEY 8d formulas):
    name -> (bound, flops or bytes[, extra]).  MFMA kernels: flops of the product.  HBM kernels (the gather /
    scatter-update side): bytes the launch must move by its interface -- every input read once, every output written once, an
    in-place accumulate = read + write, rows counted DISTINCT where the kernel touches a row once per distinct id (`uniq`:
    measured distinct / listed ratio of the input and target lists of this data set; the K negatives are taken as distinct).
    slabs: split-K slab counts the step actually used (Engine.last_counts) -- a slab operand is an input of the launch that
    reads it.  extra (dict): secondary figures, e.g. the HBM rate of the rows an MFMA kernel gathers."""
    H, D = cfgd["H"], cfgd["D"]
    G = 3 if cfgd["cell"] == "gru" else (4 if cfgd["cell"] == "lstm" else 1)
    tied = bool(cfgd.get("tied", False))
    sl = dict(dX=1, dEneg=1, wgrad=1)
    sl.update({k: max(1, int(v)) for k, v in (slabs or {}).items()})
    u_in, u_tg = (uniq or (1.0, 1.0))
    dense = float(G * H * (D + H + 1))                                    # W, U, b
    # the step's scatter lists: input rows (E, width D), target rows and negative rows (Eout, width H)
    lists = [(n_tok, D, sl["dX"]), (n_tok, H, 1), (K, H, sl["dEneg"])]
    listed = sum(n * w for n, w, _ in lists)                               # row elements listed
    distinct = n_tok * u_in * D + n_tok * u_tg * H + K * H                 # row elements of distinct rows
    scatter_b = sum(4.0 * n * w * s for n, w, s in lists) + 4.0 * sum(n for n, _, _ in lists) + 8.0 * listed
    norm_b = 4.0 * dense * (sl["wgrad"] + 1) + 4.0 * distinct + 8.0 * sum(n for n, _, _ in lists) + 4.0 * n_tok
    apply_b = 20.0 * dense + 24.0 * distinct + 8.0 * sum(n for n, _, _ in lists)
    ce_b = 8.0 * n_tok * K + 8.0 * n_tok * H + 16.0 * n_tok + 8.0 * K
    pack_b = 4.0 * G * H * H + 4.0 * upack_floats + 8.0 * K * H + 4.0 * K
    batch_b = 20.0 * n_tok + 12.0 * B
    m = {
        "seqrec_rnn_fwd": ("mfma", 2.0 * G * H * H * n_tok),
        "seqrec_rnn_bwd": ("mfma", 2.0 * G * H * H * n_tok),
        # step-wise entry points: one C-ABI call = the whole scan (cluster form: one launch); work and time are per CALL
        "seqrec_rnn_fwd_stepwise": ("mfma", 2.0 * G * H * H * n_tok),
        "seqrec_rnn_bwd_stepwise": ("mfma", 2.0 * G * H * H * n_tok),
        "seqrec_gemm_f32[xw]": ("mfma", 2.0 * n_tok * D * G * H),
        # embedding rows read through the ids: the gather of north_star's "HBM GB/s for the gather" lives in this launch
        "seqrec_gemm_f32_fused[xw]": ("mfma", 2.0 * n_tok * D * G * H, {"gather_bytes": 4.0 * D * n_tok}),
        "seqrec_gemm_f32_fused[dH]": ("mfma", 2.0 * n_tok * K * H, {"gather_bytes": 4.0 * H * n_tok}),     # + the target-row term in the final write
        "seqrec_gemm_f32[logits]": ("mfma", 2.0 * n_tok * K * H),
        # dH = dlogits . Eneg (+ the target-row term) and dEneg = dlogits^T . H in ONE launch of two layout bodies (+ dH's reduce launch)
        "seqrec_gemm_f32_pair[dH+dEneg]": ("mfma", 4.0 * n_tok * K * H, {"gather_bytes": 4.0 * H * n_tok}),
        "seqrec_gemm_f32[dH]": ("mfma", 2.0 * n_tok * K * H),
        "seqrec_gemm_f32[dEneg]": ("mfma", 2.0 * n_tok * K * H),
        "seqrec_gemm_f32_slabs[dEneg]": ("mfma", 2.0 * n_tok * K * H),       # split-K slabs left for the row scatter: no reduce launch
        "seqrec_gemm_f32_slabs[dX]": ("mfma", 2.0 * n_tok * D * G * H),
        "seqrec_gemm_f32[dW]": ("mfma", 2.0 * n_tok * D * G * H),
        "seqrec_gemm_f32[dX]": ("mfma", 2.0 * n_tok * D * G * H),
        "seqrec_gemm_f32[dU]": ("mfma", 2.0 * n_tok * H * G * H / (2 if G == 3 else 1)),   # GRU: two launches
        "seqrec_gemm_f32_grouped[dW+dU]": ("mfma", 2.0 * n_tok * G * H * (D + H)),          # dW and dU in one launch
        "seqrec_gemm_f32_grouped_slabs[dW+dU]": ("mfma", 2.0 * n_tok * G * H * (D + H),     # slabs finished by the norm launch
                                                 {"gather_bytes": 4.0 * (D + H) * n_tok}),   # E[ids] and Hout[prev] read through their index
        # ... with dEneg = dlogits^T . H riding in the same launch (same layout, same reduction over the tokens)
        "seqrec_gemm_f32_grouped_slabs[dW+dU+dEneg]": ("mfma", 2.0 * n_tok * G * H * (D + H) + 2.0 * n_tok * K * H,
                                                       {"gather_bytes": 4.0 * (D + H) * n_tok}),
        # ---- the HBM side: gather / scatter-update kernels against the HBM peak
        "seqrec_rows_scatter_add_multi": ("hbm", scatter_b),
        "seqrec_opt_sqnorm_slabs": ("hbm", norm_b),
        "seqrec_opt_sqnorm": ("hbm", 4.0 * dense + 4.0 * distinct + 8.0 * sum(n for n, _, _ in lists) + 4.0 * n_tok),
        "seqrec_opt_apply": ("hbm", apply_b),
        "seqrec_sampled_softmax_ce": ("hbm", ce_b),
        "seqrec_rnn_pack_u_sample": ("hbm", pack_b),
        "seqrec_pack_batch_host": ("hbm", batch_b),
        "seqrec_gather_rows[E]": ("hbm", 8.0 * D * n_tok),                                   # 4 B read + 4 B written / elt (dropout, or > 8 192 tokens)
        "seqrec_gather_rows[Hprev]": ("hbm", 8.0 * H * n_tok),
    }
    if tied:
        pass          # same formula: one table takes all three lists
    return m


def lib_sha16():
    """sha256[:16] of the loaded libseqrec_hip.so: stamps what a committed PMC pass was taken on."""
    import hashlib
    L = importlib.import_module("seq-recommendations_amd._lib")
    return hashlib.sha256(open(L.LIB_PATH, "rb").read()).hexdigest()[:16]


def _latest_profile(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def pmc_traffic(kernel, t_mean, a):
    """HBM bytes per call of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r*_c3_pmc_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs of
    this same command; FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads).
    Only valid for the workload those passes profiled (c3, MSNBC-shaped); None otherwise."""
    path = _latest_profile("r*_c3_pmc_hbm_traffic.json")
    if a.config != "c3" or a.saturated or not path:
        return None
    pm = json.load(open(path))
    taken_on = pm.pop("_lib_sha16", None)
    src = os.path.relpath(path, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; library %s)" % (taken_on or "unstamped")
    if taken_on != lib_sha16():
        return {"bytes_per_call": None, "source": src + " -- NOT this build (%s): dropped" % lib_sha16()}
    # cluster form of the scan: the call IS one launch
    one = {"seqrec_rnn_fwd_stepwise": "gru_cluster_fwd<4, 0", "seqrec_rnn_bwd_stepwise": "gru_cluster_bwd<4, 0"}.get(kernel)
    ent = next((v for k, v in pm.items() if one and one in k), None)
    if ent is not None and os.environ.get("SEQREC_SCAN_CLUSTER", "1") != "0":
        return {"bytes_per_call": round((ent["fetch_kib_x2"] + ent["write_kib"]) * 1024.0), "source": src}
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
    return {"bytes_per_call": round(tot), "source": src}


def pmc_mfma_util(kernel, a):
    """Counter-based matrix-pipe utilisation of the dominant kernel's launches from the committed rocprofv3 PMC
    pass (profiles/r*_c3[_saturated]_pmc_mfma.json, tools/pmc_mfma.py); c3 only."""
    path = _latest_profile("r*_c3%s_pmc_mfma.json" % ("_saturated" if a.saturated else ""))
    if a.config != "c3" or not path:
        return None
    cluster = os.environ.get("SEQREC_SCAN_CLUSTER", "1") != "0"
    frag = ({"seqrec_rnn_fwd_stepwise": "gru_cluster_fwd<4, 0", "seqrec_rnn_bwd_stepwise": "gru_cluster_bwd<4, 0"} if cluster else
            {"seqrec_rnn_fwd_stepwise": "gru_step_fwd<4, 0, ", "seqrec_rnn_bwd_stepwise": "gru_step_bwd<4, 0, "}).get(kernel)
    if not frag:
        return None
    pm = json.load(open(path))
    taken_on = pm.pop("_lib_sha16", None)
    src = os.path.relpath(path, ROOT) + " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; library %s)" % (taken_on or "unstamped")
    if taken_on != lib_sha16():
        return {"source": src + " -- NOT this build (%s): dropped" % lib_sha16()}
    out = {k[k.index("gru_"):k.index(">") + 1]: v["mfma_util"] for k, v in pm.items() if frag in k}
    if not out:
        return None
    out["source"] = src
    return out


# ------------------------------------------------------------------------------------------------
# CPU leg: the oracle on the same workload -- baseline timing AND the parity reference
# ------------------------------------------------------------------------------------------------
def padded_batch(flat, starts, sel):
    """The reference's view of a batch (preprocessor.py:16-20,67-94): x = s[i], y = s[i+1], PRE-padded."""
    sel = np.asarray(sel, dtype=np.int64)
    L = (starts[sel + 1] - starts[sel] - 1).astype(np.int64)
    T = int(max(L.max(), 1))
    B = len(sel)
    ids = np.zeros((B, T), np.int64); tgt = np.zeros((B, T), np.int64); mask = np.zeros((B, T), bool)
    for b, s in enumerate(sel):
        n = L[b]
        if n <= 0:
            continue
        seq = flat[starts[s]:starts[s + 1]]
        ids[b, T - n:] = seq[:-1]; tgt[b, T - n:] = seq[1:]; mask[b, T - n:] = True
    return {"ids": ids, "tgt": tgt, "mask": mask}


def cpu_rank_counts(onn, cd, p, batch, chunk=65536):
    """rank[i] = #items scoring strictly above the target of real token i (oracle hidden states, scores
    h . Eout[v]); token order = np.nonzero(mask) (row-major), the caller matches it with the GPU's."""
    mask = batch["mask"]
    ids = np.where(mask, batch["ids"], 0)
    xw = p["E"][ids] @ p["W"] + p["b"]
    xw = xw * mask[:, :, None].astype(xw.dtype)
    hs, _ = onn.rnn_forward(cd["cell"], "relu", xw, mask, p["U"])
    bi, ti = np.nonzero(mask)
    h = hs[bi, ti]
    Et = p["E"] if cd.get("tied", False) else p["Eout"]
    tg = batch["tgt"][bi, ti]
    ts = np.einsum("ij,ij->i", h, Et[tg])
    rank = np.zeros(len(tg), np.int64)
    for c0 in range(0, Et.shape[0], chunk):
        gt = h @ Et[c0:c0 + chunk].T > ts[:, None]
        own = np.nonzero((tg >= c0) & (tg < c0 + chunk))[0]
        gt[own, tg[own] - c0] = False                  # the target never outranks itself (seqrec_rank_count: col != tgt)
        rank += gt.sum(axis=1)
    return rank, bi, ti


def cpu_leg(cd, batch, flat, starts, sels, weights, th, al, logq, seed, seconds, parity_steps, sample_sel,
            lr=0.01, eps=1e-8, clipnorm=1.0, max_steps=200):
    """Runs the oracle from `weights` over the batches sels[0], sels[1], ... (wrapping around: an optional
    leg must never index past what was generated) with the negatives of (seed, step): returns the timing,
    the first `parity_steps` losses and the rank counts of `sample_sel` after exactly `parity_steps` steps."""
    from oracle import nn as onn
    from oracle import rng as orng
    K = cd["K"]
    p = {k: v.copy() for k, v in weights.items()}
    acc = {k: np.zeros_like(v) for k, v in p.items()}
    net = onn.OracleNet(dict(cell=cd["cell"], act="relu", input="embed", output="sampled", tied=bool(cd.get("tied", False)),
                             use_bias=True, out_bias=False), p)

    def one(i):
        neg = orng.sample_negatives(seed, i, K, th, al)
        out = net.forward(padded_batch(flat, starts, sels[i % len(sels)]), negatives=neg, logq=logq)
        g = net.backward()
        onn.adagrad_step(p, acc, g, lr=lr, eps=eps, clipnorm=clipnorm)
        return float(out["loss"])

    losses, el, n, ranks = [], 0.0, 0, None
    parity_steps = max(1, parity_steps)
    while True:
        t0 = time.perf_counter()
        l = one(n)
        dt = time.perf_counter() - t0
        if n > 0:                                    # step 0 pages the tables in: warm-up, not timed
            el += dt
        if n < parity_steps:
            losses.append(l)
        n += 1
        if n == parity_steps and sample_sel is not None and len(sample_sel):
            ranks = cpu_rank_counts(onn, cd, p, padded_batch(flat, starts, sample_sel))     # untimed
        if n >= parity_steps and (el >= seconds or n >= max_steps):
            break
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    timed = max(n - 1, 1)
    return {"value": round(timed * batch / max(el, 1e-9), 1), "unit": "sessions/s", "cores": cores, "kind": "port",
            "ms_per_step": round(el / timed * 1e3, 2), "steps_timed": timed, "losses": losses, "ranks": ranks,
            "sample": "%d training steps of the same workload (batch %d, the GPU run's own first batches and negatives, "
                      "identical initial weights; numpy fp32 oracle, BLAS threads = host cores), %.1f s"
                      % (timed, batch, el)}


def _oracle_cfg(cd):
    return dict(cell=cd["cell"], act="relu", input="embed", output="sampled", tied=bool(cd.get("tied", False)), use_bias=True, out_bias=False)


def cpu_free_run(cd, flat, starts, sels, weights, th, al, logq, seed, steps, dtype, lr=0.01, eps=1e-8, clipnorm=1.0):
    """Losses of `steps` free-running training steps of the oracle computed in `dtype` from `weights` (the fp64 ARBITER of the
    parity leg: which of two fp32 trajectories that part is the one that left the exact one?)."""
    from oracle import nn as onn
    from oracle import rng as orng
    p = {k: v.astype(dtype) for k, v in weights.items()}
    acc = {k: np.zeros_like(v) for k, v in p.items()}
    net = onn.OracleNet(_oracle_cfg(cd), p)
    out = []
    for i in range(steps):
        neg = orng.sample_negatives(seed, i, cd["K"], th, al)
        o = net.forward(padded_batch(flat, starts, sels[i % len(sels)]), negatives=neg, logq=logq)
        onn.adagrad_step(p, acc, net.backward(), lr=lr, eps=eps, clipnorm=clipnorm)
        out.append(float(o["loss"]))
    return out


class GpuSide:
    """What parity_resync needs from the device path (tests/test_bench_cpu.py substitutes a second oracle)."""

    def __init__(self, eng, ds, sels, lr=0.01, eps=1e-8, clipnorm=1.0):
        self.eng, self.ds, self.sels, self.hp = eng, ds, sels, (lr, eps, clipnorm)

    def reset(self, weights):
        for k, v in weights.items():
            self.eng.set_param(k, v)
            self.eng.A[k].zero_()

    def step(self, i):
        lr, eps, clipnorm = self.hp
        d = self.eng.upload_device(self.ds, self.sels[i % len(self.sels)])
        return float(self.eng.train_step(d, lr=lr, eps=eps, clipnorm=clipnorm, step=i).item())

    def read(self, name, rows):
        return self.eng.get_param(name) if rows is None else self.eng.get_rows(name, rows)

    def write(self, name, rows, p, a):
        if rows is None:
            self.eng.set_param(name, p)
            self.eng.set_param(name, a, accum=True)
        else:
            self.eng.set_rows(name, rows, p)
            self.eng.set_rows(name, rows, a, accum=True)

    def scale(self):
        return float(self.eng.scale.item())


def parity_resync(cd, flat, starts, sels, weights, th, al, logq, seed, steps, gpu, lr=0.01, eps=1e-8, clipnorm=1.0):
    """RE-SYNCHRONISED single-step comparison.  Three paths -- the oracle in fp64 (master), the oracle in fp32
    and the device -- run step i on the same batch and negatives FROM IDENTICAL NUMBERS: after every step the master's updated
    weights and Adagrad accumulators are rounded to fp32 and loaded into all three (only the rows the step touched, plus the
    dense tensors), so rounding cannot compound from step to step.  Compared per step: the loss (computed before the update) and,
    per tensor, the update each path applied, dX = X_after - X_before, against the master's:
      l2       |dX - d64| / |d64| over the touched rows;
      flips    elements whose update differs from the master's by more than lr / 2 -- Adagrad from a zero (or tiny) accumulator moves
               an element by -lr g / (|g| + eps) ~ -lr sign(g): a gradient element within rounding of 0 flips its +-lr move;
      l2_rest  the same l2 without the flipped elements.
    The fp32 oracle stands beside the device as the yardstick: the device is held to what fp32 arithmetic itself achieves."""
    from oracle import nn as onn
    from oracle import rng as orng
    K = cd["K"]
    ocfg = _oracle_cfg(cd)
    p64 = {k: v.astype(np.float64) for k, v in weights.items()}
    a64 = {k: np.zeros_like(v) for k, v in p64.items()}
    p32 = {k: v.astype(np.float32) for k, v in weights.items()}
    a32 = {k: np.zeros_like(v) for k, v in p32.items()}
    net64, net32 = onn.OracleNet(ocfg, p64), onn.OracleNet(ocfg, p32)
    gpu.reset(weights)
    rec = {"steps": steps, "loss_f64": [], "loss_rel_gpu": [], "loss_rel_cpu32": [], "clip_scale_rel_gpu": [], "clip_scale_rel_cpu32": [],
           "tokens_ce_clipped": [], "update": {}}
    for i in range(steps):
        batch = padded_batch(flat, starts, sels[i % len(sels)])
        neg = orng.sample_negatives(seed, i, K, th, al)
        o64 = net64.forward(batch, negatives=neg, logq=logq)
        g64 = net64.backward()
        o32 = net32.forward(batch, negatives=neg, logq=logq)
        g32 = net32.backward()
        lg = gpu.step(i)
        # tokens whose target probability sits below the Theano clip (zero gradient there: the loss is continuous, its gradient is not)
        lt, ln = o64["lt"], o64["ln"]
        mx = np.maximum(lt, ln.max(axis=1))
        pt = np.exp(lt - mx) / (np.exp(lt - mx) + np.exp(ln - mx[:, None]).sum(axis=1))
        rec["tokens_ce_clipped"].append(int((pt < 1e-7).sum()))
        rows = {k: (np.asarray(v[0]) if isinstance(v, tuple) else None) for k, v in g64.items()}
        before = {k: (p32[k].copy() if r is None else p32[k][r].copy()) for k, r in rows.items()}
        sc64 = onn.adagrad_step(p64, a64, g64, lr=lr, eps=eps, clipnorm=clipnorm)
        sc32 = onn.adagrad_step(p32, a32, g32, lr=lr, eps=eps, clipnorm=clipnorm)
        rec["loss_f64"].append(float(o64["loss"]))
        rec["loss_rel_gpu"].append(float("%.3g" % rel(lg, float(o64["loss"]))))
        rec["loss_rel_cpu32"].append(float("%.3g" % rel(float(o32["loss"]), float(o64["loss"]))))
        rec["clip_scale_rel_gpu"].append(float("%.3g" % rel(gpu.scale(), sc64)))
        rec["clip_scale_rel_cpu32"].append(float("%.3g" % rel(sc32, sc64)))
        for k, r in rows.items():
            b = before[k].astype(np.float64)
            d64 = (p64[k] if r is None else p64[k][r]) - b
            cand = {"gpu": gpu.read(k, r).astype(np.float64) - b, "cpu32": (p32[k] if r is None else p32[k][r]).astype(np.float64) - b}
            u = rec["update"].setdefault(k, {"elements": [], "gpu_l2": [], "cpu32_l2": [], "gpu_flips": [], "cpu32_flips": [],
                                             "gpu_l2_rest": [], "cpu32_l2_rest": []})
            u["elements"].append(int(d64.size))
            den = max(float(np.linalg.norm(d64)), 1e-30)
            for who, dx in cand.items():
                e = dx - d64
                flip = np.abs(e) > 0.5 * lr
                u[who + "_l2"].append(float("%.3g" % (np.linalg.norm(e) / den)))
                u[who + "_flips"].append(int(flip.sum()))
                u[who + "_l2_rest"].append(float("%.3g" % (np.linalg.norm(e[~flip]) / den)))
        # re-synchronise: the master, rounded to fp32, becomes everybody's state
        for k, r in rows.items():
            if r is None:
                p32[k][...] = p64[k]; a32[k][...] = a64[k]
                p64[k][...] = p32[k]; a64[k][...] = a32[k]
                gpu.write(k, None, p32[k], a32[k])
            else:
                p32[k][r] = p64[k][r]; a32[k][r] = a64[k][r]
                p64[k][r] = p32[k][r]; a64[k][r] = a32[k][r]
                gpu.write(k, r, p32[k][r], a32[k][r])
    return rec


RESYNC_LOSS_BOUND = 1e-5          # one step from identical numbers: loss of the device vs the fp64 master
RESYNC_REST_BOUND = 1e-3          # ... and its update outside the +-lr flips (north_star's bound, on the update itself)


def resync_verdict(rec):
    """Machine-checked reading of a parity_resync record: ok iff at EVERY step the device's loss is within RESYNC_LOSS_BOUND of the
    fp64 master's and, per tensor, its update outside the flipped elements is within RESYNC_REST_BOUND (or within 4x of what the fp32
    oracle manages), with no more flips than 4x the fp32 oracle's + 8."""
    why = []
    for i, x in enumerate(rec["loss_rel_gpu"]):
        if not x <= RESYNC_LOSS_BOUND:
            why.append("step %d: loss off by %.3g" % (i, x))
    for k, u in rec["update"].items():
        for i in range(len(u["gpu_l2"])):
            if not u["gpu_l2_rest"][i] <= max(RESYNC_REST_BOUND, 4.0 * u["cpu32_l2_rest"][i]):
                why.append("step %d: update of %s off by %.3g outside the flips (fp32 oracle: %.3g)" % (i, k, u["gpu_l2_rest"][i], u["cpu32_l2_rest"][i]))
            if u["gpu_flips"][i] > 4 * u["cpu32_flips"][i] + 8:
                why.append("step %d: %d flipped elements in %s (fp32 oracle: %d)" % (i, u["gpu_flips"][i], k, u["cpu32_flips"][i]))
    return (not why), why


def sample_order(rb):
    """Packed-token index of every real (session row, step) in the row-major order cpu_rank_counts uses."""
    order, ls, so = rb.order.astype(np.int64), rb.lengths.astype(np.int64), rb.step_off.astype(np.int64)
    inv = np.full(rb.n_sessions, -1, np.int64)
    inv[order] = np.arange(len(order))                   # original row -> sorted row
    out = []
    for b in np.sort(order):                             # sessions with >= 1 transition, original order
        sb = inv[b]
        out.append(so[:ls[sb]] + sb)
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def self_launch(a, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start `python -m torch.distributed.run --nproc-per-node N
    bench.py <same flags>` as a CHILD process -- before this process has touched the GPU (torch is not even imported yet;
    replacing a process that has initialised HIP is forbidden on the pool) -- relay rank 0's JSON line on stdout and
    exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as s:                     # a free rendezvous port (several benches may share a node)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    args = list(sys.argv[1:] if argv is None else argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + args
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the pool's host driver supports dmabuf IPC only (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)      # stderr passes through
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        sys.stderr.write("bench.py: the %d-rank child printed no JSON line\n" % a.gpus)
        return 1
    return r.returncode


def main(argv=None):
    a = parse(argv)
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a, argv))
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
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it as `python bench.py --gpus N` or under "
                         "torch.distributed.run --nproc-per-node N" % (a.gpus, world))
    if os.environ.get("SEQREC_BENCH_DRYRUN") == "1":
        # launcher rehearsal (tests/test_bench_cpu.py, no GPU): rendezvous over gloo, one collective, rank 0's line relayed
        import torch.distributed as dist_
        dist_.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist_.all_reduce(t)
        dist_.barrier()
        if rank == 0:
            os.dup2(saved_stdout, 1)
            print(json.dumps({"dryrun": True, "n_gpus": world, "ranks_seen": int(t.item()), "steps": a.steps}), flush=True)
        dist_.destroy_process_group()
        return
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
    SEED = 1234
    dev = "cuda:%d" % local
    ncfg = E.NetConfig(cell=cd["cell"], act="relu", H=H, V_in=V, V_out=V, input="embed", D=D, output="sampled", K=K,
                       tied=bool(cd.get("tied", False)), use_bias=True, out_bias=False, logq=True, seed=SEED, merge=a.merge)
    sharded = dist is not None
    notes = []          # a failing OPTIONAL leg (single process only) must not take the throughput line with it
    do_cpu = rank == 0 and world == 1 and not sharded and a.cpu_seconds > 0
    if sharded:
        Dm = importlib.import_module("seq-recommendations_amd.distributed")
        eng = Dm.ShardedEngine(ncfg, dev, dist)
    else:
        eng = E.Engine(ncfg, dev)
    weights = None
    if do_cpu:
        weights = host_weights(cd, SEED)
        for k, v in weights.items():
            eng.set_param(k, v)
    else:
        init_params_device(eng, cd, seed=SEED + rank)
    if sharded and world > 1:
        for k in ("W", "U", "b"):                       # replicated cell weights: rank 0's values everywhere
            dist.broadcast(eng.P[k], src=0)
        eng.upack_dirty = True
    gen = Sy.SyntheticSessions(V, seed=SEED)
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

    # ---- data: world * n_train training sessions (rank r owns slice r) + n_test held-out, parked in HBM
    n_train, n_test = a.train_sessions, a.test_sessions
    flat, starts = gen.generate(world * n_train + n_test, saturated=a.saturated)
    test_lo = world * n_train
    stream = BatchStream(rank * n_train, n_train, a.batch, SEED)
    ds = None if sharded else eng.put_dataset(flat, starts)
    if not sharded:
        eng.reserve(a.batch * 49)                      # no workspace growth while fresh batches stream through
    resident = []
    if a.resident:
        nb = max(1, min(a.distinct_batches, stream.per_epoch))
        resident = [eng.upload(Bt.pack_flat(flat, starts, stream.sel(i))) for i in range(nb)]

    tok_seen = []
    WINDOW = 32         # sharded: the routing of the next WINDOW batches is planned together (2 collectives, no host wait)
    planner = None
    if sharded and not a.resident:
        planner = Dm.WindowPlanner(eng, lambda j: Bt.pack_flat(flat, starts, stream.sel(j)), WINDOW)

    def next_batch(i):
        if resident:
            d = resident[i % len(resident)]
        elif sharded:
            # the next WINDOW batches are routed while the GPU still has half a window of steps queued: the count exchange is begun
            # there and ended a quarter window later (distributed.WindowPlanner) -- the loop never waits for the planner
            d = planner.get(i)
        else:
            d = eng.upload_device(ds, stream.sel(i), defer=True)       # as catalogue.SampledRNNModel's fit loop does
        return d

    def train(i):
        d = next_batch(i)
        tok_seen.append((d["n"], d["T"], d["rb"].n_sessions))
        return eng.train_step(d, lr=0.01, eps=1e-8, clipnorm=1.0, step=i)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def gpu_ranks(sel):
        d = eng.upload(Bt.pack_flat(flat, starts, sel))
        return eng.rank_counts(d), d

    step = 0
    # ---- parity leg, GPU side: the first P steps from the host-generated weights, loss fetched per step
    parity = None
    P = max(5, a.parity_steps)
    sample_sel = np.arange(test_lo, test_lo + min(a.parity_sessions, n_test))
    gpu_par = None
    if do_cpu:
        try:
            gl = []
            for i in range(P):
                gl.append(float(train(step).item()))
                step += 1
            rk, dsmp = gpu_ranks(sample_sel)
            gpu_par = {"losses": gl, "ranks": rk.cpu().numpy()[sample_order(dsmp["rb"])]}
        except Exception as e:                                   # noqa: BLE001
            notes.append("parity leg (GPU side) failed: %r" % (e,))

    # ---- settle (untimed; DESIGN.md 5 says what it covers), warm-up, timed region
    # Python's cyclic GC: the first full (generation-2) collection of a fresh process walks every object torch
    # and numpy created at import -- 40 ms in ONE training step somewhere in the first ~60 (measured with
    # gc.callbacks, tools/stall_probe.py; round 1 papered over it with 128 settle steps).  Collect once now
    # and freeze the survivors out of later collections, as a long-running trainer would after start-up.
    import gc
    gc.collect()
    gc.freeze()
    # how the scan's launches are issued is decided per box (fast host core: eager; slow one: rewritten hipGraph) on
    # untimed training steps -- Engine.autotune_scan; SEQREC_SCAN_GRAPH=0/1 pins it
    scan_issue = None
    cluster = os.environ.get("SEQREC_SCAN_CLUSTER", "1") != "0"      # every cell has a cluster form: one launch per scan call, nothing to tune
    if cluster:
        scan_issue = {"scan_issue": "one launch per call"}
    elif not sharded and "SEQREC_SCAN_GRAPH" not in os.environ and a.tune_steps > 0:
        def _one():
            nonlocal step
            train(step)
            step += 1
            return tok_seen[-1][1]
        scan_issue = eng.autotune_scan(_one, blocks=3, block_steps=max(4, a.tune_steps // 6), warm=max(8, a.tune_steps // 3))
    else:
        scan_issue = {"scan_issue": ("graph" if getattr(eng, "use_graph", False) else "eager"), "pinned": True}
    for i in range(a.settle):
        train(step)
        step += 1
    sync()
    for i in range(a.warmup):
        train(step)
        step += 1
    sync()
    first_timed = len(tok_seen)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]     # one event per step: the median next to the mean
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        loss = train(step)
        marks[i + 1].record()
        step += 1
    sync()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    ms_median = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    timed = tok_seen[first_timed:first_timed + a.steps]
    sess_timed = float(sum(t[2] for t in timed))
    n_tok_mean = float(np.mean([t[0] for t in timed]))
    t_mean = float(np.mean([t[1] for t in timed]))
    t_max = int(max(t[1] for t in timed))
    if dist is not None and world > 1:
        tt = torch.tensor([sess_timed, n_tok_mean], device=dev, dtype=torch.float64)
        dist.all_reduce(tt)
        sess_total, tok_total = float(tt[0].item()), float(tt[1].item()) * a.steps
    else:
        sess_total, tok_total = sess_timed, n_tok_mean * a.steps
    # the scan's length is the batch's LONGEST session: the mean over a whole epoch of this stream's batches stands beside the timed
    # window's (a 20-step window's T_mean wanders by +-8 %, and with it the step time: read `value` with both)
    trans = (starts[1:] - starts[:-1] - 1)
    t_mean_epoch = round(float(np.mean([int(trans[stream.sel(i)].max()) for i in range(stream.per_epoch)])), 1)
    ms_per_step = dt / a.steps * 1e3
    sessions_per_s = sess_total / dt
    tokens_per_s = tok_total / dt
    last_loss = float(loss.item())

    # ---- per-kernel device time (HIP events on the launch stream), fresh batches too
    kern = {}
    roof = None
    if a.profile_steps > 0:
        try:
            mark = len(tok_seen)
            E.profile_start()
            for i in range(a.profile_steps):
                train(step)
                step += 1
            prof = E.profile_stop()
            n_prof = float(np.mean([t[0] for t in tok_seen[mark:]]))
            t_prof = float(np.mean([t[1] for t in tok_seen[mark:]]))
            # distinct / listed rows of the input and target lists, measured on 8 batches of this data set (host)
            ui, ut = [], []
            for j in range(8):
                sel_j = stream.sel(j)
                ids_j = np.concatenate([flat[starts[x]:starts[x + 1] - 1] for x in sel_j])
                tgt_j = np.concatenate([flat[starts[x] + 1:starts[x + 1]] for x in sel_j])
                ui.append(len(np.unique(ids_j)) / max(len(ids_j), 1)); ut.append(len(np.unique(tgt_j)) / max(len(tgt_j), 1))
            uniq = (float(np.mean(ui)), float(np.mean(ut)))
            model = kernel_model(cd, n_prof, K, a.batch, slabs=getattr(eng, "last_counts", None), uniq=uniq,
                                 upack_floats=int(eng.upack.numel()))
            tot = sum(ms for _, ms in prof.values())
            for name, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                per = ms / cnt
                ent = {"launches_per_step": cnt / a.profile_steps, "avg_us": round(per * 1e3, 2),
                       "share": round(ms / tot, 4)}
                if name in model:
                    bound, work = model[name][:2]
                    extra = model[name][2] if len(model[name]) > 2 else {}
                    if "gather_bytes" in extra:      # rows this MFMA kernel reads through an index: their HBM-side rate
                        ent.update(gather_bytes=round(extra["gather_bytes"]), gather_GBps=round(extra["gather_bytes"] / (per * 1e-3) / 1e9, 1))
                    if bound == "mfma":
                        ach = work / (per * 1e-3) / 1e12
                        ent.update(bound="mfma", achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                                   frac=round(ach / PEAK_F32_MFMA_TFLOPS, 5))
                    else:
                        ach = work / (per * 1e-3) / 1e9
                        ent.update(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                                   frac=round(ach / PEAK_HBM_GBS, 5), bytes=round(work))
                kern[name] = ent
            dom = next((k for k in kern if "bound" in kern[k]), None)
            if dom:
                e = kern[dom]
                tr = pmc_traffic(dom, t_prof, a)
                roof = {"kernel": dom, "bound": e["bound"], "achieved": e["achieved"], "peak": e["peak"], "unit": e["unit"],
                        "frac": e["frac"], "traffic": tr["bytes_per_call"] if tr else None,
                        "traffic_unit": "HBM-side bytes per call (FETCH_SIZE x2 + WRITE_SIZE)" if tr and tr["bytes_per_call"] else None,
                        "traffic_source": tr["source"] if tr else None, "pmc_mfma_util": pmc_mfma_util(dom, a),
                        "avg_us": e["avg_us"], "share_of_step": e["share"], "tokens_per_call": round(n_prof, 1)}
        except Exception as e:                                   # noqa: BLE001
            if dist is not None:
                raise
            E._PROF = None
            notes.append("profile leg failed: %r" % (e,))

    # ---- Recall@20 on the held-out sessions after some more training
    recall = None
    trained_par = None
    if a.recall_steps > 0 and sharded and (world == 1 or a.sharded_recall):
        for i in range(a.recall_steps):
            train(step)
            step += 1
        per = n_test // world                                  # every rank scores the same number of held-out sessions
        base = test_lo + rank * per
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
                train(step)
                step += 1
            hits = n = 0
            for s in range(0, n_test, a.batch):
                rk, d = gpu_ranks(np.arange(test_lo + s, test_lo + min(n_test, s + a.batch)))
                hits += int((rk < 20).sum().item())
                n += d["n"]
            recall = hits / max(n, 1)
            if do_cpu:
                rk, dsmp = gpu_ranks(sample_sel)
                trained_par = {"ranks": rk.cpu().numpy()[sample_order(dsmp["rb"])], "steps": step}
        except Exception as e:                                   # noqa: BLE001
            notes.append("recall leg failed: %r" % (e,))

    # ---- CPU leg: the oracle from the SAME initial weights on the SAME batches / negatives
    cpu = None
    if do_cpu:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            sels = [stream.sel(i).copy() for i in range(min(stream.per_epoch, 200))]
            cpu = cpu_leg(cd, a.batch, flat, starts, sels, weights, th, al, logq, SEED, a.cpu_seconds, P, sample_sel)
            cl, cr = cpu.pop("losses"), cpu.pop("ranks")
            if gpu_par is not None:
                gl = gpu_par["losses"]
                worst = max(rel(g, c) for g, c in zip(gl, cl))
                per = [rel(g, c) for g, c in zip(gl, cl)]
                lead = next((i for i, x in enumerate(per) if x > PARITY_BOUND), len(per))
                parity = {"steps": P, "loss_gpu": [round(x, 6) for x in gl], "loss_cpu": [round(x, 6) for x in cl],
                          "max_rel_diff": float("%.3g" % worst), "bound": PARITY_BOUND,
                          # where two fp32 implementations part: c3 never does in these steps; c4's relu-LSTM-512 trajectory is
                          # rounding-sensitive from its 5th step on -- GPU variants that differ only in summation order part from
                          # EACH OTHER there by as much (profiles/r03_c4_parity_sensitivity.txt)
                          "rel_diff_per_step": [float("%.3g" % x) for x in per], "leading_steps_within_bound": int(lead)}
                ok = worst <= PARITY_BOUND
                if cr is not None:
                    gr = gpu_par["ranks"]
                    rg, rc = float((gr < 20).mean()), float((cr[0] < 20).mean())
                    parity["recall_at_20_sample"] = {
                        "after_steps": P, "tokens": int(len(gr)), "gpu": round(rg, 6), "cpu": round(rc, 6),
                        "rel_diff": float("%.3g" % rel(rg, rc)) if max(rg, rc) > 0 else 0.0,
                        "identical_ranks_frac": round(float((gr == cr[0]).mean()), 5)}
                    ok = ok and (rg == rc or rel(rg, rc) <= PARITY_BOUND)
                if trained_par is not None:
                    # evaluation parity at the TRAINED weights (non-trivial Recall@20): the GPU's parameters are
                    # downloaded into the oracle, which scores the same held-out sample on the host cores
                    from oracle import nn as onn
                    pw = {k: eng.get_param(k) for k in weights}
                    cr2 = cpu_rank_counts(onn, cd, pw, padded_batch(flat, starts, sample_sel))
                    gr2 = trained_par["ranks"]
                    rg, rc = float((gr2 < 20).mean()), float((cr2[0] < 20).mean())
                    parity["recall_at_20_sample_trained"] = {
                        "after_steps": trained_par["steps"], "tokens": int(len(gr2)), "gpu": round(rg, 6), "cpu": round(rc, 6),
                        "rel_diff": float("%.3g" % rel(rg, rc)) if max(rg, rc) > 0 else 0.0,
                        "identical_ranks_frac": round(float((gr2 == cr2[0]).mean()), 5),
                        "note": "GPU-trained weights scored by both paths"}
                    ok = ok and (rg == rc or rel(rg, rc) <= PARITY_BOUND)
                # ---- the arbiter: when two fp32 trajectories part, fp64 says which one left, and the
                # re-synchronised comparison says whether any single step of the device is wrong
                if a.arbiter == "on" or (a.arbiter == "auto" and worst > PARITY_BOUND):
                    l64 = cpu_free_run(cd, flat, starts, sels, weights, th, al, logq, SEED, P, np.float64)
                    g64 = [float("%.3g" % rel(g, x)) for g, x in zip(gl, l64)]
                    c64 = [float("%.3g" % rel(c, x)) for c, x in zip(cl, l64)]
                    parity["vs_fp64"] = {"loss_f64": [round(x, 6) for x in l64], "gpu_rel": g64, "cpu32_rel": c64,
                                         "note": "free-running: the same oracle in fp64 on the same batches / negatives / initial weights"}
                    rs = parity_resync(cd, flat, starts, sels, weights, th, al, logq, SEED, a.resync_steps or P, GpuSide(eng, ds, sels))
                    rs_ok, why = resync_verdict(rs)
                    rs["ok"], rs["why_not"] = bool(rs_ok), why
                    rs["bounds"] = {"loss": RESYNC_LOSS_BOUND, "update_outside_flips": RESYNC_REST_BOUND}
                    parity["resync"] = rs
                    parity["resync_ok"] = bool(rs_ok)
                    # ill-conditioned = the fp32 ORACLE cannot hold the bound against its own fp64 version over these steps
                    ill = max(c64) > PARITY_BOUND
                    # the device may leave the exact trajectory no earlier than a step before the fp32 oracle does, and while the oracle is inside
                    # the bound the device must be too
                    first = lambda xs: next((i for i, x in enumerate(xs) if x > PARITY_BOUND), len(xs))
                    fg = first(g64)
                    # ... or, where it leaves earlier (which step a chaotic trajectory departs at is itself chance: the scatter's float
                    # atomics make two runs of one build differ), the fp32 oracle's own distance to fp64 has by then grown a thousandfold
                    # from its step-0 rounding level -- the amplification is under way on the reference's side too
                    no_worse = fg + 1 >= first(c64) or (fg < len(c64) and c64[fg] >= 1e3 * max(c64[0], 1e-9))
                    parity["trajectory_ok"] = bool(worst <= PARITY_BOUND)
                    parity["ill_conditioned"] = bool(ill)
                    parity["ok_reason"] = ("trajectory within bound" if worst <= PARITY_BOUND else
                                           ("free-running trajectory ill-conditioned (the fp32 oracle leaves its fp64 version at step %d, the device at step %d); "
                                            "every single step from identical numbers agrees (resync_ok)" % (first(c64), first(g64))) if (ill and rs_ok and no_worse) else
                                           "FAILED: " + "; ".join(why[:4] or ["device leaves the fp64 trajectory before the fp32 oracle does"]))
                    if worst > PARITY_BOUND:
                        # Recall@20 after P free-running steps of an ill-conditioned trajectory is not comparable either; the trained-weights
                        # comparison (the SAME weights scored by both paths) is
                        tp = parity.get("recall_at_20_sample_trained")
                        ok = bool(ill and rs_ok and no_worse) and (tp is None or tp["gpu"] == tp["cpu"] or tp["rel_diff"] <= PARITY_BOUND)
                parity["ok"] = bool(ok)
        except Exception as e:                                   # noqa: BLE001
            notes.append("cpu_baseline leg failed: %r" % (e,))

    eng.check_status()          # a device-side failure anywhere in the run (refused update, exchange timeout, bad index) fails the bench
    if rank == 0:
        mode = ("resident: %d pre-built batches cycled" % len(resident)) if resident else \
            "fresh batches: %d-session train set in HBM, batch built on the device inside the timed region, reshuffled per epoch" % n_train
        out = {
            "metric": "sessions/sec", "value": round(sessions_per_s, 1), "unit": "sessions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median": round(ms_median, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cd["desc"] + (" (saturated: 50 items/session)" if a.saturated else " (MSNBC-shaped lengths)"),
                       "global_batch": a.batch * world, "seq_len": 50, "tokens_per_step_per_gpu": round(n_tok_mean, 1),
                       "t_mean": round(t_mean, 1), "t_max": t_max, "t_mean_epoch": t_mean_epoch, "settle_steps": a.settle, "batches": mode, "scan": scan_issue, "row_gradient_merge": a.merge,
                       "train_sessions_per_gpu": n_train, "test_sessions": n_test,
                       "routing_window": WINDOW if (sharded and not resident) else None,
                       "routing_host_ms_per_window": ({"pack": round(1e3 * planner.pack_s / max(planner.windows, 1), 2),
                                                       "plan": round(1e3 * planner.plan_s / max(planner.windows, 1), 2),
                                                       "sync_wait": round(1e3 * getattr(eng.ex, "sync_wait_s", 0.0) / max(planner.windows, 1), 2),
                                                       "windows": planner.windows} if planner is not None else None),
                       "parallelism": ("dp%d+row-sharded-tables" % world) if sharded else "single"},
            "tokens_per_s": round(tokens_per_s, 1), "final_loss": round(last_loss, 5), "recall_at_20": recall,
            "recall_after_steps": step if recall is not None else None,
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "kernels": kern, "lib_sha16": lib_sha16(),
        }
        # the cluster scan kernels wait in bounded spins; a healthy run has none that ran out
        out["config"]["scan"] = dict(scan_issue or {}, form="cluster (one launch, in-kernel exchange)" if
                                     os.environ.get("SEQREC_SCAN_CLUSTER", "1") != "0" else "step-wise",
                                     exchange_timeouts=int(importlib.import_module("seq-recommendations_amd._lib").load().seqrec_cluster_scan_errors(
                                         torch.cuda.current_stream().cuda_stream)))
        if notes:
            out["notes"] = notes
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

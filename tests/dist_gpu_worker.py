"""2-rank end-to-end check of distributed.ShardedEngine on ONE GPU: both ranks compute on cuda:0,
collectives go through gloo on CPU-staged copies (RCCL refuses two ranks on one device).  Rank 0
replays the same two training steps with the oracle on the GLOBAL model and compares losses, the
replicated cell weights and every rank's table shards.  Run under torch.distributed.run."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nn as onn, rng as orng          # noqa: E402
from helpers import make_sessions, pad_batch       # noqa: E402
from engine_helpers import oracle_drop             # noqa: E402


def _poison_empty():
    """SEQREC_POISON=1: every torch.empty on the GPU comes back filled with 1e30 (SEQREC_POISON=nan: NaN) / a huge negative
    int, so a kernel that reads memory nobody wrote cannot depend on what the caching allocator happened to hand out.
    The float poison is a huge FINITE value on purpose: in the Keras clip rule `norm >= clipnorm` is false for NaN (scale 1,
    the step goes through with NaN weights only if the NaN reaches a weight), while one 1e30 in any gradient makes the
    squared norm overflow -- the case that turns the step into a no-op (clip scale 0) and that seqrec_opt_apply now
    refuses with SEQREC_STATUS_BAD_NORM, raised by Engine.check_status().  The int poison stays negative: a negative
    row index is a filler by contract (zero row / ignored contribution); a positive wild index would be an out-of-bounds
    WRITE in the scatter, i.e. a GPU fault."""
    real = torch.empty
    fill = float("nan") if os.environ.get("SEQREC_POISON") == "nan" else 1e30

    def empty(*a, **k):
        t = real(*a, **k)
        if t.is_cuda and t.numel():
            if t.dtype.is_floating_point:
                t.fill_(fill)
            elif t.dtype in (torch.int32, torch.int64):
                t.fill_(-(2 ** 30))
        return t
    torch.empty = empty


def main():
    if os.environ.get("SEQREC_POISON"):
        _poison_empty()
    dist.init_process_group("gloo")
    rank, R = dist.get_rank(), dist.get_world_size()
    E = importlib.import_module("seq-recommendations_amd.engine")
    D = importlib.import_module("seq-recommendations_amd.distributed")
    Bt = importlib.import_module("seq-recommendations_amd.batching")
    Sm = importlib.import_module("seq-recommendations_amd.sampling")
    for case in (dict(cell="gru", V=1501, H=128, Dm=64, K=64 * R, tied=False),      # D != H: one exchange per table
                 dict(cell="gru", V=1201, H=128, Dm=128, K=160 * R, tied=False),    # unified tables, ids span 2 id rows
                 dict(cell="lstm", V=900, H=64, Dm=64, K=32 * R, tied=True),
                 dict(cell="lstm", V=1100, H=512, Dm=512, K=48 * R, tied=False),        # c4's cell: LSTM 512, row-sharded
                 # dropout in the sharded step: the reference's main run trains with z_to_y_drop = 0.3 (experiments_server.py:108-114)
                 dict(cell="lstm", V=1000, H=128, Dm=128, K=40 * R, tied=False, drop=dict(drop_out=0.3)),
                 dict(cell="gru", V=950, H=128, Dm=128, K=40 * R, tied=True, drop=dict(drop_in=0.1, drop_rec=0.2, drop_out=0.3)),
                 # a per-item output bias (RNNBaseline's Dense(n_classes) bias, model.py:257) row-sharded with the output table
                 dict(cell="gru", V=1000, H=128, Dm=128, K=40 * R, tied=False, out_bias=True)):
        V, H, Dm, K, tied, cell = case["V"], case["H"], case["Dm"], case["K"], case["tied"], case["cell"]
        G = onn.N_GATES[cell]
        rs = np.random.default_rng(4)                       # identical on every rank: the GLOBAL model
        p = {"E": rs.normal(0, 0.3, (V, Dm)), "W": rs.normal(0, 0.1, (Dm, G * H)), "U": rs.normal(0, 0.08, (H, G * H)),
             "b": rs.normal(0, 0.1, (G * H,))}
        if not tied:
            p["Eout"] = rs.normal(0, 0.3, (V, H))
        obias = bool(case.get("out_bias", False))
        if obias:
            p["bout"] = rs.normal(0, 0.3, (V,))
        p = {k: v.astype(np.float32) for k, v in p.items()}
        probs = Sm.log_uniform_probs(V)
        dropkw = case.get("drop", {})
        cfg = E.NetConfig(cell=cell, act="relu", H=H, V_in=V, V_out=V, input="embed", D=Dm, output="sampled", K=K, tied=tied,
                          logq=True, seed=9, out_bias=obias, **dropkw)
        eng = D.ShardedEngine(cfg, "cuda:0", D.HostStagedDist(dist, verify=True))
        eng.debug_capture = True             # keeps this rank's own dense gradients of the last step for the failure report
        for k in ("W", "U", "b"):
            eng.set_param(k, p[k])
        eng.set_param("E", p["E"][rank::R])
        if not tied:
            eng.set_param("Eout", p["Eout"][rank::R])
        if obias:
            eng.set_param("bout", p["bout"][rank::R])
        tables = []
        for j in range(R):
            pl = probs[j::R] / probs[j::R].sum()
            th, al = Sm.build_alias_table(pl)
            tables.append((th, al, (np.log(pl) - np.log(R)).astype(np.float32)))
        eng.set_sampler(*tables[rank])
        data_rng = np.random.default_rng(100)
        steps = []
        for s in range(2):
            per_rank = [make_sessions(data_rng, 24 + 5 * r, V, 2, 10) for r in range(R)]
            steps.append(per_rank)
        losses = []
        # both steps' batches are routed at once: two collectives + one host sync for the whole window (ShardedEngine.prepare)
        prepared = eng.prepare([Bt.pack_sessions(per_rank[rank]) for per_rank in steps])
        for s, per_rank in enumerate(steps):
            d = prepared[s]
            if s == 0:
                ev = float(eng.eval_loss(d, step=s).item())          # forward only, same negatives, same weights
            l = eng.train_step(d, lr=0.01, eps=1e-8, clipnorm=1.0, step=s)
            losses.append(float(l.item()))
            try:
                eng.check_status()   # raises if the update was refused (norm / divisor / scale) or an exchange index was out of range
            except Exception:
                print("FAILURE REPORT rank %d case %s step %d: %s" % (rank, case, s, eng.failure_report()), flush=True)
                raise
            sc_ = float(eng.scale.item())
            assert np.isfinite(sc_) and sc_ > 0.0, (case, rank, s, sc_)
            if s == 0 and not dropkw:
                assert ev == losses[0], (ev, losses[0])
        if True:
            # sharded Recall@K support: global rank of every target, counted shard by shard (unified tables and, D != H, split)
            rk = eng.rank_counts(d).cpu().numpy()
            hd = (eng._hidden(d, eng._rows_in(d, 0)[0]) if eng.unified else
                  eng._hidden(d, None, X=eng._split_rows(d, 0, negatives=False)[0])).cpu().numpy()[:, :H]
            tname = "E" if tied else "Eout"
            info = [None] * R
            dist.all_gather_object(info, (rk, hd, d["tgt"].cpu().numpy(), eng.get_param(tname), eng.get_param("bout") if obias else None))
            if rank == 0:
                table = np.zeros((V, H), np.float32)
                bvec = np.zeros(V, np.float64)
                for j in range(R):
                    table[j::R] = info[j][3]
                    if obias:
                        bvec[j::R] = info[j][4]
                for j in range(R):
                    rkj, hj, tj = info[j][:3]
                    sc = hj.astype(np.float64) @ table.T.astype(np.float64) + bvec[None, :]
                    ref = (sc > sc[np.arange(len(tj)), tj][:, None]).sum(1)
                    assert (rkj == ref).mean() > 0.97 and np.abs(rkj - ref).max() <= 2, (case, j, np.abs(rkj - ref).max())
                print("rank counts ok:", case["cell"], "tied" if tied else "untied")
            # sharded top-k prediction: global item ids of the 10 best items of every token
            ti, tv = eng.topk_rows(d, k=10, chunk=300)
            tk = [None] * R
            dist.all_gather_object(tk, (ti.cpu().numpy(), tv.cpu().numpy()))
            if rank == 0:
                for j in range(R):
                    hj = info[j][1]
                    sc = hj.astype(np.float64) @ table.T.astype(np.float64) + bvec[None, :]
                    ref = np.argsort(-sc, axis=1, kind="stable")[:, :10]
                    gi_, gv_ = tk[j]
                    assert gi_.shape == ref.shape and (gi_ == ref).mean() > 0.99, (case, j, (gi_ == ref).mean())
                    np.testing.assert_allclose(gv_, np.take_along_axis(sc, ref, axis=1), rtol=2e-5, atol=2e-5)
                print("sharded topk ok:", case["cell"])
        got = {"loss": losses, "W": eng.get_param("W"), "U": eng.get_param("U"), "b": eng.get_param("b"),
               "E": eng.get_param("E")}
        if not tied:
            got["Eout"] = eng.get_param("Eout")
        if obias:
            got["bout"] = eng.get_param("bout")
        allgot = [None] * R
        dist.all_gather_object(allgot, got)
        scales = [None] * R
        dist.all_gather_object(scales, float(eng.scale.item()))
        assert len(set(scales)) == 1, ("the clip scale must be bit-identical on every rank (replicated weights)", scales)
        for k in ("W", "U", "b"):
            for r in range(1, R):
                assert np.array_equal(allgot[r][k], allgot[0][k]), ("replicated weights differ between ranks", case, k, r)
        if rank == 0:
            ocfg = dict(cell=cell, act="relu", input="embed", output="sampled", tied=tied, use_bias=True, out_bias=obias)
            op = {k: v.copy() for k, v in p.items()}
            acc = {k: np.zeros_like(v) for k, v in op.items()}
            net = onn.OracleNet(ocfg, op)
            logq = np.zeros(V, np.float32)
            for j in range(R):
                logq[j::R] = tables[j][2]
            Kr = K // R
            for s, per_rank in enumerate(steps):
                draws = [orng.sample_negatives(9, s * R + j, R * Kr, tables[j][0], tables[j][1]) for j in range(R)]
                n_r = [sum(max(len(x) - 1, 0) for x in per_rank[r]) for r in range(R)]
                N = float(sum(n_r))
                gsum, lsum = {}, 0.0
                for r in range(R):
                    neg = np.concatenate([draws[j][r * Kr:(r + 1) * Kr].astype(np.int64) * R + j for j in range(R)]).astype(np.int32)
                    import dataclasses
                    batch_r = pad_batch(per_rank[r])
                    drop_r = oracle_drop(dataclasses.replace(cfg, seed=cfg.seed + 1000003 * (r + 1)), per_rank[r], batch_r, s)   # rank r's mask stream
                    out = net.forward(batch_r, negatives=neg, logq=logq, drop=drop_r)
                    g = net.backward()
                    wgt = n_r[r] / N
                    lsum += out["loss"] * wgt
                    for k, v in g.items():
                        if isinstance(v, tuple):
                            dense = np.zeros(op[k].shape, np.float64)
                            dense[v[0]] = v[1]
                            v = dense
                        gsum[k] = gsum.get(k, 0.0) + v * wgt
                # engine: rank r reports loss_sum_r * R / N ; the mean over ranks is the global loss
                eng_loss = float(np.mean([allgot[r]["loss"][s] for r in range(R)]))
                if abs(eng_loss - lsum) > 2e-5 * max(1.0, abs(lsum)):      # say what every rank saw before failing
                    print("LOSS MISMATCH", case, "step", s, "engine", [allgot[r]["loss"] for r in range(R)], "oracle", lsum, flush=True)
                    for r in range(R):
                        for k in ("W", "U", "b"):
                            print("  rank", r, k, "max |diff| vs oracle BEFORE this step's update",
                                  float(np.abs(allgot[r][k] - op[k]).max()), flush=True)
                assert abs(eng_loss - lsum) <= 2e-5 * max(1.0, abs(lsum)), (case, s, eng_loss, lsum)
                gfin = {}
                for k, v in gsum.items():
                    if k in ("E", "Eout"):
                        rows = np.nonzero(np.abs(v).sum(axis=1))[0]
                        gfin[k] = (rows, v[rows].astype(np.float32))
                    else:
                        gfin[k] = v.astype(np.float32)
                onn.adagrad_step(op, acc, gfin, lr=0.01, eps=1e-8, clipnorm=1.0)
            for r in range(R):
                for k in ("W", "U", "b"):
                    err = np.abs(allgot[r][k] - op[k]).max() / max(1e-6, np.abs(op[k]).max())
                    assert err < 2e-3, (case, r, k, err)
                for k in (("E",) if tied else ("E", "Eout")) + (("bout",) if obias else ()):
                    ref = op[k][r::R]
                    diff = np.abs(allgot[r][k] - ref)
                    # Adagrad's lr*sign(g) steps on numerically-zero gradients can flip: allow a vanishing fraction
                    assert (diff > 2e-3 * np.abs(ref).max()).mean() < 2e-3, (case, r, k, diff.max())
            print("case ok:", case, "global loss", lsum, "per-rank losses", [allgot[r]["loss"] for r in range(R)])
        dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()

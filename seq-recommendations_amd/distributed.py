"""Multi-GPU form of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

Partitioning -- the reference is single-process, so this is the build's design:
  * sessions are data-parallel: every rank trains on its own batch of B sessions per step
    (weak scaling; global batch = B * R);
  * the cell weights W, U, b are replicated; their gradients are summed with ONE all-reduce of a
    flat bucket (<= 8.4 MB at c4) -- small, so it is a single latency-bound collective;
  * the item tables E and Eout are ROW-SHARDED, row r on rank r mod R at local index r // R
    (interleaved, so the Zipf head spreads evenly).  Rows are moved point-to-point with
    all-to-all (every peer pair has its own xGMI link, nothing rides a ring):
      forward   owners gather the requested rows and send them      (A2A of [rows, width] fp32)
      backward  row gradients travel the same routes in reverse and are scatter-added into the
                owner's gradient table; the owner alone runs the sparse Adagrad for its rows;
  * negatives are stratified by owner: every owner draws K/R negatives per requesting rank from its
    own shard (shard-local alias table), so the negative exchange has fixed sizes and needs no id
    round trip;  Q(v) = Q_shard(v) / R is the proposal used for the log-Q correction;
  * the global gradient norm is  sum_ranks |owned table-row grads|^2 + |dense grads|^2  (one small
    all-reduce), so every rank applies the same Keras clip scale.

``RowExchange`` is device-agnostic torch code (unit-tested with gloo on CPU, world size 2 and 3);
``ShardedEngine`` wires it to the HIP kernels.
"""
import numpy as np
import torch

from . import _lib
from ._lib import ptr
from .engine import Engine, call, CELL, ACT, INT32_MAX


class RowPlan:
    """Routing of one list of global row ids (fixed per batch): who owns what, in which order."""
    __slots__ = ("n", "send_counts", "recv_counts", "perm", "inv_perm", "recv_local", "m")


class RowExchange:
    def __init__(self, dist, group=None, device="cpu"):
        self.dist, self.group = dist, group
        self.R = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dev = torch.device(device)

    # -- plan ------------------------------------------------------------------------------------
    def plan(self, ids):
        """ids: int tensor [n] of GLOBAL row ids this rank needs.  Collective (tiny): exchanges the
        per-peer counts and the requested local row indices.  Call at batch-upload time."""
        R, dist = self.R, self.dist
        ids = ids.to(self.dev).long()
        n = ids.numel()
        owner = ids % R
        perm = torch.argsort(owner, stable=True)
        send_counts = torch.bincount(owner, minlength=R)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        m = int(sum(rc))
        want = (ids[perm] // R).to(torch.int32)
        got = torch.empty(m, dtype=torch.int32, device=self.dev)
        dist.all_to_all_single(got, want, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        p = RowPlan()
        p.n, p.m = n, m
        p.send_counts, p.recv_counts = sc, rc
        p.perm = perm.to(torch.int32)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(n, device=self.dev)
        p.inv_perm = inv.to(torch.int32)
        p.recv_local = got                      # local row index of every row peers asked me for
        return p

    # -- forward: fetch rows ------------------------------------------------------------------------
    def fetch(self, plan, gather_local, width, take):
        """gather_local(idx int32[m]) -> [m, width] rows of MY shard; returns [n, width] rows in the
        order of the ids given to plan().  take(src [n,width], idx int32[n]) -> src[idx]."""
        mine = gather_local(plan.recv_local)
        out = torch.empty((plan.n, width), dtype=mine.dtype, device=self.dev)
        self.dist.all_to_all_single(out, mine, output_split_sizes=plan.send_counts,
                                    input_split_sizes=plan.recv_counts, group=self.group)
        return take(out, plan.inv_perm)

    # -- backward: push row gradients to their owners -------------------------------------------------
    def push(self, plan, grads, take):
        """grads [n, width] in plan order -> ([m, width] contributions, their local rows int32[m])."""
        sorted_g = take(grads, plan.perm)
        out = torch.empty((plan.m, grads.shape[1]), dtype=grads.dtype, device=self.dev)
        self.dist.all_to_all_single(out, sorted_g, output_split_sizes=plan.recv_counts,
                                    input_split_sizes=plan.send_counts, group=self.group)
        return out, plan.recv_local

    # -- fixed-size exchange (stratified negatives) -----------------------------------------------------
    def swap_fixed(self, x):
        """x [R, k, ...]: slice j goes to rank j; returns [R, k, ...] with slice i from rank i."""
        out = torch.empty_like(x)
        self.dist.all_to_all_single(out, x.contiguous(), group=self.group)
        return out


def shard_rows(table, rank, R):
    """Rows r = rank, rank+R, ... of a global [V, w] array (numpy or torch)."""
    return table[rank::R]


def shard_size(V, rank, R):
    return (V - rank + R - 1) // R


class ShardedEngine(Engine):
    """Engine whose item tables hold only this rank's rows.  Single-rank groups degenerate to the
    plain engine arithmetic (same kernels, the exchanges become local copies)."""

    def __init__(self, cfg, device, dist, group=None):
        if cfg.input != "embed" or cfg.output != "sampled":
            raise ValueError("ShardedEngine shards item tables: it needs input='embed', output='sampled'")
        self.dist, self.group = dist, group
        self.R = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if cfg.K % self.R:
            raise ValueError("K=%d must be divisible by the world size %d (stratified negatives)" % (cfg.K, self.R))
        self.V_global = cfg.V_out
        import dataclasses
        local = dataclasses.replace(cfg, V_in=shard_size(cfg.V_in, self.rank, self.R),
                                    V_out=shard_size(cfg.V_out, self.rank, self.R))
        Engine.__init__(self, local, device)
        self.gcfg = cfg
        self.ex = RowExchange(dist, group, self.dev)
        self.n_total = {}

    # ---- helpers -----------------------------------------------------------------------------------
    def _take(self, src, idx):
        out = torch.empty((idx.numel(), src.shape[1]), dtype=src.dtype, device=self.dev)
        call("seqrec_gather_rows", ptr(src), ptr(idx), ptr(out), idx.numel(), src.shape[1], None, None, 0, self._stream())
        return out

    def _gather_from(self, table):
        return lambda idx: self._take(table, idx)

    def set_sampler(self, thresh, alias, logq=None):
        """Shard-LOCAL alias table (proposal restricted to this rank's rows) and the log of the
        EFFECTIVE proposal Q(v) = Q_shard(v) / R for the local rows.  The per-item log-Q values are
        then replicated (V floats) so that no step needs a log-Q exchange."""
        Engine.set_sampler(self, thresh, alias, logq)
        self.logq_global = None
        if logq is not None:
            R = self.R
            mine = self.sampler[2]
            nmax = shard_size(self.V_global, 0, R)
            pad = torch.zeros(nmax, dtype=torch.float32, device=self.dev)
            pad[: mine.numel()] = mine
            allq = [torch.empty_like(pad) for _ in range(R)]
            if R > 1:
                self.dist.all_gather(allq, pad, group=self.group)
            else:
                allq = [pad]
            g = torch.zeros(self.V_global, dtype=torch.float32, device=self.dev)
            for j in range(R):
                g[j::R] = allq[j][: shard_size(self.V_global, j, R)]
            self.logq_global = g

    def upload(self, rb):
        d = Engine.upload(self, rb)
        d["plan_in"] = self.ex.plan(d["ids"])
        d["plan_tgt"] = self.ex.plan(d["tgt"])
        nt = torch.tensor([d["n"]], dtype=torch.float64, device=self.dev)
        self.dist.all_reduce(nt, group=self.group)
        d["n_total"] = float(nt.item())
        if self.cfg.logq and self.logq_global is not None:
            d["lq_tgt"] = self.logq_global[d["tgt"].long()]         # fixed per batch
        return d

    # ---- one training step ----------------------------------------------------------------------------
    def train_step(self, d, lr=0.01, eps=1e-8, clipnorm=1.0, step=None, negatives=None, apply_update=True):
        c, P, R = self.cfg, self.P, self.R
        if step is None:
            step = self.step_count
        self.step_count = step + 1
        st = self._stream()
        n, T, B = d["n"], d["T"], d["B"]
        Hp, GHp, Dp = self.Hp, self.GHp, self.Dp
        K, Kr = c.K, c.K // R
        tname = "E" if c.tied else "Eout"
        inv = 1.0 / d["n_total"]
        # -- forward: remote rows in
        X = self.ex.fetch(d["plan_in"], self._gather_from(P["E"]), Dp, self._take)
        Etgt = self.ex.fetch(d["plan_tgt"], self._gather_from(P[tname]), Hp, self._take)
        th, al, lq = self.sampler
        negl = self.buf("negl", R * Kr, dtype=torch.int32)             # local rows I draw for every requester
        call("seqrec_sample_negatives", int(c.seed), int(step) * R + self.rank, R * Kr, ptr(th), ptr(al), c.V_out, ptr(negl), st)
        # rows and their global ids travel together: [R, Kr, Hp + 1] with the id bit-cast into the last column
        ids_out = (negl.long() * R + self.rank).to(torch.int32)                # global ids of my draws
        pay = torch.empty((R * Kr, Hp + 1), dtype=torch.float32, device=self.dev)
        pay[:, :Hp] = self._take(P[tname], negl)
        pay[:, Hp] = ids_out.view(torch.float32)
        got = self.ex.swap_fixed(pay.view(R, Kr, Hp + 1)).view(K, Hp + 1)
        Eneg = got[:, :Hp].contiguous()
        neg = got[:, Hp].contiguous().view(torch.int32)
        lq_neg = self.logq_global[neg.long()] if c.logq else None
        XW = self.buf("XW", n, GHp)
        self.gemm(1, 0, n, GHp, Dp, X, Dp, P["W"], GHp, XW, GHp, bias=P.get("b"), tag="xw")
        Hout = self.buf("Hout", n, Hp); gates = self.buf("gates", n, GHp); aux = self.buf("aux", n, Hp)
        self._scan_fwd(d, XW, Hout, gates, aux)
        Hd = Hout
        ln = self.buf("ln", n, K)
        self.gemm(1, 1, n, K, Hp, Hd, Hp, Eneg, Hp, ln, K, tag="logits")
        # log-Q correction and hit masking need per-candidate vectors here (rows are remote): fold the
        # negatives' logq into the logits, the targets' logq into a per-token vector
        dlt = self.buf("dlt", n)
        loss_rows = self.buf("loss_rows", n)
        ar = self.buf("arange", n, dtype=torch.int32)
        ar.copy_(torch.arange(n, device=self.dev, dtype=torch.int32))
        lq_tgt = d.get("lq_tgt")
        call("seqrec_sampled_softmax_ce_rows", ptr(ln), K, ptr(Hd), Hp, ptr(Etgt), ptr(lq_tgt), ptr(lq_neg), ptr(d["tgt"]),
             ptr(neg), n, K, inv, ptr(loss_rows), ptr(dlt), st)
        call("seqrec_reduce_sum", ptr(loss_rows), n, ptr(self.loss_sum), 0, st)
        # -- backward
        dHd = self.buf("dHd", n, Hp)
        self.gemm(1, 0, n, Hp, K, ln, K, Eneg, Hp, dHd, Hp, splitk=self._splitk(n, Hp, K), tag="dH")
        call("seqrec_gather_rows", ptr(Etgt), ptr(ar), ptr(dHd), n, Hp, ptr(dlt), None, 1, st)
        dEneg = self.buf("dEneg", K, Hp)
        self.gemm(0, 0, K, Hp, n, ln, K, Hd, Hp, dEneg, Hp, splitk=self._splitk(K, Hp, n), tag="dEneg")
        dEtgt = self.buf("dEtgt", n, Hp)
        call("seqrec_fill_f32", ptr(dEtgt), 0.0, n * Hp, st)
        call("seqrec_gather_rows", ptr(Hd), ptr(ar), ptr(dEtgt), n, Hp, ptr(dlt), None, 0, st)
        dPre = self.buf("dPre", n, GHp)
        self._scan_bwd(d, dHd, Hout, gates, aux, dPre)
        Gd, Gt = self.Gd, self.Gt
        cs_ws = self.buf("colsum_ws", 64 * GHp)
        call("seqrec_colsum", ptr(dPre), n, GHp, GHp, ptr(Gd["b"]), 0, ptr(cs_ws), st)
        Hprev = self.buf("Hprev", n, Hp)
        call("seqrec_gather_rows", ptr(Hout), ptr(d["prev"]), ptr(Hprev), n, Hp, None, None, 0, st)
        sk = self._splitk(Hp, GHp, n)
        if c.cell == "gru":
            self.gemm(0, 0, Hp, 2 * Hp, n, Hprev, Hp, dPre, GHp, Gd["U"], GHp, splitk=sk, tag="dU")
            self.gemm(0, 0, Hp, Hp, n, aux, Hp, dPre[:, 2 * Hp:], GHp, Gd["U"][:, 2 * Hp:], GHp, splitk=sk, tag="dU")
        else:
            self.gemm(0, 0, Hp, GHp, n, Hprev, Hp, dPre, GHp, Gd["U"], GHp, splitk=sk, tag="dU")
        self.gemm(0, 0, Dp, GHp, n, X, Dp, dPre, GHp, Gd["W"], GHp, splitk=self._splitk(Dp, GHp, n), tag="dW")
        dX = self.buf("dX", n, Dp)
        self.gemm(1, 1, n, Dp, GHp, dPre, GHp, P["W"], GHp, dX, Dp, splitk=self._splitk(n, Dp, GHp), tag="dX")
        # -- row gradients back to their owners, scatter-add into the local gradient tables
        jobs = []
        g_in, r_in = self.ex.push(d["plan_in"], dX, self._take)
        g_tg, r_tg = self.ex.push(d["plan_tgt"], dEtgt, self._take)
        g_ng = self.ex.swap_fixed(dEneg.view(R, Kr, Hp)).view(R * Kr, Hp)        # grads for the rows I drew
        base = 0
        for (tab, rows, g) in ((tname, r_tg, g_tg), (tname, negl, g_ng), ("E", r_in, g_in)):
            m = rows.numel()
            if m:
                call("seqrec_rows_scatter_add", ptr(Gt[tab]), ptr(self.slot[tab]), ptr(rows), ptr(g), g.shape[1], None, m,
                     g.shape[1], base, st)
                jobs.append((tab, rows, m, g.shape[1], base, g))
            base += m
        if not apply_update:
            return jobs
        # -- dense gradients: one flat all-reduce; global norm; update
        flat = torch.cat([Gd[k].reshape(-1) for k in sorted(Gd)])
        self.dist.all_reduce(flat, group=self.group)
        o = 0
        for k in sorted(Gd):
            nk = Gd[k].numel()
            Gd[k].copy_(flat[o:o + nk].view_as(Gd[k]))
            o += nk
        self.sq.zero_()
        for (tab, rows, m, w, b, _) in jobs:
            call("seqrec_rows_sqnorm", ptr(Gt[tab]), ptr(self.slot[tab]), ptr(rows), m, w, b, ptr(self.sq), st)
        self.dist.all_reduce(self.sq, group=self.group)
        for k in sorted(Gd):
            call("seqrec_sqnorm", ptr(Gd[k]), Gd[k].numel(), ptr(self.sq), st)
        call("seqrec_clip_scale", ptr(self.sq), float(clipnorm if clipnorm else 0.0), ptr(self.scale), st)
        for k in sorted(Gd):
            call("seqrec_adagrad_dense", ptr(P[k]), ptr(self.A[k]), ptr(Gd[k]), Gd[k].numel(), lr, eps, ptr(self.scale), st)
        self.upack_dirty = True
        for (tab, rows, m, w, b, _) in jobs:
            call("seqrec_rows_adagrad", ptr(P[tab]), ptr(self.A[tab]), ptr(Gt[tab]), ptr(self.slot[tab]), ptr(rows), m, w, b,
                 lr, eps, ptr(self.scale), st)
        return self.loss_sum * (self.R / d["n_total"])      # this rank's share, scaled so the mean over ranks is the global loss
